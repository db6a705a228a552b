#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused particle-world step on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric config): simple_spread, 3 agents / 3 landmarks, 65536 worlds PER GPU
(weak scaling: rank r owns worlds [r*B, (r+1)*B); no collective on the step path), fp32,
uniform random one-hot moves resident in HBM, device-side reset every 25 steps (MADDPG episode).
One "step" = every world of the batch advanced once with all agents' obs/reward/done written.

Timed region: barrier + synchronize, K steps, synchronize + barrier; max over ranks; rank 0 prints
ONE JSON line.  `--mode graph` (default) replays the K launches from a HIP graph (no host in the
loop), `--mode eager` issues them from Python through the C ABI, `--mode api` goes through
MultiAgentEnv.step(), `--mode fused` uses the persistent T-step rollout kernel.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic (compulsory) HBM bytes per env-step, SURVEY.md 8(d): read agent pos+vel, landmark
# pos, one-hot actions; write agent pos+vel, obs, reward (fp32) and done (1 byte per agent)
def algorithmic_bytes(scenario, A, L, obs_total):
    reads = 4 * A + 2 * L + 5 * A
    writes = 4 * A + obs_total + A
    return 4 * (reads + writes) + A


HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def cpu_baseline(scenario, kw, seconds, procs):
    """The oracle's per-object fp64 loop (oracle/mpe_loop.py -- the reference's algorithm and cost
    structure; /root/reference itself cannot travel to the GPU box) timed on the host cores."""
    import multiprocessing as mp
    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(scenario, kw, seconds, i) for i in range(procs)])
    steps = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    single = res[0][0] / res[0][1]
    return steps / wall, single, time.time() - t0


def _cpu_worker(arg):
    scenario, kw, seconds, idx = arg
    import numpy as np
    from oracle import spec as ospec
    from oracle.mpe_loop import LoopEnv
    np.random.seed(idx)
    spec = ospec.by_name(scenario, **kw)
    env = LoopEnv(spec)
    A = spec.n_agents
    acts = np.eye(5)[np.random.randint(0, 5, size=(4096, A))]
    n, t0 = 0, time.perf_counter()
    while True:
        if n % 25 == 0:
            env.reset()
        env.step(list(acts[n % 4096]))
        n += 1
        if n % 50 == 0 and time.perf_counter() - t0 >= seconds:
            break
    return n, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=65536, help="worlds per GPU")
    ap.add_argument("--scenario", default="simple_spread")
    ap.add_argument("--agents", type=int, default=3)
    ap.add_argument("--episode-len", type=int, default=25)
    ap.add_argument("--mode", default="graph", choices=["graph", "eager", "api", "fused"])
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the step path has no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node == --gpus"

    import multiagent_particle_envs_amd as mpe
    from multiagent_particle_envs_amd.rollout import RandomRollout
    kw = {}
    if args.scenario == "simple_spread" and args.agents != 3:
        kw["num_agents"] = args.agents
    B, K, W = args.batch, args.steps, args.warmup
    env = mpe.make_env(args.scenario, batch_size=B, seed=args.seed, **kw)
    env.world.world_offset = rank * B          # global world numbering: shards never share an RNG stream
    A, Lm = len(env.world.agents), len(env.world.landmarks)
    roll = RandomRollout(env, episode_len=args.episode_len, pool=16)
    dev = torch.device("cuda", local)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the K-step body ---------------------------------------------------------------------------
    graph = None
    if args.mode == "graph":
        graph = roll.capture(K)

        def body():
            graph.replay()
    elif args.mode == "eager":
        def body():
            roll.enqueue(K)
    elif args.mode == "fused":
        def body():
            roll.fused(K)
    else:
        def body():
            for k in range(K):
                if args.episode_len and k % args.episode_len == 0:
                    env.reset()
                env.step(roll.pool[k % len(roll.pool)])

    # warmup: W untimed steps
    roll.enqueue(W) if args.mode != "fused" else roll.fused(W)
    barrier()
    times = []
    kern_ms = []
    for rep in range(args.repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        body()
        e1.record()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        times.append(dt)
        kern_ms.append(e0.elapsed_time(e1))
    dt = sorted(times)[len(times) // 2]                 # median of the repeats
    ev_ms = sorted(kern_ms)[len(kern_ms) // 2]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant kernel's own launch duration: back-to-back launches between two HIP events on the
    # launch stream (no resets in between); includes the ~1.5 us dependent-launch boundary ----------
    n_k = 400
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if args.mode == "fused":
        roll.episode_len, keep = 0, roll.episode_len
        torch.cuda.synchronize()
        e0.record()
        roll.fused(n_k)
        e1.record()
        torch.cuda.synchronize()
        roll.episode_len = keep
    else:
        roll.episode_len, keep = 0, roll.episode_len
        gk = roll.capture(n_k)
        torch.cuda.synchronize()
        e0.record()
        gk.replay()
        e1.record()
        torch.cuda.synchronize()
        roll.episode_len = keep
    kernel_us = e0.elapsed_time(e1) * 1e3 / n_k

    if rank == 0:
        obs_total = int(env._obs_off[-1])
        bytes_step = algorithmic_bytes(args.scenario, A, Lm, obs_total)
        total_steps = float(B) * K * world
        value = total_steps / dt
        achieved = bytes_step * B / (kernel_us * 1e-6) / 1e9
        out = {
            "metric": "env steps/sec (whole node), %s N=%d, batch=%d per GPU" % (args.scenario, A, B),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s A=%d L=%d, %d worlds/GPU, one-hot random moves, reset every %d steps"
                                   % (args.scenario, A, Lm, B, args.episode_len),
                       "batch_per_gpu": B, "global_batch": B * world, "mode": args.mode,
                       "repeats": args.repeats, "sharding": "worlds by batch index, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_env_step": bytes_step,
                         "kernel": "mpe::k_narrow" if A <= 6 else "mpe::k_wide",
                         "kernel_us_per_step": kernel_us, "env_steps_per_launch": B},
            "event_ms_timed_region": ev_ms,
        }
        if not args.no_cpu_baseline and world == 1:
            procs = os.cpu_count() or 1
            agg, single, wall = cpu_baseline(args.scenario, kw, args.cpu_seconds, procs)
            out["cpu_baseline"] = {"value": agg, "unit": "env-steps/s", "cores": procs, "kind": "port",
                                   "sample": "oracle/mpe_loop.py (per-object fp64 loop, same algorithm and cost "
                                             "structure as the reference), %d processes x %.0f s, reset every 25 "
                                             "steps; single process: %.0f env-steps/s" % (procs, args.cpu_seconds, single),
                                   "single_core": single}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
