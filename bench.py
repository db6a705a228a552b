#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused particle-world step on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric config): simple_spread, 3 agents / 3 landmarks, 65536 worlds PER GPU
(weak scaling: rank r owns worlds [r*B, (r+1)*B); no collective on the step path), fp32,
uniform random one-hot moves, device-side reset every 25 steps (MADDPG episode length).
One "step" = every world of the batch advanced once with all agents' obs/reward/done written.

Timed region: barrier + synchronize, K steps, synchronize + barrier; max over ranks; rank 0 prints
ONE JSON line.  Modes:
  graph  (default, `value`)  K `mpe_step` launches (+ resets) replayed from a HIP graph; every launch
         reads its one-hot action tensor from HBM (pool of pre-generated tensors) and writes all outputs
  eager  the same launches issued from Python through the C ABI
  api    through MultiAgentEnv.step()/reset() (the drop-in API, Python in the loop)
  fused  `mpe_rollout_random`: one launch per 25-step episode, state in registers, moves drawn in-kernel,
         every step's outputs written to its own trajectory block (reported under "extra" by default)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def algorithmic_bytes(A, L, obs_total, n_choices=0, comm_floats=0):
    """Compulsory HBM bytes per env-step: read agent pos+vel, landmark pos, one-hot
    actions (+ the per-world goal index where the scenario has one); write agent pos+vel, obs, reward
    (fp32) + done (1 byte per agent)."""
    reads = 4 * A + 2 * L + 5 * A + n_choices + comm_floats
    writes = 4 * A + obs_total + A
    return 4 * (reads + writes) + A


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json; collected and corrected as MI355X_MICROARCH.md prescribes: separate
    FETCH_SIZE / WRITE_SIZE passes, KiB units, FETCH_SIZE x2 on gfx950).  bench.py cannot run the
    profiler around itself, so this is the profile of the same command at the same sizes, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(key)
    except Exception:
        return None


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)


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


def cpu_baseline(scenario, okw, seconds, procs):
    """oracle/mpe_loop.py -- the reference's per-object Python/NumPy loop restated (the reference
    tree itself cannot travel to the GPU box) -- timed on the host cores: 1 process, then `procs`."""
    import multiprocessing as mp
    single = _cpu_worker((scenario, okw, min(seconds, 4.0), 0))
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(scenario, okw, seconds, i + 1) for i in range(procs)])
    steps = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return steps / wall, single[0] / single[1]


def cpu_baseline_c(scenario, okw, seconds, threads):
    """oracle/mpe_oracle.c -- the same algorithm in plain C (gcc -O2, OpenMP: one independent world per
    thread): what compiled host code does with it.  Returns (aggregate, single-thread) env-steps/s or None."""
    try:
        from oracle import build_c, spec as ospec
        spec = ospec.by_name(scenario, **okw)
        return build_c.bench(spec, seconds, threads), build_c.bench(spec, min(seconds, 2.0), 1)
    except Exception as e:  # the C restatement covers simple / simple_spread / simple_tag
        return None


def bench_generic(args, env, dev, rank, world, sharding):
    """Scenarios without a fused kernel: MultiAgentEnv.step() from Python -- torch
    _set_action, `mpe_world_step` (HIP), the scenario's torch observation/reward callbacks.  Host- and
    launch-bound by construction; reported as throughput only (no roofline claim)."""
    import torch
    B, K, W, EP = args.batch, args.steps, args.warmup, args.episode_len
    g = torch.Generator(device="cpu").manual_seed(args.seed + rank)
    pool = []
    for _ in range(4):   # uniform random one-hot moves / utterances per agent, in the action space's own format
        acts = []
        for agent in env.agents:
            parts = []
            if agent.movable:
                parts.append(torch.nn.functional.one_hot(torch.randint(0, 5, (B,), generator=g), 5).float())
            if not agent.silent:
                parts.append(torch.nn.functional.one_hot(torch.randint(0, env.world.dim_c, (B,), generator=g),
                                                          env.world.dim_c).float())
            acts.append(torch.cat(parts, dim=1).to(dev))
        pool.append(acts)

    def run(n):
        for k in range(n):
            if EP and k % EP == 0:
                env.reset()
            env.step(pool[k % len(pool)])
    run(W)
    walls = []
    for _ in range(args.repeats):
        sharding.barrier(dev)
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        sharding.barrier(dev)
    dt = sharding.reduce_max(sorted(walls)[len(walls) // 2], dev)
    # the same step captured once into a HIP graph (GraphedStep): one replay per step instead of ~100 launches
    from multiagent_particle_envs_amd import GraphedStep
    gs = GraphedStep(env, pool[0])

    def run_graphed(n):
        for k in range(n):
            if EP and k % EP == 0:
                env.reset()
            gs.step(pool[k % len(pool)])
    run_graphed(W)
    walls = []
    for _ in range(args.repeats):
        sharding.barrier(dev)
        t0 = time.perf_counter()
        run_graphed(K)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        sharding.barrier(dev)
    dtg = sharding.reduce_max(sorted(walls)[len(walls) // 2], dev)
    if rank == 0:
        A, Lm = len(env.world.agents), len(env.world.landmarks)
        print(json.dumps({
            "metric": "env steps/sec (whole node), %s N=%d, batch=%d per GPU" % (args.scenario, A, B),
            "value": B * K * world / dt, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s A=%d L=%d, %d worlds/GPU, generic path (torch callbacks + mpe_world_step), "
                                   "random one-hot actions, reset every %d steps" % (args.scenario, A, Lm, B, EP),
                       "batch_per_gpu": B, "global_batch": B * world, "mode": "api-generic", "repeats": args.repeats},
            "extra": {"graphed_step": {"what": "GraphedStep: the same env.step captured into a HIP graph, replayed per step "
                                               "(actions copied into the graph's static inputs every step)",
                                       "value": B * K * world / dtg, "unit": "env-steps/s", "ms_per_step": dtg * 1e3 / K}},
            "roofline": {"bound": "host", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                         "traffic": None, "note": "no fused kernel for this scenario: the step is ~100 small torch "
                                                  "launches + one HIP physics launch, bound by the Python host"},
        }))
    return None


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
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (fused-rollout) measurement")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--generic", action="store_true",
                    help="step through the generic path (torch callbacks + mpe_world_step) although a fused kernel exists")
    ap.add_argument("--streams", type=int, default=1,
                    help="cut the per-GPU batch into this many independent sub-batches, one HIP stream each")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the step path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node == --gpus"

    import multiagent_particle_envs_amd as mpe
    from multiagent_particle_envs_amd import sharding
    from multiagent_particle_envs_amd.rollout import RandomRollout, StreamedRollout, Trajectory
    kw, okw = {}, {}
    if args.scenario == "simple_spread" and args.agents != 3:
        kw["num_agents"] = args.agents
        okw["n"] = args.agents
    B, K, W, EP = args.batch, args.steps, args.warmup, args.episode_len
    S = max(1, args.streams)
    assert B % S == 0, "--batch must be a multiple of --streams"
    envs = []
    for s_ in range(S):                        # S sub-batches of B/S worlds, one HIP stream each
        e = mpe.make_env(args.scenario, batch_size=B // S, seed=args.seed, fused=False if args.generic else None, **kw)
        e.world.world_offset = rank * B + s_ * (B // S)   # global world numbering: no shared RNG streams
        envs.append(e)
    env = envs[0]
    A, Lm = len(env.world.agents), len(env.world.landmarks)
    if not env.fused:
        return bench_generic(args, env, dev, rank, world, sharding)
    rolls = [RandomRollout(e, episode_len=EP, pool=16) for e in envs]
    roll = StreamedRollout(rolls)
    obs_total = int(env._obs_off[-1])
    speakers = sum(1 for a in env.world.agents if not a.silent)
    bytes_step = algorithmic_bytes(A, Lm, obs_total, len(env.world.choice_pops), speakers * env.world.dim_c)
    can_fuse = A <= 6 or args.scenario == "simple_spread"
    trajs = None

    def fused_steps(n):
        nonlocal trajs
        T = EP if EP else 25
        if trajs is None:
            trajs = [Trajectory(e, T) for e in envs]
        done = 0
        while done < n:
            k = min(T, n - done)
            roll.fused(k, trajs)
            done += k

    def make_body(mode, n):
        if mode == "graph":
            g = roll.capture(n)
            return g.replay
        if mode == "eager":
            return lambda: roll.enqueue(n)
        if mode == "fused":
            return lambda: fused_steps(n)

        def api():
            assert S == 1, "--mode api drives one env"
            for k in range(n):
                if EP and k % EP == 0:
                    env.reset()
                env.step(rolls[0].pool[k % len(rolls[0].pool)])
        return api

    def timed(mode):
        """median over repeats of the wall time of exactly K steps, max over ranks"""
        body = make_body(mode, K)
        if mode == "fused":
            fused_steps(W)
        else:
            roll.enqueue(W)
        sharding.barrier(dev)
        walls = []
        evs = []
        for _ in range(args.repeats):
            sharding.barrier(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            body()
            e1.record()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)   # this rank's K steps, from the common start to its own completion
            sharding.barrier(dev)                      # (the MAX over ranks below is the job's time)
            evs.append(e0.elapsed_time(e1))
        if mode == args.mode:
            region_ms[:] = evs
        return sharding.reduce_max(sorted(walls)[len(walls) // 2], dev)

    region_ms = []      # HIP-event time of each timed repeat (events on the launch stream)

    def kernel_time_us(mode, n=400):
        """The dominant kernel's time per env step, from HIP events on the launch stream around n
        back-to-back steps with no resets in between (graph replay / fused launches)."""
        roll.set_episode_len(0)
        try:
            body = make_body("fused" if mode == "fused" else "graph", n)
            body()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            body()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        finally:
            roll.set_episode_len(EP)

    def copy_ceiling_gbs():
        """Device-to-device copy of 256 MiB (read + write counted), the practical streaming ceiling of this box."""
        n = 64 * 1024 * 1024
        a = torch.empty(n, dtype=torch.float32, device=dev)
        b_ = torch.empty_like(a)
        a.fill_(1.0)
        b_.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b_.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        return 10 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    dt = timed(args.mode)
    k_us = kernel_time_us(args.mode)
    extra = {}
    if not args.no_extra and can_fuse and args.mode != "fused":
        dtf = timed("fused")
        kf = kernel_time_us("fused")
        extra["fused_rollout"] = {
            "what": "mpe_rollout_random: one launch per %d-step episode, state kept on chip (registers / LDS), moves drawn "
                    "in-kernel, every step's obs/rew/done written to its own trajectory block" % (EP or 25),
            "value": B * K * world / dtf, "unit": "env-steps/s", "ms_per_step": dtf * 1e3 / K,
            "kernel_us_per_step": kf,
            "achieved_GBps_at_411B_convention": bytes_step * B / (kf * 1e-6) / 1e9,
            "compulsory_bytes_per_env_step": 4 * (obs_total + A) + A,
            "achieved_GBps_compulsory": (4 * (obs_total + A) + A) * B / (kf * 1e-6) / 1e9}

    if not args.no_extra and args.mode == "graph" and S == 1:
        # SURVEY 8d's "regenerate every step": `value` steps on moves already resident in HBM (the policy's output);
        # here each step also draws its fresh moves on the device first
        for how in ("inline",):
            rr = RandomRollout(env, episode_len=EP, pool=16, regenerate=True)
            gr = rr.capture(K)
            gr.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                sharding.barrier(dev)
                t0 = time.perf_counter()
                gr.replay()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            dtr = sharding.reduce_max(sorted(ts)[1], dev)
            extra["moves_regenerated_every_step"] = {
                "what": "graph of K x (mpe_random_actions -> mpe_step), resets every %d steps" % EP,
                "value": B * K * world / dtr, "unit": "env-steps/s", "ms_per_step": dtr * 1e3 / K}

    if rank == 0:
        achieved = bytes_step * B / (k_us * 1e-6) / 1e9
        kname = ("mpe::k_split" if os.environ.get("MPE_STEP_IMPL") != "thread" else "mpe::k_narrow") if A <= 6 \
            else ("mpe::k_multi" if max(A, Lm) <= 32 and A + Lm <= 64 else "mpe::k_wave")
        tkey = "%s_A%d_L%d_B%d" % (args.scenario, A, Lm, B)
        tr = pmc_traffic(tkey) if S == 1 and args.mode in ("graph", "eager") else None
        copy_gbs = copy_ceiling_gbs()
        out = {
            "metric": "env steps/sec (whole node), %s N=%d, batch=%d per GPU" % (args.scenario, A, B),
            "value": B * K * world / dt, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s A=%d L=%d, %d worlds/GPU, one-hot random moves in HBM, reset every %d steps"
                                   % (args.scenario, A, Lm, B, EP),
                       "batch_per_gpu": B, "global_batch": B * world, "mode": args.mode,
                       "repeats": args.repeats, "streams_per_gpu": S,
                       "sharding": "worlds by batch index, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": tr["traffic_bytes_per_launch"] if tr else None,
                         "traffic_source": tr["source"] if tr else None,
                         "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs,
                         "algorithmic_bytes_per_env_step": bytes_step,
                         "algorithmic_bytes_per_launch": bytes_step * B,
                         "kernel": kname,
                         "kernel_us_per_launch": k_us, "env_steps_per_launch": B,
                         "timed_region_us_per_step": (sorted(region_ms)[len(region_ms) // 2] * 1e3 / K) if region_ms else None,
                         "note": "achieved = algorithmic bytes per launch / kernel_us_per_launch; kernel_us_per_launch = HIP-event "
                                 "time (launch stream) of 400 back-to-back dependent step launches / 400 (agrees with the rocprofv3 "
                                 "kernel-trace average under profiles/); timed_region_us_per_step = HIP-event time of the timed "
                                 "region / steps (includes the reset launches every episode); with S>1 streams the S sub-batch "
                                 "launches of one step overlap, so both are times per full-batch step"},
        }
        if extra:
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            procs = usable_cores()
            agg, single = cpu_baseline(args.scenario, okw, args.cpu_seconds, procs)
            out["cpu_baseline"] = {
                "value": agg, "unit": "env-steps/s", "cores": procs, "kind": "port",
                "sample": "oracle/mpe_loop.py (the reference's per-object fp64 Python/NumPy loop restated; "
                          "/root/reference is absent on the GPU box), %d processes x %.0f s each, same move "
                          "distribution, reset every 25 steps; 1 process alone: %.0f env-steps/s"
                          % (procs, args.cpu_seconds, single),
                "single_core": single}
            cp = cpu_baseline_c(args.scenario, okw, min(args.cpu_seconds, 4.0), procs)
            if cp:
                out["cpu_baseline"]["c_port"] = {
                    "value": cp[0], "unit": "env-steps/s", "cores": procs, "single_core": cp[1],
                    "sample": "oracle/mpe_oracle.c (the same algorithm in plain C, gcc -O2 -fopenmp, one world per "
                              "thread, fp64), %d threads x %.0f s" % (procs, min(args.cpu_seconds, 4.0))}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
