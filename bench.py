#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused particle-world step on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # no launcher: bench.py starts the N ranks itself (one per GPU)

Headline workload (BASELINE.json metric config, C5): simple_spread, 3 agents / 3 landmarks, 65536 worlds PER GPU
(weak scaling: rank r owns worlds [r*B, (r+1)*B); no collective on the step path), fp32.
Protocol: uniform random one-hot moves `[A][B][5]` that are FRESH for every step -- one
`mpe_random_actions_block` launch per 25-step episode draws the episode's moves, inside the timed region --, a
device-side `mpe_reset` every 25 steps (MADDPG episode length), one `mpe_step` launch per step that reads its moves
from HBM and writes every agent's obs / reward / done.  One "step" = every world of the batch advanced once.

Timed region: barrier + synchronize, a HIP graph of up to 8000 CONSECUTIVE steps replayed back to back until the region
holds >= 2 s of GPU work (whatever --steps is: `--steps 20` and `--steps 1000` time the same graph), synchronize +
barrier; median of 5 repeats, max over ranks; `value` = B * timed steps * ranks / that time.  Rank 0 prints ONE JSON line
-- COMPACT (< 4096 bytes: the driver parses it; round 5's 20 KB line came back unparsed) -- and writes the full record
(every leg, raw timings, per-rank records, the box) to --full-json (default gpurun_out/bench_full.json).

Round 6: the SAME protocol is measured twice under the same bracket -- its steps LAUNCHED (one mpe_step per step, as
above) and COMMANDED to the step server (include/mpe_hip.h: mpe_step_server_*; rollout.ServedRollout: per episode one
block draw and one doorbell launch behind it on the commanding stream, the server's resident launch on a stream of its
own; fresh moves read from HBM by every step, in-launch resets, every step's rows / rewards / dones / state written
through to its own block) -- and the line's value is the faster (`config.mode`; --commands launched | served | auto);
the other stays in the line (`roofline.launched`).

N > 1 never measures fewer GPUs than asked for: with WORLD_SIZE unset `--gpus N` starts N ranks itself and fails when
the node has fewer GPUs or a rank fails; with a launcher, WORLD_SIZE must equal --gpus.  The barrier travels over RCCL
when every rank can bring it up and over gloo otherwise (`config.barrier_backend`; the step path has no collective, so
the measured work is the same); `config.ranks` lists each rank's GPU, its own rate and kernel time; `roofline` is per
GPU (slowest rank's kernel) and `cpu_baseline` is emitted on every line.

Besides the headline the line carries (N=1 only):
  roofline             the step kernel's HIP-event time over back-to-back launches -> algorithmic GB/s vs 8 TB/s.
                       At B=65536 the working set (27 MB/launch + the 98 MB move pool) sits in the 256 MiB Infinity
                       Cache: the limit there is launch + latency, and the line says so.
  extra.hbm_resident   the same kernel and protocol at B=1048576 (431 MB per launch: beyond the Infinity Cache)
  extra.configs        BASELINE.json's other single-GPU configs: C2 spread N=3 B=4096, C3 simple_tag B=16384,
                       C4 spread N=64 B=4096 -- each measured in a process of its own (as a user of that config would),
                       with its own roofline entry and min / median / max over 5 repeats
  extra.fused_rollout  `mpe_rollout_random`: one launch per episode, state on chip, moves drawn in-kernel
  extra.moves_resident the round-1 headline: moves read from a resident ring that is never redrawn
  extra.box            which GPU / clocks / power cap / partition modes / driver this line was measured on
  cpu_baseline         oracle/mpe_loop.py on the host cores (+ the C port), timed AFTER the GPU legs, with the unmodified
                       reference's own numbers from the build container (profiles/cpu_reference.json) quoted beside it
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~5.3-6.3 TB/s is the measured copy ceiling
L3_BYTES = 256 * 1024 * 1024   # Infinity Cache
MIN_REGION_MS = 2000.0      # the headline's timed region: >= 2 s of back-to-back GPU work per repeat
SIDE_REGION_MS = 300.0      # the secondary legs'
MAX_GRAPH_STEPS = 8000      # consecutive steps captured into one HIP graph (replayed to fill the region)


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
    except Exception:  # the C restatement covers simple / simple_spread / simple_tag
        return None


def cpu_reference_record(key):
    """The unmodified reference timed in the build container (tools/time_reference.py -> profiles/cpu_reference.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "cpu_reference.json")) as f:
            d = json.load(f)
        row = d["configs"][key]
        return {"source": "profiles/cpu_reference.json (tools/time_reference.py: the unmodified /root/reference, build container)",
                "cpu_model": d["cpu_model"], "cores": d["usable_cores"],
                "env_steps_per_s_1_process": row["reference"]["env_steps_per_s_1_process"],
                "env_steps_per_s_all_cores": row["reference"]["env_steps_per_s_all_cores"],
                "port_over_reference_1_process": row["port_over_reference_1_process"],
                "port_over_reference_all_cores": row["port_over_reference_all_cores"]}
    except Exception:
        return None


def bench_generic(args, env, dev, rank, world, rv):
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
        rv.barrier()
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        rv.barrier()
    dt = rv.reduce_max(sorted(walls)[len(walls) // 2])
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
        rv.barrier()
        t0 = time.perf_counter()
        run_graphed(K)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        rv.barrier()
    dtg = rv.reduce_max(sorted(walls)[len(walls) // 2])
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


class Leg(object):
    """One workload (scenario, sizes, batch) on this rank's GPU: env(s), rollout driver, measurements."""

    def __init__(self, mpe, scenario, agents, B, EP, rank, streams=1, seed=0, generic=False, scenario_kw=None):
        from multiagent_particle_envs_amd.rollout import RandomRollout, StreamedRollout
        self.kw, self.okw = {}, {}
        if scenario == "simple_spread" and agents != 3:
            self.kw["num_agents"] = agents
            self.okw["n"] = agents
        self.kw.update(scenario_kw or {})               # (tools/ab_kernels.py: team sizes of the other scenarios)
        self.scenario, self.B, self.EP, self.S = scenario, B, EP, max(1, streams)
        assert B % self.S == 0, "--batch must be a multiple of --streams"
        self.envs = []
        for s_ in range(self.S):                        # S sub-batches of B/S worlds, one HIP stream each
            e = mpe.make_env(scenario, batch_size=B // self.S, seed=seed, fused=False if generic else None, **self.kw)
            e.world.world_offset = rank * B + s_ * (B // self.S)   # global world numbering: no shared RNG streams
            self.envs.append(e)
        self.env = self.envs[0]
        self.A, self.Lm = len(self.env.world.agents), len(self.env.world.landmarks)
        self._RR, self._SR = RandomRollout, StreamedRollout
        self.rolls = {}
        self.trajs = None

    def roll(self, protocol):
        """protocol 'fresh': every step consumes moves nobody used before (block redraw per episode, in the timed
        region); 'resident': a ring of 16 move tensors drawn once."""
        if protocol not in self.rolls:
            fresh, ids = protocol.startswith("fresh"), protocol.endswith("_ids")   # "..._ids": int32 move ids instead of one-hot rows
            P = (self.EP or 16) if fresh else 16
            rs = [self._RR(e, episode_len=self.EP, pool=P, regenerate=fresh, action_ids=ids) for e in self.envs]
            self.rolls[protocol] = self._SR(rs)
        return self.rolls[protocol]

    def geometry(self):
        env = self.env
        obs_total = int(env._obs_off[-1])
        speakers = sum(1 for a in env.world.agents if not a.silent)
        bytes_step = algorithmic_bytes(self.A, self.Lm, obs_total, len(env.world.choice_pops), speakers * env.world.dim_c)
        compulsory_roll = 4 * (obs_total + self.A) + self.A    # a fused rollout keeps state on chip and draws moves in-kernel
        A, Lm = self.A, self.Lm
        # which kernel family mpe_step dispatches this shape to (csrc/mpe_abi.hip: the k_split table, else mpe_wide.hip)
        spread = self.scenario == "simple_spread"
        if (spread and A == Lm and A <= 6) or (self.scenario == "simple_tag" and (A, Lm) in ((4, 2), (2, 1), (6, 3))) or \
                self.scenario not in ("simple_spread", "simple_tag"):
            kname = "mpe::k_split"
        elif spread and max(A, Lm) <= 32 and A + Lm <= 64:
            kname = "mpe::k_multi"
        elif spread and max(A, Lm) <= 64:
            kname = "mpe::k_duo<4>"
        else:
            kname = "mpe::k_wave"
        return obs_total, bytes_step, compulsory_roll, kname

    def fused_steps(self, roll, n):
        from multiagent_particle_envs_amd.rollout import Trajectory
        T = self.EP if self.EP else 25
        if self.trajs is None:
            self.trajs = [Trajectory(e, T) for e in self.envs]
        done = 0
        while done < n:
            k = min(T, n - done)
            roll.fused(k, self.trajs)
            done += k

    def body(self, mode, protocol, n):
        roll = self.roll(protocol)
        if mode == "graph":
            return roll.capture(n).replay
        if mode == "eager":
            return lambda: roll.enqueue(n)
        if mode == "fused":
            return lambda: self.fused_steps(roll, n)
        env, r0, EP = self.env, roll.rollouts[0], self.EP
        assert self.S == 1, "--mode api drives one env"

        import torch
        if r0.pool_c is None:
            acts = [r0.pool[p] for p in range(len(r0.pool))]            # one [A, B, 5] tensor per step
        else:   # communication scenarios: agent i's row = [move (5) if it moves] + [word (dim_c) if it speaks]
            acts = [[torch.cat(([r0.pool[p][i]] if a.movable else []) + ([r0.pool_c[p][i]] if not a.silent else []), dim=1)
                     for i, a in enumerate(env.agents)] for p in range(len(r0.pool))]

        def api():
            for k in range(n):
                if EP and k % EP == 0:
                    env.reset()
                env.step(acts[k % len(acts)])      # (the first block's moves, cycled)
        if mode == "api":
            return api

        # mode "host": a caller whose policy lives on the host, as the reference's callers do -- one-hot moves arrive in
        # (pinned) host memory, observations and rewards go back to host memory, and the caller waits for them every step
        assert r0.pool_c is None, "--mode host: scenarios without a communication action"
        h_acts = [a.cpu().pin_memory() for a in acts[:min(len(acts), 4)]]
        d_act = torch.empty_like(acts[0])
        obs0 = env.reset()
        h_obs = [torch.empty(o.shape, dtype=o.dtype).pin_memory() for o in obs0]
        h_rew = [torch.empty(o.shape[:1], dtype=torch.float32).pin_memory() for o in obs0]

        def host():
            for k in range(n):
                if EP and k % EP == 0:
                    for h, o in zip(h_obs, env.reset()):
                        h.copy_(o, non_blocking=True)
                d_act.copy_(h_acts[k % len(h_acts)], non_blocking=True)
                obs, rew, _, _ = env.step(d_act)
                for h, o in zip(h_obs, obs):
                    h.copy_(o, non_blocking=True)
                for h, r in zip(h_rew, rew):
                    h.copy_(r, non_blocking=True)
                torch.cuda.synchronize()
        return host

    def timed(self, torch, rv, dev, mode, protocol, K, W, repeats, region_ms=None):
        """-> (seconds for the timed steps: median over repeats, max over ranks; R = timed steps / K; HIP-event ms of
        the median repeat; [min, median, max] env-steps/s of this rank over the repeats).
        The timed region is >= region_ms of back-to-back GPU work: a HIP graph of G = K*r1 CONSECUTIVE steps (resets
        and move draws fall every episode_len steps of the long run, whatever K is; G <= MAX_GRAPH_STEPS) replayed
        `reps` times with nothing in between."""
        region_ms = MIN_REGION_MS if region_ms is None else region_ms
        roll = self.roll(protocol)
        body = self.body(mode, protocol, K)
        if mode == "fused":
            self.fused_steps(roll, W)
        else:
            roll.enqueue(W)
        body()
        torch.cuda.synchronize()
        t0 = time.perf_counter()     # size the region
        body()
        torch.cuda.synchronize()
        once = max(time.perf_counter() - t0, 1e-6)
        R = max(1, int(math.ceil(region_ms * 1e-3 / once))) if mode in ("graph", "fused") else 1
        R = int(rv.reduce_max(R))
        r1 = max(1, min(R, MAX_GRAPH_STEPS // max(K, 1)))
        reps = int(math.ceil(R / float(r1)))
        R = r1 * reps
        if r1 > 1:
            body = self.body(mode, protocol, K * r1)
            body()
            torch.cuda.synchronize()
        if mode in ("graph", "fused"):   # the real body, timed once more: as many replays as fill the region
            t0 = time.perf_counter()
            body()
            torch.cuda.synchronize()
            reps = int(rv.reduce_max(max(1, int(math.ceil(region_ms * 1e-3 / max(time.perf_counter() - t0, 1e-6))))))
            R = r1 * reps
        walls, evs = [], []
        for _ in range(repeats):
            rv.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _r in range(reps):
                body()
            e1.record()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)   # this rank's K*R steps, from the common start to its own completion
            rv.barrier()                               # (the MAX over ranks below is the job's time)
            evs.append(e0.elapsed_time(e1))
        order = sorted(range(len(walls)), key=lambda i: walls[i])
        med = order[len(order) // 2]
        rate = [self.B * K * R / walls[i] for i in (order[-1], med, order[0])]
        return rv.reduce_max(walls[med]), R, evs[med], rate

    def kernel_time_us(self, torch, mode, n=400, protocol="resident"):
        """The dominant kernel's time per env step, from HIP events on the launch stream around back-to-back dependent
        launches with nothing else in between (no resets, no redraws): graph replay / fused launches.  Two bodies of n
        and 2n launches are timed (best of 3 each) and the SLOPE (t_2n - t_n) / n is returned: what one more launch costs.
        A single body's average also carries the replay's fixed cost (graph launch, first dispatch) -- 0.1-0.15 ms per
        replay, i.e. +0.3-0.6 us on a 3 us kernel at n = 200, which is what made C2 / C3 read 15 % slower inside the
        default run (n = 200) than in a process of their own (n = 400) in round 2 and early round 3."""
        roll = self.roll(protocol)
        roll.set_episode_len(0)
        try:
            def best(m):
                body = self.body("fused" if mode == "fused" else "graph", protocol, m)
                body()
                torch.cuda.synchronize()
                t = None
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    body()
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1)
                    t = ms if t is None else min(t, ms)
                return t
            # The slope is only as good as both points: a body that comes out slow three times in a row (seen once: the
            # n-launch graph of the int-ids leg at 6.9 us per launch next to 5.2 for the 2n one, which read as a 3.5 us kernel)
            # shows as an implausible fixed cost t_n - n * slope.  A replay's fixed cost is 0.03-0.15 ms: outside
            # [-0.05, 0.35] ms (or 2 % of the body, for long kernels) the pair is measured again (3 attempts), and failing that the 2n body's plain average -- an
            # upper bound on the slope -- is reported, flagged.
            tries = 0
            while True:
                t1, t2 = best(n), best(2 * n)
                tries += 1
                us, fixed_ms, ok = two_point_slope_us(t1, t2, n)
                if ok or tries == 3:
                    break
            self.last_kernel_timing = {"launches": [n, 2 * n], "ms": [t1, t2], "avg_us_single_body": t2 * 1e3 / (2 * n),
                                       "replay_fixed_cost_ms": fixed_ms, "attempts": tries}
            if not ok:
                self.last_kernel_timing["fallback"] = "two-point slope implausible in 3 attempts: the 2n body's average is reported"
            return us
        finally:
            roll.set_episode_len(self.EP)

    def release(self):
        self.rolls.clear()
        self.trajs = None
        self.envs = []
        self.env = None


def two_point_slope_us(t1_ms, t2_ms, n):
    """(us per launch, implied fixed cost of a replay in ms, plausible?) from the times of an n-launch and a 2n-launch body.
    Plausible: the fixed cost t_n - n * slope lies in [-0.05, 0.35] ms, or within 2 % of the body for long kernels (measurement
    noise).  When it does not, the 2n body's plain average -- an upper bound on the slope -- is returned instead of the slope."""
    slope_ms = (t2_ms - t1_ms) / n
    fixed_ms = t1_ms - slope_ms * n
    lo, hi = -max(0.05, 0.02 * t1_ms), max(0.35, 0.02 * t1_ms)
    ok = lo <= fixed_ms <= hi
    return (slope_ms * 1e3 if ok else t2_ms * 1e3 / (2 * n)), fixed_ms, ok


def _abi_action_dim():
    from multiagent_particle_envs_amd import _abi
    return _abi.MPE_ACTION_DIM


def launch_floor_us(torch, dev, n=400):
    """Empty-ish dependent launch: `mpe_episode_tick` on 64 worlds; graphs of n and 2n launches, slope per launch."""
    from multiagent_particle_envs_amd import _abi
    L = _abi.lib()
    cnt = torch.zeros(64, dtype=torch.int32, device=dev)
    done = torch.zeros((1, 64), dtype=torch.bool, device=dev)

    def timed(m):
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            L.mpe_episode_tick(cnt.data_ptr(), done.data_ptr(), 1, 64, 0, 0, _abi.raw_stream(dev))
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(m):
                    L.mpe_episode_tick(cnt.data_ptr(), done.data_ptr(), 1, 64, 0, 0, _abi.raw_stream(dev))
        torch.cuda.current_stream(dev).wait_stream(s)
        g.replay()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        return best
    return (timed(2 * n) - timed(n)) * 1e3 / n


def copy_ceiling_gbs(torch, dev):
    """Device-to-device copy of 1 GiB (read + write counted; beyond the Infinity Cache): the streaming ceiling of this box."""
    n = 256 * 1024 * 1024
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b_ = torch.empty_like(a)
    a.fill_(1.0)
    b_.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b_.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def fill_ceiling_gbs(torch, dev):
    """Linear fill of 1 GiB (writes only): what an output-dominated launch (N = 64: 96 % of its bytes are rows) can stream at."""
    a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
    a.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        a.fill_(2.0)
    e1.record()
    torch.cuda.synchronize()
    return 5 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def roofline_entry(leg, k_us, B, mode, floor_us, timing=None):
    obs_total, bytes_step, compulsory_roll, kname = leg.geometry()
    per_launch = bytes_step * B
    achieved = per_launch / (k_us * 1e-6) / 1e9
    tr = pmc_traffic("%s_A%d_L%d_B%d" % (leg.scenario, leg.A, leg.Lm, B)) if mode in ("graph", "eager") else None
    moves_pool = 25 * leg.A * B * 20
    resident = per_launch + moves_pool < L3_BYTES
    kt = (tr or {}).get("kernel_trace")
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": tr["traffic_bytes_per_launch"] if tr else None,
            "traffic_source": tr["source"] if tr else None,
            # the committed `rocprofv3 --kernel-trace --stats` durations of the same launch (profiles/), beside the live slope
            "kernel_us_rocprof": {"mean": kt["mean_us"], "median": kt["median_us"], "source": kt["source"]} if kt else None,
            "frac_at_rocprof_mean": per_launch / (kt["mean_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS if kt else None,
            "regime_label": "l3+launch" if resident else "hbm",
            "regime": ("working set (%.0f MB per launch + %.0f MB of moves) fits the 256 MiB Infinity Cache: the PMC "
                       "'traffic' is fabric requests, largely L3 hits; the launch is launch/latency-limited, not HBM-limited"
                       % (per_launch / 1e6, moves_pool / 1e6)) if resident else
                      ("working set (%.0f MB per launch) exceeds the 256 MiB Infinity Cache: HBM-resident" % (per_launch / 1e6)),
            "l3_resident": resident,
            "algorithmic_bytes_per_env_step": bytes_step, "algorithmic_bytes_per_launch": per_launch,
            "kernel": kname, "kernel_us_per_launch": k_us, "env_steps_per_launch": B,
            "kernel_timing": timing,
            "launch_floor_us": floor_us}


def served_timed(torch, rv, roll, B, region_ms, repeats):
    """Timed region of a (warmed-up) ServedRollout, with the contract's bracket (barrier + synchronize both sides, median of
    `repeats`, MAX over ranks): -> dict(dt, n, rate [min, median, max of this rank], launch_us, steps_per_launch).
    Every rank makes the SAME sequence of rendezvous calls whatever happens to it: an error is kept and raised at the end."""
    G = 2 * roll.EP * 4
    err = []

    def run(n):
        if err:
            return
        try:
            roll.enqueue(n)
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001 (kept; the rendezvous sequence goes on)
            err.append(e)
    t0 = time.perf_counter()
    run(G)
    reps = max(1, int(math.ceil(region_ms * 1e-3 / max(time.perf_counter() - t0, 1e-6))))
    t0 = time.perf_counter()      # (a second look at the region's own size: a short enqueue over-states the per-step time)
    run(G * reps)
    reps = int(rv.reduce_max(max(1, int(math.ceil(1.03 * reps * region_ms * 1e-3 / max(time.perf_counter() - t0, 1e-6))))))
    n = G * reps
    walls, launches = [], []
    for _ in range(repeats):
        roll.srv.launch_events = []
        rv.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        walls.append(time.perf_counter() - t0)
        rv.barrier()
        try:
            launches.append([(e0.elapsed_time(e1) * 1e3, T) for e0, e1, T in roll.srv.launch_events])
        except Exception as e:      # noqa: BLE001
            err.append(e)
            launches.append([(float("nan"), 1)])
        roll.srv.launch_events = None
    order = sorted(range(len(walls)), key=lambda i: walls[i])
    med = order[len(order) // 2]
    dt = rv.reduce_max(walls[med])
    if err:
        raise err[0]
    roll.srv.check()
    ev = launches[med]
    return {"dt": dt, "n": n, "rate": [B * n / walls[order[-1]], B * n / walls[med], B * n / walls[order[0]]],
            "launch_us": sum(u for u, _ in ev) / len(ev), "steps_per_launch": sum(T for _, T in ev) / float(len(ev)),
            "launches_in_region": len(ev)}


def served_leg(torch, mpe, scenario, agents, B, EP, seed, region_ms, repeats=3, kw=None, per_step_doorbells=True, rv=None):
    """The headline protocol through the STEP SERVER (rollout.ServedRollout; include/mpe_hip.h: mpe_step_server_*): fresh moves
    for every step (one block draw per episode), a reset every EP steps (the server's in-launch reset), every step's rows /
    rewards / dones / state written, every step's moves read from HBM -- but the steps COMMANDED to one resident launch instead of
    launched.  `value`: an episode's steps commanded by ONE doorbell behind the draw of their moves (they all exist then): the
    server's own rate.  `per_step_doorbells`: the same with a doorbell LAUNCH per step -- bounded by the command processor's
    dependent-launch rate (1.7-2 us each), and erratic: some stream pairings process one doorbell per ~33 us."""
    from multiagent_particle_envs_amd.rollout import ServedRollout
    from multiagent_particle_envs_amd import sharding
    rv = rv if rv is not None else sharding.Rendezvous(0, 1, torch.device("cuda", torch.cuda.current_device()))

    def run(ahead):
        env = mpe.make_env(scenario, batch_size=B, seed=seed, **(kw or {}))
        A, Lm = len(env.world.agents), len(env.world.landmarks)
        bytes_step = algorithmic_bytes(A, Lm, int(env._obs_off[-1]), len(env.world.choice_pops), 0)
        roll = ServedRollout(env, episode_len=EP, timeout_s=4.0, graphs=True, ring_ahead=ahead)
        roll.enqueue(2 * EP * 4)
        torch.cuda.synchronize()
        roll.srv.check()
        r = served_timed(torch, rv, roll, B, region_ms, repeats)
        dt, n = r["dt"], r["n"]
        return {"value": B * n / dt, "unit": "env-steps/s", "ms_per_step": dt * 1e3 / n, "timed_steps": n, "timed_region_s": dt,
                "repeats": {"min": r["rate"][0], "median": r["rate"][1], "max": r["rate"][2]},
                "frac_timed_region": bytes_step * B / (dt / n) / 1e9 / HBM_PEAK_GBS,
                "achieved_GBps": bytes_step * B / (dt / n) / 1e9, "algorithmic_bytes_per_env_step": bytes_step,
                "server_launch_us": r["launch_us"], "steps_per_server_launch": r["steps_per_launch"],
                "server_stream_probe": roll.srv.stream_probe}
    ent = run(True)
    ent["what"] = ("the headline protocol with the steps COMMANDED to a resident step server (one launch per timed repeat on its own "
                   "stream) instead of launched: per %d-step episode one block draw and ONE doorbell launch behind it on the caller's "
                   "stream (replayed as a graph); fresh moves READ FROM HBM every step, in-launch resets, every step's rows / rewards "
                   "/ dones / state written to its own block with its own completion flag" % EP)
    if per_step_doorbells:
        try:
            ent["per_step_doorbells"] = run(False)
        except Exception as e:
            ent["per_step_doorbells"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            torch.cuda.synchronize()
    return ent


CONFIG_LEGS = (("C2_spread_n3_B4096", "simple_spread", 3, 4096, 200),
               ("C3_tag_B16384", "simple_tag", 3, 16384, 200),
               ("C4_spread_n64_B4096", "simple_spread", 64, 4096, 50))


def run_config_leg(key, floor_us, seed, EP, dev_index):
    """One of BASELINE.json's other single-GPU configs, measured in this (fresh) process -> the entry of extra.configs."""
    import torch
    import multiagent_particle_envs_amd as mpe
    from multiagent_particle_envs_amd import sharding
    _, scn, ag, bb, kk = [c for c in CONFIG_LEGS if c[0] == key][0]
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rv = sharding.Rendezvous(0, 1, dev)
    SR = SIDE_REGION_MS

    def stats(rate):
        return {"min": rate[0], "median": rate[1], "max": rate[2], "unit": "env-steps/s per GPU"}
    lg = Leg(mpe, scn, ag, bb, EP, 0, 1, seed)
    d1, R1, _, r1_ = lg.timed(torch, rv, dev, "graph", "fresh", kk, 10, 5, 2 * SR)
    k1 = lg.kernel_time_us(torch, "graph", n=400 if bb * ag < 100000 else 100)
    t1_ = lg.last_kernel_timing
    ent = {"value": bb * kk * R1 / d1, "unit": "env-steps/s", "ms_per_step": d1 * 1e3 / (kk * R1), "repeats": stats(r1_),
           "timed_steps": kk * R1, "placement_probe": lg.env.placement_probe, "measured_in": "a process of its own",
           "workload": "%s A=%d L=%d, %d worlds" % (scn, lg.A, lg.Lm, bb), "roofline": roofline_entry(lg, k1, bb, "graph", floor_us, t1_)}
    d2, R2, _, r2_ = lg.timed(torch, rv, dev, "fused", "resident", kk, 10, 5, SR)
    k2 = lg.kernel_time_us(torch, "fused", n=400 if bb * ag < 100000 else 100)
    comp = lg.geometry()[2]
    ent["fused_rollout"] = {"value": bb * kk * R2 / d2, "unit": "env-steps/s", "kernel_us_per_step": k2, "repeats": stats(r2_),
                            "compulsory_bytes_per_env_step": comp,
                            "frac_compulsory": comp * bb / (k2 * 1e-6) / 1e9 / HBM_PEAK_GBS}
    if lg.geometry()[3] == "mpe::k_split":      # (the step server serves the wave-per-agent shapes)
        kw = dict(lg.kw)
        lg.release()
        torch.cuda.empty_cache()
        try:
            ent["step_server"] = served_leg(torch, mpe, scn, ag, bb, EP or 25, seed, SR, kw=kw)
        except Exception as e:
            ent["step_server"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return ent


def config_leg_subprocess(key, args, floor_us, dev_index):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg-json", key, "--leg-floor-us", repr(float(floor_us)),
                        "--seed", str(args.seed), "--episode-len", str(args.episode_len), "--leg-device", str(dev_index)],
                       capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("config leg %s failed (exit %d): %s" % (key, r.returncode, r.stderr[-1500:]))
    return json.loads(lines[-1])


def gpu_identity(rec):
    """What tells two GPUs apart in a rank record: the device UUID, else the PCI address (None when neither is known)."""
    return rec.get("uuid") or rec.get("pci")


def duplicate_gpus(recs):
    """Ranks that drive the same physical GPU, as a readable string ('' when every rank has its own).  A rank whose GPU has
    no identity at all (no UUID, no PCI address) cannot be told apart and is reported too."""
    seen, bad = {}, []
    for r in recs:
        k = gpu_identity(r)
        if k is None:
            bad.append("rank %d: no UUID / PCI address" % r["rank"])
        elif k in seen:
            bad.append("ranks %d and %d on %s" % (seen[k], r["rank"], k))
        else:
            seen[k] = r["rank"]
    return "; ".join(bad)


def per_gpu_stats(recs):
    """min / median / max over ranks of each rank's own median rate (env-steps/s per GPU)."""
    v = sorted(r["env_steps_per_s"]["median"] for r in recs)
    return {"min": v[0], "median": v[len(v) // 2], "max": v[-1], "unit": "env-steps/s per GPU", "ranks": len(v)}


def user_scenario_leg(torch, mpe, B, EP, rv, dev):
    """A scenario WITHOUT a kernel of its own (examples/corral.py: 3 agents, 3 posts, a per-world gate), stepped from Python
    through the drop-in API: described by ObsSpec / RewardSpec (World.step + the rows in one launch, mpe_step_rows -- the program
    interpreted, and compiled in) and, beside it, through its torch callbacks (the generic path: ~100 launches per step)."""
    path = os.path.join(ROOT, "examples", "corral.py")
    out = {"what": "examples/corral.py (a user scenario, no kernel of its own) from Python, env.step / env.reset every %d steps, %d worlds: "
                   "`program` = obs_spec / reward_spec interpreted by mpe_step_rows (one launch per step); `compiled` = the same program "
                   "compiled in (env.compile_program(): mpe_rows_static_source -> hipcc --genco -> mpe_rows_load_image; bit-identical "
                   "results); `generic` = the same scenario's torch observation / reward callbacks over mpe_world_step" % (EP or 25, B)}
    g = torch.Generator(device="cpu").manual_seed(0)
    for key, kw, n in (("program", {"compile_program": False}, 2000), ("compiled", {"compile_program": False}, 2000),
                       ("generic", {"fused": False}, 60)):
        env = mpe.make_env(path, batch_size=B, **kw)
        if key == "compiled":
            t0 = time.perf_counter()
            ok = env.compile_program()
            out["compile_program_s"] = time.perf_counter() - t0      # (0.0x s when lib/rows_cache/ already holds the image)
            if not ok:
                raise RuntimeError("compile_program() did not activate the image")
        acts = [torch.nn.functional.one_hot(torch.randint(0, 5, (env.n, B), generator=g), 5).float().cuda() for _ in range(4)]
        if not env.fused:
            acts = [[a[i] for i in range(env.n)] for a in acts]
        env.reset()
        for k in range(20):
            env.step(acts[k % 4])
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for k in range(n):
                if k % (EP or 25) == 0:
                    env.reset()
                env.step(acts[k % 4])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[key] = {"value": B * n / best, "unit": "env-steps/s", "us_per_step": best / n * 1e6, "steps": n,
                    "path": ("mpe_step_rows, compiled image" if env.program_compiled else "mpe_step_rows, interpreted") if env.fused
                            else "torch callbacks + mpe_world_step"}
        del env
    # the same scenario through the HEADLINE's protocol (a HIP graph of consecutive mpe_step_rows launches, fresh block-drawn moves
    # for every step, a device reset every EP steps): the device-bound rate of a user scenario, host out of the loop
    for key, pol in (("program_graph", False), ("compiled_graph", None)):
        lg = Leg(mpe, path, 3, B, EP, 0, 1, 0, scenario_kw={"compile_program": pol})
        if (key == "compiled_graph") != bool(lg.env.program_compiled):
            raise RuntimeError("user scenario leg %s: program_compiled = %r" % (key, lg.env.program_compiled))
        d, R, _, r_ = lg.timed(torch, rv, dev, "graph", "fresh", 200, 10, 3, SIDE_REGION_MS)
        out[key] = {"value": B * 200 * R / d, "unit": "env-steps/s", "ms_per_step": d * 1e3 / (200 * R), "timed_steps": 200 * R,
                    "repeats": {"min": r_[0], "median": r_[1], "max": r_[2]},
                    "path": "HIP graph of mpe_step_rows launches (%s), fresh moves per step, reset every %d steps"
                            % ("compiled image" if lg.env.program_compiled else "interpreted", EP or 25)}
        if key == "compiled_graph":      # ... and as whole episodes per launch (mpe_rollout_rows: the moves drawn in the kernel)
            d2, R2, _, r2_ = lg.timed(torch, rv, dev, "fused", "resident", 200, 10, 3, SIDE_REGION_MS)
            out["compiled_fused_rollout"] = {"value": B * 200 * R2 / d2, "unit": "env-steps/s", "us_per_step": d2 * 1e6 / (200 * R2),
                                             "repeats": {"min": r2_[0], "median": r2_[1], "max": r2_[2]},
                                             "path": "mpe_rollout_rows (compiled image): %d steps per launch, every step's rows / rewards / dones "
                                                     "kept in a trajectory" % (EP or 25)}
        lg.release()
        del lg
        torch.cuda.empty_cache()
    out["program_over_generic"] = out["program"]["value"] / out["generic"]["value"]
    out["compiled_over_generic"] = out["compiled"]["value"] / out["generic"]["value"]
    return out


def reference_style_leg(torch, mpe, B, EP, rv, dev):
    """An UNMODIFIED reference-style scenario file (tests/refstyle/convoy.py: `from multiagent.core import ...`, make_world(self),
    NumPy per-world callbacks with Python `if`s on the state, a per-world pick) through make_env(path, batch_size=B): its callbacks
    traced into a compiled row program (symtrace.py) -- one launch per step -- beside the host path (refstyle.py: the same file's
    callbacks per world on the host over the device physics)."""
    path = os.path.join(ROOT, "tests", "refstyle", "convoy.py")
    out = {"what": "tests/refstyle/convoy.py, a reference-style scenario file loaded unmodified (4 agents, 4 landmarks, a per-world goal; "
                   "rewards with contact tests and distance bands as Python ifs): `traced` = make_env(path, batch_size=%d), the file's "
                   "NumPy callbacks traced into the step kernel, env.step / env.reset every %d steps from Python; `traced_graph` = the "
                   "headline's protocol (HIP graph of step launches, fresh moves per step, device resets); `traced_fused_rollout` = whole "
                   "episodes per launch; `host_path` = traced=False: B shadow worlds, the callbacks per world on the host" % (B, EP or 25)}
    t0 = time.perf_counter()
    env = mpe.make_env(path, batch_size=B)
    out["build_s"] = time.perf_counter() - t0      # trace + verification against the file + hipcc (cached by content: 0.x s when warm)
    if not env.traced:
        raise RuntimeError("reference-style leg: not traced (%s)" % env.trace_fallback)
    out["trace"] = {"graph_nodes": env.scenario.t.graph.count, "paths": env.scenario.t.paths, "picks": list(env.scenario.t.pops),
                    "verified_max_diff_vs_file": env.scenario.t.verified, "device_reset": bool(env.scenario.device_reset)}
    g = torch.Generator(device="cpu").manual_seed(0)
    acts = [torch.nn.functional.one_hot(torch.randint(0, 5, (env.n, B), generator=g), 5).float().cuda() for _ in range(4)]
    n = 2000
    env.reset()
    for k in range(20):
        env.step(acts[k % 4])
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(n):
            if k % (EP or 25) == 0:
                env.reset()
            env.step(acts[k % 4])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["traced"] = {"value": B * n / best, "unit": "env-steps/s", "us_per_step": best / n * 1e6, "steps": n,
                     "path": "mpe_step_rows, compiled image with the file's callbacks as code"}
    del env
    lg = Leg(mpe, path, 4, B, EP, 0, 1, 0)
    k_us = lg.kernel_time_us(torch, "graph")
    _, bytes_step, _, _ = lg.geometry()
    out["roofline"] = {"bound": "hbm", "kernel": "mpe_rows_<hash>_s (k_rows, compiled image with the traced callbacks)", "kernel_us_per_launch": k_us,
                       "algorithmic_bytes_per_env_step": bytes_step, "algorithmic_bytes_per_launch": bytes_step * B,
                       "achieved": bytes_step * B / (k_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": bytes_step * B / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "kernel_timing": lg.last_kernel_timing,
                       "note": "algorithmic bytes as the headline's (state + one-hot moves in, state + rows + rewards + dones out); kernel time = "
                               "two-point slope of graphs of dependent launches, resident moves, no resets"}
    d, R, _, r_ = lg.timed(torch, rv, dev, "graph", "fresh", 200, 10, 3, SIDE_REGION_MS)
    out["traced_graph"] = {"value": B * 200 * R / d, "unit": "env-steps/s", "ms_per_step": d * 1e3 / (200 * R), "timed_steps": 200 * R,
                           "repeats": {"min": r_[0], "median": r_[1], "max": r_[2]},
                           "path": "HIP graph of mpe_step_rows launches (compiled image), fresh moves per step, reset every %d steps" % (EP or 25)}
    d2, R2, _, r2_ = lg.timed(torch, rv, dev, "fused", "resident", 200, 10, 3, SIDE_REGION_MS)
    out["traced_fused_rollout"] = {"value": B * 200 * R2 / d2, "unit": "env-steps/s", "us_per_step": d2 * 1e6 / (200 * R2),
                                   "repeats": {"min": r2_[0], "median": r2_[1], "max": r2_[2]},
                                   "path": "mpe_rollout_rows (compiled image): %d steps per launch, every step's rows / rewards / dones kept" % (EP or 25)}
    lg.release()
    del lg
    torch.cuda.empty_cache()
    HB = 256
    henv = mpe.make_env(path, batch_size=HB, traced=False)
    hact = acts[0][:, :HB].contiguous()
    hact = [hact[i] for i in range(henv.n)]
    henv.reset()
    henv.step(hact)
    t0 = time.perf_counter()
    for _ in range(3):
        henv.step(hact)
    torch.cuda.synchronize()
    hdt = (time.perf_counter() - t0) / 3
    out["host_path"] = {"value": HB / hdt, "unit": "env-steps/s", "ms_per_step": hdt * 1e3, "worlds": HB,
                        "path": "refstyle.RefScenarioAdapter: mpe_world_step + the file's callbacks per world on the host"}
    out["traced_over_host_path"] = out["traced"]["value"] / out["host_path"]["value"]
    return out


def box_fingerprint(torch, dev, smi=True):
    """What this rank's GPU is and how the box is set up: device properties from the runtime, clocks / power cap /
    partition modes / driver from rocm-smi when it answers (C4's 20 % box-to-box spread, DESIGN 2.7, needs a label)."""
    fp = {}
    try:
        pr = torch.cuda.get_device_properties(dev)
        fp.update({"name": pr.name, "arch": getattr(pr, "gcnArchName", None), "cus": pr.multi_processor_count,
                   "hbm_bytes": pr.total_memory, "uuid": str(getattr(pr, "uuid", "")) or None,
                   "clock_rate_khz": getattr(pr, "clock_rate", None), "memory_clock_rate_khz": getattr(pr, "memory_clock_rate", None),
                   "l2_bytes": getattr(pr, "L2_cache_size", None)})
        if getattr(pr, "pci_bus_id", None) is not None:
            fp["pci"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
        fp["hip"] = torch.version.hip
    except Exception as e:
        fp["error"] = repr(e)
    if not smi:
        return fp
    try:
        import subprocess
        # (rocm-smi numbers the node's cards, whatever HIP_VISIBLE_DEVICES says: a self-spawned rank knows its card as MPE_GPU_ID)
        r = subprocess.run(["rocm-smi", "-d", os.environ.get("MPE_GPU_ID", str(dev.index)), "--showclocks", "--showpower", "--showmaxpower", "--showmemorypartition",
                            "--showcomputepartition", "--showdriverversion", "--showperflevel", "--json"],
                           capture_output=True, text=True, timeout=30)
        d = json.loads(r.stdout[r.stdout.index("{"):])
        smi = {}
        for card, kv in d.items():
            for k, v in kv.items():
                smi["%s/%s" % (card, k)] = v
        fp["rocm_smi"] = smi
    except Exception as e:
        fp["rocm_smi"] = "unavailable: %s" % type(e).__name__
    return fp


LINE_BUDGET = 4096          # the driver parses rank 0's ONE line; round 5's 20 KB line came back `parsed: null`


def sig(x, n=6):
    """x rounded to n significant digits (floats only; everything else unchanged) -- the printed line is for reading."""
    if isinstance(x, bool) or not isinstance(x, float) or x == 0.0 or not math.isfinite(x):
        return x
    return round(x, n - 1 - int(math.floor(math.log10(abs(x)))))


def _pick(d, keys):
    return {k: sig(d[k]) for k in keys if isinstance(d, dict) and k in d}


def compact_line(out, full_path=None):
    """The ONE line rank 0 prints: the contract's keys + `roofline` + `cpu_baseline` + one figure per BASELINE config, under
    LINE_BUDGET bytes.  Everything else (`extra.*`, per-rank records, the box fingerprint, timing raw data, the long prose
    notes) is the full record, written to `full_path` and named in `full_record`."""
    cfg, roof, ex = out["config"], out["roofline"], out.get("extra", {})
    line = {k: sig(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["timed_steps"], line["timed_region_s"] = out.get("timed_steps"), sig(out.get("timed_region_s"), 4)
    line["config"] = _pick(cfg, ("workload", "protocol", "batch_per_gpu", "global_batch", "mode", "timed_steps", "timed_region_s",
                                 "repeats", "launcher", "barrier_backend", "ranks_seen", "distinct_gpus"))
    if cfg.get("graph_replays_in_timed_region"):
        line["config"]["graph_replays_in_timed_region"] = cfg["graph_replays_in_timed_region"]
    if out.get("n_gpus", 1) > 1:
        line["config"]["sharding"] = cfg.get("sharding")
    line["config"]["workload"] = str(cfg.get("workload_short") or cfg.get("workload", ""))[:150]
    line["config"]["timed_region"] = ("ms_per_step = timed_region_s / timed_steps; >= %.1f s of back-to-back steps, median of %s repeats, "
                                      "max over ranks (--steps / --warmup do not size it)"
                                      % (cfg.get("region_ms", MIN_REGION_MS) * 1e-3, cfg.get("repeats")))
    if cfg.get("scaling_diagnostic"):
        line["config"]["value_over_n_times_rank0_solo"] = sig(cfg["scaling_diagnostic"]["value_over_n_times_rank0_solo"], 4)
    if out.get("per_gpu_value") and out.get("n_gpus", 1) > 1:
        line["per_gpu_value"] = _pick(out["per_gpu_value"], ("min", "median", "max", "ranks"))
    line["roofline"] = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_timed_region", "regime_label",
                                    "l3_resident", "kernel", "kernel_us_per_launch", "kernel_us_per_step", "steps_per_launch", "kernel_us_rocprof", "traffic_per_step",
                                    "algorithmic_bytes_per_env_step", "algorithmic_bytes_per_launch", "env_steps_per_launch",
                                    "launch_floor_us", "measured_copy_GBps", "frac_of_measured_copy", "per_gpu"))
    if isinstance(roof.get("launched"), dict):      # the same protocol as one launch per step, when the line's value is the step server's
        line["roofline"]["launched"] = _pick(roof["launched"], ("value", "ms_per_step", "kernel_us_per_launch", "kernel_us_rocprof", "frac",
                                                              "frac_timed_region", "traffic"))
    if "kernel_us_per_launch_by_rank" in roof:
        line["roofline"]["kernel_us_per_launch_by_rank"] = [sig(x, 4) for x in roof["kernel_us_per_launch_by_rank"]]
    hb = ex.get("hbm_resident")
    if hb and "roofline" in hb:      # the same kernel where the working set cannot sit in the Infinity Cache: THE HBM fraction
        line["roofline"]["hbm_resident_frac"] = sig(hb["roofline"]["frac"])
        line["roofline"]["hbm_resident"] = {"worlds": hb["roofline"]["env_steps_per_launch"],
                                            "kernel_us_per_launch": sig(hb["roofline"]["kernel_us_per_launch"]),
                                            "kernel_us_rocprof_mean": (hb["roofline"].get("kernel_us_rocprof") or {}).get("mean"),
                                            "traffic": hb["roofline"].get("traffic"),
                                            "algorithmic_bytes_per_launch": hb["roofline"]["algorithmic_bytes_per_launch"],
                                            "value": sig(hb["value"]), "frac_timed_region": sig(
                                                hb["roofline"]["algorithmic_bytes_per_launch"] / (hb["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS)}
        if isinstance(hb.get("step_server"), dict) and "value" in hb["step_server"]:      # the same commanded to the step server
            line["roofline"]["hbm_resident"]["step_server"] = _pick(hb["step_server"], ("value", "ms_per_step", "frac", "frac_timed_region"))
            if cfg.get("mode") == "step-server":      # the line's own way of issuing steps, beyond the cache: THE HBM fraction
                line["roofline"]["hbm_resident_frac_launched"] = line["roofline"]["hbm_resident_frac"]
                line["roofline"]["hbm_resident_frac"] = sig(hb["step_server"]["frac"])
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "single_core", "value_in_reference_terms"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:130]
        if cb.get("reference_build_container"):
            line["cpu_baseline"]["reference_build_container"] = _pick(cb["reference_build_container"],
                                                                      ("cores", "env_steps_per_s_1_process", "env_steps_per_s_all_cores"))
        if cb.get("c_port"):
            line["cpu_baseline"]["c_port"] = _pick(cb["c_port"], ("value", "cores", "single_core"))
    cfgs = {}
    for key, ent in (ex.get("configs") or {}).items():
        if "roofline" not in ent:
            cfgs[key] = {"error": str(ent.get("error", "?"))[:80]}
            continue
        c = {"value": sig(ent["value"]), "kernel_us": sig(ent["roofline"]["kernel_us_per_launch"], 4),
             "frac": sig(ent["roofline"]["frac"], 4)}
        if ent.get("fused_rollout"):
            c["rollout_value"] = sig(ent["fused_rollout"]["value"])
            c["rollout_frac_compulsory"] = sig(ent["fused_rollout"]["frac_compulsory"], 4)
        if isinstance(ent.get("step_server"), dict):
            c["step_server_value"] = sig(ent["step_server"]["value"]) if "value" in ent["step_server"] else "error"
        cfgs[key] = c
    if cfgs:
        line["configs"] = cfgs
    side = {}
    for key in ("fused_rollout", "moves_resident", "int_action_ids", "python_api", "host_buffers", "step_server",
                "step_server_per_step_doorbells"):
        if isinstance(ex.get(key), dict) and "value" in ex[key]:
            side[key] = sig(ex[key]["value"])
    for key, sub in (("user_scenario", "compiled_graph"), ("reference_style_file", "traced_graph")):
        ent = ex.get(key)
        if isinstance(ent, dict):        # (the headline's protocol on a user's scenario: a compiled row program / a traced file)
            side[key] = sig(ent[sub]["value"]) if isinstance(ent.get(sub), dict) else "error"
    if side:
        line["env_steps_per_s"] = side
    if full_path:
        line["full_record"] = full_path
    s = json.dumps(line, separators=(",", ":"))
    for drop in ("env_steps_per_s", "per_gpu_value", "configs"):     # never over budget: shed the optional blocks, biggest last
        if len(s) <= LINE_BUDGET:
            break
        line.pop(drop, None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= LINE_BUDGET, len(s)
    return s


def write_full_record(out, path):
    """The whole record (every leg, every raw timing, the box) as indented JSON; returns the path written, or None."""
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        return path
    except OSError as e:
        sys.stderr.write("bench.py: could not write the full record to %s (%s)\n" % (path, e))
        return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=65536, help="worlds per GPU")
    ap.add_argument("--scenario", default="simple_spread")
    ap.add_argument("--agents", type=int, default=3)
    ap.add_argument("--episode-len", type=int, default=25)
    ap.add_argument("--mode", default="graph", choices=["graph", "eager", "api", "host", "fused"])
    ap.add_argument("--protocol", default="fresh", choices=["fresh", "resident"],
                    help="fresh: every step's moves are newly drawn (one block draw per episode, timed); resident: a ring of "
                         "16 move tensors drawn once (round-1 headline)")
    ap.add_argument("--commands", default="auto", choices=["auto", "launched", "served"],
                    help="how the headline's steps are issued (graph mode, fresh protocol): launched -- one mpe_step launch per step; "
                         "served -- commanded to the resident step server (mpe_step_server_*); auto -- both are measured, the line's "
                         "value is the faster (the other is kept in the line)")
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--region-ms", type=float, default=MIN_REGION_MS,
                    help="minimum back-to-back GPU work per timed repeat of the headline leg")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (other configs, 1M leg, fused rollout)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"],
                    help="what carries the N>1 barrier: auto = RCCL when every rank can bring it up, else gloo (the step "
                         "path has no collective either way)")
    ap.add_argument("--all-ranks-on-gpu0", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--device-map", default=None, metavar="I,J,...",
                    help="rank r drives visible GPU number <r-th entry> instead of GPU r (ranks that share a GPU are refused "
                         "unless --all-ranks-on-gpu0)")
    ap.add_argument("--dump-state", default=None, metavar="DIR",
                    help="before timing: run the rollout's episode-0 reset + step 0 and save this rank's pos / vel / obs / "
                         "rew to DIR/rank<r>.npz (the multi-rank rehearsal test compares them with one big batch)")
    ap.add_argument("--generic", action="store_true",
                    help="step through the generic path (torch callbacks + mpe_world_step) although a fused kernel exists")
    ap.add_argument("--leg-json", default=None, help=argparse.SUPPRESS)      # internal: measure one config leg, print its entry
    ap.add_argument("--leg-floor-us", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--leg-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--full-json", default=os.path.join("gpurun_out", "bench_full.json"), metavar="PATH",
                    help="where rank 0 writes the full record (every leg, raw timings, per-rank records, the box); the printed "
                         "line is its compact form (< %d bytes)" % LINE_BUDGET)
    ap.add_argument("--streams", type=int, default=1,
                    help="cut the per-GPU batch into this many independent sub-batches, one HIP stream each")
    return ap.parse_args(argv)


def device_map(args, n):
    """--device-map as a list of n visible-GPU numbers, or None."""
    if not args.device_map:
        return None
    m = [int(x) for x in args.device_map.split(",") if x.strip() != ""]
    if len(m) != n or min(m) < 0:
        raise SystemExit("bench.py: --device-map needs %d non-negative entries (one per rank), got %r" % (n, args.device_map))
    return m


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one per GPU), relay rank 0's
    line.  Never measures fewer GPUs than were asked for: too few visible GPUs, a failed rank or a line whose n_gpus is
    not N is a non-zero exit, not a mislabelled record."""
    import torch
    from multiagent_particle_envs_amd import sharding
    n = args.gpus
    seen = torch.cuda.device_count() if torch.cuda.is_available() else 0
    dmap = device_map(args, n)
    if seen < (1 if args.all_ranks_on_gpu0 else (max(dmap) + 1 if dmap is not None else n)):
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this node -- refusing to measure fewer GPUs "
                         "than requested (run with --gpus <= %d, or --all-ranks-on-gpu0 for the one-GPU rehearsal)\n"
                         % (n, seen, max(seen, 1)))
        return 2
    sys.stderr.write("bench.py: --gpus %d without a launcher (WORLD_SIZE unset): starting %d ranks, one per GPU\n" % (n, n))
    rc, out = sharding.spawn_local_ranks([os.path.abspath(__file__)] + sys.argv[1:], n, one_device=args.all_ranks_on_gpu0,
                                         gpu_slots=dmap)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if rc != 0 or len(lines) != 1:
        sys.stderr.write("bench.py: the %d-rank job failed (exit %s, %d JSON lines)\n%s\n" % (n, rc, len(lines), out[-2000:]))
        return rc or 1
    line = json.loads(lines[0])
    if line.get("n_gpus") != n:
        sys.stderr.write("bench.py: ranks reported n_gpus=%r for --gpus %d\n" % (line.get("n_gpus"), n))
        return 1
    print(lines[0])
    return 0


def main():
    args = parse_args()
    if args.leg_json:
        print(json.dumps(run_config_leg(args.leg_json, args.leg_floor_us, args.seed, args.episode_len, args.leg_device)))
        return
    env_world = int(os.environ.get("WORLD_SIZE", "1") or "1")
    if args.gpus > 1 and env_world == 1:
        sys.exit(launch_ranks(args))

    t_start = time.time()
    timeline = {}          # seconds since the start at which each part of the run was done (config.timeline_s): where a slow run lost its time

    def done_at(name):
        timeline[name] = round(time.time() - t_start, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = env_world
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d: launch with --nproc-per-node == --gpus (or drop the launcher: "
                         "`python bench.py --gpus N` starts its own ranks)" % (world, args.gpus))
    local = 0 if args.all_ranks_on_gpu0 else int(os.environ.get("LOCAL_RANK", "0"))
    if device_map(args, world) is not None and not os.environ.get("MPE_SELF_SPAWNED") and not args.all_ranks_on_gpu0:
        local = device_map(args, world)[rank]      # (self-spawned ranks were handed that GPU as their only one: cuda:0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the step path has no CPU fallback)")
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants cuda:%d, %d GPU(s) visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import multiagent_particle_envs_amd as mpe
    from multiagent_particle_envs_amd import sharding
    # this rank's place among the node's ranks (self-spawned ranks see ONE GPU each and carry their slot in MPE_LOCAL_RANK;
    # under a launcher the slot is LOCAL_RANK) -> an even slice of the allowed CPUs for its Python host
    slot = int(os.environ.get("MPE_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0")) or 0)
    cpus = sharding.pin_rank(slot, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)) or world)) if world > 1 else None
    rv = sharding.Rendezvous(rank, world, dev, backend=args.backend)
    if rv.note and rank == 0:
        sys.stderr.write("bench.py: %s\n" % rv.note)

    def leave(status=0):
        """End of this rank: flush, close the rendezvous; a job that abandoned an RCCL communicator leaves through
        os._exit WITH ITS STATUS (tearing the communicator down may block) -- decided here, not inside the library."""
        sys.stdout.flush()
        sys.stderr.flush()
        if not rv.close():
            os._exit(status)
        if status:
            sys.exit(status)

    B, K, W, EP = args.batch, args.steps, args.warmup, args.episode_len
    leg = Leg(mpe, args.scenario, args.agents, B, EP, rank, args.streams, args.seed, args.generic)
    if not leg.env.fused:
        bench_generic(args, leg.env, dev, rank, world, rv)
        return leave(0)
    A, Lm = leg.A, leg.Lm
    obs_total, bytes_step, compulsory_roll, kname = leg.geometry()
    can_fuse = A + Lm <= 16 or args.scenario in ("simple_spread", "simple_tag")

    if args.dump_state:
        import numpy as np
        r0 = leg.roll(args.protocol).rollouts[0]
        assert r0.t == 0
        out0 = r0.enqueue(1)
        torch.cuda.synchronize()
        os.makedirs(args.dump_state, exist_ok=True)
        np.savez(os.path.join(args.dump_state, "rank%d.npz" % rank), pos=leg.env.world.pos.cpu().numpy(),
                 vel=leg.env.world.vel.cpu().numpy(), obs=out0.obs.cpu().numpy(), rew=out0.rew.cpu().numpy(),
                 world_offset=np.int64(leg.env.world.world_offset))

    def stats(rate):
        return {"min": rate[0], "median": rate[1], "max": rate[2], "unit": "env-steps/s per GPU"}

    # ---- N > 1: every rank must be driving a GPU of its own -- checked BEFORE anything is measured --------------------
    me = box_fingerprint(torch, dev, smi=False)
    ids = rv.gather({"rank": rank, "uuid": me.get("uuid"), "pci": me.get("pci"), "gpu_env": os.environ.get("HIP_VISIBLE_DEVICES")})
    dup = duplicate_gpus(ids) if world > 1 else ""
    if dup and not args.all_ranks_on_gpu0:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but ranks share a GPU (%s) -- a mis-mapped job would print n_gpus=%d for fewer "
                             "GPUs; refusing (use --all-ranks-on-gpu0 for the one-GPU rehearsal)\n" % (world, dup, world))
        return leave(3)
    solo_rate = None
    if world > 1:
        # rank 0 runs one repeat of the SAME timed region alone (the others wait at a barrier): what one GPU of this node
        # does when its neighbours are idle -- the reference point of the N-GPU line (config.scaling_diagnostic)
        lone = sharding.Rendezvous(0, 1, dev)
        if rank == 0:
            d1, R1, _, _ = leg.timed(torch, lone, dev, args.mode, args.protocol, K, W, 1, min(args.region_ms, 1000.0))
            solo_rate = B * K * R1 / d1
        rv.barrier()
    dt, R, ev_ms, rate = leg.timed(torch, rv, dev, args.mode, args.protocol, K, W, args.repeats, args.region_ms)
    done_at("headline_timed_region")
    k_us = leg.kernel_time_us(torch, args.mode)
    head_timing = leg.last_kernel_timing
    head_probe = leg.env.placement_probe
    floor_us = launch_floor_us(torch, dev)
    # ---- the same protocol with the steps COMMANDED to the step server: every rank measures it under the same bracket; the line's
    # value is the faster of the two ways to issue the steps (a failure on any rank keeps the launched figures for all)
    served, served_err, solo_served = None, None, None
    if args.commands != "launched" and args.mode == "graph" and args.protocol == "fresh" and args.streams == 1 and EP and \
            kname == "mpe::k_split" and leg.roll("resident").rollouts[0].pool_c is None:
        roll_s = None
        try:      # phase 1, no rendezvous inside: build the server (stream probe, graphs) and serve a few episodes
            from multiagent_particle_envs_amd.rollout import ServedRollout
            env_s = mpe.make_env(args.scenario, batch_size=B, seed=args.seed, **leg.kw)
            env_s.world.world_offset = rank * B
            roll_s = ServedRollout(env_s, episode_len=EP, timeout_s=4.0, graphs=True, ring_ahead=True)
            roll_s.enqueue(2 * EP * 4)
            torch.cuda.synchronize()
            roll_s.srv.check()
        except Exception as e:
            served_err = "%s: %s" % (type(e).__name__, str(e)[:300])
            roll_s = None
            torch.cuda.synchronize()
        if rv.reduce_max(1.0 if roll_s is None else 0.0) > 0:      # any rank without a server: nobody measures one
            roll_s = None
            served_err = served_err or "another rank could not bring its step server up"
        if roll_s is not None:      # phase 2: every rank makes the same rendezvous calls (served_timed keeps its errors to the end)
            if world > 1:           # (rank 0 alone first, as for the launched steps: the N-GPU line's reference point)
                if rank == 0:
                    try:
                        r1 = served_timed(torch, sharding.Rendezvous(0, 1, dev), roll_s, B, min(args.region_ms, 1000.0), 1)
                        solo_served = B * r1["n"] / r1["dt"]
                    except Exception as e:
                        served_err = "%s: %s" % (type(e).__name__, str(e)[:300])
                rv.barrier()
            try:
                served = served_timed(torch, rv, roll_s, B, args.region_ms, args.repeats)
                served["probe"] = roll_s.srv.stream_probe
            except Exception as e:
                served_err = "%s: %s" % (type(e).__name__, str(e)[:300])
                torch.cuda.synchronize()
            del roll_s, env_s
        if rv.reduce_max(1.0 if served is None else 0.0) > 0:      # any rank failed: nobody switches
            served = None
        done_at("headline_step_server")
    use_served = served is not None and (args.commands == "served" or served["dt"] / served["n"] < dt / (K * R))
    extra = {}
    solo = world == 1 and not args.no_extra and args.streams == 1
    SR = SIDE_REGION_MS

    if solo and can_fuse and args.mode != "fused":
        dtf, Rf, _, rf = leg.timed(torch, rv, dev, "fused", "resident", K, W, 3, SR)
        kf = leg.kernel_time_us(torch, "fused")
        extra["fused_rollout"] = {
            "what": "mpe_rollout_random: one launch per %d-step episode, state kept on chip (registers / LDS), moves drawn "
                    "in-kernel, every step's obs/rew/done written to its own trajectory block" % (EP or 25),
            "value": B * K * Rf / dtf, "unit": "env-steps/s", "ms_per_step": dtf * 1e3 / (K * Rf), "repeats": stats(rf),
            "kernel_us_per_step": kf,
            "compulsory_bytes_per_env_step": compulsory_roll,
            "achieved_GBps_compulsory": compulsory_roll * B / (kf * 1e-6) / 1e9,
            "frac_compulsory": compulsory_roll * B / (kf * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "achieved_GBps_at_per_step_convention": bytes_step * B / (kf * 1e-6) / 1e9,
            "per_step_convention_bytes": bytes_step}
    if solo and args.mode == "graph" and args.protocol == "fresh":
        dtr, Rr, _, rr = leg.timed(torch, rv, dev, "graph", "resident", K, W, 3, SR)
        extra["moves_resident"] = {
            "what": "the same K-step graph with the moves read from a resident ring of 16 tensors drawn once (no redraw "
                    "launches in the timed region): the step as a policy-driven caller sees it",
            "value": B * K * Rr / dtr, "unit": "env-steps/s", "ms_per_step": dtr * 1e3 / (K * Rr), "repeats": stats(rr)}

    if solo and args.mode == "graph" and args.protocol == "fresh" and leg.roll("resident").rollouts[0].pool_c is None:
        # the reference's other action format (`discrete_action_input`, environment.py:161-167): int32 ids, 4 bytes per agent
        dti, Ri, _, ri = leg.timed(torch, rv, dev, "graph", "fresh_ids", K, W, 3, SR)
        ki = leg.kernel_time_us(torch, "graph", protocol="resident_ids")
        bytes_ids = bytes_step - 4 * A * (_abi_action_dim() - 1)
        extra["int_action_ids"] = {
            "what": "the headline protocol with the moves handed over as int32 ids [A][B] (discrete_action_input; SURVEY 8d: "
                    "subtract 5A*4 B, add A*4 B) instead of one-hot fp32 rows: fresh ids for every step, reset every %d" % EP,
            "value": B * K * Ri / dti, "unit": "env-steps/s", "ms_per_step": dti * 1e3 / (K * Ri), "repeats": stats(ri),
            "kernel_us_per_launch": ki, "algorithmic_bytes_per_env_step": bytes_ids,
            "achieved_GBps": B * bytes_ids / ki / 1e3, "frac": B * bytes_ids / ki / 1e3 / HBM_PEAK_GBS}
        leg.rolls.pop("fresh_ids", None)
        leg.rolls.pop("resident_ids", None)

    if solo and args.mode == "graph":
        # the drop-in API itself: env.reset() / env.step() called from Python, one launch per call, moves from a resident ring
        na = max(K, 20000)
        dta, _, _, ra = leg.timed(torch, rv, dev, "api", "resident", na, W, 3)
        extra["python_api"] = {
            "what": "MultiAgentEnv.step() / reset() called from Python (the drop-in API; host in the loop, no graph): "
                    "%d steps, reset every %d" % (na, EP),
            "value": B * na / dta, "unit": "env-steps/s", "ms_per_step": dta * 1e3 / na, "repeats": stats(ra)}

    if solo and args.mode == "graph" and leg.roll("resident").rollouts[0].pool_c is None:
        # the same API with the caller's buffers in HOST memory (PCIe both ways, a synchronisation every step): never `value`
        nh = 200
        dth, _, _, rh = leg.timed(torch, rv, dev, "host", "resident", nh, 10, 3)
        io_bytes = 4 * (A * _abi_action_dim() + obs_total + A)
        extra["host_buffers"] = {
            "what": "MultiAgentEnv.step() with one-hot moves coming from pinned host memory and observations + rewards copied "
                    "back to pinned host memory, the caller waiting for them every step (a host-side policy, as in the reference's "
                    "callers): PCIe-inclusive, %d steps" % nh,
            "value": B * nh / dth, "unit": "env-steps/s", "ms_per_step": dth * 1e3 / nh,
            "pcie_bytes_per_env_step": io_bytes, "pcie_GBps": B * nh * io_bytes / dth / 1e9}

    if solo and served is not None:
        # ... and with a doorbell LAUNCH per step instead of one per episode: bounded by the command processor's dependent-launch
        # rate, erratic by stream pairing: a secondary figure
        try:
            from multiagent_particle_envs_amd.rollout import ServedRollout
            env_p = mpe.make_env(args.scenario, batch_size=B, seed=args.seed, **leg.kw)
            roll_p = ServedRollout(env_p, episode_len=EP, timeout_s=4.0, graphs=True, ring_ahead=False)
            roll_p.enqueue(2 * EP * 4)
            torch.cuda.synchronize()
            roll_p.srv.check()
            rp = served_timed(torch, rv, roll_p, B, SR, 3)
            extra["step_server_per_step_doorbells"] = {
                "what": "the step server commanded by ONE DOORBELL LAUNCH PER STEP (25 per episode, replayed as a graph behind the draw)",
                "value": B * rp["n"] / rp["dt"], "unit": "env-steps/s", "ms_per_step": rp["dt"] * 1e3 / rp["n"], "repeats": stats(rp["rate"]),
                "server_stream_probe": roll_p.srv.stream_probe}
            del roll_p, env_p
        except Exception as e:
            extra["step_server_per_step_doorbells"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            torch.cuda.synchronize()
    done_at("headline_side_legs")
    default_line = solo and args.scenario == "simple_spread" and args.agents == 3 and B == 65536
    if default_line:
        # (secondary legs: a failure there -- no hipcc on the box for the compiled programs, say -- is reported in the entry and
        #  never costs the run its headline line)
        for key, leg_fn in (("user_scenario", user_scenario_leg), ("reference_style_file", reference_style_leg)):
            try:
                extra[key] = leg_fn(torch, mpe, B, EP, rv, dev)
            except Exception as e:
                extra[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
                torch.cuda.synchronize()
            done_at(key)
    headline_roof = roofline_entry(leg, k_us, B, args.mode, floor_us, head_timing)
    if default_line:
        leg.release()
        torch.cuda.empty_cache()
        # ---- the same kernel and protocol where the working set cannot sit in the Infinity Cache ----------------
        big = Leg(mpe, "simple_spread", 3, 1 << 20, EP, rank, 1, args.seed)
        dtb, Rb, _, rb_ = big.timed(torch, rv, dev, "graph", "fresh", 25, 5, 5, 2 * SR)
        kb = big.kernel_time_us(torch, "graph", n=100)
        rb = roofline_entry(big, kb, 1 << 20, "graph", floor_us, big.last_kernel_timing)
        extra["hbm_resident"] = {"what": "simple_spread N=3 at 1048576 worlds (431 MB per launch, 2 GB of moves per episode): "
                                         "same kernel, same protocol, HBM-resident",
                                 "value": (1 << 20) * 25 * Rb / dtb, "unit": "env-steps/s", "repeats": stats(rb_),
                                 "ms_per_step": dtb * 1e3 / (25 * Rb), "roofline": rb}
        big.release()
        del big
        torch.cuda.empty_cache()
        # ... and commanded to the step server: 16 384 workgroups cannot be resident, so each episode's steps are commanded BEFORE its
        # launch (one launch per episode behind its doorbell: nothing in it waits), the next episode's draw overlapping on the other stream
        try:
            from multiagent_particle_envs_amd.rollout import ServedRollout
            env_b = mpe.make_env("simple_spread", batch_size=1 << 20, seed=args.seed)
            roll_b = ServedRollout(env_b, episode_len=EP or 25, timeout_s=6.0, graphs=True, launch_per_episode=True)
            roll_b.enqueue(2 * (EP or 25))
            torch.cuda.synchronize()
            roll_b.srv.check()
            sb = served_timed(torch, rv, roll_b, 1 << 20, 2 * SR, 3)
            per_b = rb["algorithmic_bytes_per_launch"]
            extra["hbm_resident"]["step_server"] = {
                "what": "the same at 1048576 worlds with the steps commanded to the step server: one launch per %d-step episode behind "
                        "its doorbell (every command precedes its launch: no residency needed)" % (EP or 25),
                "value": (1 << 20) * sb["n"] / sb["dt"], "unit": "env-steps/s", "ms_per_step": sb["dt"] * 1e3 / sb["n"],
                "repeats": stats(sb["rate"]), "server_launch_us": sb["launch_us"], "steps_per_server_launch": sb["steps_per_launch"],
                "frac": per_b * sb["steps_per_launch"] / (sb["launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "frac_timed_region": per_b / (sb["dt"] / sb["n"]) / 1e9 / HBM_PEAK_GBS}
            del roll_b, env_b
        except Exception as e:
            extra["hbm_resident"]["step_server"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            torch.cuda.synchronize()
        torch.cuda.empty_cache()
        done_at("hbm_resident")
        # ---- BASELINE.json's other single-GPU configs, each in a process of its own ------------------------------------
        # (a leg measured after the 1M leg in THIS process reads 15-25 % slower on the launch-bound configs than the same
        #  leg in a fresh process -- 4.07-4.32 vs 3.45 us for C3, not thermal: a fresh process started right after this
        #  run gives 3.45 again, profiles/r3_ab_logs.txt session 9 -- so each config gets what a user of that config gets)
        cfgs = {}
        for key, scn, ag, bb, kk in CONFIG_LEGS:
            cfgs[key] = config_leg_subprocess(key, args, floor_us, local)
        extra["configs"] = cfgs
        done_at("configs_in_their_own_processes")

    # ---- per-rank records: which GPU each rank drove, its own rate and kernel time (gathered over the bookkeeping group) ----
    box = box_fingerprint(torch, dev, smi=(rank == 0))
    recs = rv.gather({"rank": rank, "device": "cuda:%d" % local, "gpu_env": os.environ.get("HIP_VISIBLE_DEVICES"),
                      "name": box.get("name"), "uuid": box.get("uuid"), "pci": box.get("pci"), "cpus": cpus,
                      "world_offset": rank * B, "kernel_us_per_launch": k_us, "launch_floor_us": floor_us,
                      "env_steps_per_s": stats(served["rate"] if use_served else rate), "env_steps_per_s_launched": stats(rate),
                      "server_launch_us": served["launch_us"] if served else None,
                      "steps_per_server_launch": served["steps_per_launch"] if served else None})
    if rank == 0:
        copy_gbs = copy_ceiling_gbs(torch, dev)
        fill_gbs = fill_ceiling_gbs(torch, dev)
        for ent in extra.get("configs", {}).values():   # the other configs against this box's measured streaming rates
            ent["roofline"].update({"measured_copy_GBps": copy_gbs, "measured_fill_GBps": fill_gbs,
                                    "frac_of_measured_fill": ent["roofline"]["achieved"] / fill_gbs})
        if world > 1:   # the per-GPU roofline of an N-GPU job is the SLOWEST rank's kernel (every rank runs the same launch)
            k_slow = max(r["kernel_us_per_launch"] for r in recs)
            headline_roof = roofline_entry(leg, k_slow, B, args.mode, floor_us, head_timing)
            headline_roof["per_gpu"] = True
            headline_roof["kernel_us_per_launch_by_rank"] = [r["kernel_us_per_launch"] for r in recs]
        headline_roof.update({
            "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": headline_roof["achieved"] / copy_gbs,
            "measured_fill_GBps": fill_gbs,
            "timed_region_us_per_step": ev_ms * 1e3 / (K * R),
            # the same algorithmic bytes over the TIMED REGION's time per step (fresh move draws and resets included): the
            # end-to-end fraction next to the kernel-only `frac` -- both are printed, the smaller one is the job's
            "frac_timed_region": bytes_step * B / (dt / (K * R)) / 1e9 / HBM_PEAK_GBS,
            "achieved_timed_region": bytes_step * B / (dt / (K * R)) / 1e9,
            "frac_definition": "frac = algorithmic bytes per launch / kernel_us_per_launch (two-point slope of resident-move, "
                               "no-reset launches) / peak; frac_timed_region = algorithmic bytes per step / ms_per_step / peak",
            "note": "achieved = algorithmic bytes per launch / kernel_us_per_launch; kernel_us_per_launch = (HIP-event time of 2n "
                    "back-to-back dependent step launches - that of n) / n on the launch stream, best of 3 each, n = 400 (the "
                    "rocprofv3 kernel-trace summary of the same command is under profiles/); launch_floor_us = the same for a 64-world bookkeeping "
                    "kernel; timed_region_us_per_step = HIP-event time of the timed region / timed_steps and "
                    "includes the per-episode reset and move-draw launches"})
        out = {
            "metric": "env steps/sec (whole node), %s N=%d, batch=%d per GPU" % (args.scenario, A, B),
            "value": B * K * R * world / dt, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt * 1e3 / (K * R), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "timed_region_s": dt, "timed_steps": K * R,
            "per_gpu_value": per_gpu_stats(recs),
            "config": {"workload": "%s A=%d L=%d, %d worlds/GPU, one-hot random moves (%s), device reset every %d steps"
                                   % (args.scenario, A, Lm, B,
                                      "fresh for every step: one block draw per episode inside the timed region"
                                      if args.protocol == "fresh" else "resident ring of 16 tensors", EP),
                       "workload_short": "%s A=%d L=%d, %d worlds/GPU, %s one-hot moves, reset every %d steps; one mpe_step launch per step"
                                         % (args.scenario, A, Lm, B, "fresh" if args.protocol == "fresh" else "resident", EP),
                       "protocol": args.protocol, "batch_per_gpu": B, "global_batch": B * world, "mode": args.mode,
                       "graph_replays_in_timed_region": R, "timed_steps": K * R,
                       "timed_region_s": dt, "repeats": args.repeats, "region_ms": args.region_ms, "streams_per_gpu": args.streams,
                       "sharding": "worlds by batch index, no collective on the step path",
                       "barrier_backend": rv.backend, "barrier_note": rv.note,
                       "launcher": "torch.distributed.run / env" if world > 1 and not os.environ.get("MPE_SELF_SPAWNED") else
                                   ("self-spawned (bench.py started the ranks)" if world > 1 else "single process"),
                       "ranks_seen": len(recs), "ranks": recs,
                       "distinct_gpus": len(set(gpu_identity(r) for r in recs)),
                       "barrier_fallbacks": rv.barrier_fallbacks,
                       "scaling_diagnostic": None if solo_rate is None else {
                           "what": "rank 0 alone (its neighbours idle at a barrier), one repeat of the same timed region, inside this "
                                   "job; value_over_n_times_rank0_solo = value / (n_gpus x that rate)",
                           "rank0_solo_env_steps_per_s": solo_rate,
                           "value_over_n_times_rank0_solo": (B * K * R * world / dt) / (world * solo_rate)},
                       "wall_s_since_start": time.time() - t_start, "timeline_s": timeline,
                       "placement_probe": head_probe},
            "roofline": headline_roof,
            "repeats": stats(rate),
        }
        if served is not None or served_err:
            extra["step_server"] = {"error": served_err} if served is None else {
                "what": "the headline protocol with the steps COMMANDED to the resident step server (ServedRollout, ring_ahead: one "
                        "doorbell launch per episode behind the draw of its moves); max over ranks of the median repeat",
                "value": B * served["n"] * world / served["dt"], "unit": "env-steps/s", "ms_per_step": served["dt"] * 1e3 / served["n"],
                "timed_steps": served["n"], "timed_region_s": served["dt"], "server_launch_us": served["launch_us"],
                "steps_per_server_launch": served["steps_per_launch"], "server_stream_probe": served.get("probe"),
                "rank0_solo_env_steps_per_s": solo_served}
        if use_served:
            # ---- the line's value: the step server's figures; the launched steps' stay in roofline.launched ----------------
            n_s, dt_s = served["n"], served["dt"]
            l_us = max(r["server_launch_us"] for r in recs)          # per GPU: the slowest rank's server launch
            spl = served["steps_per_launch"]
            per_launch = bytes_step * B * spl
            kt = (pmc_traffic("served_%s_A%d_L%d_B%d" % (args.scenario, A, Lm, B)) or {})
            launched = {"value": out["value"], "ms_per_step": out["ms_per_step"], "timed_steps": out["timed_steps"],
                        "kernel": headline_roof["kernel"], "kernel_us_per_launch": headline_roof["kernel_us_per_launch"],
                        "kernel_us_rocprof": headline_roof.get("kernel_us_rocprof"), "frac": headline_roof["frac"],
                        "frac_timed_region": headline_roof["frac_timed_region"], "traffic": headline_roof.get("traffic"),
                        "algorithmic_bytes_per_launch": headline_roof["algorithmic_bytes_per_launch"]}
            headline_roof.update({
                "achieved": per_launch / (l_us * 1e-6) / 1e9, "frac": per_launch / (l_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "kernel": "mpe::k_split<SERVE> (step server)", "steps_per_launch": spl,
                "kernel_us_per_launch": l_us, "kernel_us_per_step": l_us / spl, "env_steps_per_launch": int(B * spl),
                "algorithmic_bytes_per_launch": int(per_launch), "kernel_timing": None,
                # the committed PMC passes / kernel trace profiled a launch of kt["steps_per_launch"] steps: scaled to this launch's
                "traffic": int(kt["traffic_bytes_per_launch"] / kt["steps_per_launch"] * spl) if kt.get("traffic_bytes_per_launch") else None,
                "traffic_per_step": kt["traffic_bytes_per_launch"] / kt["steps_per_launch"] if kt.get("traffic_bytes_per_launch") else None,
                "traffic_source": (kt["source"] + "; that launch served %d steps: scaled by steps_per_launch" % kt["steps_per_launch"])
                                  if kt.get("source") else None,
                "kernel_us_rocprof": ({"mean_per_step": kt["kernel_trace"]["mean_us"] / kt["steps_per_launch"],
                                       "median_per_step": kt["kernel_trace"]["median_us"] / kt["steps_per_launch"],
                                       "steps_per_launch": kt["steps_per_launch"],
                                       "source": kt["kernel_trace"]["source"]} if kt.get("kernel_trace") else None),
                "timed_region_us_per_step": dt_s * 1e6 / n_s,
                "frac_timed_region": bytes_step * B / (dt_s / n_s) / 1e9 / HBM_PEAK_GBS,
                "achieved_timed_region": bytes_step * B / (dt_s / n_s) / 1e9,
                "frac_of_measured_copy": per_launch / (l_us * 1e-6) / 1e9 / copy_gbs,
                "frac_definition": "frac = algorithmic bytes per server launch (bytes per env-step x worlds x commanded steps) / the "
                                   "launch's HIP-event duration on the server's stream / peak; frac_timed_region = algorithmic bytes "
                                   "per step / ms_per_step / peak (draws and doorbells included)",
                "note": "the dominant kernel is the step server's launch: its duration is measured with HIP events on the server's "
                        "stream around each launch of the timed region (mean of the median repeat; slowest rank); roofline.launched "
                        "holds the same protocol issued as one mpe_step launch per step",
                "launched": launched})
            if world > 1:
                headline_roof["kernel_us_per_launch_by_rank"] = [r["server_launch_us"] for r in recs]
            out.update({"value": B * n_s * world / dt_s, "ms_per_step": dt_s * 1e3 / n_s, "timed_region_s": dt_s, "timed_steps": n_s,
                        "per_gpu_value": per_gpu_stats(recs), "repeats": stats(served["rate"])})
            out["config"].update({
                "workload": "%s A=%d L=%d, %d worlds/GPU, one-hot random moves fresh for every step (one block draw per episode inside "
                            "the timed region, read from HBM by every step), reset every %d steps; steps COMMANDED to the resident "
                            "step server" % (args.scenario, A, Lm, B, EP),
                "workload_short": "%s A=%d L=%d, %d worlds/GPU, fresh one-hot moves read from HBM every step (block draw per episode, "
                                  "timed), reset every %d steps; steps commanded to the step server" % (args.scenario, A, Lm, B, EP),
                "protocol": "fresh, step-server (one doorbell per episode)", "mode": "step-server",
                "graph_replays_in_timed_region": None, "timed_steps": n_s, "timed_region_s": dt_s,
                "server_launches_in_timed_region": served["launches_in_region"], "server_stream_probe": served.get("probe"),
                "scaling_diagnostic": None if solo_served is None else {
                    "what": "rank 0 alone (its neighbours idle at a barrier), one repeat of the same timed region, inside this job",
                    "rank0_solo_env_steps_per_s": solo_served,
                    "value_over_n_times_rank0_solo": (B * n_s * world / dt_s) / (world * solo_served)}})
        extra["box"] = box
        out["extra"] = extra
        # the CPU baseline runs LAST (rank 0 only; the other ranks wait in close()): every GPU leg is behind us
        if not args.no_cpu_baseline and args.scenario not in ("simple", "simple_spread", "simple_tag"):
            out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                   "sample": "none: the per-object CPU restatement (oracle/mpe_loop.py) covers simple / simple_spread / "
                                             "simple_tag; the batched oracle of the other scenarios is test infrastructure only"}
        elif not args.no_cpu_baseline:
            procs = usable_cores()
            if world > 1:      # the N-GPU line keeps the baseline but not its length: the job's wall time is bounded
                args.cpu_seconds = min(args.cpu_seconds, 3.0)
            agg, single = cpu_baseline(args.scenario, leg.okw, args.cpu_seconds, procs)
            refkey = {"simple": "simple", "simple_tag": "simple_tag"}.get(
                args.scenario, "simple_spread_n%d" % A if args.scenario == "simple_spread" else None)
            ref = cpu_reference_record(refkey) if refkey else None
            out["cpu_baseline"] = {
                "value": agg, "unit": "env-steps/s", "cores": procs, "kind": "port",
                "sample": "oracle/mpe_loop.py (the reference's per-object fp64 Python/NumPy loop restated; "
                          "/root/reference is absent on the GPU box), %d processes x %.0f s each, same move "
                          "distribution, reset every 25 steps; 1 process alone: %.0f env-steps/s.  The unmodified "
                          "reference itself, same protocol, in the build container: see `reference_build_container`"
                          % (procs, args.cpu_seconds, single),
                "single_core": single}
            if ref:
                out["cpu_baseline"]["reference_build_container"] = ref
                out["cpu_baseline"]["value_in_reference_terms"] = agg / ref["port_over_reference_all_cores"]
            cp = cpu_baseline_c(args.scenario, leg.okw, min(args.cpu_seconds, 4.0), procs)
            if cp:
                out["cpu_baseline"]["c_port"] = {
                    "value": cp[0], "unit": "env-steps/s", "cores": procs, "single_core": cp[1],
                    "sample": "oracle/mpe_oracle.c (the same algorithm in plain C, gcc -O2 -fopenmp, one world per "
                              "thread, fp64), %d threads x %.0f s" % (procs, min(args.cpu_seconds, 4.0))}
        full = write_full_record(out, args.full_json if os.path.isabs(args.full_json) else os.path.join(ROOT, args.full_json))
        print(compact_line(out, args.full_json if full else None))
        sys.stdout.flush()
    leave(0)


if __name__ == "__main__":
    main()
