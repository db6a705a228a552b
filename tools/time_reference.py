#!/usr/bin/env python3
"""Time the UNMODIFIED reference (MultiAgentEnv.step, environment.py:80-104 -> core.py:117-131 ->
Scenario.observation/reward) on this machine's host cores -- SURVEY.md 8(d)'s CPU baseline.

    python tools/time_reference.py [--seconds 10]      -> profiles/cpu_reference.json

Runs only where /root/reference exists (the build container); the tree is imported read-only through the
shape-only gym stub of tests/golden/_gym_stub (nothing is written into it).  Protocol = bench.py's CPU leg:
uniform random one-hot moves, env.reset() every 25 steps, 1 process and then one process per usable core
(independent envs, aggregate env-steps/s).  The same protocol is run on oracle/mpe_loop.py (the restatement
bench.py times on the GPU box, where the reference tree is absent), so the JSON also records the
port / reference speed ratio that lets a reader convert the GPU box's `cpu_baseline` into reference terms.
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SUPPRESS_MA_PROMPT"] = "1"
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)


def ref_worker(arg):
    scenario, n_agents, seconds, idx = arg
    sys.path[:0] = [os.path.join(ROOT, "tests", "golden", "_gym_stub"), "/root/reference"]
    import warnings
    warnings.filterwarnings("ignore")
    import numpy as np
    if scenario == "simple_spread" and n_agents != 3:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import gen_golden
        env = gen_golden.spread_n(n_agents)   # make_world with N agents / N landmarks, every other method the reference's own
    else:
        from make_env import make_env
        env = make_env(scenario)
    np.random.seed(idx)
    A = env.n
    acts = np.eye(5)[np.random.randint(0, 5, size=(4096, A))]
    n, t0 = 0, time.perf_counter()
    while True:
        if n % 25 == 0:
            env.reset()
        env.step(list(acts[n % 4096]))
        n += 1
        if n % 10 == 0 and time.perf_counter() - t0 >= seconds:
            break
    return n, time.perf_counter() - t0


def port_worker(arg):
    scenario, n_agents, seconds, idx = arg
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import spec as ospec
    from oracle.mpe_loop import LoopEnv
    kw = {"n": n_agents} if scenario == "simple_spread" and n_agents != 3 else {}
    env = LoopEnv(ospec.by_name(scenario, **kw))
    np.random.seed(idx)
    A = env.spec.n_agents if hasattr(env, "spec") else n_agents
    acts = np.eye(5)[np.random.randint(0, 5, size=(4096, A))]
    n, t0 = 0, time.perf_counter()
    while True:
        if n % 25 == 0:
            env.reset()
        env.step(list(acts[n % 4096]))
        n += 1
        if n % 10 == 0 and time.perf_counter() - t0 >= seconds:
            break
    return n, time.perf_counter() - t0


def rate(worker, scenario, n_agents, seconds, procs):
    if procs == 1:
        n, t = worker((scenario, n_agents, seconds, 0))
        return n / t
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(worker, [(scenario, n_agents, seconds, i + 1) for i in range(procs)])
    return sum(r[0] for r in res) / max(r[1] for r in res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0)
    args = ap.parse_args()
    if not os.path.isdir("/root/reference/multiagent"):
        raise SystemExit("the reference tree is not here; this script runs in the build container only")
    cores = usable_cores()
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), platform.processor())
    out = {"what": "the unmodified reference's MultiAgentEnv.step timed in the build container (tools/time_reference.py); "
                   "uniform random one-hot moves, env.reset() every 25 steps",
           "cpu_model": cpu, "usable_cores": cores, "python": platform.python_version(),
           "numpy": __import__("numpy").__version__, "seconds_per_leg": args.seconds, "configs": {}}
    for key, scenario, n_agents, secs in (("simple", "simple", 1, args.seconds),
                                          ("simple_spread_n3", "simple_spread", 3, args.seconds),
                                          ("simple_tag", "simple_tag", 4, args.seconds),
                                          ("simple_spread_n64", "simple_spread", 64, max(args.seconds, 20.0))):
        row = {}
        for name, worker in (("reference", ref_worker), ("port_oracle_mpe_loop", port_worker)):
            one = rate(worker, scenario, n_agents, secs, 1)
            allc = rate(worker, scenario, n_agents, secs, cores)
            row[name] = {"env_steps_per_s_1_process": one, "env_steps_per_s_all_cores": allc, "processes": cores}
        row["port_over_reference_1_process"] = row["port_oracle_mpe_loop"]["env_steps_per_s_1_process"] / \
            row["reference"]["env_steps_per_s_1_process"]
        row["port_over_reference_all_cores"] = row["port_oracle_mpe_loop"]["env_steps_per_s_all_cores"] / \
            row["reference"]["env_steps_per_s_all_cores"]
        out["configs"][key] = row
        print(key, json.dumps(row), flush=True)
    with open(os.path.join(ROOT, "profiles", "cpu_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote profiles/cpu_reference.json")


if __name__ == "__main__":
    main()
