#!/usr/bin/env python3
"""What a USER scenario costs per step on each path (round-4): env-steps/s at 65 536 worlds of

    program   obs_spec / reward_spec -> mpe_step_rows: 1 launch per step, the program interpreted (eager env.step, and GraphedStep)
    compiled  the same program compiled in (env.compile_program(): mpe_rows_static_source -> hipcc --genco -> mpe_rows_load_image)
    generic   torch observation / reward callbacks over mpe_world_step: ~100 launches (eager, and GraphedStep)
    fused     the built-in's own kernel: 1 launch (built-ins only)

for the custom Corral scenario of tests/test_rowspec.py (no kernel of its own) and the built-ins written as specs.

    python tools/rowspec_rate.py [--batch 65536] [--steps 400] > profiles/r4_rowspec_rate.txt
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi  # noqa: E402
import test_rowspec as tr  # noqa: E402


def rate(step, acts, B, n, reset=None, every=25):
    for k in range(10):
        step(acts[k % len(acts)])
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(n):
            if reset is not None and k % every == 0:
                reset()
            step(acts[k % len(acts)])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return B * n / best, best / n * 1e6


def measure(label, env, B, n, out):
    rs = np.random.RandomState(0)
    acts = [tr.rand_actions(env, rs, B) for _ in range(4)]
    env.reset()
    if env.fused and not env._comm_kind:      # one [A, B, 5] tensor: the zero-copy form
        acts = [torch.stack(a).contiguous() for a in acts]
    elif env.fused and env._comm_kind:        # (moves, utterances): the batched form of the per-agent [move | utterance] rows
        acts = [tr.as_tuple(env, a, B) for a in acts]
    r, us = rate(env.step, acts, B, n, env.reset)
    out.append({"what": label + " eager env.step", "env_steps_per_s": r, "us_per_step": us})
    if EAGER_ONLY:
        return
    gs = mpe.GraphedStep(env, acts[0])
    r, us = rate(gs.step, acts, B, n)
    out.append({"what": label + " GraphedStep", "env_steps_per_s": r, "us_per_step": us})
    r, us = rate(lambda a: gs.graph.replay(), acts, B, n)
    out.append({"what": label + " graph replay only (actions resident)", "env_steps_per_s": r, "us_per_step": us})


EAGER_ONLY = False


def main():
    global EAGER_ONLY
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--scenarios", default="corral,simple_spread,simple_tag,simple_world_comm")
    ap.add_argument("--no-generic", action="store_true")
    ap.add_argument("--eager-only", action="store_true")
    ap.add_argument("--compiled", action="store_true", help="add a row per program with the program compiled in")
    args = ap.parse_args()
    EAGER_ONLY = args.eager_only
    B, out = args.batch, []
    for name in args.scenarios.split(","):
        if name == "corral":
            measure("corral  program (1 launch) ", tr.corral_env(B), B, args.steps, out)
            if args.compiled:
                e = tr.corral_env(B)
                t0 = time.perf_counter()
                assert e.compile_program()
                sys.stderr.write("corral: compile_program %.1f s\n" % (time.perf_counter() - t0))
                measure("corral  program COMPILED IN", e, B, args.steps, out)
            measure("corral  generic (torch callbacks)", tr.corral_env(B, fused=False), B, max(50, args.steps // 8), out)
        else:       # name[:key=value ...]: scenario kwargs (team sizes)
            parts = name.split(":")
            name, kw = parts[0], {k: int(v) for k, v in (p.split("=") for p in parts[1:])}
            tag = name.replace("simple_", "") + ("(" + ",".join(str(v) for v in kw.values()) + ")" if kw else "")
            e = mpe.make_env(name, batch_size=B, **kw)
            if e._prog is None and e.fused:
                measure("%-18s fused (1 launch)" % tag, e, B, args.steps, out)
            measure("%-18s program (1 launch) " % tag, tr.make_spec_env(name, B, scenario_kw=kw), B, args.steps, out)
            if args.compiled:
                e = tr.make_spec_env(name, B, scenario_kw=kw)
                t0 = time.perf_counter()
                try:
                    ok = e.compile_program()
                    sys.stderr.write("%s: compile_program %.1f s, %d ops\n" % (tag, time.perf_counter() - t0, e._prog.n_ops))
                    assert ok
                    measure("%-18s program COMPILED IN" % tag, e, B, args.steps, out)
                except _abi.MpeError as err:
                    sys.stderr.write("%s: not compiled: %s\n" % (tag, err))
            if not args.no_generic:
                measure("%-18s generic (torch callbacks)" % tag, mpe.make_env(name, batch_size=B, fused=False, **kw), B, max(50, args.steps // 8), out)
    print("# env-steps/s at %d worlds per path (tools/rowspec_rate.py); best of 3 x %d steps, reset every 25 in the eager rows" % (B, args.steps))
    for o in out:
        print("%-62s %10.1f M env-steps/s   %8.2f us / step" % (o["what"], o["env_steps_per_s"] / 1e6, o["us_per_step"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
