#!/usr/bin/env python3
"""Where a row-program step's cycles go, per wave (instrumented build libmpe_hip_rowsclock.so, -DMPE_ROWS_CLOCK).

    python -m multiagent_particle_envs_amd._build --ab rowsclock
    MPE_HIP_LIB=multiagent_particle_envs_amd/lib/libmpe_hip_rowsclock.so python tools/rows_clock.py [scenario]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi  # noqa: E402
import test_rowspec as tr  # noqa: E402

PHASES = ["entry -> state loads parked in LDS", "barrier 1", "World.step of own agents", "barrier 2 + new state to LDS / HBM + barrier 3",
          "observation programs + row flush", "reward programs", "reward / done stores"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "simple_spread"
    B = 65536
    env = tr.make_spec_env(name, B)
    env.reset()
    act = torch.nn.functional.one_hot(torch.randint(0, 5, (env.n, B), device="cuda"), 5).float().contiguous()
    stamps = torch.zeros((8, 16, 8), dtype=torch.int64, device="cuda")
    b = env._sets[0].bufs
    b.act, b.ids, b.u = act.data_ptr(), None, None
    b.force = stamps.data_ptr()
    L, st = _abi.lib(), _abi.raw_stream(env.world.device)
    for _ in range(50):
        L.mpe_step_rows(C.byref(env._desc), C.byref(b), env._prog.ref, B, st)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    W = min(env.n, 16)
    print("# %s, %d worlds, %d ops; shader-clock ticks per phase, workgroups 0-7 averaged (lane 0 of each wave)" % (name, B, env._prog.n_ops))
    for w in range(W):
        d = (s[:, w, 1:] - s[:, w, :-1]).mean(axis=0)
        tot = (s[:, w, 7] - s[:, w, 0]).mean()
        print("wave %d: total %7.0f ticks | %s" % (w, tot, " | ".join("%6.0f" % x for x in d)))
    print("phases: " + " | ".join(PHASES))
    t0 = s[:, :W, 0].min(axis=1)
    t7 = s[:, :W, 7].max(axis=1)
    print("workgroup span (first entry -> last exit): %s ticks" % " ".join("%.0f" % x for x in (t7 - t0)))


if __name__ == "__main__":
    main()
