#!/usr/bin/env python3
"""C4 (simple_spread N=64, B=4096) against the reference's own recorded worlds: WHERE the margin goes (round-5 verdict #3).

tests/test_gpu_parity.py::test_spread64_reference_worlds_inside_the_real_grid hands the kernel the fp32 rounding of an fp64
golden state while the reference stepped the unrounded one, so the recorded 9.67e-6 (of a 1e-5 bar) mixes the kernel's
arithmetic error with the input rounding (x contact stiffness x dt x up to 63 partners).  This tool separates them, on the
same 64 worlds inside the same 4096-world grid, per recorded step:

    gpu  vs  reference golden (fp64 from the UNROUNDED state)      -- what the test measures
    gpu  vs  fp64 oracle stepped from the fp32-ROUNDED state        -- the kernel's arithmetic error alone
    that oracle  vs  the reference golden                           -- the input rounding alone
    fp32 NumPy oracle (the reference's arithmetic in float32)  vs  the fp64 oracle, both from the rounded state
                                                                    -- what ANY fp32 evaluation in the reference's order costs

and prints the step kernel's time per launch for the library in use (MPE_HIP_LIB: the product build, or the
-DMPE_CONTACT_EXACT build of mpe_wide.hip -- contact_force in the reference's operation order with IEEE sqrt / divisions).

    python tools/c4_parity_ab.py                      # product library
    tools/ab_build.sh exact wide -DMPE_CONTACT_EXACT=1 && MPE_HIP_LIB=.../libmpe_hip_ab_exact.so python tools/c4_parity_ab.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiagent_particle_envs_amd as mpe  # noqa: E402
from oracle import spec as ospec  # noqa: E402
from oracle.mpe_batched import BatchedOracle  # noqa: E402


def scaled(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    e = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    k = np.unravel_index(int(e.argmax()), e.shape)
    return float(e.max()), k, float(b[k])


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "simple_spread_n64_w64.npz")))
    T, W, N = g["rew"].shape
    B = 4096
    spec = ospec.simple_spread(N)
    slots = (np.arange(W) * 61 + 5) % B
    rs = np.random.RandomState(8)
    pos = rs.uniform(-1, 1, (B, 2 * N, 2))
    vel = np.zeros((B, N, 2))
    env = mpe.make_env("simple_spread", batch_size=B, num_agents=N, benchmark=True)
    lib = os.path.basename(os.environ.get("MPE_HIP_LIB", "libmpe_hip.so"))
    print("# library: %s   worlds %d of %d, %d recorded steps" % (lib, W, B, T))
    worst = {"gpu_vs_reference": 0.0, "gpu_vs_oracle_from_rounded_state": 0.0, "rounding_alone": 0.0, "numpy_fp32_vs_oracle": 0.0}
    for t in range(T):
        p0, v0 = (g["pos0"], g["vel0"]) if t == 0 else (g["pos"][t - 1], g["vel"][t - 1])
        pos[slots], vel[slots] = p0, v0
        env.world.set_state(pos, vel)
        act = np.eye(5)[rs.randint(0, 5, size=(N, B))]
        act[:, slots] = np.transpose(g["act"][t], (1, 0, 2))
        obs_n, rew_n, _, _ = env.step(torch.as_tensor(act, dtype=torch.float32).cuda().contiguous())
        pg, vg = env.world.get_state()
        pg, vg = pg.astype(np.float64), vg.astype(np.float64)
        rew = np.stack([r.cpu().numpy() for r in rew_n], axis=1)[slots]
        # the control: fp64 (and fp32 NumPy) oracle from the fp32-ROUNDED state -- exactly what the kernel was handed
        p32, v32 = p0.astype(np.float32), v0.astype(np.float32)
        out = {}
        for name, dt in (("f64", np.float64), ("f32", np.float32)):
            o = BatchedOracle(spec, W, dtype=dt)
            o.set_state(p32.astype(np.float64), v32.astype(np.float64))
            _, r_o, _, _ = o.step(act[:, slots])
            out[name] = (np.asarray(o.pos, np.float64), np.asarray(o.vel, np.float64), np.stack(r_o, axis=1))
        po, vo, ro = out["f64"]
        rows = []
        for what, a, b in (("pos", pg[slots], g["pos"][t]), ("vel", vg[slots], g["vel"][t]), ("rew", rew, g["rew"][t])):
            rows.append(("gpu_vs_reference", what) + scaled(a, b))
        for what, a, b in (("pos", pg[slots], po), ("vel", vg[slots], vo[:, :N]), ("rew", rew, ro)):
            rows.append(("gpu_vs_oracle_from_rounded_state", what) + scaled(a, b))
        for what, a, b in (("pos", po, g["pos"][t]), ("vel", vo[:, :N], g["vel"][t]), ("rew", ro, g["rew"][t])):
            rows.append(("rounding_alone", what) + scaled(a, b))
        p3, v3, r3 = out["f32"]
        if not np.array_equal(p3, po):
            for what, a, b in (("pos", p3, po), ("vel", v3[:, :N], vo[:, :N]), ("rew", r3, ro)):
                rows.append(("numpy_fp32_vs_oracle", what) + scaled(a, b))
        for key, what, e, k, ref in rows:
            worst[key] = max(worst[key], e)
            print("t=%d  %-34s %-4s max scaled err %.3e at (world, agent[, xy]) %s  ref value %+.4f" % (t, key, what, e, tuple(int(x) for x in k), ref))
    print("# worst over the %d steps:" % T)
    for k, v in worst.items():
        print("#   %-34s %.3e" % (k, v))
    # the step kernel's time per launch (graph-replayed dependent launches, two-point slope), probe off: this allocation as it is
    import bench
    leg = bench.Leg(mpe, "simple_spread", 64, 4096, 25, 0, 1, 0)
    k = leg.kernel_time_us(torch, "graph", n=100)
    print("# k_duo<4> per launch: %.2f us  (placement probe: %s)" % (k, leg.env.placement_probe))


if __name__ == "__main__":
    main()
