import sys, os, ctypes as C, torch
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'tests')]
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi
import test_rowspec as tr
sys.path.insert(0, os.path.join(ROOT,'tools'))
import rows_ablate as ra
B = 1 << 20
print("# simple_spread at %d worlds (HBM-resident: 431 MB per launch), us per launch, 100 back-to-back launches, best of 3" % B)
for compiled in (False, True):
    env = tr.make_spec_env("simple_spread", B)
    if compiled: assert env.compile_program()
    env._ensure_buffers()
    t = ra.time_env(env, B, n=100)
    print("   row program %-12s %7.2f us   -> %.2f of 8 TB/s on the 411 algorithmic bytes per env-step" % ("compiled in" if compiled else "interpreted", t, 411.0 * B / (t * 1e-6) / 8e12))
    del env; torch.cuda.empty_cache()
e = mpe.make_env("simple_spread", batch_size=B)
e.reset()
act = torch.nn.functional.one_hot(torch.randint(0, 5, (3, B), device="cuda"), 5).float().contiguous()
e.step(act); torch.cuda.synchronize()
best = None
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): e.step(act)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 10.0
    best = t if best is None else min(best, t)
print("   the fused kernel (env.step)  %7.2f us   -> %.2f" % (best, 411.0 * B / (best * 1e-6) / 8e12))
