"""An unmodified reference-style scenario file at kernel speed.

    python examples/run_reference_style_file.py [path/to/scenario.py] [worlds]

The file is written against the REFERENCE's plug-in contract (multiagent/scenario.py:4-10: `from multiagent.core import World,
Agent, Landmark`, `make_world(self)`, `reset_world(self, world)`, NumPy `reward` / `observation` per world) -- any of the
reference's own nine scenario files works, and so does tests/refstyle/convoy.py (the default).  make_env traces its callbacks into a
compiled row program (multiagent_particle_envs_amd/symtrace.py): `env.step` is one kernel launch for all worlds.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiagent_particle_envs_amd import make_env  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "refstyle", "convoy.py")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536

env = make_env(path, batch_size=B)                      # the same call as for a built-in scenario
if env.traced:
    print(env.scenario.report())
else:
    print("on the host path (the file's callbacks per world):", env.trace_fallback)
    B = min(B, 256)
    env = make_env(path, batch_size=B, traced=False)

obs_n = env.reset()
print("observation shapes:", [tuple(o.shape) for o in obs_n])
moves = torch.nn.functional.one_hot(torch.randint(0, 5, (env.n, B)), 5).float().cuda()     # [A, B, 5] one-hot moves
action = moves if all(a.silent for a in env.agents) else \
    (moves, torch.nn.functional.one_hot(torch.randint(0, env.world.dim_c, (env.n, B)), env.world.dim_c).float().cuda())
if not env.traced:
    action = [torch.cat(([moves[i]] if a.movable else []) + ([action[1][i]] if not a.silent else []), dim=1) for i, a in enumerate(env.agents)]
for _ in range(10):
    obs_n, reward_n, done_n, info_n = env.step(action)
torch.cuda.synchronize()
steps = 500 if env.traced else 5
t0 = time.perf_counter()
for k in range(steps):
    if k % 25 == 0:
        env.reset()
    obs_n, reward_n, done_n, info_n = env.step(action)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("%d worlds: %.1f us per env.step = %.3g env-steps/s; mean reward of agent 0: %.4f" % (B, dt * 1e6, B / dt, float(reward_n[0].mean())))
