"""Helper of tests/test_gpu_race.py::test_shared_reward_values_wait_for_a_late_wave, run as a subprocess so that MPE_ROWS_IMAGE_FLAGS
can select how the traced program's image is compiled: steps a 6-agent cooperative-navigation file (its rewards share the
distance of the nearest agent to every landmark: symtrace.shared_tasks -> traced_shared, six values through LDS) and prints a
SHA-256 over the rewards and rows of every step."""
import hashlib
import json
import sys

import numpy as np
import torch

import multiagent_particle_envs_amd as mpe

path, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
env = mpe.make_env(path, batch_size=B, seed=3)
assert env.traced and env._prog.struct.n_shared >= 2, (env.trace_fallback, env._prog.struct.n_shared if env.traced else None)
rs = np.random.RandomState(5)
env.reset()
env.world.pos.mul_(0.4)
h = hashlib.sha256()
for t in range(steps):
    act = torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(env.n, B))]).cuda()
    obs, rew, _, _ = env.step(act)
    for x in list(rew) + list(obs):
        h.update(x.detach().cpu().numpy().tobytes())
print("RACE_PROBE " + json.dumps({"sha": h.hexdigest(), "n_shared": int(env._prog.struct.n_shared)}))
