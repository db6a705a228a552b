"""Helper of tests/test_gpu_race.py, run as a subprocess so that MPE_HIP_LIB can select the library build:
steps a few scenarios with crowded worlds and prints one SHA-256 per scenario over everything the step wrote
(pos, vel, obs, rew after every step).  Same seeds, same moves in every process."""
import hashlib
import json
import sys

import numpy as np
import torch

import multiagent_particle_envs_amd as mpe


def run(name, kw, B, steps):
    env = mpe.make_env(name, batch_size=B, seed=7, **kw)
    assert env.fused
    w = env.world
    A, E = len(w.agents), len(w.entities)
    rs = np.random.RandomState(3)
    pos = rs.uniform(-1, 1, (B, E, 2)).astype(np.float32) * 0.25     # crowded: most agent pairs in contact
    w.set_state(pos, rs.uniform(-0.5, 0.5, (B, A, 2)).astype(np.float32))
    h = hashlib.sha256()
    for t in range(steps):
        acts = []
        for agent in env.agents:
            parts = []
            if agent.movable:
                parts.append(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=B)])
            if not agent.silent:
                parts.append(np.eye(w.dim_c, dtype=np.float32)[rs.randint(0, w.dim_c, size=B)])
            acts.append(torch.as_tensor(np.concatenate(parts, axis=1)).cuda())
        obs, rew, done, _ = env.step(acts)
        for x in [w.pos, w.vel] + list(obs) + list(rew):
            h.update(x.detach().cpu().numpy().tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    out = {}
    for name, kw in (("simple_spread", {}), ("simple_tag", {}), ("simple_world_comm", {}), ("simple_spread", {"num_agents": 6})):
        out["%s%s" % (name, kw.get("num_agents", ""))] = run(name, kw, int(sys.argv[1]), int(sys.argv[2]))
    print("RACE_PROBE " + json.dumps(out))
