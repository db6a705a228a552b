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
    env = mpe.make_env(name, batch_size=B, seed=7, compile_program=False, **kw)      # (a row program: interpreted by THIS library build)
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


def run_rollout(name, kw, B, steps):
    """a row-program env's fused T-step rollout (mpe_rollout_rows): every step's rows and rewards, the final state"""
    from multiagent_particle_envs_amd.rollout import RandomRollout, Trajectory
    env = mpe.make_env(name, batch_size=B, seed=7, compile_program=False, **kw)
    assert env.fused and env._prog is not None
    rr = RandomRollout(env, episode_len=0, pool=2, regenerate=False)
    w = env.world
    rs = np.random.RandomState(3)
    w.set_state(rs.uniform(-1, 1, (B, len(w.entities), 2)).astype(np.float32) * 0.25, rs.uniform(-0.5, 0.5, (B, len(w.agents), 2)).astype(np.float32))
    traj = Trajectory(env, steps)
    rr.fused(steps, traj)
    h = hashlib.sha256()
    for x in [w.pos, w.vel, traj.obs_flat, traj.rew]:
        h.update(x.detach().cpu().numpy().tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    out = {}
    import os
    corral = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "corral.py")   # colliding agents, steps
    for name, kw in (("simple_spread", {}), ("simple_tag", {}), ("simple_world_comm", {}), ("simple_spread", {"num_agents": 6}),   # through
                     (corral, {})):                                                                                # its row program (k_rows)
        out["%s%s" % (os.path.basename(name), kw.get("num_agents", ""))] = run(name, kw, int(sys.argv[1]), int(sys.argv[2]))
    out["rollout_corral"] = run_rollout(corral, {}, int(sys.argv[1]), int(sys.argv[2]))
    print("RACE_PROBE " + json.dumps(out))
