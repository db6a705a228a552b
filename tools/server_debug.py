import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd.rollout import StepServer
from test_gpu_server import reference_steps
name, kw, B, EP = (sys.argv[1] if len(sys.argv) > 1 else "simple_spread"), {}, int(sys.argv[2]) if len(sys.argv) > 2 else 4096, 25
T, ring = 3, 16
moves, ref = reference_steps(name, kw, B, T, EP, ring)
env = mpe.make_env(name, batch_size=B, seed=3, **kw)
srv = StepServer(env, moves, slots=T, episode_len=EP, timeout_s=5.0)
srv.start(T); srv.ring(T); srv.join(); torch.cuda.synchronize(); srv.check()
for g in range(T):
    o_s, r_s, d_s = srv.outputs(g)
    obs, rew, done, pos, vel, ch = ref[g]
    for i in range(len(obs)):
        bad = (o_s[i] != obs[i])
        w = bad.any(dim=1).nonzero().flatten()
        c = bad.any(dim=0).nonzero().flatten()
        print("step", g, "agent", i, "worlds wrong", len(w), w[:12].tolist(), "cols", c.tolist()[:18])
        if len(w):
            k = int(w[0]); print("   world", k, "served", [round(x, 4) for x in o_s[i][k].tolist()]); print("   ref   ", [round(x, 4) for x in obs[i][k].tolist()], "move", moves[g % ring][i][k].tolist())
    print("   rew wrong", int((r_s != rew).sum()))
if name == "simple_tag":
    g = 0
    o_s, r_s, d_s = srv.outputs(g)
    obs = ref[g][0]
    for k in (3, 5, 8):
        print("world", k)
        for i in range(4):
            print("  agent", i, "served", [round(x, 4) for x in o_s[i][k].tolist()])
            print("          ref   ", [round(x, 4) for x in obs[i][k].tolist()])
    # which (lane, col) pairs of agent 3 differ, over the first wave
    bad = (o_s[3][:64] != obs[3][:64])
    print("agent 3, wave 0: bad (lane, col):", [(int(a), int(b)) for a, b in bad.nonzero().tolist()])
    print("flat index of bad floats in the wave's 64 x 14 block:", sorted(set(int(a) * 14 + int(b) for a, b in bad.nonzero().tolist())))
