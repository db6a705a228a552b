import sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import multiagent_particle_envs_amd as mpe
which = sys.argv[1]
B = 513
def acts_for(env, rs):
    return [torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, B)]).cuda() for _ in env.agents]
rs = np.random.RandomState(0)
env = mpe.make_env("simple_adversary", batch_size=B, seed=8, fused=False)
env.reset()
gs = mpe.GraphedStep(env, acts_for(env, rs))
print("captured", flush=True)
torch.cuda.synchronize()
print("synced", flush=True)
if which == "replay_only":
    gs.graph.replay(); torch.cuda.synchronize(); print("replayed", flush=True)
elif which == "step":
    out = gs.step(acts_for(env, rs)); torch.cuda.synchronize(); print("stepped", flush=True)
elif which == "getstate":
    env.world.get_state(); print("got state", flush=True)
    out = gs.step(acts_for(env, rs)); torch.cuda.synchronize(); print("stepped", flush=True)
elif which == "twin":
    e2 = mpe.make_env("simple_adversary", batch_size=B, seed=8, fused=False); e2.reset()
    a = acts_for(env, rs)
    out = gs.step(a); torch.cuda.synchronize(); print("stepped", flush=True)
    o2 = e2.step(a); torch.cuda.synchronize(); print("eager stepped", flush=True)
    print(torch.equal(out[0][0], o2[0][0]))
print(which, "OK")
