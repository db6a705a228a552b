import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import multiagent_particle_envs_amd as mpe
which = sys.argv[1]
B = 513
env = mpe.make_env("simple_adversary", batch_size=B, seed=8, fused=False)
env.reset()
rs = np.random.RandomState(0)
acts = [torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, B)]).cuda() for _ in env.agents]
def body():
    if which == "set_action":
        for i, a in enumerate(env.agents): env._set_action(acts[i], a, env.action_space[i])
    elif which == "world_step":
        env.world.step()
    elif which == "obs":
        return [env._get_obs(a) for a in env.agents]
    elif which == "obs0":
        return env._get_obs(env.agents[0])
    elif which == "obs1":
        return env._get_obs(env.agents[1])
    elif which == "rew":
        return [env._get_reward(a) for a in env.agents]
    elif which == "all":
        return env.step(acts)
for i, a in enumerate(env.agents): env._set_action(acts[i], a, env.action_space[i])
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = body()
torch.cuda.current_stream().wait_stream(side)
g.replay(); torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print(which, "OK")
