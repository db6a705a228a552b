import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd.rollout import ServedRollout
scn, B, ahead, N = sys.argv[1], int(sys.argv[2]), sys.argv[3] in ("ahead", "episode"), int(sys.argv[4])
per_episode = sys.argv[3] == "episode"
res = []
for k in range(N):
    env2 = mpe.make_env(scn, batch_size=B, seed=1)
    roll = ServedRollout(env2, episode_len=25, graphs=True, ring_ahead=ahead, timeout_s=3.0, launch_per_episode=per_episode)
    roll.enqueue(100); torch.cuda.synchronize()
    t0 = time.perf_counter(); roll.enqueue(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    roll.srv.check()
    res.append((round(dt * 1e6 / 2000, 2), roll.srv.stream_probe["picked"], roll.srv.stream.priority, hex(roll.srv.stream.cuda_stream)[-5:], hex(roll.cmd.cuda_stream)[-5:]))
print(scn, B, sys.argv[3], "prio env", os.environ.get("MPE_SERVER_PRIORITY", "1"), res, flush=True)
