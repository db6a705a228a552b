import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multiagent_particle_envs_amd as mpe
from multiagent_particle_envs_amd import _abi
from multiagent_particle_envs_amd.rollout import StepServer, ServedRollout
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for trial in range(4):
    env = mpe.make_env("simple_spread", batch_size=B, seed=trial)
    moves = torch.zeros((8, 3, B, 5), device="cuda"); moves[..., 0] = 1
    srv = StepServer(env, moves, slots=2, episode_len=25, timeout_s=5.0)
    T = 500
    for mode in ("rings then wait", "ring+wait per step"):
        srv.start(T)
        torch.cuda.current_stream().synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if mode == "rings then wait":
            for _ in range(T):
                srv.ring()
            srv.wait()
        else:
            for _ in range(T):
                srv.ring(); srv.wait()
        e1.record()
        torch.cuda.synchronize()
        srv.check()
        print("trial", trial, "server stream", hex(srv.stream.cuda_stream), "prio", srv.stream.priority, mode, "%.2f us/step" % (e0.elapsed_time(e1) * 1e3 / T), srv.stream_probe, flush=True)
    # the graph path
    env2 = mpe.make_env("simple_spread", batch_size=B, seed=trial)
    roll = ServedRollout(env2, episode_len=25, graphs=True)
    roll.enqueue(100); torch.cuda.synchronize()
    t0 = time.perf_counter(); roll.enqueue(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    roll.srv.check()
    print("   ServedRollout graphs: %.2f us/step" % (dt * 1e6 / 2000), "server stream", hex(roll.srv.stream.cuda_stream), roll.srv.stream_probe, flush=True)
