#!/usr/bin/env python3
"""Step rate of an env whose Scenario overrides `observation` only: partial fusion (one launch + the Python rows) against
the generic path (mpe_world_step + every callback in torch), eager env.step() and GraphedStep replay, 65 536 worlds."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multiagent_particle_envs_amd as mpe  # noqa: E402


def rate(env, acts, n, graphed):
    step = mpe.GraphedStep(env, acts).step if graphed else (lambda: env.step(acts))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def main():
    B = 65536
    for name in ("simple_spread", "simple_tag"):
        Base = mpe.scenarios.load(name + ".py").Scenario

        class Mine(Base):
            def observation(self, agent, world):
                return torch.cat([Base.observation(self, agent, world), agent.state.p_pos.norm(dim=1, keepdim=True)], dim=1)
        out = []
        for fused in (None, False):
            sc = Mine()
            w = sc.make_world(batch_size=B)
            sc.reset_world(w)
            env = mpe.MultiAgentEnv(w, sc.reset_world, sc.reward, sc.observation, fused=fused)
            acts = [torch.nn.functional.one_hot(torch.randint(0, 5, (B,), device="cuda"), 5).float() for _ in env.agents]
            out.append("%s: eager %.0f steps/s, graphed %.0f steps/s" % ("partial fusion" if env.fused else "generic path",
                                                                        rate(env, acts, 300, False), rate(env, acts, 300, True)))
        full = mpe.make_env(name, batch_size=B)
        acts = [torch.nn.functional.one_hot(torch.randint(0, 5, (B,), device="cuda"), 5).float() for _ in full.agents]
        out.append("built-in (fully fused): eager %.0f steps/s" % rate(full, acts, 300, False))
        print("%s, %d worlds, observation overridden | %s" % (name, B, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
