#!/usr/bin/env python3
"""Where does a row-program step's time go?  The simple_spread programs with parts removed, 65 536 worlds, HIP-event time of
400 back-to-back mpe_step_rows launches (tools/rows_ablate.py > profiles/r4_rows_ablation.txt)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi, rowspec  # noqa: E402
import test_rowspec as tr  # noqa: E402


def time_env(env, B, n=400):
    env.reset()
    act = torch.nn.functional.one_hot(torch.randint(0, 5, (env.n, B), device="cuda"), 5).float().contiguous()
    out = env._sets[0]
    b = out.bufs
    b.act, b.ids, b.u = act.data_ptr(), None, None
    L, st = _abi.lib(), _abi.raw_stream(env.world.device)
    fn = lambda: L.mpe_step_rows(C.byref(env._desc), C.byref(b), env._prog.ref, B, st)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / n
        best = t if best is None else min(best, t)
    return best


def variant(name, B, keep_obs, keep_rew, scenario_kw=None):
    sc = tr.spec_scenario(name)
    base_specs = sc._specs

    def specs(world):
        obs, rew, rg = rowspec.builtin_specs(name, world)
        if not keep_obs:
            obs = [rowspec.ObsSpec(world, a).vel() for a in world.agents]       # one cheap op: a row must have some width
        elif keep_obs == "half":
            for o in obs:
                k = len(o.ops) // 2
                o.width -= sum(2 for _ in o.ops[k:])          # (all spread ops after the first two are 2 or 1 wide; recomputed below)
            obs = []
            for a in world.agents:
                o = rowspec.ObsSpec(world, a).vel().pos()
                for l in world.landmarks:
                    o.rel(l)
                obs.append(o)
        if not keep_rew:
            rew = [rowspec.RewardSpec(world, a).value(1.0).add(1.0) for a in world.agents]
        return obs, rew, rg
    sc._specs = lambda world: specs(world)
    w = sc.make_world(batch_size=B, **(scenario_kw or {}))
    sc._cache = None

    class S2(type(sc)):
        def _specs(self, world):
            if getattr(self, "_c2", None) is None:
                self._c2 = specs(world)
            return self._c2
    s2 = S2()
    w = s2.make_world(batch_size=B, **(scenario_kw or {}))
    s2.reset_world(w)
    env = mpe.MultiAgentEnv(w, s2.reset_world, s2.reward, s2.observation, compile_program=False)      # interpreted until asked
    env._ensure_buffers()
    return env


def main():
    B = 65536
    print("# mpe_step_rows at %d worlds, us per launch (400 back-to-back launches, best of 3)" % B)
    for name, kw in (("simple_spread", None), ("simple_tag", None), ("simple_adversary", {"num_agents": 6, "num_adversaries": 2})):
        rows = []
        for label, ko, kr in (("World.step + 2-column rows + constant rewards", False, False), ("+ the observation programs", True, False),
                              ("+ the reward programs (= the full step)", True, True), ("reward programs without the observation programs", False, True)):
            env = variant(name, B, ko, kr, kw)
            rows.append((label, time_env(env, B), env._prog.n_ops))
            if ko and kr:      # the full step once more, the program compiled in (env.compile_program())
                try:
                    if env.compile_program():
                        rows.append(("   the same, the program COMPILED IN", time_env(env, B), env._prog.n_ops))
                except _abi.MpeError as err:
                    rows.append(("   (not compiled: %s)" % str(err)[:60], float("nan"), env._prog.n_ops))
        fused = None
        try:
            e = mpe.make_env(name, batch_size=B, **(kw or {}))
            if e._prog is None and e.fused:
                e.reset()
                act = torch.nn.functional.one_hot(torch.randint(0, 5, (e.n, B), device="cuda"), 5).float().contiguous()
                e.step(act)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(400):
                    e.step(act)
                e1.record()
                torch.cuda.synchronize()
                fused = e0.elapsed_time(e1) * 1e3 / 400
        except Exception:
            pass
        print("%s %s" % (name, kw or ""))
        for label, us, nops in rows:
            print("   %-58s %7.2f us   (%d ops)" % (label, us, nops))
        if fused:
            print("   %-58s %7.2f us" % ("the scenario's own fused kernel (env.step)", fused))


if __name__ == "__main__":
    main()
