"""Reference-style scenario files (tests/refstyle/*.py, or any path): the traced path (symtrace.py: the file's callbacks compiled
into the step kernel) against the host path (refstyle.py: the callbacks per world on the host) -- agreement and env-steps/s."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import multiagent_particle_envs_amd as mpe  # noqa: E402


def actions(env, rs, B, dev):
    acts = []
    for a in env.agents:
        parts = []
        if a.movable:
            parts.append(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=B)])
        if not a.silent:
            parts.append(np.eye(env.world.dim_c, dtype=np.float32)[rs.randint(0, env.world.dim_c, size=B)])
        acts.append(torch.as_tensor(np.concatenate(parts, axis=1)).to(dev))
    return acts


def fast_actions(env, rs, B, dev):
    """The batched input forms of env.step: one [A, B, 5] tensor of moves (+ one [A, B, dim_c] tensor of utterances)."""
    A, dc = env.n, int(env.world.dim_c)
    moves = torch.as_tensor(np.eye(5, dtype=np.float32)[rs.randint(0, 5, size=(A, B))]).to(dev)
    if all(a.silent for a in env.agents):
        return moves
    words = torch.as_tensor(np.eye(dc, dtype=np.float32)[rs.randint(0, dc, size=(A, B))]).to(dev)
    return (moves, words)


def device_rates(env, worlds, n=500):
    """(us per step as a HIP graph of step launches with fresh moves [+ resets], us per step as 25-step rollouts with a trajectory)"""
    from multiagent_particle_envs_amd.rollout import RandomRollout, Trajectory
    EP = 25 if env._device_restart_ok else 0
    roll = RandomRollout(env, episode_len=EP, pool=25, regenerate=True)
    graph = roll.capture(n)
    graph.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.time()
        for _ in range(4):
            graph.replay()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / (4 * n)
        best = dt if best is None else min(best, dt)
    traj = Trajectory(env, 25)
    roll.fused(25, traj)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(40):
        roll.fused(25, traj)
    torch.cuda.synchronize()
    return best * 1e6, (time.time() - t0) / (40 * 25) * 1e6, EP


_NAV = '''
import numpy as np
from multiagent.core import World, Agent, Landmark
from multiagent.scenario import BaseScenario


class Scenario(BaseScenario):
    """cooperative navigation against the reference's contract, N agents and N landmarks (what a user gets who raises N in a copy
    of the reference's scenario; np.linalg.norm instead of sqrt(sum(square())))"""
    def make_world(self):
        world = World()
        world.dim_c = 2
        world.collaborative = True
        world.agents = [Agent() for _ in range(N_)]
        for i, agent in enumerate(world.agents):
            agent.name, agent.collide, agent.silent, agent.size = "agent %d" % i, True, True, 0.15
        world.landmarks = [Landmark() for _ in range(N_)]
        for i, lm in enumerate(world.landmarks):
            lm.name, lm.collide, lm.movable = "landmark %d" % i, False, False
        self.reset_world(world)
        return world

    def reset_world(self, world):
        for e in world.agents + world.landmarks:
            e.state.p_pos = np.random.uniform(-1, +1, world.dim_p)
            e.state.p_vel = np.zeros(world.dim_p)
        for a in world.agents:
            a.state.c = np.zeros(world.dim_c)

    def reward(self, agent, world):
        rew = 0
        for lm in world.landmarks:
            rew -= min(np.linalg.norm(a.state.p_pos - lm.state.p_pos) for a in world.agents)
        for a in world.agents:
            if np.linalg.norm(a.state.p_pos - agent.state.p_pos) < a.size + agent.size:
                rew -= 1
        return rew

    def observation(self, agent, world):
        lms = [lm.state.p_pos - agent.state.p_pos for lm in world.landmarks]
        others = [o.state.p_pos - agent.state.p_pos for o in world.agents if o is not agent]
        return np.concatenate([agent.state.p_vel, agent.state.p_pos] + lms + others + [np.zeros(2 * (len(world.agents) - 1))])
'''


def team_sizes(args):
    """How far straight-line traced code carries: N x N cooperative navigation as a reference-style file."""
    import tempfile
    for n in args.nav:
        path = os.path.join(tempfile.gettempdir(), "mpe_nav%d.py" % n)
        with open(path, "w") as fh:
            fh.write(_NAV.replace("N_", str(n)))
        t0 = time.time()
        env = mpe.make_env(path, batch_size=args.worlds)
        build = time.time() - t0
        if not env.traced:
            print("N=%-3d host path: %s" % (n, env.trace_fallback))
            continue
        t = env.scenario.t
        g, f, _ = device_rates(env, args.worlds, n=200)
        b = mpe.make_env("simple_spread", batch_size=args.worlds, num_agents=n)
        g2, f2, _ = device_rates(b, args.worlds, n=200)
        print("N=%-3d traced: build %.1f s (trace + verify + hipcc%s), %d graph nodes, %d statements; graph protocol %.2f us per step = %.3g "
              "env-steps/s, 25-step rollouts %.2f us per step   | built-in simple_spread N=%d : %.2f / %.2f us  (traced / built-in: %.2f / %.2f)"
              % (n, build, "" if build > 2 else ": cached", t.graph.count, len(env.scenario.row_source(env.world).splitlines()), g,
                 args.worlds / (g * 1e-6), f, n, g2, f2, g / g2, f / f2))
        del env, b
        torch.cuda.empty_cache()


def special(args):
    import json
    from multiagent_particle_envs_amd import refstyle
    here = os.path.dirname(os.path.abspath(__file__))
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)
    envs = []
    for name in args.json:
        with open(os.path.join(here, "..", "tests", "golden", "traced_%s.json" % name)) as fh:
            envs.append(("%s.py (the reference's file, traced)" % name, refstyle.make_traced_env(json.load(fh), args.worlds), name))
    for path in args.files:
        e = mpe.make_env(path, batch_size=args.worlds)
        assert e.traced, e.trace_fallback
        envs.append((os.path.basename(path) + " (traced)", e, None))
    if args.profile_steps:
        for label, env, _ in envs:
            act = [fast_actions(env, rs, args.worlds, dev) for _ in range(4)]
            env.reset()
            for k in range(args.profile_steps):
                env.step(act[k % 4])
            torch.cuda.synchronize()
            print("%-44s %d eager env.step calls at %d worlds" % (label, args.profile_steps, args.worlds))
        return
    for label, env, name in envs:
        g, f, EP = device_rates(env, args.worlds)
        line = "%-44s graph of step launches: %.2f us per step = %.3g env-steps/s; 25-step rollouts: %.2f us per step" % (label, g, args.worlds / (g * 1e-6), f)
        if name is not None:
            b = mpe.make_env(name, batch_size=args.worlds)
            g2, f2, _ = device_rates(b, args.worlds)
            line += "   | built-in %s (hand-fused kernel): %.2f / %.2f us  (traced / fused: %.2f / %.2f)" % (name, g2, f2, g / g2, f / f2)
        print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--check-worlds", type=int, default=256)
    ap.add_argument("--worlds", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--host-worlds", type=int, default=256)
    ap.add_argument("--json", action="append", default=[], metavar="NAME",
                    help="a committed trace of one of the reference's nine files (tests/golden/traced_NAME.json): its graph-protocol and "
                         "rollout rates beside the built-in scenario of that name (the hand-fused kernel)")
    ap.add_argument("--nav", action="append", type=int, default=[], metavar="N",
                    help="cooperative navigation with N agents and N landmarks written as a reference-style file (generated into /tmp): "
                         "trace / compile times and rates beside the built-in simple_spread at that size")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="only: build the traced env(s) at --worlds and run this many eager env.step calls (for rocprofv3)")
    args = ap.parse_args()
    if args.nav:
        return team_sizes(args)
    if args.json or args.profile_steps:
        return special(args)
    here = os.path.dirname(os.path.abspath(__file__))
    files = args.files or [os.path.join(here, "..", "tests", "refstyle", f) for f in ("herd.py", "relay.py", "convoy.py", "patrol.py")]
    dev = torch.device("cuda", 0)
    for path in files:
        name = os.path.basename(path)
        t0 = time.time()
        a = mpe.make_env(path, batch_size=args.check_worlds, seed=1)
        t_build = time.time() - t0
        if not a.traced:
            print("%-24s host path only: %s" % (name, a.trace_fallback))
            continue
        b = mpe.make_env(path, batch_size=args.check_worlds, seed=1, traced=False)
        rs = np.random.RandomState(0)
        seeds = list(range(100, 100 + args.check_worlds))
        oa, ob = a.reset(seeds=seeds), b.reset(seeds=seeds)
        worst = max(float((x.double() - y.double()).abs().max()) for x, y in zip(oa, ob))
        for t in range(12):
            if t == 4:      # crowd the worlds: contacts
                pa, va = a.world.get_state(all_entities=True)
                a.world.set_state(pa * 0.3, va)
                b.world.set_state(pa * 0.3, va)
            act = actions(a, rs, args.check_worlds, dev)
            (oa, ra, da, _), (ob, rb, db, _) = a.step(act), b.step(act)
            pa, va = a.world.get_state(all_entities=True)
            pb, _ = b.world.get_state(all_entities=True)
            assert np.array_equal(pa, pb), "state"
            # (fp32 kernel vs the fp64 host callbacks: compared outside a 2e-6 band around the file's own thresholds)
            from multiagent_particle_envs_amd import symtrace
            tr = a.scenario.t
            Cw = np.zeros((args.check_worlds, tr.A, tr.dim_c))
            for i, ag in enumerate(a.world.agents):
                if tr.dim_c and not ag.silent:
                    Cw[:, i] = a._comm[i].cpu().numpy()
            K = a.world.choice_i32.cpu().numpy().T if tr.pops else np.zeros((args.check_worlds, 0), np.int64)
            ok = torch.as_tensor(symtrace.decision_margin([n for row in tr.obs for n in row] + list(tr.rew), args.check_worlds,
                                                          P=pa.astype(np.float64), V=va.astype(np.float64), Cw=Cw, K=K) > 2e-6)
            for x, y in zip(oa, ob):
                worst = max(worst, float(((x.double() - y.double()).abs() / y.double().abs().clamp(min=1.0))[ok].max()))
            for x, y in zip(ra, rb):
                e = (x.double() - y.double()).abs() / y.double().abs().clamp(min=1.0)
                worst = max(worst, float(e[ok].max()))
        # rates
        big = mpe.make_env(path, batch_size=args.worlds, seed=2)
        act = [fast_actions(big, rs, args.worlds, dev) for _ in range(4)]
        big.reset()
        for k in range(10):
            big.step(act[k % 4])
        torch.cuda.synchronize()
        t0 = time.time()
        for k in range(args.steps):
            big.step(act[k % 4])
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.steps
        # the device-bound rates: a HIP graph of consecutive step launches with fresh block-drawn moves (bench.py's protocol), and
        # whole episodes per launch (mpe_rollout_rows)
        from multiagent_particle_envs_amd.rollout import RandomRollout, Trajectory
        EP = 25 if big._device_restart_ok else 0
        roll = RandomRollout(big, episode_len=EP, pool=25, regenerate=True)
        n = 500
        graph = roll.capture(n)
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(4):
            graph.replay()
        torch.cuda.synchronize()
        gdt = (time.time() - t0) / (4 * n)
        traj = Trajectory(big, 25)
        roll.fused(25, traj)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(40):
            roll.fused(25, traj)
        torch.cuda.synchronize()
        fdt = (time.time() - t0) / (40 * 25)
        print("%-24s graph of step launches (fresh moves%s): %.2f us per step = %.3g env-steps/s; %d-step rollouts in one launch "
              "(trajectory kept): %.2f us per step = %.3g env-steps/s" % (name, ", reset every 25 steps" if EP else ", no resets: reset_world "
              "is not the device draw", gdt * 1e6, args.worlds / gdt, 25, fdt * 1e6, args.worlds / fdt))
        hb = mpe.make_env(path, batch_size=args.host_worlds, seed=2, traced=False)
        hact = actions(hb, rs, args.host_worlds, dev)
        hb.reset()
        hb.step(hact)
        t0 = time.time()
        for k in range(3):
            hb.step(hact)
        torch.cuda.synchronize()
        hdt = (time.time() - t0) / 3
        print("%-24s traced == host path within %.2e over 12 steps x %d worlds; build (trace + verify + hipcc) %.1f s; traced: %.2f us per "
              "env.step at %d worlds = %.3g env-steps/s; host path: %.1f ms per step at %d worlds = %.3g env-steps/s  (x %.0f)"
              % (name, worst, args.check_worlds, t_build, dt * 1e6, args.worlds, args.worlds / dt, hdt * 1e3, args.host_worlds,
                 args.host_worlds / hdt, (args.worlds / dt) / (args.host_worlds / hdt)))


if __name__ == "__main__":
    main()
