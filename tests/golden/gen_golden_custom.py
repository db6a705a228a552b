#!/usr/bin/env python3
"""Golden vectors for CUSTOMISED entity / world constants, recorded from the unmodified reference
(runs only in the build container, like gen_golden.py).

    python tests/golden/gen_golden_custom.py        # writes tests/golden/custom_simple_tag.npz, custom_simple_spread.npz
    python tests/golden/gen_golden_custom.py --f3   # writes tests/golden/f3c_<scenario>.npz for the six other scenarios
    python tests/golden/gen_golden_custom.py --modes   # mode_force_discrete.npz, mode_continuous.npz (_set_action modes)

The reference keeps sizes, masses, collide flags, speed limits, action gains and the integration constants as plain
attributes (core.py:27-51, 94-99) that a user may change after make_world; the step must honour them.  The
constants used are stored in the .npz next to the trajectories."""
import os
import numpy as np

import gen_golden as G   # sets up the import path / gym stub, provides record()

HERE = os.path.dirname(os.path.abspath(__file__))


def customise(env, seed, world_consts):
    r = np.random.RandomState(seed)
    w = env.world
    consts = {"size": [], "mass": [], "collide": [], "max_speed": [], "accel": []}
    for e in w.entities:
        e.size = float(r.uniform(0.03, 0.2))
        e.initial_mass = float(r.uniform(0.5, 2.0))
        if r.rand() < 0.3:
            e.collide = not e.collide
        consts["size"].append(e.size)
        consts["mass"].append(e.mass)
        consts["collide"].append(bool(e.collide))
    for a in w.agents:
        a.max_speed = None if r.rand() < 0.4 else float(r.uniform(0.4, 1.5))
        a.accel = None if r.rand() < 0.4 else float(r.uniform(2.0, 6.0))
        consts["max_speed"].append(-1.0 if a.max_speed is None else a.max_speed)    # -1 encodes None
        consts["accel"].append(-1.0 if a.accel is None else a.accel)
    w.dt, w.damping, w.contact_force, w.contact_margin = world_consts
    out = {"c_" + k: np.array(v) for k, v in consts.items()}
    out["c_world"] = np.array(world_consts, dtype=np.float64)
    return out


def main():
    for name, seed, wc in (("simple_tag", 11, (0.05, 0.4, 250.0, 4e-3)), ("simple_spread", 12, (0.1, 0.25, 1e2, 1e-3))):
        env = G.make_env(name, benchmark=True)
        consts = customise(env, seed, wc)
        data = G.record(name, env, list(range(300, 324)), 12, squeeze_every=2, squeeze=0.3)
        data.update(consts)
        path = os.path.join(HERE, "custom_%s.npz" % name)
        np.savez_compressed(path, **data)
        print("%-28s %8.1f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024.0))


def main_f3():
    """The six other scenarios with customised constants: tests/golden/f3c_<scenario>.npz."""
    import gen_golden_scenarios as S
    for k, name in enumerate(("simple_adversary", "simple_push", "simple_speaker_listener", "simple_reference",
                              "simple_crypto", "simple_world_comm")):
        wc = (0.05, 0.4, 250.0, 4e-3) if k % 2 else (0.1, 0.25, 1e2, 1e-3)
        data = S.record(name, list(range(500, 516)), 8, squeeze_every=2, squeeze=0.3,
                        customise=lambda env, k=k, wc=wc: customise(env, 20 + k, wc))
        path = os.path.join(HERE, "f3c_%s.npz" % name)
        np.savez_compressed(path, **data)
        print("%-28s %8.1f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024.0))


def main_modes():
    """_set_action modes not covered elsewhere: force_discrete_action (environment.py:169-172, argmax -> one-hot, the
    caller's rows are soft) and a continuous action space (environment.py:176-177: u = action itself, 2 numbers)."""
    env = G.make_env("simple_spread", benchmark=True)
    env.force_discrete_action = True
    data = G.record("simple_spread", env, list(range(700, 716)), 8, squeeze_every=2, squeeze=0.3, soft_every=1)
    np.savez_compressed(os.path.join(HERE, "mode_force_discrete.npz"), **data)
    env = G.make_env("simple_tag")
    env.discrete_action_space = False
    W, T, A = 16, 8, env.n
    rng = np.random.RandomState(77)
    out = {"pos0": np.zeros((W, 6, 2)), "vel0": np.zeros((W, A, 2)), "act2": np.zeros((T, W, A, 2)),
           "pos": np.zeros((T, W, 6, 2)), "vel": np.zeros((T, W, A, 2)), "rew": np.zeros((T, W, A))}
    for i in range(A):
        out["obs%d" % i] = np.zeros((T, W, env.observation_space[i].shape[0]))
    for w in range(W):
        np.random.seed(800 + w)
        env.reset()
        if w % 2:
            for ent in env.world.entities:
                ent.state.p_pos = ent.state.p_pos * 0.3
        out["pos0"][w], out["vel0"][w] = G.state_of(env)
        for t in range(T):
            act = rng.uniform(-1, 1, (A, 2))
            out["act2"][t, w] = act
            obs, rew, done, info = env.step([a.copy() for a in act])
            out["pos"][t, w], out["vel"][t, w] = G.state_of(env)
            out["rew"][t, w] = rew
            for i in range(A):
                out["obs%d" % i][t, w] = obs[i]
    np.savez_compressed(os.path.join(HERE, "mode_continuous.npz"), **out)
    print("mode_force_discrete.npz, mode_continuous.npz written")


def main_noise():
    """Agent.u_noise / Agent.c_noise (core.py:138, :176): Gaussian noise drawn from the process-global np.random inside
    World.step -- the u draws (movable agents, in order) before the physics, the c draws (speaking agents) after.  Each
    world: np.random.seed(s); env.reset(); T steps; the noise and the next reset come out of that one stream."""
    import gen_golden_scenarios as S   # noqa: F401  (import path for the comm scenarios' action rows)
    for name, setup in (("simple_spread", lambda env: [setattr(a, "u_noise", 0.3) for a in env.world.agents]),
                        ("simple_reference", lambda env: [setattr(env.world.agents[0], "u_noise", 0.1)] +
                                                         [setattr(a, "c_noise", 0.2) for a in env.world.agents])):
        env = G.make_env(name)
        setup(env)
        W, T, A = 6, 6, env.n
        E = len(env.world.entities)
        rng = np.random.RandomState(31)
        widths = [env.action_space[i].n if hasattr(env.action_space[i], "n") else
                  int(sum(env.action_space[i].high - env.action_space[i].low + 1)) for i in range(A)]
        out = {"seeds": np.arange(900, 900 + W), "pos0": np.zeros((W, E, 2)), "vel0": np.zeros((W, A, 2)),
               "pos": np.zeros((T, W, E, 2)), "vel": np.zeros((T, W, A, 2)), "rew": np.zeros((T, W, A)),
               "u_noise": np.array([a.u_noise or 0.0 for a in env.world.agents]),
               "c_noise": np.array([a.c_noise or 0.0 for a in env.world.agents])}
        for i in range(A):
            out["act%d" % i] = np.zeros((T, W, widths[i]))
            out["obs%d" % i] = np.zeros((T, W, env.observation_space[i].shape[0]))
            out["c%d" % i] = np.zeros((T, W, env.world.dim_c))
        for w in range(W):
            np.random.seed(int(out["seeds"][w]))
            env.reset()
            out["pos0"][w], out["vel0"][w] = G.state_of(env)
            for t in range(T):
                acts = []
                for i in range(A):
                    a = np.zeros(widths[i])
                    k = 0
                    if env.world.agents[i].movable:
                        a[rng.randint(0, 5)] = 1.0
                        k = 5
                    if not env.world.agents[i].silent:
                        a[k + rng.randint(0, env.world.dim_c)] = 1.0
                    acts.append(a)
                    out["act%d" % i][t, w] = a
                obs, rew, done, info = env.step([a.copy() for a in acts])
                out["pos"][t, w], out["vel"][t, w] = G.state_of(env)
                out["rew"][t, w] = rew
                for i in range(A):
                    out["obs%d" % i][t, w] = obs[i]
                    out["c%d" % i][t, w] = env.world.agents[i].state.c
        path = os.path.join(HERE, "noise_%s.npz" % name)
        np.savez_compressed(path, **out)
        print("%-28s %8.1f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024.0))


def main_movable():
    """A landmark made MOVABLE after make_world (core.py:54-56: a Landmark is an Entity; World.integrate_state
    core.py:158-169 integrates every movable entity, and get_collision_force :194-195 pushes both sides of a contact):
      movable_simple_tag     obstacle 0 becomes a pushable ball of mass 3 (it collides already, size 0.2)
      movable_simple_spread  landmark 1 (of 0, 1, 2) becomes a colliding ball of mass 0.5 and size 0.1 -- an immovable
                             landmark in front of it and one behind it in the entity order
    Velocities are recorded for every entity (vel0 [W,E,2], vel [T,W,E,2])."""
    for name, setup in (("simple_tag", lambda w: (setattr(w.landmarks[0], "movable", True), setattr(w.landmarks[0], "initial_mass", 3.0))),
                        ("simple_spread", lambda w: (setattr(w.landmarks[1], "movable", True), setattr(w.landmarks[1], "collide", True),
                                                     setattr(w.landmarks[1], "initial_mass", 0.5), setattr(w.landmarks[1], "size", 0.1)))):
        env = G.make_env(name, benchmark=True)
        setup(env.world)
        data = G.record(name, env, list(range(1100, 1132)), 12, squeeze_every=2, squeeze=0.3, all_vel=True)
        w = env.world
        data.update({"c_size": np.array([e.size for e in w.entities]), "c_mass": np.array([e.mass for e in w.entities]),
                     "c_collide": np.array([bool(e.collide) for e in w.entities]),
                     "c_movable": np.array([bool(e.movable) for e in w.entities])})
        moved = np.abs(data["vel"][:, :, env.n:]).max()
        path = os.path.join(HERE, "movable_%s.npz" % name)
        np.savez_compressed(path, **data)
        print("%-28s %8.1f KiB   fastest landmark %.3f" % (os.path.basename(path), os.path.getsize(path) / 1024.0, moved))


if __name__ == "__main__":
    import sys
    (main_noise() if "--noise" in sys.argv else main_modes() if "--modes" in sys.argv else
     main_f3() if "--f3" in sys.argv else main_movable() if "--movable" in sys.argv else main())
