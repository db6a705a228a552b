#!/usr/bin/env python3
"""What done_callback + auto_reset costs when no world finishes (round 4), 65 536 worlds.

  device side   HIP-event time of n back-to-back launch groups through the C ABI (the host out of the way):
                  [mpe_step]                                   the plain fused step
                  [mpe_step; mpe_episode_finish]               round 4: one more launch, every workgroup leaves after its flags
                  [mpe_step; tick; mask; masked reset; observe] round 3's sequence (mpe_episode_tick, any(), masked_fill, masked
                                                               mpe_reset, full mpe_observe) on every step
  row programs  the same for a scenario whose done condition is part of its program (examples/corral.py with `arena`: a done_spec):
                  [mpe_step_rows]                              no episodes
                  [mpe_step_rows_episode]                      step, done tests, counters, restart of the finished worlds: ONE launch
                  [mpe_step_rows; mpe_episode_finish]          the two-launch form a Python done callback needs
  from Python   wall time per env.step of the same three configurations (a done callback that is one preallocated all-False
                row: the callback's own cost is the user's), eager.

    python tools/finish_cost.py > profiles/r4_finish_cost.txt
"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import multiagent_particle_envs_amd as mpe  # noqa: E402
from multiagent_particle_envs_amd import _abi  # noqa: E402


def event_time(fn, n=400):
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


def wall_time(step, n=400):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) * 1e6 / n
        best = t if best is None else min(best, t)
    return best


def main():
    B = 65536
    print("# done_callback + auto_reset when no world finishes, %d worlds (tools/finish_cost.py)" % B)
    for name in ("simple_spread", "simple_tag"):
        false_row = torch.zeros(B, dtype=torch.bool, device="cuda")

        def never(agent, world):
            return false_row

        def build(finish, cb=True, compile_program=None):
            env = mpe.make_env(name, batch_size=B, seed=1, max_episode_steps=1000000, auto_reset=True, compile_program=compile_program)
            if cb:
                env.done_callback = never
                env._py_done = True
            env.finish_launch = finish
            env.reset()
            return env
        plain = mpe.make_env(name, batch_size=B, seed=1)
        plain.reset()
        act = torch.nn.functional.one_hot(torch.randint(0, 5, (plain.n, B), device="cuda"), 5).float().contiguous()
        acts = [act[i] for i in range(plain.n)]
        # ---- device side, through the C ABI
        L, st = _abi.lib(), _abi.raw_stream(plain.world.device)
        plain.step(act)
        out = plain._sets[0]
        b = out.bufs
        b.act, b.ids, b.u = act.data_ptr(), None, None
        t_step = event_time(lambda: L.mpe_step(plain._desc_ref, out.bufs_ref, B, st))
        rows = [("[mpe_step]", t_step)]
        for label, pol in (("[mpe_step; mpe_episode_finish]  finish program interpreted", False),
                           ("[mpe_step; mpe_episode_finish]  finish program compiled in", True)):
            env = build(True, compile_program=pol)
            prog = env._finish_program()
            env.step(acts)
            o = env._sets[0]
            bb = o.bufs
            bb.act, bb.ids, bb.u = act.data_ptr(), None, None
            fb = _abi.MpeBuffers()
            C.memmove(C.byref(fb), C.byref(bb), C.sizeof(fb))
            done = torch.zeros((env.n, B), dtype=torch.bool, device="cuda")
            fb.done = done.data_ptr()
            fb.act = fb.ids = fb.u = None
            es = env.episode_step.data_ptr()

            def pair():
                L.mpe_step(env._desc_ref, o.bufs_ref, B, st)
                L.mpe_episode_finish(C.byref(env._desc), C.byref(fb), prog.ref, B, es, 1000000, 1.0, 1, 7, 0, st)
            rows.append((label + (" (image active)" if prog.image_active(env._desc) else ""), event_time(pair)))
        print("%s   device side, us per step (HIP events, 400 back-to-back groups, best of 3)" % name)
        for label, t in rows:
            print("   %-72s %6.2f us   %.2fx" % (label, t, t / t_step))
        # ---- from Python
        rows = [("env.step(tensor)   plain fused step, fast path", wall_time(lambda: plain.step(act))),
                ("env.step(list)     plain fused step", wall_time(lambda: plain.step(acts)))]
        e_h = build(True, cb=False)
        rows.append(("env.step(list)     max_episode_steps + auto_reset (horizon only)", wall_time(lambda: e_h.step(acts))))
        e_new, e_old = build(True), build(False)
        rows.append(("env.step(list)     + done_callback: mpe_episode_finish (round 4)", wall_time(lambda: e_new.step(acts))))
        rows.append(("env.step(list)     + done_callback: separate launches (round 3)", wall_time(lambda: e_old.step(acts))))
        base = rows[1][1]
        print("%s   from Python, us per env.step (wall clock, eager, 400 steps, best of 3)" % name)
        for label, t in rows:
            print("   %-72s %6.2f us   %.2fx" % (label, t, t / base))


def programs():
    import test_rowspec as tr
    B = 65536
    L = _abi.lib()
    for compiled in (False, True):
        env = tr.corral_env(B, arena=50.0, max_episode_steps=1000000, auto_reset=True)      # (an arena nobody leaves)
        assert env._prog.has_done and env._episode_in_launch
        if compiled:
            assert env.compile_program()
        env.reset()
        st = _abi.raw_stream(env.world.device)
        act = torch.nn.functional.one_hot(torch.randint(0, 5, (env.n, B), device="cuda"), 5).float().contiguous()
        env.step([act[i] for i in range(env.n)])
        o = env._sets[0]
        b = o.bufs
        b.act, b.ids, b.u = act.data_ptr(), None, None
        es = env.episode_step.data_ptr()
        fb = _abi.MpeBuffers()
        C.memmove(C.byref(fb), C.byref(b), C.sizeof(fb))
        fb.act = fb.ids = fb.u = None
        prog = env._prog
        t0 = event_time(lambda: L.mpe_step_rows(env._desc_ref, o.bufs_ref, prog.ref, B, st))
        t1 = event_time(lambda: L.mpe_step_rows_episode(env._desc_ref, o.bufs_ref, prog.ref, B, es, 1000000, 0.9, 1, 7, 0, st))

        def pair():
            L.mpe_step_rows(env._desc_ref, o.bufs_ref, prog.ref, B, st)
            L.mpe_episode_finish(C.byref(env._desc), C.byref(fb), prog.ref, B, es, 1000000, 0.9, 1, 7, 0, st)
        t2 = event_time(pair)
        print("examples/corral.py with a done_spec, program %s   device side, us per step" % ("COMPILED IN" if compiled else "interpreted"))
        for label, t in (("[mpe_step_rows]", t0), ("[mpe_step_rows_episode]   (episodes end inside the launch)", t1),
                         ("[mpe_step_rows; mpe_episode_finish]", t2)):
            print("   %-72s %6.2f us   %.2fx" % (label, t, t / t0))
        plain = tr.corral_env(B, arena=50.0)
        if compiled:
            plain.compile_program()
        plain.reset()
        acts = [act[i] for i in range(env.n)]
        w0 = wall_time(lambda: plain.step(acts))
        w1 = wall_time(lambda: env.step(acts))
        cb = tr.corral_env(B, max_episode_steps=1000000, auto_reset=True, done_callback=tr._strayed)
        if compiled:
            cb.compile_program()
        cb.reset()
        w2 = wall_time(lambda: cb.step(acts))
        print("   from Python (eager env.step(list)): no episodes %.2f us | done_spec + auto_reset %.2f us (%.2fx) | torch done_callback + auto_reset %.2f us (%.2fx)"
              % (w0, w1, w1 / w0, w2, w2 / w0))


if __name__ == "__main__":
    programs()
    main()
