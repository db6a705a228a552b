#!/usr/bin/env python3
"""Per-launch kernel times from the DEVICE's own clock (round-4).

    python tools/device_span.py [C2 C3 C5 1M] [--launches 1000] [--out profiles/]

Why: for the launch-bound configs (C2 spread N=3 B=4096, C3 tag B=16384) the rocprofv3 kernel-trace median exceeds the
bench's own per-step time -- the profiler's per-dispatch packets cost a dependent 3 us launch 1.4-2.7 us -- so the only
evidence for the bench's kernel times were the bench's own HIP events.  This tool needs neither: it loads the
instrumented build (libmpe_hip_span.so, -DMPE_DEVICE_SPAN: lane 0 of every wave of k_split stamps s_memrealtime -- the
chip-wide constant-rate counter -- at its first instruction and, after its stores are acknowledged, at its last), captures
the SAME dependent launch chain the bench times (resident moves, no resets, ping-pong output sets) into one HIP graph
with a different stamp block per launch, replays it, and reduces the stamps per launch:

    span    max(end) - min(start) over the launch's waves     what the kernel occupies of the device
    gap     min(start of launch l+1) - max(end of launch l)   dispatch-to-dispatch dead time of a dependent chain
    period  start-to-start = span + gap                        == the bench's `kernel_us_per_launch` (two-point slope)

and, for the cross-check, times the same graph with HIP events (n and 2n launches -> slope) with BOTH libraries.
The wall clock's rate is read from the runtime (hipDeviceAttributeWallClockRate) and calibrated against HIP events over a
200 ms host sleep between two replays.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {"C2": ("simple_spread", {}, 4096), "C3": ("simple_tag", {}, 16384), "C5": ("simple_spread", {}, 65536),
           "1M": ("simple_spread", {}, 1 << 20)}


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))]


def chain(mpe, _abi, torch, env, pool, n, span=None, ack=False):
    """Capture n dependent mpe_step launches (moves from the resident pool, ping-pong output sets); launch l stamps block l."""
    L = _abi.lib()
    B = env.batch_size
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    desc = C.byref(env._desc)

    def enqueue(m):
        st = _abi.raw_stream(env.world.device)
        for l in range(m):
            out = env._sets[l & 1]
            b = out.bufs
            b.act, b.ids, b.u = pool[l % len(pool)].data_ptr(), None, None
            out.act_ptr = None
            b.force = (span[l].data_ptr() | (1 if ack else 0)) if span is not None else None     # bit 0: end stamp after the stores' acknowledgement
            _abi.check(L.mpe_step(desc, C.byref(b), B, st), "mpe_step")
    with torch.cuda.stream(s):
        enqueue(2)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            enqueue(n)
    torch.cuda.current_stream().wait_stream(s)
    for out in env._sets:
        out.bufs.force = None
    return g


def event_slope_us(torch, mk, n):
    def best(m):
        g = mk(m)
        g.replay()
        torch.cuda.synchronize()
        t = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            t = ms if t is None else min(t, ms)
        return t
    t1, t2 = best(n), best(2 * n)
    return (t2 - t1) * 1e3 / n, t1, t2


def wall_clock_khz(dev_index):
    try:
        hip = C.CDLL("libamdhip64.so")
        v = C.c_int(0)
        rc = hip.hipDeviceGetAttribute(C.byref(v), 10017, dev_index)      # hipDeviceAttributeWallClockRate
        return int(v.value) if rc == 0 and v.value > 0 else None
    except Exception:
        return None


def run(key, n, product_slope):
    import torch
    import multiagent_particle_envs_amd as mpe
    from multiagent_particle_envs_amd import _abi
    scn, kw, B = CONFIGS[key]
    env = mpe.make_env(scn, batch_size=B, seed=0, **kw)
    env.reset()
    env._ensure_buffers()
    A = env.n
    L = _abi.lib()
    pool_t = torch.empty((16, A, B, 5), dtype=torch.float32, device="cuda")
    _abi.check(L.mpe_random_actions_block(pool_t.data_ptr(), None, A, B, 0, 0, 16, 0, _abi.raw_stream(env.world.device)), "moves")
    pool = [pool_t[p] for p in range(16)]
    grid = (B + 63) // 64
    out = {"config": key, "scenario": scn, "worlds": B, "launches": n, "grid": grid, "waves_per_workgroup": A + 1,
           "library": os.path.basename(_abi.LIB_PATH)}
    if product_slope:
        us, t1, t2 = event_slope_us(torch, lambda m: chain(mpe, _abi, torch, env, pool, m), 400 if B < (1 << 20) else 100)
        out["event_slope_us"] = us
        out["event_ms"] = [t1, t2]
        return out
    span = torch.zeros((n, grid, 8, 2), dtype=torch.int64, device="cuda")
    g = chain(mpe, _abi, torch, env, pool, n, span)
    # ---- clock calibration: two replays a host sleep apart, HIP events recorded in front of each --------------------
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    torch.cuda.synchronize()
    first_a = int(span[0, :, :A + 1, 0].min())
    time.sleep(0.2)
    e1.record()
    g.replay()
    torch.cuda.synchronize()
    first_b = int(span[0, :, :A + 1, 0].min())
    ticks_per_us_cal = (first_b - first_a) / (e0.elapsed_time(e1) * 1e3)
    khz = wall_clock_khz(torch.cuda.current_device())
    tpu = (khz / 1e3) if khz else ticks_per_us_cal
    out["wall_clock_khz_runtime"] = khz
    out["ticks_per_us_calibrated"] = ticks_per_us_cal
    out["tick_ns"] = 1e3 / tpu

    def st3(v):
        return {"median": pct(v, 0.5), "p10": pct(v, 0.1), "p90": pct(v, 0.9), "mean": sum(v) / len(v), "n": len(v)}
    # ---- the measured replays: end stamps at store ISSUE (the launch is not lengthened), then at store ACKNOWLEDGEMENT ------
    for flavour, ack in (("issue", False), ("ack", True)):
        g = chain(mpe, _abi, torch, env, pool, n, span, ack)
        spans, gaps, periods, skews = [], [], [], []
        for rep in range(3):
            span.zero_()
            g.replay()          # two replays in front, back to back: the measured one runs at the clocks of a busy GPU, as the
            g.replay()          # bench's best-of-3 bodies do (each replay overwrites the stamps: the last one's are read)
            g.replay()
            torch.cuda.synchronize()
            st = span[:, :, :A + 1, 0]
            en = span[:, :, :A + 1, 1]
            assert int((st == 0).sum()) == 0 and int((en == 0).sum()) == 0, "a wave did not stamp"
            s0 = st.reshape(n, -1).min(dim=1).values.cpu().double()
            s1 = st.reshape(n, -1).max(dim=1).values.cpu().double()      # the last wave to START (dispatch ramp)
            e_ = en.reshape(n, -1).max(dim=1).values.cpu().double()
            spans += ((e_ - s0) / tpu).tolist()[1:]
            periods += ((s0[1:] - s0[:-1]) / tpu).tolist()
            gaps += ((s0[1:] - e_[:-1]) / tpu).tolist()
            skews += ((s1 - s0) / tpu).tolist()[1:]
        ev, _, _ = event_slope_us(torch, lambda m: chain(mpe, _abi, torch, env, pool, m,
                                                         torch.zeros((m, grid, 8, 2), dtype=torch.int64, device="cuda"), ack),
                                  400 if B < (1 << 20) else 100)
        out[flavour] = {"span_us": st3(spans), "gap_us": st3(gaps), "period_us": st3(periods), "start_skew_us": st3(skews),
                        "event_slope_us": ev}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["C2", "C3", "C5"])
    ap.add_argument("--launches", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "span"))
    ap.add_argument("--one", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--product", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.one:                       # one config, one library, in a process of its own (as bench.py measures C2 / C3)
        print(json.dumps(run(args.one, args.launches, args.product)))
        return
    import subprocess
    os.makedirs(args.out, exist_ok=True)
    span_lib = os.path.join(ROOT, "multiagent_particle_envs_amd", "lib", "libmpe_hip_span.so")
    assert os.path.exists(span_lib), "build the instrumented library first (python -m multiagent_particle_envs_amd._build)"
    for key in args.configs:
        res = {}
        for tag, env_extra, extra in (("instrumented", {"MPE_HIP_LIB": span_lib}, []), ("product", {}, ["--product"])):
            env = dict(os.environ)
            env.pop("MPE_HIP_LIB", None)
            env.update(env_extra)
            n = args.launches if key != "1M" else min(args.launches, 200)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", key, "--launches", str(n)] + extra,
                               capture_output=True, text=True, env=env, timeout=900)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                raise SystemExit("device_span %s (%s) failed: %s" % (key, tag, r.stderr[-3000:]))
            res[tag] = json.loads(lines[-1])
        i, p = res["instrumented"], res["product"]
        ev = p["event_slope_us"]
        txt = []
        txt.append("# device-clock stamps of %d graph-replayed dependent mpe_step launches x 3 replays -- %s, %d worlds (tools/device_span.py)"
                   % (i["launches"], i["scenario"], i["worlds"]))
        txt.append("# instrumented build: %s (-DMPE_DEVICE_SPAN); grid %d workgroups x %d waves; wall clock %s kHz (runtime), "
                   "%.4f ticks/us calibrated against HIP events over a 200 ms sleep; tick = %.1f ns"
                   % (i["library"], i["grid"], i["waves_per_workgroup"], i["wall_clock_khz_runtime"], i["ticks_per_us_calibrated"], i["tick_ns"]))
        for flavour, what in (("issue", "end stamp when the wave has ISSUED its last store (launch not lengthened; span is a lower bound)"),
                              ("ack", "end stamp after s_waitcnt vmcnt(0): the wave's stores ACKNOWLEDGED (+ one store round trip on the launch)")):
            f = i[flavour]
            txt.append("## %s" % what)
            txt.append("%-34s %9s %9s %9s %9s" % ("per launch, us", "median", "p10", "p90", "mean"))
            for name, k in (("period start-to-start", "period_us"), ("span   max(end) - min(start)", "span_us"),
                            ("gap    next start - this end", "gap_us"), ("start skew (last - first wave)", "start_skew_us")):
                v = f[k]
                txt.append("%-34s %9.3f %9.3f %9.3f %9.3f" % (name, v["median"], v["p10"], v["p90"], v["mean"]))
            txt.append("HIP-event two-point slope of the same instrumented chain: %.3f us per launch" % f["event_slope_us"])
        per = i["issue"]["period_us"]["median"]
        txt.append("## cross-check")
        txt.append("HIP-event two-point slope, same chain, PRODUCT build (= bench.py's kernel_us_per_launch protocol): %.3f us per launch" % ev)
        txt.append("device period (issue flavour, median) / product event slope = %.3f" % (per / ev))
        body = "\n".join(txt) + "\n"
        with open(os.path.join(args.out, "r4_device_span_%s.txt" % key), "w") as f:
            f.write(body)
        with open(os.path.join(args.out, "r4_device_span_%s.json" % key), "w") as f:
            json.dump(res, f, indent=1)
        print(body)


if __name__ == "__main__":
    main()
