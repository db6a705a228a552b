#!/usr/bin/env python3
"""Compile the row programs of the shipped scenarios (and examples/corral.py) in, on any machine with hipcc (no GPU needed):
fills multiagent_particle_envs_amd/lib/rows_cache/, which travels with the tree, so the first env.compile_program() on the
GPU box finds its image.    python tools/precompile_rows.py [--tests] [scenario[:key=value...] ...]
(--tests: also the programs the GPU tests compile -- the example with an arena, with resized agents, the random programs)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from multiagent_particle_envs_amd import _abi, _build  # noqa: E402
import test_rowspec as tr  # noqa: E402

DEFAULT = list(tr.NINE) + ["corral", "simple_adversary:num_agents=4:num_adversaries=2", "simple_adversary:num_agents=6:num_adversaries=2",
                           "simple_world_comm:num_good_agents=2:num_adversaries=3", "simple_world_comm:num_good_agents=3:num_adversaries=5",
                           "simple_adversary:num_agents=10:num_adversaries=3", "simple_world_comm:num_good_agents=5:num_adversaries=6"]
def test_programs():
    """The images tests/test_rowspec.py, test_gpu_rollout.py and tools/finish_cost.py would otherwise compile on the GPU box."""
    envs = []
    for arena in (0.95, 50.0):
        envs.append(("corral arena=%g" % arena, tr.corral_env(4, device="cpu", arena=arena)))
    for who, size in ((0, 0.07),):
        e = tr.corral_env(4, device="cpu")
        e.world.agents[who].size = size
        e.refresh_constants()
        envs.append(("corral agent %d size %g" % (who, size), e))
    for seed in range(1, 7):
        sc = tr._RandomScenario(seed)
        w = sc.make_world(batch_size=4, device="cpu")
        w.seed = seed
        import multiagent_particle_envs_amd as mpe
        e = mpe.MultiAgentEnv(w, sc.reset_world, None, None, compile_program=False)
        envs.append(("random program %d" % seed, e))
    # reference-style files on the traced path (symtrace.py): the fixtures, and the committed traces of the reference's nine
    import json
    from multiagent_particle_envs_amd import refstyle, symtrace, scenarios
    import multiagent_particle_envs_amd as mpe
    class _Info(object):      # (benchmark_data's program of a traced scenario: a second image)
        def __init__(self, ts):
            ip = refstyle._InfoProgram(ts, ts.make_world(4, "cpu"), compile=False)
            self._prog, self._desc = ip.prog, ip.desc
    for name in ("herd", "relay", "convoy", "survey", "mesh", "scatter"):
        sc = scenarios.load(os.path.join(ROOT, "tests", "refstyle", name + ".py")).Scenario()
        for info in ((False, True) if hasattr(sc, "benchmark_data") else (False,)):
            ts = refstyle.trace_ref_scenario(sc, want_info=info)
            envs.append(("traced " + name, mpe.MultiAgentEnv(ts.make_world(4, "cpu"), ts.reset_world, None, None, compile_program=False)))
            if info:
                envs.append(("traced %s benchmark_data" % name, _Info(ts)))
    for name in tr.NINE:
        with open(os.path.join(ROOT, "tests", "golden", "traced_%s.json" % name)) as fh:
            ts = refstyle.TracedRefScenario(None, symtrace.from_dict(json.load(fh)))
        envs.append(("traced reference " + name, mpe.MultiAgentEnv(ts.make_world(4, "cpu"), ts.reset_world, None, None, compile_program=False)))
        if ts.t.info is not None:
            envs.append(("traced reference %s benchmark_data" % name, _Info(ts)))
    return envs


from concurrent.futures import ThreadPoolExecutor  # noqa: E402

if "--race-variant" in sys.argv:
    # (internal) the traced 6-agent navigation program of tests/test_gpu_race.py compiled with the flags in MPE_ROWS_IMAGE_FLAGS
    import tempfile
    import test_gpu_race as tg
    from multiagent_particle_envs_amd import refstyle, scenarios
    import multiagent_particle_envs_amd as mpe
    path = os.path.join(tempfile.mkdtemp(), "nav6.py")
    with open(path, "w") as fh:
        fh.write(tg._NAV6)
    ts = refstyle.trace_ref_scenario(scenarios.load(path).Scenario())
    env = mpe.MultiAgentEnv(ts.make_world(4, "cpu"), ts.reset_world, None, None, compile_program=False)
    t0 = time.time()
    img = _build.compile_rows_image(env._prog.static_source(env._desc))
    print("race variant %-60r %8d bytes %6.1f s" % (os.environ.get("MPE_ROWS_IMAGE_FLAGS", ""), len(img), time.time() - t0))
    sys.exit(0)

jobs = []      # (label, n_ops, header text)
if "--tests" in sys.argv:
    jobs += [(label, env._prog.n_ops, env._prog.static_source(env._desc)) for label, env in test_programs()]
if "--tests" in sys.argv:      # the three builds of the race test's traced program (flags are part of the cache key: one process each)
    import subprocess
    for flags in ("", "-DMPE_STRESS_DELAY_WAVE=1", "-DMPE_STRESS_DELAY_WAVE=1 -DMPE_STRESS_NO_SHARED_BARRIER"):
        e = dict(os.environ)
        e.pop("MPE_ROWS_IMAGE_FLAGS", None)
        if flags:
            e["MPE_ROWS_IMAGE_FLAGS"] = flags
        subprocess.run([sys.executable, os.path.abspath(__file__), "--race-variant"], env=e, check=True)
specs = [a for a in sys.argv[1:] if a != "--tests"]
for spec in (specs or ([] if "--tests" in sys.argv else DEFAULT)):
    parts = spec.split(":")
    kw = {k: int(v) for k, v in (p.split("=") for p in parts[1:])}
    env = tr.corral_env(4, device="cpu") if parts[0] == "corral" else tr.make_spec_env(parts[0], 4, device="cpu", scenario_kw=kw)
    try:
        jobs.append((spec, env._prog.n_ops, env._prog.static_source(env._desc)))
    except _abi.MpeError as err:
        print("%-60s %4d ops  not compiled: %s" % (spec, env._prog.n_ops, str(err)[:200]))


def compile_one(job):
    label, n_ops, src = job
    t0 = time.time()
    try:
        image = _build.compile_rows_image(src)
        return "%-60s %4d ops  %7d bytes  %5.1f s" % (label, n_ops, len(image), time.time() - t0)
    except RuntimeError as err:
        return "%-60s %4d ops  hipcc failed: %s" % (label, n_ops, str(err)[-300:])


with ThreadPoolExecutor(max_workers=max(2, (os.cpu_count() or 4) - 1)) as ex:      # one hipcc process per image
    for line in ex.map(compile_one, jobs):
        print(line, flush=True)
