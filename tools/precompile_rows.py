#!/usr/bin/env python3
"""Compile the row programs of the shipped scenarios (and examples/corral.py) in, on any machine with hipcc (no GPU needed):
fills multiagent_particle_envs_amd/lib/rows_cache/, which travels with the tree, so the first env.compile_program() on the
GPU box finds its image.    python tools/precompile_rows.py [scenario[:key=value...] ...]"""
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
for spec in (sys.argv[1:] or DEFAULT):
    parts = spec.split(":")
    kw = {k: int(v) for k, v in (p.split("=") for p in parts[1:])}
    env = tr.corral_env(4, device="cpu") if parts[0] == "corral" else tr.make_spec_env(parts[0], 4, device="cpu", scenario_kw=kw)
    t0 = time.time()
    try:
        image = _build.compile_rows_image(env._prog.static_source(env._desc))
        print("%-60s %4d ops  %7d bytes  %5.1f s" % (spec, env._prog.n_ops, len(image), time.time() - t0))
    except (_abi.MpeError, RuntimeError) as err:
        print("%-60s %4d ops  not compiled: %s" % (spec, env._prog.n_ops, str(err)[:200]))
