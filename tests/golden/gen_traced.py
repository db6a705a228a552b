"""Traces of the reference's nine scenario files (symtrace.trace on the unmodified files, loaded by path) as JSON fixtures:
tests/golden/traced_<name>.json.  Build container only (reads /root/reference); the GPU box, which has no reference tree, runs the
traced programs of these nine from the committed data (tests/test_gpu_traced.py) against the goldens the reference's own env
recorded, and the CPU suite holds them to the same goldens with NumPy (tests/test_symtrace.py) -- and, here, to a fresh trace.

    python tests/golden/gen_traced.py            # writes the nine files
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/multiagent/scenarios"
NINE = ["simple", "simple_spread", "simple_tag", "simple_adversary", "simple_push", "simple_speaker_listener", "simple_reference",
        "simple_crypto", "simple_world_comm"]


def main():
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    import multiagent_particle_envs_amd as mpe
    from multiagent_particle_envs_amd import symtrace
    for name in NINE:
        sc = mpe.scenarios.load(os.path.join(REF, name + ".py")).Scenario()
        try:          # with benchmark_data where the file's works (simple_speaker_listener.py:61 references an undefined name, Q19)
            t = symtrace.trace(sc, want_info=hasattr(sc, "benchmark_data"))
        except symtrace.TraceUnsupported:
            t = symtrace.trace(sc)
        worst = symtrace.verify(sc, t, worlds=128)
        d = symtrace.to_dict(t)
        d["source"] = "symtrace.trace(%s/%s.py), verified against the file's own callbacks on 128 random worlds (max scaled difference %.1e)" % (REF, name, worst)
        out = os.path.join(HERE, "traced_%s.json" % name)
        with open(out, "w") as fh:
            json.dump(d, fh, separators=(",", ":"))
        print("%-26s %5d nodes, %6d bytes, paths obs %s rew %s, benchmark_data %s" % (name, len(d["nodes"]), os.path.getsize(out), t.paths["obs"], t.paths["rew"],
                                                                                         "-" if t.info is None else t.info_desc[0]))


if __name__ == "__main__":
    main()
