"""CPU oracle for the batched particle-world hot path.  TEST INFRASTRUCTURE ONLY.

This package is a restatement, in plain NumPy / C, of the arithmetic the reference performs on
the path  MultiAgentEnv.step -> World.step -> Scenario.observation/reward
(reference: multiagent/environment.py:80-104, multiagent/core.py:117-196,
multiagent/scenarios/simple_spread.py:31-100, simple_tag.py:39-147, simple.py:24-50).

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
-- and there only as the checker / the reported CPU baseline.  The product package
(`multiagent_particle_envs_amd`) never imports anything from here and has no CPU fallback: it
raises if the HIP library is missing.

Parity status: PINNED.  The reference has no tests or golden vectors of its own (SURVEY.md
section 4), so the pin is the reference itself executed in the build container:
`tests/golden/gen_golden.py` imports the unmodified `/root/reference` (through a shape-only
`gym` stub), records states/actions/outputs into `tests/golden/*.npz`, and
`tests/test_oracle_golden.py` checks every module here against those files (<=1e-12 for the
fp64 paths, counts/dones exact).  SURVEY.md appendix A.3's known-answer vectors are checked too.

Modules
  spec.py         scenario constants (per-entity size/movable/collide/accel/max_speed ...)
  mpe_loop.py     per-object fp64 loop, one world at a time -- same cost structure as the
                  reference (Python loop + tiny NumPy ops per pair); the `cpu_baseline` "port"
  mpe_batched.py  the same arithmetic vectorised over B worlds, fp64 or fp32 -- the scalable truth
  philox.py       Philox4x32-10 + the uniform mapping used by the device reset / random actions
  mpe_oracle.c    the fp64 step in plain C (gcc; second checker + compiled-code CPU baseline; built by build_c.py)
  build_c.py      gcc build + ctypes binding of mpe_oracle.c
"""
