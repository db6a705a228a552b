"""make_env: the reference's factory (make_env.py:15-44) with the batch made explicit.

    env = make_env('simple_spread')                          # reference-compatible: one world,
                                                             # NumPy in/out, np.random-seeded resets
    env = make_env('simple_spread', batch_size=65536)        # B worlds, torch tensors on the GPU
    env = make_env('simple_spread', batch_size=4096, num_agents=64)   # scenario kwargs pass through

    env = make_env('simple_tag', batch_size=16384, max_episode_steps=25, auto_reset=True)
                                                             # new: done at the horizon + device-side auto-reset

    env = make_env('/path/to/my_scenario.py')                # a REFERENCE-STYLE scenario file, unmodified (`from multiagent.core
    env = make_env('/path/to/my_scenario.py', batch_size=65536)  # import World ...`, make_world(self), NumPy callbacks): batched,
                                                             # the file's callbacks are TRACED into a compiled row program
                                                             # (symtrace.py: one launch per step; env.traced) -- or, where a file
                                                             # is outside what the tracer models / traced=False / one world, run
                                                             # per world on the host over the device physics (refstyle.py)

Everything on the step path runs in libmpe_hip.so on a HIP device; there is no CPU fallback.
"""


def make_env(scenario_name, benchmark=False, batch_size=None, device=None, seed=0, fresh_outputs=False,
             fused=None, max_episode_steps=None, auto_reset=False, probe_placement=True, compile_program=None, traced=None,
             **scenario_kwargs):
    from .environment import MultiAgentEnv
    from . import scenarios

    scenario = scenarios.load(scenario_name if scenario_name.endswith(".py") else scenario_name + ".py").Scenario()
    from . import refstyle
    if refstyle.is_reference_style(scenario):
        # written against the reference's own contract (scenario.py:4-10: make_world(self), NumPy per-world callbacks)
        if scenario_kwargs or fused:
            raise TypeError("a reference-style scenario takes no scenario kwargs and has no fused kernel (got %r, fused=%r)"
                            % (sorted(scenario_kwargs), fused))
        return refstyle.make_ref_env(scenario, benchmark=benchmark, batch_size=batch_size, device=device, seed=seed,
                                     max_episode_steps=max_episode_steps, auto_reset=auto_reset, traced=traced,
                                     fresh_outputs=fresh_outputs)
    compat = batch_size is None
    world = scenario.make_world(batch_size=1 if compat else int(batch_size), device=device, **scenario_kwargs)
    world.seed = seed
    world.rng_mode = "numpy" if compat else "device"
    # the reference's make_world ends with reset_world(world) (simple_spread.py:28): keep that, it
    # also means a compat-mode env consumes the global NumPy stream exactly like the reference does
    if world.pos.is_cuda or compat:
        scenario.reset_world(world)
    info_cb = getattr(scenario, "benchmark_data", None) if benchmark else None
    # (a scenario that DESCRIBES its rows -- obs_spec / reward_spec, rowspec.py -- need not have Python callbacks at all)
    env = MultiAgentEnv(world, scenario.reset_world, getattr(scenario, "reward", None), getattr(scenario, "observation", None), info_cb,
                        numpy_io=compat, fresh_outputs=fresh_outputs or compat, fused=fused,
                        max_episode_steps=max_episode_steps, auto_reset=auto_reset, probe_placement=probe_placement,
                        compile_program=compile_program)
    env.scenario = scenario
    return env
