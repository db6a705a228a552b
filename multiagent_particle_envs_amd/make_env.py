"""make_env: the reference's factory (make_env.py:15-44) with the batch made explicit.

    env = make_env('simple_spread')                          # reference-compatible: one world,
                                                             # NumPy in/out, np.random-seeded resets
    env = make_env('simple_spread', batch_size=65536)        # B worlds, torch tensors on the GPU
    env = make_env('simple_spread', batch_size=4096, num_agents=64)   # scenario kwargs pass through

    env = make_env('simple_tag', batch_size=16384, max_episode_steps=25, auto_reset=True)
                                                             # new: done at the horizon + device-side auto-reset

Everything on the step path runs in libmpe_hip.so on a HIP device; there is no CPU fallback.
"""


def make_env(scenario_name, benchmark=False, batch_size=None, device=None, seed=0, fresh_outputs=False,
             fused=None, max_episode_steps=None, auto_reset=False, probe_placement=True, **scenario_kwargs):
    from .environment import MultiAgentEnv
    from . import scenarios

    scenario = scenarios.load(scenario_name + ".py").Scenario()
    compat = batch_size is None
    world = scenario.make_world(batch_size=1 if compat else int(batch_size), device=device, **scenario_kwargs)
    world.seed = seed
    world.rng_mode = "numpy" if compat else "device"
    # the reference's make_world ends with reset_world(world) (simple_spread.py:28): keep that, it
    # also means a compat-mode env consumes the global NumPy stream exactly like the reference does
    if world.pos.is_cuda or compat:
        scenario.reset_world(world)
    info_cb = getattr(scenario, "benchmark_data", None) if benchmark else None
    env = MultiAgentEnv(world, scenario.reset_world, scenario.reward, scenario.observation, info_cb,
                        numpy_io=compat, fresh_outputs=fresh_outputs or compat, fused=fused,
                        max_episode_steps=max_episode_steps, auto_reset=auto_reset, probe_placement=probe_placement)
    env.scenario = scenario
    return env
