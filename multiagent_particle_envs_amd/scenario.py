"""Scenario plug-in protocol.

The reference's contract (multiagent/scenario.py:4-10 and the README's "Creating new environments"): a
scenario builds a World and supplies the reset / reward / observation callbacks that `make_env` hands to
`MultiAgentEnv`.  Here every per-world scalar of the reference becomes a leading-B tensor:

    make_world(batch_size, device, **kw) -> World      entities created, `world.allocate()` called
    reset_world(world, mask=None, seeds=None)           initial conditions (all worlds, a masked subset, or
                                                         reference-exact per-world NumPy seeds)
    reward(agent, world) -> [B] tensor                   observation(agent, world) -> [B, D] tensor
    benchmark_data(agent, world)  (optional)             what `make_env(..., benchmark=True)` returns as info

A scenario whose class attribute `kind` names one of the `MPE_SCN_*` kernels, with its callbacks left
unmodified, is stepped by one fused launch.  A scenario of your own has two ways to run:

    obs_spec(agent, world) -> rowspec.ObsSpec            DESCRIBE the row (segments) and the reward (ordered terms):
    reward_spec(agent, world) -> rowspec.RewardSpec      two launches per step (mpe_world_step + mpe_rows), any scenario
    observation(agent, world) / reward(agent, world)     COMPUTE them with torch ops on [B, .] views: the generic path,
                                                         ~100 small launches per step (GraphedStep replays them as one)

Specs win when a scenario has both (fused=False keeps the Python callbacks) -- they stand in for the scenario's OWN
observation / reward methods only: callbacks handed to MultiAgentEnv that are not this scenario's methods run as given.

Restarts on the device.  With `auto_reset`, worlds that finish are restarted inside the step launch (row programs) or by
one small launch behind it (`mpe_episode_finish`), and RandomRollout resets episodes inside its launches; those draws are
exactly `world.reset_uniform(landmark_range, choices=choice_pops)` (agents U[-1,1)^2, landmarks U[-r,r)^2, vel = 0, uniform
picks, utterances zeroed) and never call `reset_world`.  They are used for the built-in scenarios' own reset_world and for
a scenario that declares `device_reset = True` (its reset_world is that placement -- examples/corral.py), in rng_mode
'device'; any other reset_world (fixed posts, per-world state of its own) is called with a mask.  A restricted spawn area is
still a device-side draw: `reset_boxes(world) -> [(lo_x, hi_x, lo_y, hi_y) per entity]` beside `device_reset = True` makes the
program's restarts place every entity uniformly in its own box, and `world.reset_boxes(boxes, mask, choices, seeds)` is the
matching reset_world (the same draws).  A file written against the REFERENCE's contract (`make_world(self)`, NumPy callbacks)
loads unmodified through refstyle.py -- traced into the step kernel (symtrace.py) where that is possible.
"""


class BaseScenario(object):
    #: `_abi.MPE_SCN_*` of the fused kernel that implements this scenario's callbacks (None: generic path)
    kind = None
    #: True: reset_world is `world.reset_uniform(self.landmark_range, mask, choices=world.choice_pops, seeds=seeds)` and nothing
    #: else that matters -- finished worlds may be restarted by the device-side draw instead of a masked reset_world call
    device_reset = False

    def make_world(self, batch_size=1, device=None):
        """Create the World: agents, landmarks, their constants; end with `world.allocate()`."""
        raise NotImplementedError("%s.make_world" % type(self).__name__)

    def reset_world(self, world, mask=None, seeds=None):
        """Draw initial conditions (`world.reset_uniform(...)` implements the reference's uniform placement)."""
        raise NotImplementedError("%s.reset_world" % type(self).__name__)
