"""Reference-style Scenario files, unmodified, over the device physics.

The reference's plug-in contract (multiagent/scenario.py:4-10, scenarios/__init__.py:5-7, make_env.py:36-43, README
"Creating new environments") is a file whose `Scenario` builds ONE world of Python objects holding NumPy 2-vectors and
answers `reward(agent, world)` / `observation(agent, world)` with NumPy scalars / 1-D arrays.  This package's own
protocol (scenario.py) is the batched one -- `make_world(batch_size, device)`, `[B]` torch tensors -- because that is
what a GPU can evaluate.  `RefScenarioAdapter` lets the first kind run on the second kind's engine:

    B shadow worlds   `scenario.make_world()` called B times: the file's own objects (`multiagent.core` classes, see
                      compat/), one per world, holding whatever per-world Python state the scenario keeps on them
                      (`agent.goal_a`, colours, keys ...)
    one device World  the SoA tensors of all B worlds; `World.step` = `mpe_world_step` (libmpe_hip.so), action decode
                      = the env's `_set_action` in torch -- exactly the generic path of the torch protocol
    after every step  ONE device-to-host copy of the state; every shadow entity's `state.p_pos / p_vel / c` and
                      `action.u / c` become NumPy views of its row; the file's callbacks are then evaluated per world
                      ON THE HOST, in the reference's order (environment.py:92-97), and stacked into [B, .] outputs
    reset             `scenario.reset_world(shadow_b)` per (masked) world -- the file draws from `np.random` exactly as
                      it does in the reference -- then one upload

Correct and slow by construction: B x A Python calls per step (about 4 us per NumPy op; a 4096-world simple_spread step
is ~0.3 s).  It is the COMPATIBILITY path -- one world with NumPy in / NumPy out is the reference's own use, seed for
seed -- and the migration path: a scenario that matters gets rewritten against the torch protocol (or ObsSpec /
RewardSpec, rowspec.py) and runs 3-5 orders of magnitude faster.  The physics never runs on the CPU either way.
"""
import inspect
import os

import numpy as np
import torch

from . import _abi
from .core import World, Agent, Landmark, Action

# constants World.step reads (core.py:27-51, 59-79): identical in every shadow world, or the batch cannot be one launch
_ENTITY_KEYS = ("size", "movable", "collide", "max_speed", "accel", "mass")
_AGENT_KEYS = ("silent", "u_noise", "c_noise", "u_range", "blind")
_WORLD_KEYS = ("dim_c", "dim_p", "dt", "damping", "contact_force", "contact_margin")


def is_reference_style(scenario):
    """A Scenario written against the reference's contract: `make_world(self)` takes no batch size."""
    try:
        params = inspect.signature(scenario.make_world).parameters
    except (TypeError, ValueError):
        return False
    return "batch_size" not in params and not any(p.kind == p.VAR_KEYWORD for p in params.values())


class _MirroredWorld(World):
    """The device World of a reference-style env: after every step the shadows are brought up to date."""

    def __init__(self, adapter, batch_size, device):
        super(_MirroredWorld, self).__init__(batch_size, device)
        self.__dict__["_adapter"] = adapter

    def step(self):
        super(_MirroredWorld, self).step()
        self._adapter.pull()

    def set_state(self, pos, vel=None):
        super(_MirroredWorld, self).set_state(pos, vel)
        if self.__dict__.get("_adapter_ready"):
            self._adapter.pull()


class RefScenarioAdapter(object):
    """Presents a reference-style Scenario to MultiAgentEnv as a (generic-path) batched one."""
    kind = None            # no fused kernel: the callbacks are the file's own Python
    reference_style = True

    def __init__(self, scenario, batch_size=1, device=None, host_outputs=False):
        self.scenario = scenario
        self.B = int(batch_size)
        self.host_outputs = bool(host_outputs)     # compat mode (one world, NumPy I/O): fp64 host tensors, no round trip
        self.shadows = [scenario.make_world() for _ in range(self.B)]
        w0 = self.shadows[0]
        self._agents = [list(w.agents) for w in self.shadows]          # "no agents created / destroyed at runtime" (environment.py:8)
        self._entities = [list(w.agents) + list(w.landmarks) for w in self.shadows]
        self._check_uniform()
        world = _MirroredWorld(self, self.B, device)
        for k in _WORLD_KEYS:
            setattr(world, k, getattr(w0, k))
        for k in ("collaborative", "discrete_action"):                  # optional attributes the env looks for (environment.py:34-36)
            if hasattr(w0, k):
                setattr(world, k, getattr(w0, k))
        world.agents = [self._mirror(a, Agent(), i) for i, a in enumerate(w0.agents)]
        world.landmarks = [self._mirror(l, Landmark(), None) for l in w0.landmarks]
        world.allocate()
        self.world = world
        for w in self.shadows:                       # environment.py:70: agent.action.c = np.zeros(world.dim_c)
            for a in w.agents:
                a.action.c = np.zeros(w.dim_c)
                if a.action.u is None:
                    a.action.u = np.zeros(w.dim_p)
        self.push()
        world.__dict__["_adapter_ready"] = True

    # ---- construction ------------------------------------------------------------------------------------------------
    def _check_uniform(self):
        w0 = self.shadows[0]
        ref = [tuple(getattr(e, k) for k in _ENTITY_KEYS) for e in self._entities[0]]
        refa = [tuple(getattr(a, k, None) for k in _AGENT_KEYS) + (a.action_callback is not None,) for a in self._agents[0]]
        refw = tuple(getattr(w0, k) for k in _WORLD_KEYS)
        for b, w in enumerate(self.shadows[1:], 1):
            if len(w.agents) != len(w0.agents) or len(w.landmarks) != len(w0.landmarks) or \
                    tuple(getattr(w, k) for k in _WORLD_KEYS) != refw or \
                    [tuple(getattr(e, k) for k in _ENTITY_KEYS) for e in self._entities[b]] != ref or \
                    [tuple(getattr(a, k, None) for k in _AGENT_KEYS) + (a.action_callback is not None,) for a in self._agents[b]] != refa:
                raise _abi.MpeError("reference-style scenario: make_world() built world %d with other entity counts or physics "
                                    "constants (size / movable / collide / max_speed / accel / mass / noise) than world 0 -- B worlds "
                                    "step in one launch and share them; per-world randomness belongs in reset_world" % b)

    def _mirror(self, src, dst, agent_index):
        dst.name = src.name
        dst.size, dst.movable, dst.collide = float(src.size), bool(src.movable), bool(src.collide)
        dst.max_speed, dst.accel, dst.initial_mass = src.max_speed, src.accel, float(src.mass)
        dst.density, dst.color = getattr(src, "density", 25.0), None
        if agent_index is not None:
            dst.silent, dst.blind = bool(src.silent), bool(getattr(src, "blind", False))
            dst.u_noise, dst.c_noise, dst.u_range = src.u_noise, src.c_noise, getattr(src, "u_range", 1.0)
            if src.action_callback is not None:     # a scripted agent: its action comes from the file's callback, world by world
                dst.action_callback = lambda agent, world, i=agent_index: self._scripted_action(i)
        return dst

    def refresh_constants(self):
        """Re-read the physics constants from shadow world 0 (a caller changed them on the reference-style objects)."""
        self._check_uniform()
        w0 = self.shadows[0]
        for k in _WORLD_KEYS[2:]:
            setattr(self.world, k, getattr(w0, k))
        for src, dst in zip(self._entities[0], self.world.entities):
            dst.size, dst.movable, dst.collide = float(src.size), bool(src.movable), bool(src.collide)
            dst.max_speed, dst.accel, dst.initial_mass = src.max_speed, src.accel, float(src.mass)
        for src, dst in zip(self._agents[0], self.world.agents):
            dst.silent, dst.u_noise, dst.c_noise = bool(src.silent), src.u_noise, src.c_noise

    # ---- state traffic ---------------------------------------------------------------------------------------------
    def push(self):
        """Shadow worlds -> device: positions, velocities, utterances (after make_world / reset_world), then pull, so
        the shadows hold exactly what the device holds (fp32-rounded)."""
        w = self.world
        E, A, dc = len(w.entities), len(w.agents), int(w.dim_c)
        pos = np.zeros((self.B, E, 2), np.float64)
        vel = np.zeros((self.B, E, 2), np.float64)
        c = np.zeros((A, self.B, dc), np.float32)
        for b, ents in enumerate(self._entities):
            for e, ent in enumerate(ents):
                if ent.state.p_pos is not None:
                    pos[b, e] = ent.state.p_pos
                if ent.state.p_vel is not None:
                    vel[b, e] = ent.state.p_vel
            if dc:
                for i, a in enumerate(self._agents[b]):
                    if a.state.c is not None:
                        c[i, b] = a.state.c
        World.set_state(w, pos, vel)
        for i, agent in enumerate(w.agents):
            agent.state.c = torch.as_tensor(c[i]).to(w.device)
        self.pull()

    def pull(self):
        """Device -> shadow worlds: every entity's state row and every agent's utterance / action become NumPy views of
        ONE host copy (fp64, as reference-style callbacks expect)."""
        w = self.world
        if w.pos.is_cuda and torch.cuda.is_current_stream_capturing():
            raise _abi.MpeError("a reference-style scenario's callbacks run on the host: its step cannot be captured into a HIP graph")
        A = len(w.agents)
        pos = w.pos.permute(2, 0, 1).contiguous().cpu().numpy().astype(np.float64)          # [B, E, 2]
        vel = w._vel_all.permute(2, 0, 1).contiguous().cpu().numpy().astype(np.float64)

        def host(t, width):
            if t is None or not torch.is_tensor(t) or width == 0:
                return np.zeros((self.B, width), np.float64)
            return t.detach().to("cpu", torch.float64).reshape(-1, width).expand(self.B, width).contiguous().numpy()
        dc, dp = int(w.dim_c), int(w.dim_p)
        sc = [host(a.state.c, dc) for a in w.agents]
        au = [host(a.action.u, dp) for a in w.agents]
        ac = [host(a.action.c, dc) for a in w.agents]
        for b, ents in enumerate(self._entities):
            pb, vb = pos[b], vel[b]
            for e, ent in enumerate(ents):
                st = ent.state
                st.p_pos, st.p_vel = pb[e], vb[e]
            for i in range(A):
                a = ents[i]
                a.state.c = sc[i][b]
                a.action.u, a.action.c = au[i][b], ac[i][b]

    def _scripted_action(self, i):
        """World.step for a scripted agent (core.py:119-121): the file's action_callback per world -> one batched Action."""
        u = np.zeros((self.B, int(self.world.dim_p)), np.float32)
        c = np.zeros((self.B, int(self.world.dim_c)), np.float32)
        for b, w in enumerate(self.shadows):
            a = self._agents[b][i]
            act = a.action_callback(a, w)
            a.action = act
            if act.u is not None:
                u[b] = act.u
            if act.c is not None and c.shape[1]:
                c[b] = act.c
        out = Action()
        out.u, out.c = torch.as_tensor(u).to(self.world.device), torch.as_tensor(c).to(self.world.device)
        return out

    # ---- the callbacks MultiAgentEnv calls -----------------------------------------------------------------------------
    def _index(self, agent):
        return self.world.agents.index(agent)

    def _tensor(self, arr, dtype=torch.float32):
        if self.host_outputs:
            return torch.from_numpy(np.ascontiguousarray(arr))
        return torch.as_tensor(np.ascontiguousarray(arr)).to(self.world.device, dtype)

    def reset_world(self, world, mask=None, seeds=None):
        """Scenario.reset_world for every (masked) world; `seeds`: `np.random.seed(seeds[b])` right before world b's --
        the reference's `np.random.seed(s); env.reset()` per world."""
        m = None if mask is None else torch.as_tensor(mask).cpu().numpy().astype(bool).reshape(-1)
        for b, w in enumerate(self.shadows):
            if m is not None and not m[b]:
                continue
            if seeds is not None:
                np.random.seed(int(seeds[b]))
            self.scenario.reset_world(w)
        self.push()

    def observation(self, agent, world):
        i = self._index(agent)
        f = self.scenario.observation
        return self._tensor(np.stack([np.asarray(f(self._agents[b][i], w), np.float64).reshape(-1) for b, w in enumerate(self.shadows)]))

    def reward(self, agent, world):
        i = self._index(agent)
        f = self.scenario.reward
        return self._tensor(np.array([float(f(self._agents[b][i], w)) for b, w in enumerate(self.shadows)], np.float64))

    def done(self, agent, world):
        i = self._index(agent)
        f = self.scenario.done
        v = np.array([bool(f(self._agents[b][i], w)) for b, w in enumerate(self.shadows)])
        return torch.from_numpy(v) if self.host_outputs else torch.from_numpy(v).to(self.world.device)

    def benchmark_data(self, agent, world):
        """The file's benchmark_data per world: numbers (or tuples of numbers) are stacked into [B] tensors (tuples of
        them), anything else comes back as the list of the B per-world objects."""
        i = self._index(agent)
        f = self.scenario.benchmark_data
        vals = [f(self._agents[b][i], w) for b, w in enumerate(self.shadows)]
        return _stack_info(vals, self)


# ---- the fast path: the file's callbacks traced into a compiled row program (symtrace.py) ----------------------------------
def _mirror_entity(src, dst, is_agent):
    dst.name = src.name
    dst.size, dst.movable, dst.collide = float(src.size), bool(src.movable), bool(src.collide)
    dst.max_speed, dst.accel, dst.initial_mass = src.max_speed, src.accel, float(src.mass)
    dst.density, dst.color = getattr(src, "density", 25.0), None
    if is_agent:
        dst.silent, dst.blind = bool(src.silent), bool(getattr(src, "blind", False))
        dst.u_noise, dst.c_noise, dst.u_range = src.u_noise, src.c_noise, getattr(src, "u_range", 1.0)
    return dst


class TracedRefScenario(object):
    """A reference-style Scenario presented to MultiAgentEnv as a row-program scenario of this package's protocol: its
    `observation` / `reward` (/ `done`) were traced into expression graphs (symtrace.trace), verified against the file's own
    callbacks, and are compiled into the step kernel as straight-line code -- `env.step` is ONE launch, no host callbacks, no
    per-world Python objects.  `reset_world` is the traced reset: drawn on the device where every coordinate is a uniform draw
    of its own -- `World.reset_uniform`'s placement (all nine reference files) or a box per entity (`reset_boxes`: a restricted
    spawn area); then the episodes also end and restart inside the step launch -- else drawn and evaluated with torch ops on the
    device for all worlds at once; `reset(seeds=...)` replays the file's own random stream per world (the reference's
    `np.random.seed(s); env.reset()`).  `benchmark_data` (make_env(..., benchmark=True)) runs as a second program."""
    kind = None
    reference_style = True
    traced_style = True

    def __init__(self, scenario, traced):
        self.scenario, self.t = scenario, traced
        self.landmark_range, self.device_reset, self._boxes = self._uniform_pattern()
        self._source = None
        self._info = None          # benchmark_data's program, built when first asked for
        self._env = None           # weak reference to the env (its utterance buffer), set by make_traced_env

    def report(self):
        """One paragraph: what was traced and how it runs."""
        t = self.t
        return ("traced: %d agents, %d landmarks, dim_c %d; observation widths %s; %d graph nodes; control-flow paths per callback: obs %s, "
                "reward %s%s (%s); per-world picks of reset_world: %s; reset_world %s; verified against the file's own callbacks: max "
                "difference %s" % (
                    t.A, t.E - t.A, t.dim_c, [len(r) for r in t.obs], t.graph.count, t.paths["obs"], t.paths["rew"],
                    ", done %s" % t.paths["done"] if t.paths.get("done") else "",
                    "value-only control flow predicated" if getattr(t, "predicated", False) else "by forking",
                    (str(list(t.pops[:t.real_picks()]) or "none") +
                     (" (+ %d random numbers it keeps outside the state and the callbacks read: per-world parameters)" % len(t.params) if getattr(t, "params", ()) else "")),
                    ("is World.reset_uniform's placement (landmarks on [-%g, %g)^2): restarts are drawn on the device, inside the step launch"
                     % (self.landmark_range, self.landmark_range) if self._boxes is None else
                     "places every entity uniformly in a box of its own: restarts are drawn on the device, inside the step launch")
                    if self.device_reset else
                    ("is not a formula of its draws (%s): the file's own reset_world runs on the host, once per world that restarts"
                     % t.host_reset) if getattr(t, "host_reset", None) else
                    "is the file's own placement: evaluated with torch ops on the device for all worlds at once",
                    getattr(t, "verified", "not run")))

    def spot_check(self, env, obs_n, reward_n, worlds=64, band=2e-6):
        """After `obs_n, reward_n, _, _ = env.step(...)`: run the FILE'S OWN callbacks on the env's current state for a sample of
        worlds (on the host, concretely) and compare with what the kernel delivered.  Returns (max scaled difference, worlds
        checked); worlds within `band` of one of the file's own thresholds are skipped (fp32 vs fp64: either branch is right).
        A check on live data, for trust -- the trace was already verified against the file on random worlds when it was made."""
        from . import symtrace
        if self.scenario is None:
            raise _abi.MpeError("spot_check needs the scenario object (this env was built from trace data)")
        t, sc, w = self.t, self.scenario, env.world
        B = w.batch_size
        pos, vel = w.get_state(all_entities=True)
        picks = w.choice_i32.cpu().numpy() if t.pops else np.zeros((0, B), np.int64)
        Cw = np.zeros((B, t.A, t.dim_c))
        for i, a in enumerate(w.agents):
            if t.dim_c and not a.silent and env._comm is not None:
                Cw[:, i] = env._comm[i].cpu().numpy()
        margin = symtrace.decision_margin([n for row in t.obs for n in row] + list(t.rew), B, P=pos.astype(np.float64),
                                          V=vel.astype(np.float64), Cw=Cw, K=picks.T)
        state = np.random.get_state()
        try:
            cw = sc.make_world()
        finally:
            np.random.set_state(state)
        shared = bool(t.collaborative)
        worst, checked = 0.0, 0
        for b in np.linspace(0, B - 1, num=min(int(worlds), B)).astype(int):
            if margin[b] <= band:
                continue
            if getattr(t, "host_reset", None):          # (the picks of world b in place: the file's own reset, its choices answered)
                state = np.random.get_state()
                try:
                    with symtrace.logged_picks(symtrace.PickLogger(forced=picks[:, b])):
                        sc.reset_world(cw)
                finally:
                    np.random.set_state(state)
            else:
                u = np.zeros(max(t.n_u, 1))
                for j, idx in enumerate(getattr(t, "params", ())):          # (the draws the callbacks read: what their slots carry)
                    u[idx] = float(picks[t.real_picks() + j, b]) / symtrace.PARAM_POP
                with symtrace.patched_random(symtrace._Replayer(u, picks[:, b])):
                    sc.reset_world(cw)
            for k, e in enumerate(list(cw.agents) + list(cw.landmarks)):
                e.state.p_pos, e.state.p_vel = pos[b, k].astype(np.float64), vel[b, k].astype(np.float64)
            for i, a in enumerate(cw.agents):
                a.state.c = np.zeros(t.dim_c) if a.silent else Cw[b, i].copy()
            rews = []
            for i, a in enumerate(cw.agents):
                o = np.asarray(sc.observation(a, cw), np.float64).reshape(-1)
                worst = max(worst, float(np.abs(obs_n[i][b].detach().cpu().numpy() - o).max()) if o.size else 0.0)
                rews.append(float(sc.reward(a, cw)))
            if shared:
                rews = [sum(rews)] * len(rews)
            for i, r in enumerate(rews):
                worst = max(worst, abs(float(reward_n[i][b]) - r) / max(1.0, abs(r)))
            checked += 1
        return worst, checked

    # ---- reset_world ------------------------------------------------------------------------------------------------------
    def _uniform_pattern(self):
        """(landmark_range, device_reset, boxes): how the traced reset_world places the entities.  Every coordinate a uniform draw of
        its own (`np.random.uniform(lo, hi, dim_p)` per entity), zero velocities and utterances -> restarts can be drawn on the
        device: as World.reset_uniform's placement (agents on [-1,1)^2, every landmark on [-r,r)^2 with one r: boxes None -- all
        nine reference files) or in per-entity boxes [(lo_x, hi_x, lo_y, hi_y)] (a restricted spawn area ...).  Anything else
        (positions that depend on a pick or on each other): (1.0, False, None) -- evaluated with torch ops at reset time."""
        from . import symtrace
        t = self.t
        if getattr(t, "host_reset", None):      # not a program at all: the file's own reset_world, per world, at reset time
            return 1.0, False, None
        used, boxes = set(), []
        for e in range(t.E):
            box = []
            for c in range(2):
                n = t.reset_pos[e][c]
                # lo + (hi - lo) * U, as np.random.uniform(lo, hi) is traced ...
                if n.op == "add" and n.args[0].op == "const" and n.args[1].op == "mul" and n.args[1].args[0].op == "const" and \
                        n.args[1].args[1].op == "U":
                    lo, span, u = n.args[0].value, n.args[1].args[0].value, n.args[1].args[1].value[0]
                else:      # ... or any other way of writing an affine function of ONE draw (`np.random.rand(2) * 2 - 1`, `0.5 - rand()`)
                    us = [x for x in symtrace.topo([n]) if x.op in ("U", "K", "P", "V", "C")]
                    if len(us) != 1 or us[0].op != "U":
                        return 1.0, False, None
                    u = us[0].value[0]
                    probe = np.zeros((5, max(t.n_u, 1)))
                    probe[:, u] = (0.0, 0.25, 0.5, 0.75, 1.0)
                    f = symtrace.evaluate([n], 5, U=probe, K=np.zeros((5, len(t.pops)), np.int64))[0]
                    if not np.all(np.isfinite(f)) or np.abs(f - (f[0] + (f[4] - f[0]) * probe[:, u])).max() > 1e-12 * max(1.0, np.abs(f).max()):
                        return 1.0, False, None
                    lo, span = float(min(f[0], f[4])), float(abs(f[4] - f[0]))      # (a falling line: the same distribution)
                if u in used or span < 0:
                    return 1.0, False, None
                used.add(u)
                box += [lo, lo + span]
            boxes.append(tuple(box))
        zeros = all(n.op == "const" and n.value == 0.0 for v in t.reset_vel for n in v) and \
            all(n.op == "const" and n.value == 0.0 for v in t.reset_c for n in v)
        if not zeros or used & set(getattr(t, "params", ())):          # (a draw the callbacks read that ALSO places an entity: one source for both)
            return 1.0, False, None

        def sym(b, r):
            return all(abs(x - y) <= 1e-12 * max(1.0, abs(r)) for x, y in zip(b, (-r, r, -r, r)))
        r_lm = boxes[t.A][1] if t.E > t.A else 1.0
        if all(sym(b, 1.0) for b in boxes[:t.A]) and r_lm > 0 and all(sym(b, r_lm) for b in boxes[t.A:]):
            return float(r_lm), True, None
        return 1.0, True, boxes

    def reset_boxes(self, world):
        """rowspec.compile_scenario: the per-entity placement boxes of the program's restarts (None: the reference's placement)"""
        return self._boxes

    def make_world(self, batch_size=1, device=None):
        t, w0 = self.t, self.t.world
        world = World(batch_size, device)
        for k in _WORLD_KEYS:
            setattr(world, k, getattr(w0, k))
        world.collaborative = bool(t.collaborative)
        if hasattr(w0, "discrete_action"):
            world.discrete_action = w0.discrete_action
        world.choice_pops = list(t.pops)
        world.agents = [_mirror_entity(a, Agent(), True) for a in w0.agents]
        world.landmarks = [_mirror_entity(l, Landmark(), False) for l in w0.landmarks]
        world.allocate()
        return world

    def _merge_picks(self, world, idx, mask):
        if world.choice_i32 is None or not self.t.pops:
            return
        for k in range(len(self.t.pops)):
            world.choice_i32[k].copy_(World.merge_choice(world.choice_i32[k].long(), idx[:, k].to(world.device), mask).to(torch.int32))

    def reset_world(self, world, mask=None, seeds=None):
        t, B = self.t, world.batch_size
        if seeds is None and self.device_reset and world.rng_mode == "device" and world.pos.is_cuda:
            if self._boxes is None:
                idx = world.reset_uniform(self.landmark_range, mask, choices=list(t.pops))
            else:
                idx = world.reset_boxes(self._boxes, mask, choices=list(t.pops))
            if idx is None:
                idx = torch.zeros((B, 0), dtype=torch.long, device=world.device)
            self._merge_picks(world, idx, mask)
            return
        from . import symtrace
        if getattr(t, "host_reset", None):
            return self._host_reset(world, mask, seeds)
        flat = [n for e in t.reset_pos for n in e] + [n for e in t.reset_vel for n in e]
        if seeds is None and world.rng_mode == "device" and world.pos.is_cuda:
            # not reset_uniform's placement (a restricted spawn area, positions that depend on a pick ...): the traced reset
            # program drawn and evaluated with torch ops for all worlds at once -- ~50 small launches, nothing leaves the device
            dev = world.device
            # counter-based draws keyed per (seed, GLOBAL world number, episode, draw): no two triples share a stream, and a
            # world's draws do not depend on how the batch is cut across ranks (as mpe_reset's Philox key)
            nu = max(t.n_u, 1)
            R = _keyed_uniform(int(world.seed), int(world._episode), int(world.world_offset), B, nu + len(t.pops), dev)
            world._episode += 1
            U = R[:nu].to(torch.float32)
            K = torch.stack([(R[nu + j] * n).floor().to(torch.int64).clamp_(0, n - 1) for j, n in enumerate(t.pops)]) if t.pops else \
                torch.zeros((0, B), dtype=torch.int64, device=dev)
            for j, idx in enumerate(getattr(t, "params", ())):      # the draws the callbacks read travel in the slots after the picks
                K[t.real_picks() + j] = (U[idx].double() * symtrace.PARAM_POP).floor().to(torch.int64).clamp_(0, symtrace.PARAM_POP - 1)
            vals = symtrace.evaluate_torch(flat, B, K=K, U=U, device=dev)
            pos = torch.stack(vals[:2 * t.E]).reshape(t.E, 2, B)
            vel = torch.stack(vals[2 * t.E:]).reshape(t.E, 2, B)
            if mask is None:
                world.pos.copy_(pos)
                world._vel_all.copy_(vel)
            else:
                mk = torch.as_tensor(mask, device=dev).bool().reshape(1, 1, B)
                world.pos.copy_(torch.where(mk, pos, world.pos))
                world._vel_all.copy_(torch.where(mk, vel, world._vel_all))
            for a in world.agents:      # every reset_world of the reference zeroes the utterances (traced: reset_c, zeros or not modelled)
                if torch.is_tensor(a.state.c) and a.state.c.numel():
                    if mask is None:
                        a.state.c.zero_()
                    else:
                        a.state.c.mul_((~torch.as_tensor(mask, device=dev).bool()).to(a.state.c.dtype)[:, None])
            self._merge_picks(world, K.t(), mask)
            return
        # the traced reset program evaluated for all worlds at once (fp64) on the host, then one upload: seeded resets (the file's
        # own random stream per world) and compatibility modes
        m = None if mask is None else torch.as_tensor(mask).cpu().numpy().astype(bool).reshape(-1)
        U = np.zeros((B, max(t.n_u, 1)))
        K = np.zeros((B, len(t.pops)), np.int64)
        if seeds is not None:      # the file's own random stream per world: np.random.seed(s); reset_world(world)
            assert len(seeds) == B
            for b in range(B):
                if m is not None and not m[b]:
                    continue
                rs, iu, ik = np.random.RandomState(int(seeds[b])), 0, 0
                for d in t.draws:
                    if d[0] == "uniform":
                        U[b, iu:iu + d[3]] = rs.random_sample(d[3])       # uniform(lo, hi, n) = lo + (hi - lo) * random_sample(n)
                        iu += d[3]
                    else:
                        K[b, ik] = rs.randint(0, d[1])                    # == np.random.choice(list of n) (same stream)
                        ik += 1
        else:
            if world.rng_mode == "numpy":
                rs = np.random
            else:
                rs = np.random.RandomState([int(world.seed) & 0x7FFFFFFF, int(world._episode) & 0x7FFFFFFF, int(world.world_offset) & 0x7FFFFFFF])
                world._episode += 1
            U = rs.random_sample(U.shape)
            for k, n in enumerate(t.pops[:t.real_picks()]):
                K[:, k] = rs.randint(0, n, B)
        for j, idx in enumerate(getattr(t, "params", ())):
            K[:, t.real_picks() + j] = np.clip(np.floor(U[:, idx] * symtrace.PARAM_POP), 0, symtrace.PARAM_POP - 1).astype(np.int64)
        vals = symtrace.evaluate(flat, B, K=K, U=U)
        pos = np.stack(vals[:2 * t.E], axis=1).reshape(B, t.E, 2)
        vel = np.stack(vals[2 * t.E:], axis=1).reshape(B, t.E, 2)
        if m is not None:
            old_p, old_v = world.get_state(all_entities=True)
            pos = np.where(m[:, None, None], pos, old_p)
            vel = np.where(m[:, None, None], vel, old_v)
        world.set_state(pos, vel)
        self._merge_picks(world, torch.as_tensor(K), mask)

    def _host_reset(self, world, mask, seeds):
        """reset_world of a file whose placement is not a formula of its draws (rejection sampling, shuffles, normal draws): the
        file's OWN reset_world, run on the host on one shadow world, once per world that restarts -- np.random seeded per world
        (`seeds`: the reference's `np.random.seed(s); env.reset()`; else from (env seed, episode, world)) --, the positions,
        velocities and utterances it leaves uploaded in one copy.  Only resets pay for this; steps stay one launch."""
        t, B, sc = self.t, world.batch_size, self.scenario
        if sc is None:
            raise _abi.MpeError("this trace's reset_world runs on the host (%s): it needs the scenario file, not only trace data" % t.host_reset)
        m = np.ones(B, bool) if mask is None else torch.as_tensor(mask).cpu().numpy().astype(bool).reshape(-1)
        pos, vel = (np.array(x, np.float64) for x in world.get_state(all_entities=True))
        keep = world.rng_mode != "numpy"          # (numpy mode: the process-global stream is consumed, as the reference does)
        state = np.random.get_state() if keep else None
        try:
            if getattr(self, "_shadow", None) is None:
                self._shadow = sc.make_world()
            cw = self._shadow
            ents = list(cw.agents) + list(cw.landmarks)
            utter = np.zeros((B, t.A, max(t.dim_c, 1)))
            K = np.zeros((B, len(t.pops)), np.int64)
            from . import symtrace
            for b in np.flatnonzero(m):
                if seeds is not None:
                    np.random.seed(int(seeds[b]) & 0xFFFFFFFF)
                elif keep:
                    np.random.seed([int(world.seed) & 0x7FFFFFFF, int(world._episode) & 0x7FFFFFFF, (int(world.world_offset) + int(b)) & 0x7FFFFFFF])
                with symtrace.logged_picks(symtrace.PickLogger()) as lg:      # (np.random.choice as NumPy's own, its outcomes noted)
                    sc.reset_world(cw)
                if len(lg.log) != t.real_picks():
                    raise _abi.MpeError("reset_world of world %d made %d picks, the trace has %d" % (b, len(lg.log), t.real_picks()))
                K[b, :t.real_picks()] = lg.log
                for k, e in enumerate(ents):
                    pos[b, k] = np.asarray(e.state.p_pos, np.float64).reshape(2)
                    vel[b, k] = 0.0 if e.state.p_vel is None else np.asarray(e.state.p_vel, np.float64).reshape(2)
                for i, a in enumerate(cw.agents):
                    if t.dim_c and a.state.c is not None:
                        utter[b, i, :t.dim_c] = np.asarray(a.state.c, np.float64).reshape(-1)[:t.dim_c]
        finally:
            if keep:
                np.random.set_state(state)
        world._episode += 1
        world.set_state(pos, vel)
        mt = torch.as_tensor(m, device=world.device)
        for i, a in enumerate(world.agents):
            if torch.is_tensor(a.state.c) and a.state.c.numel():
                new = torch.as_tensor(utter[:, i, :a.state.c.shape[-1]], dtype=a.state.c.dtype, device=world.device)
                a.state.c.copy_(torch.where(mt[:, None], new, a.state.c))
        self._merge_picks(world, torch.as_tensor(K), mask)

    # ---- the row program ------------------------------------------------------------------------------------------------------
    def row_source(self, world):
        if self._source is None:
            from . import symtrace
            self._source = symtrace.hip_source(self.t)
        return self._source

    def row_shared(self, world):
        """how many values the source's traced_shared computes once per world (symtrace.shared_tasks)"""
        self.row_source(world)
        return int(getattr(self.t, "n_shared", 0))

    def _hash(self, world):
        import hashlib
        return int.from_bytes(hashlib.sha256(self.row_source(world).encode()).digest()[:8], "little")

    def obs_spec(self, agent, world):
        from . import rowspec
        i = world.agents.index(agent)
        return rowspec.ObsSpec(world, agent).code(len(self.t.obs[i]), self._hash(world))

    def reward_spec(self, agent, world):
        from . import rowspec
        return rowspec.RewardSpec(world, agent).code(self._hash(world))

    def done_spec(self, agent, world):
        from . import rowspec
        i = world.agents.index(agent)
        if self.t.done[i] is None:
            return None
        return rowspec.DoneSpec(world, agent).code(self._hash(world))

    # ---- benchmark_data (make_env(..., benchmark=True)): traced too, evaluated by one more launch after the step -------------
    def benchmark_data(self, agent, world):
        """The file's benchmark_data(agent, world) for all worlds: the info graphs run as a second row program over the post-step
        state (one `mpe_rows` launch per step, issued when the env asks for the first agent's), delivered in the file's own
        structure.  (With auto_reset, a world that restarted in this step reports its new episode's first state.)"""
        if self.t.info is None:
            return {}
        i = world.agents.index(agent)
        if self._info is None:
            self._info = _InfoProgram(self, world)
        if i == 0:
            env = self._env() if self._env is not None else None
            comm = getattr(env, "_comm", None) if env is not None else None
            self._info.launch(world, comm, _abi.raw_stream(world.device))
        return self._info.deliver(i, self.t.info_desc[i], world.batch_size)


def _trace_cache_path(scenario, want_done):
    """lib/rows_cache/trace_<sha256 of the scenario's source files + the tracer's>.json, or None when the sources cannot be read."""
    import hashlib
    import inspect
    from . import _build, symtrace
    h = hashlib.sha256(("trace format 1, %r;" % (want_done,)).encode())
    try:
        files = [inspect.getsourcefile(symtrace)]
        for klass in type(scenario).__mro__:
            if klass.__module__ not in ("builtins",) and not klass.__module__.startswith(__name__.rsplit(".", 1)[0] + "."):
                f = symtrace.source_file(klass)
                if f not in files:
                    files.append(f)
        for f in files:
            with open(f, "rb") as fh:
                h.update(fh.read())
        import pickle
        h.update(pickle.dumps(sorted(scenario.__dict__.items())))      # a scenario parametrised through its attributes: part of the key
    except Exception:      # (sources not readable, attributes that do not pickle: no cache)
        return None
    return os.path.join(_build.ROWS_CACHE, "trace_%s.json" % h.hexdigest()[:32])


class _InfoProgram(object):
    """benchmark_data of a traced scenario as a second row program: its "rows" are the info values (one launch of mpe_rows after
    the step), delivered in the structure the file returns -- numbers as [B] tensors (ints / bools as int32), arrays as [B, n],
    tuples as tuples."""

    def __init__(self, ts, world, compile=True):
        import ctypes as C
        import hashlib
        from . import rowspec, symtrace
        t = ts.t

        class _T(object):
            pass
        ti = _T()
        ti.obs, ti.rew, ti.done = t.info, [t.graph.const(0.0)] * t.A, [None] * t.A
        self.source = symtrace.hip_source(ti)
        h = int.from_bytes(hashlib.sha256(self.source.encode()).digest()[:8], "little")
        self.widths = [len(row) for row in t.info]
        obs = [rowspec.ObsSpec(world, a).code(self.widths[i], h) if self.widths[i] else rowspec.ObsSpec(world, a)
               for i, a in enumerate(world.agents)]
        rew = [rowspec.RewardSpec(world, a) for a in world.agents]
        if not any(self.widths):
            self.prog = None
            return
        self.prog = rowspec.RowProgram(world, obs, rew, source=self.source)
        src = world.scenario_desc(_abi.MPE_SCN_GENERIC)
        d = _abi.MpeScenarioDesc()
        C.memmove(C.byref(d), C.byref(src), C.sizeof(d))
        off = [0]
        for wd in self.widths:
            off.append(off[-1] + wd)
        for i, o in enumerate(off):
            d.obs_off[i] = o
        d.collaborative = 0
        self.desc, self.off = d, off
        self.prog.validate(d)
        if compile:
            self.prog.compile(d)
        self.rows = torch.zeros(off[-1] * world.batch_size, dtype=torch.float32, device=world.device)

    def launch(self, world, comm, stream):
        import ctypes as C
        if self.prog is None:
            return
        b = _abi.MpeBuffers()
        b.pos, b.vel, b.obs = world.pos.data_ptr(), world.vel.data_ptr(), self.rows.data_ptr()
        if world.choice_i32 is not None:
            b.choice = world.choice_i32.data_ptr()
        if comm is not None:
            b.comm = comm.data_ptr()
        _abi.check(_abi.lib().mpe_rows(C.byref(self.desc), C.byref(b), self.prog.ref, world.batch_size, stream), "mpe_rows (benchmark_data)")

    def deliver(self, i, desc, B):
        if self.prog is None or not self.widths[i]:
            return {}
        rows = self.rows[self.off[i] * B: self.off[i + 1] * B].view(B, self.widths[i])
        pos = [0]

        def build(d):
            if d[0] == "none":
                return {}
            if d[0] == "s":
                col = rows[:, pos[0]]
                pos[0] += 1
                return col.round().to(torch.int32) if d[1] == "i" else col.clone()
            if d[0] == "a":
                blk = rows[:, pos[0]:pos[0] + d[1]]
                pos[0] += d[1]
                return blk.round().to(torch.int32) if d[2] == "i" else blk.clone()
            return tuple(build(x) for x in d[1])
        return build(desc)


def trace_ref_scenario(scenario, want_done=False, verify_worlds=64, cache=True, want_info=False):
    """symtrace.trace + symtrace.verify -> TracedRefScenario; raises symtrace.TraceUnsupported (with the reason) when the
    file is outside what the tracer models or the trace does not reproduce the file's own callbacks.  The trace is cached by
    the content of the file (simple_world_comm's 4096 paths per agent take seconds); the verification runs every time."""
    import json
    from . import symtrace
    path = _trace_cache_path(scenario, (bool(want_done), bool(want_info))) if cache else None
    t = None
    if path is not None and os.path.exists(path):
        try:
            with open(path) as fh:
                t = symtrace.from_dict(json.load(fh))
            symtrace.verify(scenario, t, worlds=8)        # (a stale or foreign cache entry: traced afresh)
        except Exception:
            t = None
    if t is None:
        t = symtrace.trace(scenario, want_done=want_done, want_info=want_info)
        if path is not None:
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                tmp = path + ".tmp%d" % os.getpid()
                with open(tmp, "w") as fh:
                    json.dump(symtrace.to_dict(t), fh, separators=(",", ":"))
                os.replace(tmp, path)
            except OSError:
                pass
    if len(t.pops) > _abi.MPE_MAX_CHOICES:
        raise symtrace.TraceUnsupported("reset_world makes %d np.random.choice draws (at most %d per-world picks)" % (len(t.pops), _abi.MPE_MAX_CHOICES))
    if t.E > _abi.MPE_ROWS_MAX_ENTITIES:
        raise symtrace.TraceUnsupported("%d entities (row programs cover %d)" % (t.E, _abi.MPE_ROWS_MAX_ENTITIES))
    try:
        t.verified = symtrace.verify(scenario, t, worlds=verify_worlds)
    except symtrace.TraceUnsupported as e:
        if getattr(t, "host_reset", None):      # (say which reset it was: what such a reset keeps outside the state vectors cannot be followed)
            raise symtrace.TraceUnsupported("%s [reset_world could not be traced (%s) and was run as it is: whatever it draws per world and "
                                            "keeps outside the entities' state vectors -- a goal, a role -- is hidden state to the trace]"
                                            % (e, t.host_reset))
        raise
    ts = TracedRefScenario(scenario, t)
    ts.row_source(None)          # (generated now: a program too large for straight-line code is refused here, with the reason)
    return ts


def make_traced_env(ts, batch_size, device=None, seed=0, max_episode_steps=None, auto_reset=False, fresh_outputs=False, benchmark=False):
    """The env of a TracedRefScenario (or of a symtrace.Traced / its to_dict() data: a trace made elsewhere)."""
    from . import symtrace
    from .environment import MultiAgentEnv
    if isinstance(ts, dict):
        ts = symtrace.from_dict(ts)
    if isinstance(ts, symtrace.Traced):
        ts = TracedRefScenario(None, ts)
    world = ts.make_world(int(batch_size), device)
    if not world.pos.is_cuda:
        raise symtrace.TraceUnsupported("traced programs run compiled in on a HIP device (this world is on %s)" % world.device)
    world.seed = seed
    world.rng_mode = "device"
    ts.reset_world(world)
    info_cb = ts.benchmark_data if benchmark and ts.t.info is not None else None
    env = MultiAgentEnv(world, ts.reset_world, None, None, info_cb, None, fresh_outputs=fresh_outputs, fused=True,
                        max_episode_steps=max_episode_steps, auto_reset=auto_reset, compile_program=True)
    env.scenario, env.ref_scenario, env.traced, env.trace_fallback = ts, ts.scenario, True, None
    import weakref
    ts._env = weakref.ref(env)
    if info_cb is not None:
        ts.benchmark_data(world.agents[0], world)      # (build and compile the info program now, not at the first step)
    return env


def _is_num(x):
    return isinstance(x, (int, float, bool, np.integer, np.floating, np.bool_)) or (isinstance(x, np.ndarray) and x.ndim == 0)


def _stack_info(vals, ad):
    if all(_is_num(v) for v in vals):
        arr = np.asarray(vals)
        return ad._tensor(arr, torch.int32 if arr.dtype.kind in "iub" else torch.float32)
    if all(isinstance(v, tuple) for v in vals) and len(set(len(v) for v in vals)) == 1 and \
            all(_is_num(x) for v in vals for x in v):
        return tuple(_stack_info([v[k] for v in vals], ad) for k in range(len(vals[0])))
    return list(vals)


def _keyed_uniform(seed, episode, world_offset, B, n, dev):
    """[n, B] float64 uniforms in [0, 1) with 24 random bits each: draw j of world b = a splitmix64-style hash of
    (seed, world_offset + b, episode, j), computed with torch integer ops on `dev` (int64 wrap-around arithmetic)."""
    M = (1 << 64) - 1

    def s64(v):      # a Python int as the int64 with the same low 64 bits
        v &= M
        return v - (1 << 64) if v >= (1 << 63) else v

    def shr(x, k):   # logical shift right of int64 tensors
        return (x >> k) & ((1 << (64 - k)) - 1)
    key = (seed * 0x9E3779B97F4A7C15 + (episode + 1) * 0xD1B54A32D192ED03) & M
    w = torch.arange(B, dtype=torch.int64, device=dev) + int(world_offset)
    j = torch.arange(n, dtype=torch.int64, device=dev)[:, None]
    x = (w[None, :] * s64(0xBF58476D1CE4E5B9)) ^ ((j + 1) * s64(0x94D049BB133111EB)) ^ s64(key)
    for mul in (0xBF58476D1CE4E5B9, 0x94D049BB133111EB):
        x = (x ^ shr(x, 30)) * s64(mul)
        x = x ^ shr(x, 27)
    x = x ^ shr(x, 31)
    return shr(x, 40).to(torch.float64) * (2.0 ** -24)


def make_ref_env(scenario, benchmark=False, batch_size=None, device=None, seed=0, max_episode_steps=None, auto_reset=False,
                 done_callback=False, traced=None, fresh_outputs=False, verbose=False):
    """make_env for a reference-style Scenario object (make_env.py:36-43): one world with NumPy in / NumPy out when
    `batch_size` is None -- the reference's own use, np.random stream included --, B worlds with [B, .] tensors otherwise.

    traced (batched envs on a HIP device): None -- trace the file's callbacks into a compiled row program when that works
    (one launch per step; `env.traced` says whether, `env.trace_fallback` why not), else the host path; True -- trace or
    raise; False -- always the host path (B shadow worlds, the file's callbacks per world).  benchmark=True: the file's
    benchmark_data is traced as well (one more launch per step); on the host path it is evaluated per world."""
    from .environment import MultiAgentEnv
    compat = batch_size is None
    why = None
    if traced is not False and not compat:
        from . import symtrace
        try:
            want_done = bool(done_callback) and hasattr(scenario, "done")
            want_info = bool(benchmark) and hasattr(scenario, "benchmark_data")
            ts = trace_ref_scenario(scenario, want_done=want_done, want_info=want_info)
            return make_traced_env(ts, batch_size, device=device, seed=seed, max_episode_steps=max_episode_steps,
                                   auto_reset=auto_reset, fresh_outputs=fresh_outputs, benchmark=want_info)
        except Exception as e:      # auto mode: whatever stops the trace (TraceUnsupported, no hipcc: OSError, the file's own
            if traced:              # callback raising on the verifier's worlds ...) the host path is always correct
                raise
            why = "%s: %s" % (type(e).__name__, e)
            if verbose or int(batch_size) >= 1024:      # (a large batch on the host path is orders of magnitude slower: say so once)
                import warnings
                warnings.warn("reference-style scenario on the HOST path -- its callbacks run per world in Python (%s); "
                              "env.trace_fallback holds the reason" % why, RuntimeWarning, stacklevel=3)
    elif traced:
        raise _abi.MpeError("traced=True needs batch_size (a batched env)")
    ad = RefScenarioAdapter(scenario, 1 if compat else int(batch_size), device, host_outputs=compat)
    world = ad.world
    world.seed = seed
    world.rng_mode = "numpy" if compat else "device"      # where action / communication noise is drawn from (core.py:138,176)
    info_cb = ad.benchmark_data if benchmark and hasattr(scenario, "benchmark_data") else None
    # (the reference's make_env passes no done_callback, make_env.py:41-43: done stays False unless asked for)
    done_cb = ad.done if done_callback and hasattr(scenario, "done") else None
    env = MultiAgentEnv(world, ad.reset_world, ad.reward, ad.observation, info_cb, done_cb,
                        numpy_io=compat, fresh_outputs=True, fused=False,
                        max_episode_steps=max_episode_steps, auto_reset=auto_reset)
    env.scenario = ad
    env.ref_scenario = scenario
    env.ref_worlds = ad.shadows          # the file's own world objects, one per world, kept current after every step / reset
    env.traced, env.trace_fallback = False, why
    return env
