/*
 * mpe_hip.h -- C ABI of libmpe_hip.so: the batched particle-world hot path for MI355X (gfx950).
 *
 * The reference (openai/multiagent-particle-envs) has no native/FFI boundary: its hot path is
 * Python.  Each entry point below therefore names the reference *Python*
 * function(s) whose body it replaces; INTEGRATION.md shows the ctypes stub that binds it.
 *
 * Conventions (all entry points)
 *   - plain C, no C++/torch types; every array argument is a raw DEVICE pointer into memory the
 *     caller owns (torch tensors in our host layer).  The library never allocates device memory,
 *     never keeps a pointer after returning and has no global state besides a thread-local
 *     error string.
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *     no hidden synchronisation.  The device is whatever the caller made current.
 *   - returns 0 on success, a negative MPE_E* code for argument errors, or a positive hipError_t.
 *     Never throws, never exits.  mpe_last_error() describes the last failure on this thread.
 *   - B independent worlds are stepped in lock-step; world b never reads another world's data.
 *
 * Device data layout (fp32, structure-of-arrays, batch index innermost => coalesced over b)
 *   pos   [E][2][B]   entity positions, agents first then landmarks   (EntityState.p_pos, core.py:4-9)
 *   vel   [A][2][B]   agent velocities (EntityState.p_vel); the shipped scenarios' landmarks are immovable and never
 *                     integrated (core.py:160).  With MOVABLE landmarks (core.py:158-169 integrates every movable
 *                     entity) vel is [n_dyn][2][B], n_dyn = index of the last movable entity + 1: see mpe_world_step
 *   act   [A][B][5]   per-agent action rows as the reference takes them  (environment.py:174-175)
 *   ids   [A][B]      int32 action ids, the discrete_action_input form    (environment.py:161-167)
 *   obs   agent i's block starts at float offset B*obs_off[i]; inside it [B][D_i] row-major, i.e.
 *         exactly the [B, D_i] array obs_n[i] of the drop-in API        (environment.py:93)
 *   rew   [A][B]   done [A][B] (uint8, always 0: environment.py:132-135)
 *   info_* [A][B]  benchmark_data columns                                (simple_spread.py:47-63, simple_tag.py:57-66)
 *   comm  [A][B][dim_c] communication action rows of the agents that speak (environment.py:183-190)
 *   choice [K][B]  int32 per-world picks of reset_world, e.g. the goal landmark index (simple_adversary.py:44)
 */
#ifndef MPE_HIP_H_
#define MPE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPE_ABI_VERSION 4 /* layout of the structs below + meaning of existing entry points; new entry points are additive
                             (4: MpeRowProgram.traced, reset_boxes) */
#define MPE_MAX_ENTITIES 512 /* agents + landmarks per world */
#define MPE_ACTION_DIM 5     /* Discrete(dim_p*2+1), environment.py:45 */
#define MPE_MAX_CHOICES 4    /* np.random.choice draws a reset_world makes before the positions */

/* error codes (negative); positive return values are hipError_t */
#define MPE_OK 0
#define MPE_EINVAL (-1)       /* NULL / inconsistent argument */
#define MPE_EUNSUPPORTED (-2) /* shape or feature this build has no kernel for */

/* Which Scenario's observation()/reward() the fused output stage computes. */
enum MpeScenarioKind {
  MPE_SCN_GENERIC = 0, /* physics only (World.step); obs/reward are the caller's business      */
  MPE_SCN_SIMPLE = 1,  /* multiagent/scenarios/simple.py:41-50                                 */
  MPE_SCN_SPREAD = 2,  /* multiagent/scenarios/simple_spread.py:47-100                         */
  MPE_SCN_TAG = 3,     /* multiagent/scenarios/simple_tag.py:57-147                            */
  MPE_SCN_ADVERSARY = 4, /* multiagent/scenarios/simple_adversary.py:76-139 (per-world goal landmark) */
  MPE_SCN_PUSH = 5,      /* multiagent/scenarios/simple_push.py:60-96      (per-world goal landmark) */
  MPE_SCN_SPEAKER_LISTENER = 6, /* multiagent/scenarios/simple_speaker_listener.py:63-92 (comm, dim_c 3)   */
  MPE_SCN_REFERENCE = 7,        /* multiagent/scenarios/simple_reference.py:57-83        (comm, dim_c 10)  */
  MPE_SCN_CRYPTO = 8,           /* multiagent/scenarios/simple_crypto.py:97-169          (comm only, dim_c 4) */
  MPE_SCN_WORLD_COMM = 9        /* multiagent/scenarios/simple_world_comm.py:143-289     (leader comm, dim_c 4) */
};

/*
 * Per-scenario constants: what the reference keeps as attributes on World / Entity / Agent
 * objects (core.py:27-51, 59-79, 83-99) and sets in Scenario.make_world.  Host POD, read during
 * the call only.  Entities are ordered agents [0,A) then landmarks [A,A+L) (core.py:103-104).
 */
typedef struct MpeScenarioDesc {
  int32_t kind;          /* enum MpeScenarioKind */
  int32_t n_agents;      /* A >= 1 */
  int32_t n_landmarks;   /* L >= 0 */
  int32_t dim_c;         /* World.dim_c; in-scope agents are silent so only obs width uses it */
  int32_t n_adversaries; /* simple_tag: agents [0,n_adversaries) are adversaries              */
  int32_t collaborative; /* world.collaborative: every agent gets the SUM (environment.py:100-102) */
  float dt;              /* World.dt              0.1  */
  float damping;         /* World.damping         0.25 */
  float contact_force;   /* World.contact_force   1e2  */
  float contact_margin;  /* World.contact_margin  1e-3 */
  float size[MPE_MAX_ENTITIES];      /* Entity.size                                            */
  float mass[MPE_MAX_ENTITIES];      /* Entity.mass (= initial_mass, 1.0)                      */
  float accel[MPE_MAX_ENTITIES];     /* action sensitivity: Agent.accel or 5.0 (environment.py:178-181) */
  float max_speed[MPE_MAX_ENTITIES]; /* Entity.max_speed; < 0 means None (no clamp)            */
  uint8_t movable[MPE_MAX_ENTITIES]; /* Entity.movable (landmarks: kind GENERIC / mpe_world_step only) */
  uint8_t collide[MPE_MAX_ENTITIES]; /* Entity.collide                                         */
  int32_t obs_off[MPE_MAX_ENTITIES + 1]; /* prefix sums of per-agent obs widths D_i, [A+1] used */
  int32_t n_choices;                     /* per-world picks drawn at reset (goal = np.random.choice(landmarks), */
  int32_t choice_pop[MPE_MAX_CHOICES];   /* simple_adversary.py:44): how many, and the population size of each  */
} MpeScenarioDesc;

/* Device buffers of one batch of worlds (see layout above).  NULL = not used by this call. */
typedef struct MpeBuffers {
  float *pos;
  float *vel;
  const float *act;   /* exactly one of act / ids / u for calls that consume actions */
  const int32_t *ids;
  const float *u;     /* [A][2][B] already-decoded Action.u (World.step() entered directly, core.py:117) */
  float *obs;
  float *rew;
  uint8_t *done;
  float *info_rew;          /* spread: per-agent reward before the shared sum; simple_adversary: [A][B] squared distance
                               to the goal landmark (benchmark_data, simple_adversary.py:57-67)                          */
  int32_t *info_collisions; /* spread: #agents in contact incl. self (Q1); tag and simple_world_comm: an adversary's
                               #contacts with good agents (0 for the good agents; simple_world_comm.py:115-124)          */
  float *info_min_dists;    /* spread: [A][B]; simple_adversary: [L][A][B] squared distance to landmark l (with info_rew) */
  int32_t *info_occupied;   /* spread */
  float *force;             /* [A][2][B] scratch, only for the phase-level entry points        */
  const float *entity_table; /* device copy of mpe_fill_entity_table(); needed when A+L > 16   */
  const float *comm;         /* [A][B][dim_c] communication action rows (Action.c, environment.py:183-190); an agent's
                                row becomes its AgentState.c (core.py:171-177, no c_noise); rows of silent agents unused */
  int32_t *choice;           /* [n_choices][B] per-world picks (landmark indices): read by step/observe of the
                                scenarios that have them, written by mpe_reset / in-kernel resets             */
} MpeBuffers;

/* ---- library / binding sanity -------------------------------------------------------------- */
int mpe_abi_version(void);
const char *mpe_last_error(void);
size_t mpe_sizeof_desc(void);
size_t mpe_sizeof_buffers(void);
size_t mpe_sizeof_row_program(void);
size_t mpe_sizeof_step_server(void);
/* Observation widths of the built-in scenarios: fills desc->obs_off[0..A]; returns D_total or <0. */
int mpe_fill_obs_layout(MpeScenarioDesc *desc);
/* Per-entity constants as one float table for the wave-per-world (large N) kernel:
 * returns the number of floats; writes them to host_out when it is not NULL. */
int mpe_fill_entity_table(const MpeScenarioDesc *desc, float *host_out);

/* ---- the fused hot path --------------------------------------------------------------------
 * mpe_step: one MultiAgentEnv.step for B worlds in ONE kernel launch --
 *   _set_action (environment.py:144-181) -> World.step (core.py:117-131: apply_action_force
 *   :134-140, apply_environment_force :143-155 / get_collision_force :180-196, integrate_state
 *   :158-169, update_agent_state :171-177) -> per agent observation/reward/done/info
 *   (environment.py:92-97 -> Scenario.observation/reward/benchmark_data) -> shared-reward sum
 *   (environment.py:100-102).  Reads pos, vel, act|ids; writes pos, vel, obs, rew, done, info_*. */
int mpe_step(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, void *stream);

/* mpe_step_thread: the same step, same arguments, bit-identical results, on the thread-per-world kernel
 * family (one lane owns one world) instead of the wave-per-agent one -- the independent second
 * implementation the parity tests hold mpe_step to.  Shapes without a thread-per-world kernel (the
 * communication scenarios, A + L > 16) run the kernel mpe_step runs.                                  */
int mpe_step_thread(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, void *stream);

/* mpe_observe: the output half only (used by reset(): environment.py:106-116 -> _get_obs, and by
 * the phase-level tests): obs (+ rew/done/info when those pointers are set) from the current state. */
int mpe_observe(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, void *stream);

/* mpe_world_step: World.step only (core.py:117-131) incl. action decode; for user-written
 * scenarios whose observation/reward stay host-side tensor code.
 * Movable landmarks (desc->movable[e] != 0 for some e >= n_agents; a Landmark is an Entity, core.py:54-56, and
 * integrate_state core.py:158-169 integrates every movable entity, get_collision_force :194-195 pushes both
 * sides of a contact): with n_dyn = (index of the last movable entity) + 1, the entities [n_agents, n_dyn) are
 * stepped like agents without an action force -- same entity order, so forces are summed in the reference's
 * order.  The caller passes vel [n_dyn][2][B] and the decoded forces u [n_dyn][2][B] (rows [n_agents, n_dyn)
 * zero; act / ids cannot carry them).                                                                        */
int mpe_world_step(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, void *stream);

/* ---- phase-level entry points (one reference function each; tests + generic path) ----------- */
/* _set_action + World.apply_action_force (environment.py:144-181, core.py:134-140): force = u   */
int mpe_apply_action_force(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, void *stream);
/* World.apply_environment_force + get_collision_force (core.py:143-155,180-196): force += contacts */
int mpe_collision_force(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, void *stream);
/* World.integrate_state (core.py:158-169): vel,pos <- force */
int mpe_integrate_state(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, void *stream);

/* ---- device-side reset and synthetic actions (counter-based Philox4x32-10) -------------------
 * mpe_reset: Scenario.reset_world (simple_spread.py:31-45, simple_tag.py:39-54, simple.py:24-39)
 *   for the worlds whose mask byte is non-zero (mask == NULL: all): agents then landmarks,
 *   pos ~ U[-1,1)^2 (landmarks U[-r,r)^2 with r = landmark_range), vel = 0, and -- when desc->n_choices > 0
 *   and bufs->choice is set -- the per-world picks choice[k][b] ~ U{0..choice_pop[k]-1}.  The reference draws
 *   from NumPy's global MT19937; this draws from Philox keyed by (seed, world, episode) -- same
 *   distribution, different stream (seed-exact resets are done host-side).
 *   The generator is indexed by the GLOBAL world number world_offset + b, so a batch sharded over
 *   several GPUs (rank r owns worlds [r*B, (r+1)*B)) draws exactly what one big batch would.     */
int mpe_reset(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, const uint8_t *mask,
              float landmark_range, uint64_t seed, uint64_t episode, int64_t world_offset, void *stream);
/* mpe_reset_random_actions_block: mpe_reset(mask = NULL, ..., episode) AND mpe_random_actions_block(act, ids, ..., step0, T) in
 * ONE launch -- the episode boundary of a synthetic-policy rollout (Scenario.reset_world, then the moves of the episode's T steps:
 * what a caller of the reference does between `env.reset()` and the first `env.step()`); the same draws as the two calls.      */
int mpe_reset_random_actions_block(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, float landmark_range,
                                   uint64_t episode, float *act, int32_t *ids, uint64_t seed, uint64_t step0, int32_t T,
                                   int64_t world_offset, void *stream);
/* Uniform random moves: one-hot rows into act [A][B][5] and/or ids [A][B] (either may be NULL). */
int mpe_random_actions(float *act, int32_t *ids, int32_t n_agents, int64_t B, uint64_t seed,
                       uint64_t step, int64_t world_offset, void *stream);
/* The same for T consecutive global steps step0 .. step0+T-1 in one launch: act holds T consecutive
 * [A][B][5] tensors, ids T consecutive [A][B] tensors; tensor s is exactly what
 * mpe_random_actions(step0 + s) writes.  (A rollout that wants fresh moves for every step draws one
 * episode's worth per launch instead of paying a launch per step.)                               */
int mpe_random_actions_block(float *act, int32_t *ids, int32_t n_agents, int64_t B, uint64_t seed,
                             uint64_t step0, int32_t T, int64_t world_offset, void *stream);

/* Uniform random WORDS for the communication scenarios, for T consecutive global steps step0 .. step0+T-1: comm holds
 * T consecutive [A][B][dim_c] tensors; one-hot rows for the agents whose bit is set in `speakers` (bit a = agent a;
 * rows of the others are left alone) -- the communication half of a random action (environment.py:183-190), keyed
 * by (seed, global world, step, agent) like the moves.                                                        */
int mpe_random_comm(float *comm, int32_t n_agents, int64_t B, int32_t dim_c, uint32_t speakers, uint64_t seed,
                    uint64_t step0, int32_t T, int64_t world_offset, void *stream);

/* 1 when mpe_step / mpe_observe have a fused kernel for this descriptor (kind, agent / landmark /
 * adversary counts, dim_c), 0 when the caller must keep Scenario.observation / reward itself and
 * use mpe_world_step (a user Scenario: kind GENERIC), < 0 on an invalid descriptor or one no fused
 * kernel can take (MPE_EUNSUPPORTED: simple_speaker_listener / simple_reference / simple_crypto at other
 * than the reference's shapes, simple_world_comm with other than its one obstacle, two food items and
 * two forests, a built-in scenario with a movable landmark).  simple_spread and simple_tag are fused at
 * every size and every team split up to MPE_MAX_ENTITIES entities; simple_adversary and simple_world_comm
 * at the reference's team sizes and every other of a grid (csrc/mpe_split.hip, kSplitTable: 2-6 agents with
 * 1-2 adversaries; 1-3 good agents with 2-5 adversaries), 0 beyond it.                                */
int mpe_step_supported(const MpeScenarioDesc *desc);

/* Episode bookkeeping -- NEW API, no reference counterpart: `done` is always False in the reference
 * (environment.py:132-135, make_env.py:41-43: no done_callback) and the caller counts steps itself
 * (bin/interactive.py, MADDPG's 25-step episodes).  After a step: episode_step[w] += 1 for every
 * world; worlds that reach max_episode_steps (> 0) get done[a][w] = 1 for all n_agents rows (other
 * entries are left as the step / the done_callback wrote them); with clear_finished != 0 their
 * counter restarts at 0 (the caller resets them next: mpe_reset with mask = a done row).          */
int mpe_episode_tick(int32_t *episode_step, uint8_t *done, int32_t n_agents, int64_t B,
                     int32_t max_episode_steps, int32_t clear_finished, void *stream);

/* mpe_rollout_random: T consecutive env steps in ONE launch, with in-kernel uniform random moves
 * (the rows mpe_random_actions(step0+t) would write) and an in-kernel reset whenever the global
 * step index step0+t is a multiple of `episode_len` (0 = never; episode = (step0+t)/episode_len,
 * as mpe_reset draws it).  State stays on chip (registers / LDS) between steps; pos/vel are written after the
 * last step.  Every step's obs/rew/done/info ARE written: with trajectory != 0 the output buffers
 * hold T consecutive per-step blocks (obs: T x [B*obs_off[A]] floats; rew/done/info: T x [A][B]),
 * with trajectory == 0 step t overwrites the single block.
 * Bit-identical to T x { [mpe_reset]; mpe_random_actions(step0+t); mpe_step }.
 * Communication scenarios (kinds 6-9): the speaking agents' words are drawn in-kernel too (the rows
 * mpe_random_comm(step0+t) writes: speaker_listener agent 0, reference agents 0-1, crypto agents 0-2, world_comm
 * agent 0), per-world picks are re-drawn by the in-kernel resets, and bufs->comm receives the words of the LAST
 * step (the agents' comm state afterwards) -- this one entry point writes through the `comm` pointer.          */
int mpe_rollout_random(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, int32_t T,
                       int32_t episode_len, float landmark_range, uint64_t seed, uint64_t step0,
                       int64_t world_offset, int32_t trajectory, void *stream);

/* mpe_rollout_actions / mpe_rollout_rows_actions: mpe_rollout_random / mpe_rollout_rows with the CALLER's moves -- act_seq = T
 * consecutive [A][B][5] one-hot (or soft) action tensors, step t of the launch reads tensor t -- instead of moves drawn in the
 * kernel: the caller's `for t in range(T): env.step(actions[t])` loop (bin/interactive.py:27-36 pattern; environment.py:80-104 per
 * step) as ONE launch, bit-identical to the T mpe_step / mpe_step_rows launches.  mpe_rollout_actions covers simple_spread /
 * simple_tag beyond 16 entities (the wave-per-world kernels; the wave-per-agent shapes take the caller's moves through the step
 * server below: ring, then start, with srv->ahead = 1); mpe_rollout_rows_actions every row-program env whose agents do not speak.  */
int mpe_rollout_actions(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, int32_t T, int32_t episode_len,
                        float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset, int32_t trajectory,
                        const float *act_seq, void *stream);
/* (mpe_rollout_rows_actions is declared behind mpe_rollout_rows, below) */

/* ---- the step server: per-step commands to ONE resident launch -- NEW API, no reference counterpart ----------------------
 * The reference's caller loop is one env.step per policy decision (bin/interactive.py:27-36, environment.py:80-104); as a
 * launch per step every step pays the dependent-launch gap (1.0-2.0 us on top of the kernel's span).  A step server is ONE
 * launch that performs the same steps ON COMMAND: mpe_step_server_start puts it on a stream of its own for up to T steps;
 * the caller then commands steps one by one with mpe_step_server_ring -- a one-thread launch on the CALLER's stream, so it
 * is ordered behind whatever produced that step's moves; it adds n_steps to the doorbell word -- and the server executes
 * global step g as soon as the doorbell exceeds g (steps are numbered from 0 over the life of the doorbell word):
 *   moves    one-hot rows [A][B][5] in tensor g % ring of `act_ring` (ring consecutive tensors)
 *   outputs  obs / rew / done / info_* block g % slots of bufs' output buffers (blocks of obs_off[A] * B floats / A * B
 *            entries: `slots` consecutive blocks, as mpe_rollout_random's trajectory blocks); pos / vel after every step
 *   resets   every episode_len global steps (0 = never) inside the launch, mpe_reset's draws (as mpe_rollout_random)
 * and flag[wg] = g + 1 tells when workgroup wg's outputs of step g are in memory (mpe_step_server_wait: a launch on the
 * caller's stream that ends when all flags have reached a step count; launches behind it read that step's outputs).
 * Results are bit-identical to the T launches {mpe_reset at the boundaries; mpe_step}.  A server never outlives its
 * commander: a wave that waits longer than timeout_us for its next command sets *status (1; a timed-out wait: 2) and the
 * launch ends.  Who gains: callers whose moves for step g + 1 exist before step g has finished (recorded or scripted action
 * sequences, action repeat, several env instances interleaved, the random-action benchmark) -- their steps run back to back
 * at the kernel's span; a closed loop (policy reads step g's rows before it can command g + 1) pays ring + wait launches
 * and is better served by mpe_step.  All nine scenarios at the shapes with a wave-per-agent kernel (the communication scenarios'
 * utterances come from `comm_ring` as the moves from `act_ring`); a launch that may wait needs every 64-world workgroup resident
 * (mpe_step_server_start says so), one whose commands precede it (`ahead`) takes any batch size.
 * The two streams must map to different HARDWARE queues: HIP multiplexes streams onto a few of them, and a ring queued behind
 * the resident server on the same hardware queue never starts (the server then times out).  Probe before trusting a pair of
 * streams: mpe_step_server_wait on the candidate server stream with a short timeout_us, mpe_step_server_ring on the caller's
 * stream with door == flag == one scratch word -- *status stays 0 only if the two ran concurrently (rollout.StepServer).   */
typedef struct MpeStepServer {
  uint64_t *door;          /* device, 1 word, zeroed by the caller once: steps commanded so far (absolute, monotonic)   */
  uint64_t *flag;          /* device, mpe_step_server_flags(B) words, zeroed once: steps completed, per workgroup      */
  uint32_t *status;        /* device, 1 word, zeroed once: 0 = fine, 1 = the server timed out, 2 = a wait timed out    */
  const float *act_ring;   /* device, `ring` consecutive [A][B][5] move tensors                                         */
  int32_t ring, slots;
  uint64_t timeout_us;
  const float *comm_ring;  /* communication scenarios (kinds 5-8): `ring` consecutive [A][B][dim_c] utterance tensors (the speaking
                              agents' rows, Action.c: environment.py:183-190); step g reads tensor g % ring; NULL otherwise          */
  int32_t ahead;           /* != 0: every step of a launch is commanded BEFORE the launch starts (ring, then start, in stream order):
                              the launch never waits, so its workgroups need not all be resident -- any batch size            */
  int32_t reserved_;
} MpeStepServer;
int mpe_step_server_supported(const MpeScenarioDesc *desc, int64_t B);   /* 1 / 0 (< 0: invalid descriptor) */
int64_t mpe_step_server_flags(int64_t B);
int mpe_step_server_start(const MpeScenarioDesc *desc, const MpeBuffers *bufs, int64_t B, int32_t T, int32_t episode_len,
                          float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                          const MpeStepServer *srv, void *server_stream);
int mpe_step_server_ring(const MpeStepServer *srv, uint64_t n_steps /* door += n_steps */, void *caller_stream);
int mpe_step_server_wait(const MpeStepServer *srv, int64_t B, uint64_t steps_completed, void *caller_stream);

/* ---- composable output stage: a USER scenario's observation / reward as a row program ---------------------------------
 * The reference's plug-in promise (README "Creating new environments", scenario.py:4-10) is that new scenarios are the
 * normal use; every shipped observation is a concatenation of a few segment kinds and every shipped reward an ordered
 * sum of a few term kinds (simple_spread.py:72-100, simple_tag.py:84-147, simple_adversary.py:76-139, simple_push.py:60-96,
 * simple_speaker_listener.py:63-92, simple_reference.py:57-83, simple_crypto.py:97-169, simple_world_comm.py:143-289).
 * mpe_rows interprets such a list -- 16 bytes per op -- on the post-step state: a scenario nobody wrote a kernel for steps
 * in two launches, mpe_world_step + mpe_rows.  Entities are indexed agents [0, A) then landmarks; A + L <= 64 (and at
 * most 32 agents when regions hide agents from each other).
 *
 * An op is four int32 words: w0 = code | a0 << 8 | a1 << 16 | a2 << 24, w1 = an integer argument, w2 / w3 = float bits.
 * Observation ops append columns to the agent's row, in program order (a0 = MPE_ROW_SELF: the observing agent):          */
#define MPE_ROWS_MAX_ENTITIES 64
#define MPE_ROW_SELF 255
enum MpeRowOp {
  MPE_ROW_OBS_VEL = 1,       /* p_vel of entity a0 (2)                                                  */
  MPE_ROW_OBS_POS = 2,       /* p_pos of entity a0 (2)                                                  */
  MPE_ROW_OBS_REL = 3,       /* p_pos[a0] - p_pos[self] (2)                                             */
  MPE_ROW_OBS_REL_PICK = 4,  /* p_pos[w1 + choice[a1]] - p_pos[self] (2): the per-world goal            */
  MPE_ROW_OBS_COMM = 5,      /* AgentState.c of agent a0: a1 floats of its MpeBuffers.comm row          */
  MPE_ROW_OBS_CONST = 6,     /* the float w2 (1)                                                        */
  MPE_ROW_OBS_ONEHOT = 7,    /* a1 floats: w3 where choice[a0] + w1 == column, w2 elsewhere (colours)   */
  MPE_ROW_OBS_REL_VIS = 8,   /* OBS_REL, zeroed when self cannot see a0 (regions, simple_world_comm.py:231-261) */
  MPE_ROW_OBS_VEL_VIS = 9,   /* OBS_VEL, likewise                                                       */
  MPE_ROW_OBS_IN_REGION = 10,/* +1 / -1: is entity a0 inside region a1 (1)                              */
  /* range forms (what the host's peephole pass turns runs of the ops above into: one decode, an inner loop):
   * entities a0 .. a0 + a1 - 1, a2 & 1: skip the observing agent itself                                  */
  MPE_ROW_OBS_REL_RANGE = 11, MPE_ROW_OBS_VEL_RANGE = 12, MPE_ROW_OBS_REL_VIS_RANGE = 13, MPE_ROW_OBS_VEL_VIS_RANGE = 14,
  MPE_ROW_OBS_CONST_N = 15,  /* the float w2, a1 times                                                  */
  /* reward machine: a value register v, accumulators acc0 / acc1 (a2 & 1 selects), 8 slots.  Arithmetic in program
   * order is what fixes the rounding: the same order as the reference gives the reference's float (in fp32).         */
  MPE_ROW_R_D2 = 32,         /* v = |p[a0] - p[a1]|^2                                                   */
  MPE_ROW_R_MIN_D2 = 33,     /* v = min(v, |p[a0] - p[a1]|^2)                                           */
  MPE_ROW_R_D2_PICK = 34,    /* v = |p[a0] - p[w1 + choice[a1]]|^2                                      */
  MPE_ROW_R_MIN_D2_PICK = 35,
  MPE_ROW_R_SQRT = 36,       /* v = sqrt(v)                                                             */
  MPE_ROW_R_BOUND = 37,      /* v = bound(|coordinate a1 of p[a0]|), simple_tag.py:103-108              */
  MPE_ROW_R_COMM_ERR = 38,   /* v = sum_c (c[a0][c] - onehot(choice[a1])[c])^2, 0 for an all-zero utterance (simple_crypto.py:97-124) */
  MPE_ROW_R_COMM_SUM = 39,   /* v = sum_c c[a0][c]                                                      */
  MPE_ROW_R_CONST = 40,      /* v = w2                                                                  */
  MPE_ROW_R_SAVE = 41,       /* slot[a0] = v                                                            */
  MPE_ROW_R_LOAD = 42,       /* v = slot[a0]                                                            */
  MPE_ROW_R_ZERO = 43,       /* acc = 0                                                                 */
  MPE_ROW_R_ADD = 44,        /* acc = acc + w2 * v                                                      */
  MPE_ROW_R_ADD_IF_HIT = 45, /* if |p[a0] - p[a1]| < size[a0] + size[a1] (strict, exact): acc = acc + w2 */
  MPE_ROW_R_ADD_ACC = 46,    /* acc0 = acc0 + acc1                                                      */
  MPE_ROW_R_STORE = 47,      /* reward of agent a0 = acc0 (a0 = the agent whose program this is)        */
  MPE_ROW_R_MIN_D2_RANGE = 48,    /* v = min over a in a0 .. a0 + w1 - 1 of |p[a] - p[a1]|^2, in that order     */
  MPE_ROW_R_MIN_D2_TO_RANGE = 49, /* v = min over b in a1 .. a1 + w1 - 1 of |p[a0] - p[b]|^2                     */
  MPE_ROW_R_ADD_IF_HIT_GRID = 50, /* for a in a0 .. a0 + (w1 & 255) - 1, b in a1 .. a1 + (w1 >> 8) - 1: ADD_IF_HIT(a, b, w2) */
  MPE_ROW_R_ADD_MIN_DIST_GRID = 51,/* for b in a1 .. a1 + (w1 >> 8) - 1: v = sqrt(min over a in a0 .. a0 + (w1 & 255) - 1 of |p[a] - p[b]|^2);
                                      acc = acc + w2 * v   (simple_spread.py:72-77: minus the distance of the nearest agent, per landmark) */
  /* ---- done programs (MpeRowProgram.done_begin): Scenario.done of a user scenario (environment.py:132-135: done_callback) as
   * ordered tests on the same value machine; an agent's done = the OR of its tests (no test: False, the reference's default)  */
  MPE_ROW_R_ABS_POS = 52,       /* v = |p[a0][a1]|  (coordinate a1 of entity a0)                          */
  MPE_ROW_R_DONE_IF_GT = 53,    /* done = done or v > w2                                                  */
  MPE_ROW_R_DONE_IF_LT = 54,    /* done = done or v < w2                                                  */
  MPE_ROW_R_DONE_IF_HIT = 55,   /* done = done or |p[a0] - p[a1]| < size[a0] + size[a1] (strict, exact)   */
  /* ---- TRACED code (a program compiled in only, prog->traced = 1): the op calls a device function that the caller appended
   * to the generated header of mpe_rows_static_source -- `mpe::traced_obs(i, row, P, V, W, K)` writes the w1 columns of agent i's
   * row, `mpe::traced_rew(i, P, V, W, K, S)` / `traced_done(i, P, V, W, K)` return its reward / done -- where P(e, c) / V(e, c) read the staged
   * post-step position / velocity of entity e, W(j, c) agent j's utterance, K(k) the per-world pick k, S(k) shared value k (n_shared).  This is how an unmodified
   * reference-style Scenario file (multiagent/scenario.py:4-10: NumPy callbacks) runs in the step launch: its callbacks traced
   * into expression graphs (multiagent_particle_envs_amd/symtrace.py), one statement per arithmetic step.  w2 / w3 carry a hash
   * of the appended source (it is part of the program's identity: image name, cache key).                                   */
  MPE_ROW_OBS_CODE = 16,        /* observation program: w1 columns written by traced_obs (agent = the program's)           */
  MPE_ROW_R_CODE = 56,          /* reward program: acc0 = traced_rew                                                       */
  MPE_ROW_R_DONE_CODE = 57      /* done program: done = done or traced_done                                                */
};
/* Host POD describing one env's programs; the ops live in DEVICE memory the caller owns (uploaded once).               */
#define MPE_ROWS_HEADER_BYTES 4096 /* >= the kernel-side header (row layout, program ranges, per-entity constants) */
typedef struct MpeRowProgram {
  const int32_t *ops_device;  /* n_ops x 4 int32 words                                                  */
  void *header_device;        /* MPE_ROWS_HEADER_BYTES of caller-owned DEVICE memory: the library keeps the kernel-side header
                                 there (as kernel arguments its ~30 cache lines would be fetched from host memory one by one) and
                                 re-uploads it -- one 64-thread launch on the call's stream -- whenever its content changes       */
  uint64_t header_hash;       /* in / out: what header_device holds (0 = nothing yet).  A program is used by one thread at a time. */
  int32_t n_ops;
  int32_t obs_begin[MPE_ROWS_MAX_ENTITIES + 1]; /* agent i's observation ops: [obs_begin[i], obs_begin[i+1]); its row width is
                                                   desc->obs_off[i+1] - desc->obs_off[i] (filled by the caller)            */
  int32_t rew_begin[MPE_ROWS_MAX_ENTITIES + 1]; /* agent i's reward ops: [rew_begin[i], rew_begin[i+1]), ending in its STORE */
  int32_t n_vel;              /* rows of MpeBuffers.vel: entities [0, n_vel) have a velocity, the rest read 0      */
  int32_t n_regions;          /* 0..2 landmarks that hide what is inside them                            */
  int32_t region_entity[2];
  uint32_t all_seeing;        /* bit i: agent i sees everybody (simple_world_comm.py:253: the leader)   */
  void *image;                /* NULL, or what mpe_rows_load_image attached: the program compiled in (owned by the library)       */
  int32_t done_begin[MPE_ROWS_MAX_ENTITIES + 1]; /* agent i's done ops: [done_begin[i], done_begin[i+1]) -- value ops and DONE_IF_*
                                                    tests, run after its reward program on a fresh machine; all zero: no done programs */
  int32_t reset_boxes;        /* 1: reset_world places entity e uniformly in its own box -- x in [reset_box[e][0], + reset_box[e][1]),
                                 y in [reset_box[e][2], + reset_box[e][3]) (`np.random.uniform(lo, hi, dim_p)` per entity: a restricted
                                 spawn area, landmarks off-centre) -- for every restart the library draws for this program (in-launch
                                 episode ends, rollouts, mpe_episode_finish, mpe_reset_rows); 0: agents on [-1,1)^2, landmarks on
                                 [-landmark_range, landmark_range)^2 (the reference's nine scenarios)                               */
  float reset_box[MPE_ROWS_MAX_ENTITIES][4];   /* lo_x, span_x, lo_y, span_y per entity                                            */
  int32_t n_shared;           /* traced programs: how many values the appended `mpe::traced_shared(k, out, P, V, W, K)` computes -- parts of
                                 the agents' reward graphs that several agents share (the distance of the nearest agent to every
                                 landmark in a cooperative reward): evaluated ONCE per world, the tasks dealt to the workgroup's waves,
                                 parked in LDS, read by traced_rew through its accessor S(k).  0: none (<= 64)                      */
  int32_t traced;             /* 1: the program contains *_CODE ops -- it runs compiled in ONLY (an entry point called without a
                                 matching image returns MPE_EUNSUPPORTED instead of interpreting); 0 otherwise                  */
} MpeRowProgram;
/* Checks a program against the descriptor (entity / pick / slot indices, row widths == obs_off): ops_host are the same
 * n_ops x 4 words in HOST memory.  0 or MPE_EINVAL with mpe_last_error() naming the op.                               */
int mpe_rows_validate(const MpeScenarioDesc *desc, const MpeRowProgram *prog, const int32_t *ops_host);
/* Scenario.observation / reward / done of every agent from the CURRENT state (environment.py:92-102 after World.step, or
 * :113-115 after a reset): obs rows, rew (shared sum when desc->collaborative), done = 0.  desc->kind is ignored (GENERIC
 * is what a user scenario has); reads pos, vel, comm, choice; bufs->rew / done may be NULL.                            */
int mpe_rows(const MpeScenarioDesc *desc, const MpeBuffers *bufs, MpeRowProgram *prog, int64_t B, void *stream);
/* mpe_reset_rows: Scenario.reset_world of a USER scenario (README "Creating new environments"; the shape of simple_tag.py:39-54:
 * `entity.state.p_pos = np.random.uniform(lo, hi, world.dim_p)` per entity, zero velocities and utterances, np.random.choice picks)
 * -- mpe_reset with the PROGRAM's placement -- prog->reset_boxes ? its per-entity boxes : (agents [-1,1)^2, landmarks
 * [-landmark_range, landmark_range)^2) -- the same draws, keyed by (seed, world_offset + b, episode, entity), that the in-kernel
 * restarts of this program make: a rollout's per-step form {mpe_reset_rows at the boundaries; moves; mpe_step_rows} stays
 * bit-identical to mpe_rollout_rows whatever the placement.                                                              */
int mpe_reset_rows(const MpeScenarioDesc *desc, const MpeBuffers *bufs, const MpeRowProgram *prog, int64_t B, const uint8_t *mask,
                   float landmark_range, uint64_t seed, uint64_t episode, int64_t world_offset, void *stream);
/* mpe_step_rows: one MultiAgentEnv.step of a user scenario in ONE launch -- _set_action + World.step (environment.py:144-181,
 * core.py:117-177; exactly one of bufs->act / ids / u, no movable landmarks: those go through mpe_world_step) by the
 * agents' waves, then the row programs on the post-step state as in mpe_rows.  State bit-identical to mpe_world_step's.  */
int mpe_step_rows(const MpeScenarioDesc *desc, const MpeBuffers *bufs, MpeRowProgram *prog, int64_t B, void *stream);
/* mpe_episode_finish: what follows a step when episodes end on the device -- NEW API like mpe_episode_tick (the reference's
 * done is always False, environment.py:132-135) -- in ONE launch: count the step (episode_step[w] += 1; worlds at
 * max_episode_steps > 0 get done = 1 in every agent's row), find the finished worlds (any done row set, by the horizon or by
 * the caller's done callback), and for those: counter = 0, Scenario.reset_world (the draws of
 * mpe_reset(mask, landmark_range, seed, episode, world_offset): positions, zero velocities, per-world picks, utterances
 * zeroed) and the observation rows of the new episode's first state.  A workgroup (64 worlds) without a finished world
 * returns after reading its counters and flags: when nothing finished the call costs a launch and little else.         */
int mpe_episode_finish(const MpeScenarioDesc *desc, const MpeBuffers *bufs, MpeRowProgram *prog, int64_t B,
                       int32_t *episode_step, int32_t max_episode_steps, float landmark_range, uint64_t seed,
                       uint64_t episode, int64_t world_offset, void *stream);
/* mpe_step_rows_episode: mpe_step_rows and mpe_episode_finish in ONE launch, for a scenario whose done condition is part of
 * the program (done_begin) or that ends at the horizon only: the step, every agent's row / reward / done, then -- in the same
 * workgroup -- the step count, the finished worlds (an agent's done test fired, or max_episode_steps > 0 reached: done = 1 in
 * every agent's row), and for those reset_world (the draws of mpe_reset(mask, landmark_range, seed, episode, world_offset))
 * and the rows of the new episode's first state, exactly as the two calls would leave them.  A workgroup without a finished
 * world is done after one more barrier: an episode end inside the launch costs a step that ends nothing about nothing.    */
int mpe_step_rows_episode(const MpeScenarioDesc *desc, const MpeBuffers *bufs, MpeRowProgram *prog, int64_t B,
                          int32_t *episode_step, int32_t max_episode_steps, float landmark_range, uint64_t seed,
                          uint64_t episode, int64_t world_offset, void *stream);

/* mpe_rollout_rows: mpe_rollout_random for a USER scenario -- T consecutive steps of a row-program env in ONE launch: the worlds'
 * state stays in LDS, every agent's move is drawn in the kernel (the one-hot row mpe_random_actions_block(seed, step0 + t) would
 * write), every episode_len global steps (0 = never) every world is reset (mpe_reset(mask = NULL, landmark_range, seed, episode =
 * (step0 + t) / episode_len, world_offset)); step t's rows / rewards / dones go to trajectory block t of bufs->obs / rew / done
 * (trajectory != 0: blocks of obs_off[A] * B floats / A * B entries) or over block 0.  Bit-identical to the T launches of
 * {mpe_reset at the boundaries; mpe_random_actions_block; mpe_random_comm; mpe_step_rows}.  speakers: bit a = agent a says a drawn
 * one-hot word every step (mpe_random_comm's draws); after the launch bufs->comm holds the speakers' last words.               */
int mpe_rollout_rows(const MpeScenarioDesc *desc, const MpeBuffers *bufs, MpeRowProgram *prog, int64_t B, int32_t T,
                     int32_t episode_len, float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                     int32_t trajectory, uint32_t speakers, void *stream);
/* ... with the CALLER's moves (act_seq: T consecutive [A][B][5] tensors; see mpe_rollout_actions): agents that do not speak */
int mpe_rollout_rows_actions(const MpeScenarioDesc *desc, const MpeBuffers *bufs, MpeRowProgram *prog, int64_t B, int32_t T,
                             int32_t episode_len, float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                             int32_t trajectory, const float *act_seq, void *stream);

/* mpe_rollout_rows_episode: T consecutive mpe_step_rows_episode steps in ONE launch, the moves (and words) drawn in the kernel as in
 * mpe_rollout_rows: the episodes end, per world, where the program's done tests or max_episode_steps say -- the finished worlds
 * restart inside the step that ended them (episode number episode0 + t for step t: one per step, as a caller of
 * mpe_step_rows_episode counts them), their rows in that step's block are the new episode's first.  episode_step: the worlds'
 * step counters, read at the start, left as after step T - 1.                                                                  */
int mpe_rollout_rows_episode(const MpeScenarioDesc *desc, const MpeBuffers *bufs, MpeRowProgram *prog, int64_t B, int32_t T,
                             int32_t *episode_step, int32_t max_episode_steps, float landmark_range, uint64_t seed,
                             uint64_t step0, uint64_t episode0, int64_t world_offset, int32_t trajectory, uint32_t speakers,
                             void *stream);

/* ---- a row program COMPILED IN: the interpreter specialised away ---------------------------------------------------------
 * mpe_rows / mpe_step_rows / mpe_episode_finish interpret a program op by op.  For a program that stays the same for the
 * life of an env the same kernel source can be compiled WITH the program as constants (every op code, entity index, column
 * and per-entity constant a literal): the interpreter folds into straight-line code in the same arithmetic order -- results
 * bit-identical to the interpreted launch, at about the cost of a hand-fused kernel.  Three steps, each plain C:
 *   1. mpe_rows_static_source writes a generated header (a few #defines: name, dims, tables, ops) into `buf`
 *      (a traced program: the caller appends `#define MPE_ROWS_TRACED 1`, `#include "mpe_internal.h"` and its
 *      mpe::traced_obs / traced_rew / traced_done templates -- symtrace.hip_source writes exactly that);
 *   2. the caller compiles csrc/mpe_rows.hip with it:  hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
 *      -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=14 -include <header> -I include -I csrc mpe_rows.hip -o prog.hsaco
 *      (multiagent_particle_envs_amd/_build.py: compile_rows_image does exactly this and caches by content);
 *   3. mpe_rows_load_image attaches the code object to the program; the three entry points above then launch it whenever
 *      the call's descriptor still equals the one it was compiled for (constants are part of the image: after a change
 *      -- an entity resized, dt edited -- calls fall back to the interpreter, still correct, until a new image is loaded).
 * Programs of more than MPE_ROWS_STATIC_MAX_OPS ops stay interpreted (MPE_EUNSUPPORTED from steps 1 and 3).              */
#define MPE_ROWS_STATIC_MAX_OPS 512
int mpe_rows_static_source(const MpeScenarioDesc *desc, const MpeRowProgram *prog, const int32_t *ops_host, char *buf,
                           size_t cap, size_t *needed);
int mpe_rows_load_image(const MpeScenarioDesc *desc, MpeRowProgram *prog, const int32_t *ops_host, const void *image,
                        size_t bytes);
int mpe_rows_unload_image(MpeRowProgram *prog); /* 0; a program without an image is left alone */
/* 1 if the next mpe_rows / mpe_step_rows call with this descriptor would launch the compiled image, else 0.              */
int mpe_rows_image_active(const MpeScenarioDesc *desc, const MpeRowProgram *prog);

#ifdef __cplusplus
}
#endif
#endif /* MPE_HIP_H_ */
