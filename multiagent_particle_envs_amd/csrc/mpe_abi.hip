// mpe_abi.hip -- the extern "C" surface declared in include/mpe_hip.h (host code only).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "mpe_internal.h"

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_result(int rc, const char *what) {
  if (rc == 0) return 0;
  if (rc == MPE_EUNSUPPORTED) return fail(rc, "%s: no kernel built for this scenario shape", what);
  return fail(rc, "%s: %s", what, hipGetErrorString((hipError_t)rc));
}

// Which kernel family serves a step on the small-entity shapes: wave-per-agent (mpe_split.hip) is what
// mpe_step launches; thread-per-world (mpe_narrow.hip) is reachable through its own entry point
// mpe_step_thread -- an explicit argument of the call, not process state.  Both give bit-identical results.
enum class StepImpl { Split, Thread };

int check_desc(const MpeScenarioDesc *d, const char *what) {
  if (!d) return fail(MPE_EINVAL, "%s: desc is NULL", what);
  if (d->n_agents < 1 || d->n_landmarks < 0 || d->n_agents + d->n_landmarks > MPE_MAX_ENTITIES)
    return fail(MPE_EINVAL, "%s: need 1 <= A, 0 <= L, A+L <= %d (got A=%d L=%d)", what, MPE_MAX_ENTITIES,
                d->n_agents, d->n_landmarks);
  if (d->kind < MPE_SCN_GENERIC || d->kind > MPE_SCN_WORLD_COMM) return fail(MPE_EINVAL, "%s: bad kind %d", what, d->kind);
  // Movable landmarks (core.py:158-169 integrates EVERY movable entity; a Landmark is just an Entity, core.py:54-56) exist
  // on the physics path only -- kind GENERIC, i.e. mpe_world_step and the phase-free generic path above it; the built-in
  // scenarios' fused output stages read agent velocities only, so for them a movable landmark is "no kernel for this"
  // (mpe_step_supported < 1: the env steps such a world through mpe_world_step + the scenario's callbacks).
  for (int e = d->n_agents; e < d->n_agents + d->n_landmarks; ++e)
    if (d->movable[e] && d->kind != MPE_SCN_GENERIC)
      return fail(MPE_EUNSUPPORTED, "%s: a movable landmark (entity %d) is stepped by mpe_world_step (kind GENERIC) only", what, e);
  for (int e = 0; e < d->n_agents + d->n_landmarks; ++e)
    if ((e < d->n_agents || d->movable[e]) && !(d->mass[e] > 0.f)) return fail(MPE_EINVAL, "%s: mass[%d] must be > 0", what, e);
  const bool teams = d->kind == MPE_SCN_TAG || d->kind == MPE_SCN_ADVERSARY || d->kind == MPE_SCN_PUSH ||
                     d->kind == MPE_SCN_WORLD_COMM;
  // the communication scenarios exist in the reference's shapes only -- except simple_world_comm's team sizes, which its
  // callbacks leave open (simple_world_comm.py:126-289 loop over good_agents / adversaries): one obstacle, two food items
  // and two forests as in its make_world (the observation reads forests[0] and [1] by index, :241-248), any A and
  // n_adversaries that mpe_split.hip has an entry for (A = 0 below: not checked here)
  struct Shape { int kind, A, L, dim_c, n_choices, nadv; };
  static const Shape kComm[] = {{MPE_SCN_SPEAKER_LISTENER, 2, 3, 3, 1, 0}, {MPE_SCN_REFERENCE, 2, 3, 10, 2, 0},
                                {MPE_SCN_CRYPTO, 3, 2, 4, 2, 1}, {MPE_SCN_WORLD_COMM, 0, 5, 4, 0, 0}};
  for (const Shape &sh : kComm)
    if (d->kind == sh.kind && ((sh.A && d->n_agents != sh.A) || d->n_landmarks != sh.L || d->dim_c != sh.dim_c ||
                               d->n_choices != sh.n_choices || (sh.nadv && d->n_adversaries != sh.nadv)))
      return fail(MPE_EUNSUPPORTED, "%s: kind %d is built for A=%d L=%d dim_c=%d n_choices=%d (got A=%d L=%d dim_c=%d n_choices=%d)",
                  what, sh.kind, sh.A, sh.L, sh.dim_c, sh.n_choices, d->n_agents, d->n_landmarks, d->dim_c, d->n_choices);
  if (teams && (d->n_adversaries < 1 || d->n_adversaries >= d->n_agents))
    return fail(MPE_EINVAL, "%s: this scenario needs 1 <= n_adversaries < A", what);
  if ((d->kind == MPE_SCN_SPREAD || d->kind == MPE_SCN_TAG) && d->dim_c != 2)
    return fail(MPE_EUNSUPPORTED, "%s: built-in spread/tag kernels assume dim_c == 2 (got %d)", what, d->dim_c);
  if (d->n_choices < 0 || d->n_choices > MPE_MAX_CHOICES) return fail(MPE_EINVAL, "%s: bad n_choices %d", what, d->n_choices);
  for (int k = 0; k < d->n_choices; ++k)
    if (d->choice_pop[k] < 1) return fail(MPE_EINVAL, "%s: choice_pop[%d] must be >= 1", what, k);
  if ((d->kind == MPE_SCN_ADVERSARY || d->kind == MPE_SCN_PUSH) &&
      (d->n_landmarks < 1 || d->n_choices != 1 || d->choice_pop[0] != d->n_landmarks))
    return fail(MPE_EINVAL, "%s: adversary/push pick one goal among the landmarks (n_choices = 1, choice_pop[0] = L)", what);
  return 0;
}

bool use_narrow(const MpeScenarioDesc *d) {
  return d->n_agents + d->n_landmarks <= mpe::kNarrowMaxE &&
         mpe::narrow_supports(d->kind, d->n_agents, d->n_landmarks, d->n_adversaries);
}

mpe::NarrowDesc make_narrow(const MpeScenarioDesc *d, const MpeBuffers *b, size_t B) {
  mpe::NarrowDesc n;
  std::memset(&n, 0, sizeof(n));
  const int E = d->n_agents + d->n_landmarks;
  for (int e = 0; e < E; ++e) {
    n.size[e] = d->size[e];
    n.inv_mass[e] = d->mass[e] > 0.f ? 1.0f / d->mass[e] : 1.0f;
    n.accel[e] = d->accel[e];
    n.max_speed[e] = d->max_speed[e];
    if (d->movable[e]) n.movable |= 1u << e;
    if (d->collide[e]) n.collide |= 1u << e;
  }
  bool vec4 = b->obs && (reinterpret_cast<uintptr_t>(b->obs) % 16 == 0);
  for (int i = 0; i <= d->n_agents; ++i) {
    n.obs_off[i] = d->obs_off[i];
    if (((size_t)d->obs_off[i] * B) % 4 != 0) vec4 = false;
  }
  n.dt = d->dt;
  n.damp = 1.0f - d->damping;  // (1 - self.damping), core.py:161 (0.75 exactly for the default)
  n.cforce = d->contact_force;
  n.cmargin = d->contact_margin;
  n.cmargin_inv = 1.0f / d->contact_margin;
  n.collaborative = d->collaborative;
  n.vec4 = vec4 ? 1 : 0;
  n.n_choices = d->n_choices;
  for (int k = 0; k < MPE_MAX_CHOICES; ++k) n.choice_pop[k] = k < d->n_choices ? d->choice_pop[k] : 1;
  return n;
}

mpe::WideDesc make_wide(const MpeScenarioDesc *d) {
  mpe::WideDesc w;
  w.kind = d->kind;
  w.A = d->n_agents;
  w.L = d->n_landmarks;
  w.dim_c = d->dim_c;
  w.nadv = d->n_adversaries;
  w.collaborative = d->collaborative;
  w.D = d->obs_off[1] - d->obs_off[0];
  w.dt = d->dt;
  w.damp = 1.0f - d->damping;
  w.cforce = d->contact_force;
  w.cmargin = d->contact_margin;
  w.cmargin_inv = 1.0f / d->contact_margin;
  const int A = d->n_agents, E = d->n_agents + d->n_landmarks;
  bool homo = true;
  for (int i = 1; i < A; ++i)
    homo = homo && d->size[i] == d->size[0] && d->mass[i] == d->mass[0] && d->accel[i] == d->accel[0] &&
           d->max_speed[i] == d->max_speed[0] && d->movable[i] == d->movable[0] && d->collide[i] == d->collide[0];
  for (int e = A; e < E; ++e) homo = homo && !d->collide[e];
  w.homo = homo ? 1 : 0;
  w.rows_nt = 0;   // launch_wide decides (it knows the buffers)
  w.a_flags = (d->movable[0] ? 1 : 0) | (d->collide[0] ? 2 : 0);
  w.a_size = d->size[0];
  w.a_inv_mass = 1.0f / d->mass[0];
  w.a_accel = d->accel[0];
  w.a_max_speed = d->max_speed[0];
  return w;
}

int need(const void *p, const char *what, const char *name) {
  return p ? 0 : fail(MPE_EINVAL, "%s: bufs->%s is NULL", what, name);
}

int check_state(const MpeBuffers *b, int64_t B, const char *what) {
  if (!b) return fail(MPE_EINVAL, "%s: bufs is NULL", what);
  if (B < 0) return fail(MPE_EINVAL, "%s: B < 0", what);
  if (int rc = need(b->pos, what, "pos")) return rc;
  if (int rc = need(b->vel, what, "vel")) return rc;
  return 0;
}

int check_actions(const MpeBuffers *b, const char *what) {
  if ((b->act != nullptr) + (b->ids != nullptr) + (b->u != nullptr) != 1)
    return fail(MPE_EINVAL, "%s: exactly one of bufs->act / bufs->ids / bufs->u must be set", what);
  return 0;
}

int check_info(const MpeScenarioDesc *d, const MpeBuffers *b, const char *what) {
  if (d->kind == MPE_SCN_SPREAD && b->info_rew &&
      !(b->info_collisions && b->info_min_dists && b->info_occupied))
    return fail(MPE_EINVAL, "%s: spread benchmark_data needs all four info_* buffers", what);
  return 0;
}

}  // namespace

extern "C" {

int mpe_abi_version(void) { return MPE_ABI_VERSION; }
const char *mpe_last_error(void) { return g_err; }
size_t mpe_sizeof_desc(void) { return sizeof(MpeScenarioDesc); }
size_t mpe_sizeof_buffers(void) { return sizeof(MpeBuffers); }
size_t mpe_sizeof_row_program(void) { return sizeof(MpeRowProgram); }
size_t mpe_sizeof_step_server(void) { return sizeof(MpeStepServer); }

int mpe_fill_obs_layout(MpeScenarioDesc *d) {
  if (int rc = check_desc(d, "mpe_fill_obs_layout")) return rc;
  const int A = d->n_agents, L = d->n_landmarks;
  d->obs_off[0] = 0;
  for (int i = 0; i < A; ++i) {
    int D = 0;
    switch (d->kind) {
      case MPE_SCN_SIMPLE: D = 2 + 2 * L; break;                                   // simple.py:45-50
      case MPE_SCN_SPREAD: D = 4 + 2 * L + 2 * (A - 1) + d->dim_c * (A - 1); break;  // simple_spread.py:84-100
      case MPE_SCN_TAG: {                                                          // simple_tag.py:131-147
        const int good_others = (A - d->n_adversaries) - (i >= d->n_adversaries ? 1 : 0);
        D = 4 + 2 * L + 2 * (A - 1) + 2 * good_others;
        break;
      }
      case MPE_SCN_ADVERSARY:                                                      // simple_adversary.py:121-139
        D = (i < d->n_adversaries ? 0 : 2) + 2 * L + 2 * (A - 1);
        break;
      case MPE_SCN_PUSH:                                                           // simple_push.py:78-96
        D = i < d->n_adversaries ? 2 + 2 * L + 2 * (A - 1) : 2 + 2 + 3 + 2 * L + 3 * L + 2 * (A - 1);
        break;
      case MPE_SCN_SPEAKER_LISTENER: D = i == 0 ? 3 : 2 + 2 * L + d->dim_c; break;  // simple_speaker_listener.py:69-92
      case MPE_SCN_REFERENCE: D = 2 + 2 * L + 3 + d->dim_c; break;                  // simple_reference.py:63-83
      case MPE_SCN_CRYPTO: D = i == 0 ? d->dim_c : 2 * d->dim_c; break;             // simple_crypto.py:127-169
      case MPE_SCN_WORLD_COMM:                                                      // simple_world_comm.py:231-289
        D = i < d->n_adversaries ? 4 + 2 * L + 2 * (A - 1) + 2 * (A - d->n_adversaries) + 2 + d->dim_c
                                 : 4 + 2 * L + 2 * (A - 1) + 2 + 2 * (A - d->n_adversaries - 1);
        break;
      default: D = 0;
    }
    d->obs_off[i + 1] = d->obs_off[i] + D;
  }
  return d->obs_off[A];
}

int mpe_fill_entity_table(const MpeScenarioDesc *d, float *out) {
  if (int rc = check_desc(d, "mpe_fill_entity_table")) return rc;
  const int E = d->n_agents + d->n_landmarks;
  if (out) {
    for (int e = 0; e < E; ++e) {
      out[0 * E + e] = d->size[e];
      out[1 * E + e] = d->mass[e];
      out[2 * E + e] = d->accel[e];
      out[3 * E + e] = d->max_speed[e];
      out[4 * E + e] = d->movable[e] ? 1.f : 0.f;
      out[5 * E + e] = d->collide[e] ? 1.f : 0.f;
    }
  }
  return mpe::kEntityTableCols * E;
}

static int run(const char *what, bool phys, bool out, const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B,
               void *stream, StepImpl impl = StepImpl::Split) {
  if (int rc = check_desc(d, what)) return rc;
  if (int rc = check_state(b, B, what)) return rc;
  if (phys) if (int rc = check_actions(b, what)) return rc;
  if (out) {
    if (d->kind == MPE_SCN_GENERIC) return fail(MPE_EINVAL, "%s: kind GENERIC has no output stage", what);
    if (int rc = need(b->obs, what, "obs")) return rc;
    if (int rc = check_info(d, b, what)) return rc;
    if (d->n_choices > 0 && d->kind >= MPE_SCN_ADVERSARY)
      if (int rc = need(b->choice, what, "choice (the per-world picks of reset_world)")) return rc;
    if (d->kind >= MPE_SCN_SPEAKER_LISTENER)
      if (int rc = need(b->comm, what, "comm (communication action rows / current comm state)")) return rc;
  }
  if (B == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int kind = out ? d->kind : MPE_SCN_GENERIC;
  MpeScenarioDesc dd;
  const MpeScenarioDesc *use = d;
  if (!out && d->kind != MPE_SCN_GENERIC) {  // physics-only call on a built-in scenario
    dd = *d;
    dd.kind = MPE_SCN_GENERIC;
    use = &dd;
  }
  // ---- movable landmarks: the entities up to the last movable one are stepped as action-less agents ----------------------
  // World.step treats agents and landmarks alike except for the action force (core.py:134-140 adds it for agents only):
  // with n_dyn = (index of the last movable entity) + 1, entities [A, n_dyn) enter the physics kernels as agents whose
  // action force is zero -- same entity order, so every force is summed in the reference's order (Q9) -- and entities
  // [n_dyn, E) stay landmarks.  The caller's vel and u are [n_dyn][2][B] (rows [A, n_dyn) of u must be zero).
  int n_dyn = d->n_agents;
  for (int e = d->n_agents; e < d->n_agents + d->n_landmarks; ++e)
    if (d->movable[e]) n_dyn = e + 1;
  if (n_dyn > d->n_agents) {
    if (out) return fail(MPE_EUNSUPPORTED, "%s: movable landmarks are stepped by mpe_world_step only", what);
    if (!b->u) return fail(MPE_EINVAL, "%s: movable landmarks need bufs->u ([n_dyn][2][B], n_dyn = %d: zero rows for the landmarks)", what, n_dyn);
    if (use != &dd) dd = *d;
    dd.kind = MPE_SCN_GENERIC;
    dd.n_landmarks = d->n_agents + d->n_landmarks - n_dyn;
    dd.n_agents = n_dyn;
    use = &dd;
  }
  const bool comm_kind = kind >= MPE_SCN_SPEAKER_LISTENER;   // these exist as wave-per-agent kernels only
  // (an observe-only call goes to the thread-per-world kernel where one exists -- no barrier, no exchange block needed --
  //  and to the wave-per-agent kernel's observe half for the shapes that exist there only)
  if (out && (phys || comm_kind || !use_narrow(use)) && d->n_agents + d->n_landmarks <= mpe::kNarrowMaxE &&
      (comm_kind || impl != StepImpl::Thread) &&
      mpe::split_supports(kind, d->n_agents, d->n_landmarks, d->n_adversaries)) {
    // the fused step: wave-per-agent / lane-per-world (mpe_split.hip)
    const mpe::NarrowDesc n = make_narrow(use, b, (size_t)B);
    mpe::RollArgs ra;
    std::memset(&ra, 0, sizeof(ra));
    ra.T = 1;
    ra.observe_only = phys ? 0 : 1;
    return hip_result(mpe::launch_split(false, kind, d->n_agents, d->n_landmarks, d->n_adversaries, n, *b, (size_t)B,
                                        ra, s), what);
  }
  if (use_narrow(use)) {
    const mpe::NarrowDesc n = make_narrow(use, b, (size_t)B);
    return hip_result(mpe::launch_narrow(phys ? mpe::NarrowOp::Step : mpe::NarrowOp::Observe, kind, use->n_agents,
                                         use->n_landmarks, d->n_adversaries, n, *b, (size_t)B, s), what);
  }
  if (kind == MPE_SCN_GENERIC || kind == MPE_SCN_SPREAD || kind == MPE_SCN_TAG) {
    if (int rc = need(b->entity_table, what, "entity_table (required by the wave-per-world kernel)")) return rc;
    mpe::WideDesc w = make_wide(use);
    w.kind = kind;
    return hip_result(mpe::launch_wide(phys, out, w, *b, (size_t)B, s), what);
  }
  return fail(MPE_EUNSUPPORTED, "%s: no kernel for kind=%d A=%d L=%d n_adv=%d", what, d->kind, d->n_agents,
              d->n_landmarks, d->n_adversaries);
}

int mpe_step_supported(const MpeScenarioDesc *d) {
  if (int rc = check_desc(d, "mpe_step_supported")) return rc;
  if (d->kind == MPE_SCN_GENERIC) return 0;   // World.step only (mpe_world_step); callbacks stay with the caller
  const int A = d->n_agents, L = d->n_landmarks;
  if (A + L <= mpe::kNarrowMaxE && mpe::split_supports(d->kind, A, L, d->n_adversaries)) return 1;
  if (d->kind >= MPE_SCN_SPEAKER_LISTENER) return 0;
  if (use_narrow(d)) return 1;
  if (d->kind != MPE_SCN_SPREAD && d->kind != MPE_SCN_TAG) return 0;   // the wave-per-world kernel: any team sizes
  mpe::WideDesc w = make_wide(d);
  w.kind = d->kind;
  return mpe::wide_supports(w, true) ? 1 : 0;
}

int mpe_step(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, void *stream) {
  return run("mpe_step", true, true, d, b, B, stream);
}
int mpe_step_thread(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, void *stream) {
  return run("mpe_step_thread", true, true, d, b, B, stream, StepImpl::Thread);
}
int mpe_observe(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, void *stream) {
  return run("mpe_observe", false, true, d, b, B, stream);
}
int mpe_world_step(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, void *stream) {
  return run("mpe_world_step", true, false, d, b, B, stream);
}

static int run_phase(const char *what, int phase, const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B,
                     void *stream) {
  if (int rc = check_desc(d, what)) return rc;
  if (int rc = check_state(b, B, what)) return rc;
  if (int rc = need(b->force, what, "force")) return rc;
  if (phase == 0) if (int rc = check_actions(b, what)) return rc;
  if (d->n_agents + d->n_landmarks > mpe::kNarrowMaxE)
    return fail(MPE_EUNSUPPORTED, "%s: phase-level entry points cover A+L <= %d", what, mpe::kNarrowMaxE);
  if (B == 0) return 0;
  const mpe::NarrowDesc n = make_narrow(d, b, (size_t)B);
  return hip_result(mpe::launch_phase(phase, d->n_agents, d->n_landmarks, n, *b, (size_t)B,
                                      static_cast<hipStream_t>(stream)), what);
}
int mpe_apply_action_force(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, void *stream) {
  return run_phase("mpe_apply_action_force", 0, d, b, B, stream);
}
int mpe_collision_force(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, void *stream) {
  return run_phase("mpe_collision_force", 1, d, b, B, stream);
}
int mpe_integrate_state(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, void *stream) {
  return run_phase("mpe_integrate_state", 2, d, b, B, stream);
}

int mpe_reset(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, const uint8_t *mask, float landmark_range,
              uint64_t seed, uint64_t episode, int64_t world_offset, void *stream) {
  const char *what = "mpe_reset";
  if (int rc = check_desc(d, what)) return rc;
  if (int rc = check_state(b, B, what)) return rc;
  if (d->n_choices > 0 && b->choice == nullptr)
    return fail(MPE_EINVAL, "%s: desc->n_choices = %d but bufs->choice is NULL", what, d->n_choices);
  if (B == 0) return 0;
  int32_t pop[MPE_MAX_CHOICES] = {1, 1, 1, 1};
  for (int k = 0; k < d->n_choices; ++k) pop[k] = d->choice_pop[k];
  return hip_result(mpe::launch_reset(d->n_agents, d->n_landmarks, *b, (size_t)B, mask, landmark_range, seed,
                                      episode, (uint64_t)world_offset, d->n_choices, pop,
                                      static_cast<hipStream_t>(stream)), what);
}

int mpe_reset_rows(const MpeScenarioDesc *d, const MpeBuffers *b, const MpeRowProgram *p, int64_t B, const uint8_t *mask,
                   float landmark_range, uint64_t seed, uint64_t episode, int64_t world_offset, void *stream) {
  const char *what = "mpe_reset_rows";
  if (!p) return fail(MPE_EINVAL, "%s: prog is NULL", what);
  if (!p->reset_boxes) return mpe_reset(d, b, B, mask, landmark_range, seed, episode, world_offset, stream);
  if (int rc = check_desc(d, what)) return rc;
  if (int rc = check_state(b, B, what)) return rc;
  if (d->n_agents + d->n_landmarks > MPE_ROWS_MAX_ENTITIES) return fail(MPE_EINVAL, "%s: more than %d entities", what, MPE_ROWS_MAX_ENTITIES);
  if (d->n_choices > 0 && b->choice == nullptr)
    return fail(MPE_EINVAL, "%s: desc->n_choices = %d but bufs->choice is NULL", what, d->n_choices);
  if (B == 0) return 0;
  int32_t pop[MPE_MAX_CHOICES] = {1, 1, 1, 1};
  for (int k = 0; k < d->n_choices; ++k) pop[k] = d->choice_pop[k];
  mpe::ResetBoxes boxes;
  std::memset(&boxes, 0, sizeof(boxes));
  for (int e = 0; e < d->n_agents + d->n_landmarks; ++e)
    for (int k = 0; k < 4; ++k) boxes.box[e][k] = p->reset_box[e][k];
  return hip_result(mpe::launch_reset_box(d->n_agents, d->n_landmarks, *b, (size_t)B, mask, boxes, seed, episode,
                                          (uint64_t)world_offset, d->n_choices, pop, static_cast<hipStream_t>(stream)), what);
}

int mpe_reset_random_actions_block(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, float landmark_range, uint64_t episode,
                                   float *act, int32_t *ids, uint64_t seed, uint64_t step0, int32_t T, int64_t world_offset,
                                   void *stream) {
  const char *what = "mpe_reset_random_actions_block";
  if (int rc = check_desc(d, what)) return rc;
  if (int rc = check_state(b, B, what)) return rc;
  if (d->n_choices > 0 && b->choice == nullptr)
    return fail(MPE_EINVAL, "%s: desc->n_choices = %d but bufs->choice is NULL", what, d->n_choices);
  if (!act && !ids) return fail(MPE_EINVAL, "%s: act and ids are both NULL", what);
  if (T < 1 || T > 65535) return fail(MPE_EINVAL, "%s: T = %d (the reset rides on the block's first step: 1..65535)", what, T);
  if (B == 0) return 0;
  int32_t pop[MPE_MAX_CHOICES] = {1, 1, 1, 1};
  for (int k = 0; k < d->n_choices; ++k) pop[k] = d->choice_pop[k];
  return hip_result(mpe::launch_reset_random_actions(d->n_agents, d->n_landmarks, *b, (size_t)B, landmark_range, episode,
                                                     d->n_choices, pop, act, ids, seed, step0, T, (uint64_t)world_offset,
                                                     static_cast<hipStream_t>(stream)), what);
}

int mpe_random_actions_block(float *act, int32_t *ids, int32_t n_agents, int64_t B, uint64_t seed, uint64_t step0,
                             int32_t T, int64_t world_offset, void *stream) {
  const char *what = "mpe_random_actions_block";
  if (!act && !ids) return fail(MPE_EINVAL, "%s: act and ids are both NULL", what);
  if (n_agents < 1 || B < 0 || T < 0 || T > 65535) return fail(MPE_EINVAL, "%s: bad n_agents/B/T", what);
  if (B == 0 || T == 0) return 0;
  return hip_result(mpe::launch_random_actions(act, ids, n_agents, (size_t)B, seed, step0, T, (uint64_t)world_offset,
                                               static_cast<hipStream_t>(stream)), what);
}

int mpe_random_actions(float *act, int32_t *ids, int32_t n_agents, int64_t B, uint64_t seed, uint64_t step,
                       int64_t world_offset, void *stream) {
  return mpe_random_actions_block(act, ids, n_agents, B, seed, step, 1, world_offset, stream);
}

int mpe_random_comm(float *comm, int32_t n_agents, int64_t B, int32_t dim_c, uint32_t speakers, uint64_t seed,
                    uint64_t step0, int32_t T, int64_t world_offset, void *stream) {
  const char *what = "mpe_random_comm";
  if (!comm) return fail(MPE_EINVAL, "%s: comm is NULL", what);
  if (n_agents < 1 || n_agents > 32 || B < 0 || dim_c < 1 || T < 0 || T > 65535)
    return fail(MPE_EINVAL, "%s: bad n_agents/B/dim_c/T", what);
  if (B == 0 || speakers == 0 || T == 0) return 0;
  return hip_result(mpe::launch_random_comm(comm, n_agents, (size_t)B, dim_c, speakers, seed, step0, T,
                                            (uint64_t)world_offset, static_cast<hipStream_t>(stream)), what);
}

int mpe_episode_tick(int32_t *episode_step, uint8_t *done, int32_t n_agents, int64_t B, int32_t max_episode_steps,
                     int32_t clear_finished, void *stream) {
  const char *what = "mpe_episode_tick";
  if (!episode_step || !done) return fail(MPE_EINVAL, "%s: episode_step and done must be device pointers", what);
  if (n_agents < 1 || B < 0 || max_episode_steps < 0) return fail(MPE_EINVAL, "%s: bad n_agents/B/max_episode_steps", what);
  if (B == 0) return 0;
  return hip_result(mpe::launch_episode_tick(episode_step, done, n_agents, (size_t)B, max_episode_steps,
                                             clear_finished, static_cast<hipStream_t>(stream)), what);
}

static int rollout_fused(const char *what, const float *act_seq, const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, int32_t T,
                         int32_t episode_len, float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                         int32_t trajectory, void *stream);
int mpe_rollout_random(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, int32_t T, int32_t episode_len,
                       float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                       int32_t trajectory, void *stream) {
  return rollout_fused("mpe_rollout_random", nullptr, d, b, B, T, episode_len, landmark_range, seed, step0, world_offset, trajectory, stream);
}
int mpe_rollout_actions(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, int32_t T, int32_t episode_len,
                        float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                        int32_t trajectory, const float *act_seq, void *stream) {
  if (!act_seq) return fail(MPE_EINVAL, "mpe_rollout_actions: act_seq is NULL");
  return rollout_fused("mpe_rollout_actions", act_seq, d, b, B, T, episode_len, landmark_range, seed, step0, world_offset, trajectory, stream);
}
static int rollout_fused(const char *what, const float *act_seq, const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, int32_t T,
                         int32_t episode_len, float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                         int32_t trajectory, void *stream) {
  if (int rc = check_desc(d, what)) return rc;
  if (int rc = check_state(b, B, what)) return rc;
  if (int rc = need(b->obs, what, "obs")) return rc;
  if (int rc = check_info(d, b, what)) return rc;
  if (d->n_choices > 0 && d->kind >= MPE_SCN_ADVERSARY)
    if (int rc = need(b->choice, what, "choice (the per-world picks of reset_world)")) return rc;
  if (d->kind >= MPE_SCN_SPEAKER_LISTENER)   // the speaking agents' words are drawn in-kernel; their last words are left in comm
    if (int rc = need(b->comm, what, "comm (receives the agents' comm state after the last step)")) return rc;
  if (T < 0 || episode_len < 0) return fail(MPE_EINVAL, "%s: T, episode_len must be >= 0", what);
  if (B == 0 || T == 0) return 0;
  mpe::RollArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  ra.T = T;
  ra.episode_len = episode_len;
  ra.trajectory = trajectory ? 1 : 0;
  ra.landmark_range = landmark_range;
  ra.seed = seed;
  ra.step0 = step0;
  ra.world_offset = (uint64_t)world_offset;
  ra.act_seq = act_seq;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->n_agents + d->n_landmarks > mpe::kNarrowMaxE ||
      !mpe::split_supports(d->kind, d->n_agents, d->n_landmarks, d->n_adversaries)) {
    if (d->kind != MPE_SCN_SPREAD && d->kind != MPE_SCN_TAG)
      return fail(MPE_EUNSUPPORTED, "%s: the fused rollout exists for the wave-per-agent shapes and for simple_spread / simple_tag of any size", what);
    if (int rc = need(b->entity_table, what, "entity_table (required by the wave-per-world kernel)")) return rc;
    const mpe::WideDesc w = make_wide(d);
    return hip_result(mpe::launch_wide(true, true, w, *b, (size_t)B, s, &ra), what);
  }
  if (act_seq)      // (the wave-per-agent shapes take the caller's moves through the step server: its launch with every command ahead)
    return fail(MPE_EUNSUPPORTED, "%s: this shape has a wave-per-agent kernel -- its T-step launch with the caller's moves is the step "
                "server's (mpe_step_server_ring, then mpe_step_server_start with srv->ahead = 1)", what);
  const mpe::NarrowDesc n = make_narrow(d, b, (size_t)B);
  return hip_result(mpe::launch_split(true, d->kind, d->n_agents, d->n_landmarks, d->n_adversaries, n, *b, (size_t)B,
                                      ra, s), what);
}

// ---- the step server (mpe_split.hip, SERVE) ------------------------------------------------------------------------------
static int check_server(const MpeStepServer *srv, const char *what) {
  if (!srv) return fail(MPE_EINVAL, "%s: srv is NULL", what);
  if (!srv->door || !srv->flag || !srv->status) return fail(MPE_EINVAL, "%s: srv->door / flag / status must be device words", what);
  return 0;
}
int mpe_step_server_supported(const MpeScenarioDesc *d, int64_t B) {
  if (int rc = check_desc(d, "mpe_step_server_supported")) return rc;
  if (B <= 0 || d->n_agents + d->n_landmarks > mpe::kNarrowMaxE) return 0;
  return mpe::serve_supports(d->kind, d->n_agents, d->n_landmarks, d->n_adversaries) ? 1 : 0;
}
int64_t mpe_step_server_flags(int64_t B) { return B > 0 ? (int64_t)mpe::serve_grid((size_t)B) : 0; }
int mpe_step_server_start(const MpeScenarioDesc *d, const MpeBuffers *b, int64_t B, int32_t T, int32_t episode_len,
                          float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset, const MpeStepServer *srv,
                          void *stream) {
  const char *what = "mpe_step_server_start";
  if (int rc = check_desc(d, what)) return rc;
  if (int rc = check_state(b, B, what)) return rc;
  if (int rc = need(b->obs, what, "obs")) return rc;
  if (int rc = check_info(d, b, what)) return rc;
  if (int rc = check_server(srv, what)) return rc;
  if (d->n_choices > 0 && d->kind >= MPE_SCN_ADVERSARY)
    if (int rc = need(b->choice, what, "choice (the per-world picks of reset_world)")) return rc;
  if (!srv->act_ring || srv->ring < 1 || srv->slots < 1) return fail(MPE_EINVAL, "%s: srv->act_ring, ring >= 1, slots >= 1", what);
  if (T < 0 || episode_len < 0) return fail(MPE_EINVAL, "%s: T, episode_len must be >= 0", what);
  if (B == 0 || T == 0) return 0;
  if (d->n_agents + d->n_landmarks > mpe::kNarrowMaxE || !mpe::serve_supports(d->kind, d->n_agents, d->n_landmarks, d->n_adversaries))
    return fail(MPE_EUNSUPPORTED, "%s: the step server exists for the shapes with a wave-per-agent kernel", what);
  mpe::RollArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  ra.T = T;
  ra.episode_len = episode_len;
  ra.trajectory = 1;
  ra.landmark_range = landmark_range;
  ra.seed = seed;
  ra.step0 = step0;
  ra.world_offset = (uint64_t)world_offset;
  mpe::ServeHandles h;
  h.door = srv->door;
  h.flag = srv->flag;
  h.status = srv->status;
  h.act_ring = srv->act_ring;
  h.comm_ring = srv->comm_ring;
  if (d->kind >= MPE_SCN_SPEAKER_LISTENER && !srv->comm_ring)
    return fail(MPE_EINVAL, "%s: srv->comm_ring (the utterance tensors of the communication scenarios) is NULL", what);
  h.ring = srv->ring;
  h.slots = srv->slots;
  h.timeout_ticks = srv->timeout_us * 100ull;      // the 100 MHz wall clock (s_memrealtime)
  h.ahead = srv->ahead != 0;
  const mpe::NarrowDesc n = make_narrow(d, b, (size_t)B);
  const int rc = mpe::launch_split_serve(d->kind, d->n_agents, d->n_landmarks, d->n_adversaries, n, *b, (size_t)B, ra, h,
                                         static_cast<hipStream_t>(stream));
  if (rc == mpe::MPE_ESERVER_TOO_LARGE)
    return fail(MPE_EUNSUPPORTED, "%s: %lld worlds = %u workgroups cannot all be resident on this GPU (a waiting workgroup holds "
                "its slots): serve a smaller batch per server", what, (long long)B, mpe::serve_grid((size_t)B));
  return hip_result(rc, what);
}
int mpe_step_server_ring(const MpeStepServer *srv, uint64_t n_steps, void *stream) {
  if (int rc = check_server(srv, "mpe_step_server_ring")) return rc;
  return hip_result(mpe::launch_serve_ring(srv->door, n_steps, static_cast<hipStream_t>(stream)), "mpe_step_server_ring");
}
int mpe_step_server_wait(const MpeStepServer *srv, int64_t B, uint64_t steps_completed, void *stream) {
  if (int rc = check_server(srv, "mpe_step_server_wait")) return rc;
  if (B <= 0) return 0;
  return hip_result(mpe::launch_serve_wait(srv->flag, mpe::serve_grid((size_t)B), steps_completed, srv->status,
                                           srv->timeout_us * 100ull, static_cast<hipStream_t>(stream)), "mpe_step_server_wait");
}

// ---- the composable output stage (mpe_rows.hip) -------------------------------------------------------------------------
static int rows_header(const char *what, const MpeScenarioDesc *d, const MpeRowProgram *p, mpe::RowDims *h, mpe::RowTables *t) {
  if (int rc = check_desc(d, what)) return rc;
  if (!p) return fail(MPE_EINVAL, "%s: prog is NULL", what);
  const int A = d->n_agents, E = d->n_agents + d->n_landmarks;
  if (E > MPE_ROWS_MAX_ENTITIES) return fail(MPE_EUNSUPPORTED, "%s: row programs cover A + L <= %d (got %d)", what, MPE_ROWS_MAX_ENTITIES, E);
  if (p->n_ops < 0 || (p->n_ops > 0 && !p->ops_device)) return fail(MPE_EINVAL, "%s: prog->ops_device is NULL", what);
  if (p->n_vel < 0 || p->n_vel > E) return fail(MPE_EINVAL, "%s: prog->n_vel = %d, need 0 .. A + L", what, p->n_vel);
  if (p->n_regions < 0 || p->n_regions > 2) return fail(MPE_EINVAL, "%s: prog->n_regions = %d, need 0 .. 2", what, p->n_regions);
  for (int r = 0; r < p->n_regions; ++r)
    if (p->region_entity[r] < 0 || p->region_entity[r] >= E) return fail(MPE_EINVAL, "%s: region %d names entity %d", what, r, p->region_entity[r]);
  int prev = 0;
  for (int i = 0; i <= A; ++i) {
    if (p->obs_begin[i] < prev || p->obs_begin[i] > p->n_ops) return fail(MPE_EINVAL, "%s: obs_begin[%d] = %d out of order / range", what, i, p->obs_begin[i]);
    prev = p->obs_begin[i];
  }
  prev = 0;
  for (int i = 0; i <= A; ++i) {
    if (p->rew_begin[i] < prev || p->rew_begin[i] > p->n_ops) return fail(MPE_EINVAL, "%s: rew_begin[%d] = %d out of order / range", what, i, p->rew_begin[i]);
    prev = p->rew_begin[i];
  }
  bool has_done = false;
  for (int i = 0; i <= A; ++i) has_done = has_done || p->done_begin[i] != 0;
  if (has_done) {
    prev = 0;
    for (int i = 0; i <= A; ++i) {
      if (p->done_begin[i] < prev || p->done_begin[i] > p->n_ops) return fail(MPE_EINVAL, "%s: done_begin[%d] = %d out of order / range", what, i, p->done_begin[i]);
      prev = p->done_begin[i];
    }
  }
  if (p->n_regions > 0 && A > 32) return fail(MPE_EUNSUPPORTED, "%s: regions hide agents from each other for A <= 32 (got %d)", what, A);
  std::memset(h, 0, sizeof(*h));
  std::memset(t, 0, sizeof(*t));
  h->n_agents = A;
  h->n_entities = E;
  h->n_vel = p->n_vel;
  h->dim_c = d->dim_c;
  h->collaborative = d->collaborative;
  h->n_picks = d->n_choices;
  h->n_ops = p->n_ops;
  h->d_max = 1;
  for (int i = 0; i < A; ++i) {
    const int D = d->obs_off[i + 1] - d->obs_off[i];
    if (D < 0) return fail(MPE_EINVAL, "%s: desc->obs_off is not a prefix sum", what);
    if (D > h->d_max) h->d_max = D;
  }
  h->n_regions = p->n_regions;
  h->region_entity[0] = p->n_regions > 0 ? p->region_entity[0] : 0;
  h->region_entity[1] = p->n_regions > 1 ? p->region_entity[1] : 0;
  h->all_seeing = p->all_seeing;
  // (the physics constants are part of the tables whether or not a call steps the world: mpe_step_rows, mpe_rows and
  //  mpe_episode_finish of one program share ONE table content, and alternating between them uploads nothing)
  for (int e = 0; e < E; ++e) {
    t->size[e] = d->size[e];
    t->inv_mass[e] = d->mass[e] > 0.f ? 1.0f / d->mass[e] : 1.0f;
    t->accel[e] = d->accel[e];
    t->max_speed[e] = d->max_speed[e];
    if (d->movable[e]) h->movable |= 1ull << e;
    if (d->collide[e]) h->collide |= 1ull << e;
  }
  for (int i = 0; i <= MPE_ROWS_MAX_ENTITIES; ++i) {
    t->obs_off[i] = i <= A ? d->obs_off[i] : d->obs_off[A];
    t->obs_begin[i] = i <= A ? p->obs_begin[i] : p->obs_begin[A];
    t->rew_begin[i] = i <= A ? p->rew_begin[i] : p->rew_begin[A];
    t->done_begin[i] = i <= A ? p->done_begin[i] : p->done_begin[A];
  }
  if (p->n_shared < 0 || p->n_shared > 64 || (p->n_shared > 0 && !p->traced))
    return fail(MPE_EINVAL, "%s: prog->n_shared = %d (0..64, traced programs only)", what, p->n_shared);
  h->n_shared = p->n_shared;
  h->reset_boxes = p->reset_boxes ? 1 : 0;
  if (p->reset_boxes)
    for (int e = 0; e < E; ++e)
      for (int k = 0; k < 4; ++k) t->reset_box[e][k] = p->reset_box[e][k];
  h->dt = d->dt;
  h->damp = 1.0f - d->damping;
  h->cforce = d->contact_force;
  h->cmargin = d->contact_margin;
  h->cmargin_inv = 1.0f / d->contact_margin;
  return 0;
}

int mpe_rows_validate(const MpeScenarioDesc *d, const MpeRowProgram *p, const int32_t *ops) {
  const char *what = "mpe_rows_validate";
  mpe::RowDims h;
  mpe::RowTables tabs;
  if (int rc = rows_header(what, d, p, &h, &tabs)) return rc;
  if (p->n_ops > 0 && !ops) return fail(MPE_EINVAL, "%s: ops_host is NULL", what);
  const int A = d->n_agents, E = A + d->n_landmarks;
  auto ent = [&](int a, bool self_ok) { return (self_ok && a == MPE_ROW_SELF) || (a >= 0 && a < E); };
  bool code_ops = false;
  for (int i = 0; i < A; ++i) {
    int width = 0;
    for (int pc = p->obs_begin[i]; pc < p->obs_begin[i + 1]; ++pc) {
      const int32_t w0 = ops[4 * pc], w1 = ops[4 * pc + 1];
      const int code = w0 & 0xff, a0 = (w0 >> 8) & 0xff, a1 = (w0 >> 16) & 0xff;
      switch (code) {
        case MPE_ROW_OBS_VEL: case MPE_ROW_OBS_POS: case MPE_ROW_OBS_REL: case MPE_ROW_OBS_REL_VIS: case MPE_ROW_OBS_VEL_VIS:
          if (!ent(a0, true)) return fail(MPE_EINVAL, "%s: op %d (agent %d): entity %d out of range", what, pc, i, a0);
          if ((code == MPE_ROW_OBS_REL_VIS || code == MPE_ROW_OBS_VEL_VIS) && (a0 == MPE_ROW_SELF ? i : a0) >= A)
            return fail(MPE_EINVAL, "%s: op %d: visibility is defined between agents", what, pc);
          width += 2;
          break;
        case MPE_ROW_OBS_REL_PICK:
          if (a1 >= d->n_choices || w1 < 0 || w1 + d->choice_pop[a1 < MPE_MAX_CHOICES ? a1 : 0] > E)
            return fail(MPE_EINVAL, "%s: op %d (agent %d): pick %d with base %d leaves the entity list", what, pc, i, a1, w1);
          width += 2;
          break;
        case MPE_ROW_OBS_COMM:
          if (!ent(a0, true) || (a0 == MPE_ROW_SELF ? i : a0) >= A || a1 > d->dim_c)
            return fail(MPE_EINVAL, "%s: op %d (agent %d): utterance of agent %d, %d floats (dim_c = %d)", what, pc, i, a0, a1, d->dim_c);
          width += a1;
          break;
        case MPE_ROW_OBS_CONST: width += 1; break;
        case MPE_ROW_OBS_CONST_N: width += a1; break;
        case MPE_ROW_OBS_REL_RANGE: case MPE_ROW_OBS_VEL_RANGE: case MPE_ROW_OBS_REL_VIS_RANGE: case MPE_ROW_OBS_VEL_VIS_RANGE: {
          const bool vis = code == MPE_ROW_OBS_REL_VIS_RANGE || code == MPE_ROW_OBS_VEL_VIS_RANGE;
          if (a1 < 1 || a0 + a1 > (vis ? A : E))
            return fail(MPE_EINVAL, "%s: op %d (agent %d): entities %d .. %d leave the %s list", what, pc, i, a0, a0 + a1 - 1, vis ? "agent" : "entity");
          const bool skip = ((w0 >> 24) & 1) && i >= a0 && i < a0 + a1;
          width += 2 * (a1 - (skip ? 1 : 0));
          break;
        }
        case MPE_ROW_OBS_ONEHOT:
          if (a0 >= d->n_choices) return fail(MPE_EINVAL, "%s: op %d (agent %d): pick %d of %d", what, pc, i, a0, d->n_choices);
          width += a1;
          break;
        case MPE_ROW_OBS_IN_REGION:
          if (!ent(a0, true) || a1 >= p->n_regions) return fail(MPE_EINVAL, "%s: op %d (agent %d): region %d of %d", what, pc, i, a1, p->n_regions);
          width += 1;
          break;
        case MPE_ROW_OBS_CODE:
          if (w1 < 0 || w1 > 4096) return fail(MPE_EINVAL, "%s: op %d (agent %d): traced code writing %d columns", what, pc, i, w1);
          width += w1;
          code_ops = true;
          break;
        default: return fail(MPE_EINVAL, "%s: op %d (agent %d): code %d is not an observation op", what, pc, i, code);
      }
    }
    if (width != d->obs_off[i + 1] - d->obs_off[i])
      return fail(MPE_EINVAL, "%s: agent %d's program emits %d columns, desc->obs_off says %d", what, i, width, d->obs_off[i + 1] - d->obs_off[i]);
  }
  for (int pass = 0; pass < 2; ++pass)      // 0: the reward programs, 1: the done programs (value ops + DONE_IF_* tests, no STORE)
  for (int i = 0; i < A; ++i) {
    bool stored = false;
    const int32_t *begin = pass ? p->done_begin : p->rew_begin;
    for (int pc = begin[i]; pc < begin[i + 1]; ++pc) {
      const int32_t w0 = ops[4 * pc], w1 = ops[4 * pc + 1];
      const int code = w0 & 0xff, a0 = (w0 >> 8) & 0xff, a1 = (w0 >> 16) & 0xff;
      if (pass && (code == MPE_ROW_R_STORE || code == MPE_ROW_R_ADD_IF_HIT || code == MPE_ROW_R_ADD_IF_HIT_GRID || code == MPE_ROW_R_ADD_MIN_DIST_GRID))
        return fail(MPE_EINVAL, "%s: done op %d (agent %d): code %d belongs to reward programs", what, pc, i, code);
      if (!pass && (code == MPE_ROW_R_DONE_IF_GT || code == MPE_ROW_R_DONE_IF_LT || code == MPE_ROW_R_DONE_IF_HIT))
        return fail(MPE_EINVAL, "%s: reward op %d (agent %d): a DONE_IF test belongs to the done program", what, pc, i);
      switch (code) {
        case MPE_ROW_R_ABS_POS:
          if (!ent(a0, false) || a1 > 1) return fail(MPE_EINVAL, "%s: op %d: coordinate %d of entity %d", what, pc, a1, a0);
          break;
        case MPE_ROW_R_DONE_IF_HIT:
          if (!ent(a0, false) || !ent(a1, false)) return fail(MPE_EINVAL, "%s: done op %d: entities %d, %d out of range", what, pc, a0, a1);
          break;
        case MPE_ROW_R_DONE_IF_GT: case MPE_ROW_R_DONE_IF_LT: break;
        case MPE_ROW_R_D2: case MPE_ROW_R_MIN_D2: case MPE_ROW_R_ADD_IF_HIT:
          if (!ent(a0, false) || !ent(a1, false)) return fail(MPE_EINVAL, "%s: reward op %d: entities %d, %d out of range", what, pc, a0, a1);
          break;
        case MPE_ROW_R_D2_PICK: case MPE_ROW_R_MIN_D2_PICK:
          if (!ent(a0, false) || a1 >= d->n_choices || w1 < 0 || w1 + d->choice_pop[a1 < MPE_MAX_CHOICES ? a1 : 0] > E)
            return fail(MPE_EINVAL, "%s: reward op %d: pick %d with base %d leaves the entity list", what, pc, a1, w1);
          break;
        case MPE_ROW_R_BOUND:
          if (!ent(a0, false) || a1 > 1) return fail(MPE_EINVAL, "%s: reward op %d: coordinate %d of entity %d", what, pc, a1, a0);
          break;
        case MPE_ROW_R_COMM_ERR:
          if (a0 >= A || a1 >= d->n_choices) return fail(MPE_EINVAL, "%s: reward op %d: utterance of agent %d against pick %d", what, pc, a0, a1);
          break;
        case MPE_ROW_R_COMM_SUM:
          if (a0 >= A) return fail(MPE_EINVAL, "%s: reward op %d: utterance of agent %d", what, pc, a0);
          break;
        case MPE_ROW_R_SAVE: case MPE_ROW_R_LOAD:
          if (a0 >= mpe::kRowSlots) return fail(MPE_EINVAL, "%s: reward op %d: slot %d of %d", what, pc, a0, mpe::kRowSlots);
          break;
        case MPE_ROW_R_STORE:
          if (a0 != i || stored) return fail(MPE_EINVAL, "%s: reward op %d: agent %d's program stores agent %d (or twice)", what, pc, i, a0);
          stored = true;
          break;
        case MPE_ROW_R_MIN_D2_RANGE:
          if (w1 < 1 || a0 + w1 > E || !ent(a1, false)) return fail(MPE_EINVAL, "%s: reward op %d: entities %d .. %d against %d", what, pc, a0, a0 + w1 - 1, a1);
          break;
        case MPE_ROW_R_MIN_D2_TO_RANGE:
          if (w1 < 1 || a1 + w1 > E || !ent(a0, false)) return fail(MPE_EINVAL, "%s: reward op %d: entity %d against %d .. %d", what, pc, a0, a1, a1 + w1 - 1);
          break;
        case MPE_ROW_R_ADD_IF_HIT_GRID: case MPE_ROW_R_ADD_MIN_DIST_GRID: {
          const int na = w1 & 255, nb = (w1 >> 8) & 255;
          if (na < 1 || nb < 1 || a0 + na > E || a1 + nb > E) return fail(MPE_EINVAL, "%s: reward op %d: contact grid %d+%d x %d+%d leaves the entity list", what, pc, a0, na, a1, nb);
          break;
        }
        case MPE_ROW_R_SQRT: case MPE_ROW_R_CONST: case MPE_ROW_R_ZERO: case MPE_ROW_R_ADD: case MPE_ROW_R_ADD_ACC: break;
        case MPE_ROW_R_CODE:
          if (pass) return fail(MPE_EINVAL, "%s: done op %d (agent %d): R_CODE belongs to reward programs", what, pc, i);
          code_ops = true;
          break;
        case MPE_ROW_R_DONE_CODE:
          if (!pass) return fail(MPE_EINVAL, "%s: reward op %d (agent %d): R_DONE_CODE belongs to the done program", what, pc, i);
          code_ops = true;
          break;
        default: return fail(MPE_EINVAL, "%s: op %d: code %d is not a reward op", what, pc, code);
      }
    }
    if (!pass && p->rew_begin[i + 1] > p->rew_begin[i] && !stored)
      return fail(MPE_EINVAL, "%s: agent %d's reward program has no STORE", what, i);
  }
  if (code_ops != (p->traced != 0))
    return fail(MPE_EINVAL, "%s: prog->traced = %d but the program %s *_CODE ops (a traced program runs compiled in only and says so)",
                what, p->traced, code_ops ? "contains" : "has no");
  return 0;
}

static uint64_t fnv1a(uint64_t hash, const void *data, size_t n) {
  const unsigned char *bytes = static_cast<const unsigned char *>(data);
  for (size_t k = 0; k < n; ++k) hash = (hash ^ bytes[k]) * 1099511628211ull;
  return hash;
}
static constexpr uint64_t kFnvSeed = 1469598103934665603ull;
static uint64_t tables_hash(const mpe::RowTables &tabs) { return fnv1a(kFnvSeed, &tabs, sizeof(tabs)) | 1ull; }

// a program compiled in (mpe_rows_load_image): the module, its five entry points, and what it was compiled for
struct RowImage {
  hipModule_t module;
  void *fns[5];            // _s, _r, _e, _l, _m
  mpe::RowDims dims;
  uint64_t tables;
  const int32_t *ops_device;
  int32_t n_ops;
};
static bool image_matches(const MpeRowProgram *p, const mpe::RowDims &h, uint64_t tabs_hash) {
  const RowImage *im = static_cast<const RowImage *>(p->image);
  return im && im->tables == tabs_hash && std::memcmp(&im->dims, &h, sizeof(h)) == 0 && im->ops_device == p->ops_device &&
         im->n_ops == p->n_ops;
}
// geometry + name of the compiled form
static int static_identity(const char *what, const mpe::RowDims &h, const mpe::RowTables &tabs, const int32_t *ops, int waves[2],
                           size_t lds[2], char name[40]) {
  for (int phys = 0; phys < 2; ++phys)
    if (int rc = mpe::rows_geometry(h, phys != 0, &waves[phys], &lds[phys], 0))
      return fail(rc, "%s: the program does not fit a workgroup's LDS", what);
  if (h.n_ops > MPE_ROWS_STATIC_MAX_OPS)
    return fail(MPE_EUNSUPPORTED, "%s: %d ops; programs of more than %d ops stay interpreted (every op becomes code: compile time and "
                "instruction-cache footprint grow with it)", what, h.n_ops, MPE_ROWS_STATIC_MAX_OPS);
  uint64_t hash = fnv1a(kFnvSeed, &h, sizeof(h));
  hash = fnv1a(hash, &tabs, sizeof(tabs));
  hash = fnv1a(hash, ops, (size_t)h.n_ops * 16);
  hash = fnv1a(hash, waves, 2 * sizeof(int));
  std::snprintf(name, 40, "mpe_rows_%016llx", (unsigned long long)hash);
  return 0;
}

static int rows_call(const char *what, bool phys, const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B,
                     const mpe::RowEpisode *episode, void *stream, const mpe::RollArgs *roll = nullptr) {
  static_assert(sizeof(mpe::RowTables) <= MPE_ROWS_HEADER_BYTES, "MPE_ROWS_HEADER_BYTES too small");
  mpe::RowDims h;
  mpe::RowTables tabs;
  if (int rc = rows_header(what, d, p, &h, &tabs)) return rc;
  if (!p->header_device) return fail(MPE_EINVAL, "%s: prog->header_device is NULL (MPE_ROWS_HEADER_BYTES of device memory)", what);
  if (int rc = check_state(b, B, what)) return rc;
  if (int rc = need(b->obs, what, "obs")) return rc;
  if (d->n_choices > 0) if (int rc = need(b->choice, what, "choice (the per-world picks of reset_world)")) return rc;
  // (bufs->comm may be NULL: every utterance then reads as zero -- the state of agents that never speak)
  if (phys) {
    if (!roll)
      if (int rc = check_actions(b, what)) return rc;
    for (int e = d->n_agents; e < d->n_agents + d->n_landmarks; ++e)
      if (d->movable[e]) return fail(MPE_EUNSUPPORTED, "%s: a movable landmark (entity %d) is stepped by mpe_world_step only", what, e);
  }
  mpe::RowEpisode ep;
  std::memset(&ep, 0, sizeof(ep));
  if (episode) ep = *episode;
  if (B == 0) return 0;
  bool vec4 = reinterpret_cast<uintptr_t>(b->obs) % 16 == 0;
  for (int i = 0; i <= d->n_agents; ++i)
    if (((size_t)d->obs_off[i] * (size_t)B) % 4 != 0) vec4 = false;
  // the tables in device memory: uploaded when their content differs from what the program holds (FNV-1a over the bytes)
  const uint64_t hash = tables_hash(tabs);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (image_matches(p, h, hash))      // the program compiled in, and still the descriptor it was compiled for
    return hip_result(mpe::launch_rows_image(static_cast<RowImage *>(p->image)->fns, *b, h, tabs, phys, vec4 ? 1 : 0, ep, (size_t)B, s, roll), what);
  if (p->traced)      // *_CODE ops exist as code of the image only: nothing to interpret
    return fail(MPE_EUNSUPPORTED, "%s: a traced program runs compiled in only -- no image is attached, or the descriptor is not the "
                "one it was compiled for (mpe_rows_static_source + the traced source -> hipcc --genco -> mpe_rows_load_image)", what);
  if (hash != p->header_hash) {
    if (int rc = mpe::launch_rows_header(tabs, p->header_device, s)) return hip_result(rc, what);
    p->header_hash = hash;
  }
  return hip_result(mpe::launch_rows(*b, h, tabs, p->header_device, phys, vec4 ? 1 : 0, ep, p->ops_device, (size_t)B, s, roll), what);
}

int mpe_rows(const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B, void *stream) {
  return rows_call("mpe_rows", false, d, b, p, B, nullptr, stream);
}

int mpe_step_rows(const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B, void *stream) {
  return rows_call("mpe_step_rows", true, d, b, p, B, nullptr, stream);
}

static int episode_args(const char *what, int mode, const MpeScenarioDesc *d, const MpeBuffers *b, int32_t *episode_step,
                        int32_t max_episode_steps, float landmark_range, uint64_t seed, uint64_t episode, int64_t world_offset,
                        mpe::RowEpisode *ep) {
  if (!episode_step) return fail(MPE_EINVAL, "%s: episode_step is NULL", what);
  if (!b || !b->done) return fail(MPE_EINVAL, "%s: bufs->done (the agents' done rows) is NULL", what);
  if (max_episode_steps < 0) return fail(MPE_EINVAL, "%s: max_episode_steps < 0", what);
  if (!d) return fail(MPE_EINVAL, "%s: desc is NULL", what);
  std::memset(ep, 0, sizeof(*ep));
  ep->enabled = mode;
  ep->max_steps = max_episode_steps;
  ep->episode_step = episode_step;
  ep->landmark_range = landmark_range;
  ep->n_choices = d->n_choices;
  for (int k = 0; k < MPE_MAX_CHOICES; ++k) ep->choice_pop[k] = k < d->n_choices ? d->choice_pop[k] : 1;
  ep->seed = seed;
  ep->episode = episode;
  ep->world_offset = (uint64_t)world_offset;
  return 0;
}

static int rollout_rows(const char *what, const float *act_seq, const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B,
                        int32_t T, int32_t episode_len, float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                        int32_t trajectory, uint32_t speakers, void *stream);
int mpe_rollout_rows(const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B, int32_t T, int32_t episode_len,
                     float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset, int32_t trajectory, uint32_t speakers,
                     void *stream) {
  return rollout_rows("mpe_rollout_rows", nullptr, d, b, p, B, T, episode_len, landmark_range, seed, step0, world_offset, trajectory,
                      speakers, stream);
}
int mpe_rollout_rows_actions(const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B, int32_t T, int32_t episode_len,
                             float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset, int32_t trajectory,
                             const float *act_seq, void *stream) {
  if (!act_seq) return fail(MPE_EINVAL, "mpe_rollout_rows_actions: act_seq is NULL");
  return rollout_rows("mpe_rollout_rows_actions", act_seq, d, b, p, B, T, episode_len, landmark_range, seed, step0, world_offset,
                      trajectory, 0u, stream);
}
static int rollout_rows(const char *what, const float *act_seq, const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B,
                        int32_t T, int32_t episode_len, float landmark_range, uint64_t seed, uint64_t step0, int64_t world_offset,
                        int32_t trajectory, uint32_t speakers, void *stream) {
  if (T < 0 || episode_len < 0) return fail(MPE_EINVAL, "%s: T, episode_len must be >= 0", what);
  if (!d) return fail(MPE_EINVAL, "%s: desc is NULL", what);
  if (speakers != 0) {
    if (d->dim_c <= 0) return fail(MPE_EINVAL, "%s: speakers 0x%x but desc->dim_c = %d", what, speakers, d->dim_c);
    if (d->n_agents < 32 && (speakers >> d->n_agents) != 0) return fail(MPE_EINVAL, "%s: speakers 0x%x names agents beyond %d", what, speakers, d->n_agents);
    if (int rc = need(b ? b->comm : nullptr, what, "comm (receives the speaking agents' last words)")) return rc;
  }
  if (T == 0) return 0;
  mpe::RollArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  ra.T = T;
  ra.episode_len = episode_len;
  ra.trajectory = trajectory ? 1 : 0;
  ra.landmark_range = landmark_range;
  ra.seed = seed;
  ra.step0 = step0;
  ra.world_offset = (uint64_t)world_offset;
  ra.act_seq = act_seq;
  mpe::RowEpisode ep;      // (off; carries the picks' populations and the world numbering the in-kernel resets draw with)
  std::memset(&ep, 0, sizeof(ep));
  ep.n_choices = d->n_choices;
  for (int k = 0; k < MPE_MAX_CHOICES; ++k) ep.choice_pop[k] = k < d->n_choices ? d->choice_pop[k] : 1;
  ep.world_offset = (uint64_t)world_offset;
  ep.speakers = speakers;
  return rows_call(what, true, d, b, p, B, &ep, stream, &ra);
}

int mpe_rollout_rows_episode(const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B, int32_t T,
                             int32_t *episode_step, int32_t max_episode_steps, float landmark_range, uint64_t seed, uint64_t step0,
                             uint64_t episode0, int64_t world_offset, int32_t trajectory, uint32_t speakers, void *stream) {
  const char *what = "mpe_rollout_rows_episode";
  if (T < 0) return fail(MPE_EINVAL, "%s: T must be >= 0", what);
  mpe::RowEpisode ep;
  if (int rc = episode_args(what, 2, d, b, episode_step, max_episode_steps, landmark_range, seed, episode0, world_offset, &ep)) return rc;
  if (speakers != 0) {
    if (d->dim_c <= 0) return fail(MPE_EINVAL, "%s: speakers 0x%x but desc->dim_c = %d", what, speakers, d->dim_c);
    if (int rc = need(b->comm, what, "comm (receives the speaking agents' last words)")) return rc;
  }
  if (T == 0) return 0;
  ep.speakers = speakers;
  mpe::RollArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  ra.T = T;
  ra.episode_len = 0;      // (no clocked resets: the episodes end where the done programs / the horizon say)
  ra.trajectory = trajectory ? 1 : 0;
  ra.landmark_range = landmark_range;
  ra.seed = seed;
  ra.step0 = step0;
  ra.world_offset = (uint64_t)world_offset;
  return rows_call(what, true, d, b, p, B, &ep, stream, &ra);
}

int mpe_step_rows_episode(const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B, int32_t *episode_step,
                          int32_t max_episode_steps, float landmark_range, uint64_t seed, uint64_t episode, int64_t world_offset,
                          void *stream) {
  const char *what = "mpe_step_rows_episode";
  mpe::RowEpisode ep;
  if (int rc = episode_args(what, 2, d, b, episode_step, max_episode_steps, landmark_range, seed, episode, world_offset, &ep)) return rc;
  return rows_call(what, true, d, b, p, B, &ep, stream);
}

int mpe_episode_finish(const MpeScenarioDesc *d, const MpeBuffers *b, MpeRowProgram *p, int64_t B, int32_t *episode_step,
                       int32_t max_episode_steps, float landmark_range, uint64_t seed, uint64_t episode, int64_t world_offset,
                       void *stream) {
  const char *what = "mpe_episode_finish";
  mpe::RowEpisode ep;
  if (int rc = episode_args(what, 1, d, b, episode_step, max_episode_steps, landmark_range, seed, episode, world_offset, &ep)) return rc;
  MpeBuffers bb = *b;
  bb.rew = nullptr;          // rewards belong to the step that just ran; only rows (and the finished worlds' state) change here
  return rows_call(what, false, d, &bb, p, B, &ep, stream);
}

int mpe_rows_static_source(const MpeScenarioDesc *d, const MpeRowProgram *p, const int32_t *ops, char *buf, size_t cap, size_t *needed) {
  const char *what = "mpe_rows_static_source";
  if (int rc = mpe_rows_validate(d, p, ops)) return rc;
  mpe::RowDims h;
  mpe::RowTables tabs;
  if (int rc = rows_header(what, d, p, &h, &tabs)) return rc;
  int waves[2];
  size_t lds[2];
  char name[40];
  if (int rc = static_identity(what, h, tabs, ops, waves, lds, name)) return rc;
  std::string out;
  char line[256];
  auto put = [&](const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(line, sizeof(line), fmt, ap);
    va_end(ap);
    out += line;
  };
  auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  put("// generated by mpe_rows_static_source for csrc/mpe_rows.hip (-include this file): a row program as constants\n");
  put("#define MPE_ROWS_STATIC 1\n#define MPE_ROWS_STATIC_NAME %s\n", name);
  put("#define MPE_ROWS_STATIC_WAVES_ROWS %d\n#define MPE_ROWS_STATIC_WAVES_STEP %d\n", waves[0], waves[1]);
  put("#define MPE_ROWS_STATIC_LDS_ROWS %zu\n#define MPE_ROWS_STATIC_LDS_STEP %zu\n", lds[0], lds[1]);
  // register budget: as many workgroups per CU as the LDS admits, four at most (65 536 worlds = 1024 workgroups = one round on
  // 256 CUs): that many times W waves on four SIMDs -- the compiler is told, or it spends registers the workgroup count pays for
  int occ[2];
  for (int k = 0; k < 2; ++k) {
    int wg = (int)((160u * 1024u) / lds[k]);
    wg = wg > 4 ? 4 : (wg < 1 ? 1 : wg);
    occ[k] = (wg * waves[k] + 3) / 4;
    occ[k] = occ[k] > 8 ? 8 : (occ[k] < 1 ? 1 : occ[k]);
  }
  put("#define MPE_ROWS_STATIC_OCC_ROWS %d\n#define MPE_ROWS_STATIC_OCC_STEP %d\n", occ[0], occ[1]);
  put("#define MPE_ROWS_STATIC_DIMS { %d, %d, %d, %d, %d, %d, %d, %d, %d, {%d, %d}, 0x%xu, 0x%llxull, 0x%llxull, ", h.n_agents, h.n_entities,
      h.n_vel, h.dim_c, h.collaborative, h.d_max, h.n_picks, h.n_ops, h.n_regions, h.region_entity[0], h.region_entity[1], h.all_seeing,
      (unsigned long long)h.movable, (unsigned long long)h.collide);
  put("__builtin_bit_cast(float, 0x%08xu), __builtin_bit_cast(float, 0x%08xu), __builtin_bit_cast(float, 0x%08xu), "
      "__builtin_bit_cast(float, 0x%08xu), __builtin_bit_cast(float, 0x%08xu), %d, %d }\n", fbits(h.dt), fbits(h.damp), fbits(h.cforce),
      fbits(h.cmargin), fbits(h.cmargin_inv), h.reset_boxes, h.n_shared);
  out += "#define MPE_ROWS_STATIC_TABLES { ";
  const uint32_t *tw = reinterpret_cast<const uint32_t *>(&tabs);
  for (size_t k = 0; k < sizeof(tabs) / 4; ++k) put("0x%xu,%s", tw[k], k % 16 == 15 ? " \\\n  " : " ");
  out += "}\n#define MPE_ROWS_STATIC_OPS { ";
  for (int pc = 0; pc < h.n_ops; ++pc) put("{%d, %d, %d, %d},%s", ops[4 * pc], ops[4 * pc + 1], ops[4 * pc + 2], ops[4 * pc + 3], pc % 4 == 3 ? " \\\n  " : " ");
  if (h.n_ops == 0) out += "{0, 0, 0, 0} ";
  out += "}\n";
  if (needed) *needed = out.size() + 1;
  if (!buf || cap < out.size() + 1) {
    if (buf) return fail(MPE_EINVAL, "%s: the header needs %zu bytes, cap = %zu", what, out.size() + 1, cap);
    return 0;      // (size query)
  }
  std::memcpy(buf, out.c_str(), out.size() + 1);
  return 0;
}

int mpe_rows_unload_image(MpeRowProgram *p) {
  if (!p || !p->image) return 0;
  RowImage *im = static_cast<RowImage *>(p->image);
  if (im->module) (void)hipModuleUnload(im->module);
  delete im;
  p->image = nullptr;
  return 0;
}

int mpe_rows_load_image(const MpeScenarioDesc *d, MpeRowProgram *p, const int32_t *ops, const void *image, size_t bytes) {
  const char *what = "mpe_rows_load_image";
  if (!image || bytes == 0) return fail(MPE_EINVAL, "%s: image is empty", what);
  if (int rc = mpe_rows_validate(d, p, ops)) return rc;
  mpe::RowDims h;
  mpe::RowTables tabs;
  if (int rc = rows_header(what, d, p, &h, &tabs)) return rc;
  int waves[2];
  size_t lds[2];
  char name[40];
  if (int rc = static_identity(what, h, tabs, ops, waves, lds, name)) return rc;
  mpe_rows_unload_image(p);
  RowImage *im = new RowImage();
  std::memset(static_cast<void *>(im), 0, sizeof(*im));
  hipError_t rc = hipModuleLoadData(&im->module, image);
  if (rc != hipSuccess) {
    (void)hipGetLastError();
    delete im;
    return fail((int)rc, "%s: hipModuleLoadData: %s", what, hipGetErrorString(rc));
  }
  static const char *const suffix[5] = {"_s", "_r", "_e", "_l", "_m"};
  for (int k = 0; k < 5; ++k) {
    const std::string fn = std::string(name) + suffix[k];
    hipFunction_t f = nullptr;
    rc = hipModuleGetFunction(&f, im->module, fn.c_str());
    if (rc != hipSuccess || !f) {
      (void)hipModuleUnload(im->module);
      (void)hipGetLastError();      // (the failed lookup is reported through our return code, not left behind for the next HIP call)
      delete im;
      return fail(MPE_EINVAL, "%s: the image has no kernel %s: it was compiled for another program, descriptor or library version", what, fn.c_str());
    }
    im->fns[k] = f;
  }
  im->dims = h;
  im->tables = tables_hash(tabs);
  im->ops_device = p->ops_device;
  im->n_ops = p->n_ops;
  p->image = im;
  return 0;
}

int mpe_rows_image_active(const MpeScenarioDesc *d, const MpeRowProgram *p) {
  if (!d || !p || !p->image) return 0;
  mpe::RowDims h;
  mpe::RowTables tabs;
  if (rows_header("mpe_rows_image_active", d, p, &h, &tabs)) return 0;
  return image_matches(p, h, tables_hash(tabs)) ? 1 : 0;
}

}  // extern "C"
