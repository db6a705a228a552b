/*
 * mpe_oracle.c -- CPU restatement in plain C (fp64) of the reference's hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): used by tests/ as a second checker and by bench.py's cpu_baseline leg as the
 * "what the same algorithm does in compiled code on the host cores" figure.  The product never links it.
 *
 * Follows, in the reference's operation order:
 *   MultiAgentEnv._set_action        multiagent/environment.py:144-181  (default one-hot / soft rows)
 *   World.step                       multiagent/core.py:117-131
 *     apply_action_force             :134-140
 *     apply_environment_force        :143-155  (a < b over ALL entities; f_a = F + f_a, f_b = -F + f_b)
 *     get_collision_force            :180-196  (((C * delta) / dist) * penetration, logaddexp(0, x) * k)
 *     integrate_state                :158-169
 *   Scenario.observation / reward    simple.py:41-50, simple_spread.py:66-100, simple_tag.py:69-147
 *   shared reward                    environment.py:100-102
 * Pinned to the golden vectors recorded from the reference (tests/test_oracle_golden.py).
 *
 *   gcc -O2 -fPIC -shared -fopenmp -o _build/libmpe_oracle.so mpe_oracle.c -lm     (oracle/build_c.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_MAX_E 256
enum { ORC_SIMPLE = 1, ORC_SPREAD = 2, ORC_TAG = 3 };

typedef struct OrcSpec {
  int32_t kind, n_agents, n_landmarks, dim_c, n_adversaries, collaborative;
  double dt, damping, contact_force, contact_margin;
  double size[ORC_MAX_E], mass[ORC_MAX_E], accel[ORC_MAX_E] /* sensitivity: agent.accel or 5.0 */,
      max_speed[ORC_MAX_E] /* < 0: None */;
  uint8_t movable[ORC_MAX_E], collide[ORC_MAX_E];
} OrcSpec;

size_t orc_sizeof_spec(void) { return sizeof(OrcSpec); }

static double logaddexp0(double x) { /* np.logaddexp(0, x) */
  if (x > 0) return x + log1p(exp(-x));
  return log1p(exp(x));
}

static int obs_dim(const OrcSpec *s, int i) {
  const int A = s->n_agents, L = s->n_landmarks;
  if (s->kind == ORC_SIMPLE) return 2 + 2 * L;
  if (s->kind == ORC_SPREAD) return 4 + 2 * L + 2 * (A - 1) + s->dim_c * (A - 1);
  const int good_others = (A - s->n_adversaries) - (i >= s->n_adversaries ? 1 : 0);
  return 4 + 2 * L + 2 * (A - 1) + 2 * good_others;
}
int orc_obs_total(const OrcSpec *s) {
  int t = 0;
  for (int i = 0; i < s->n_agents; ++i) t += obs_dim(s, i);
  return t;
}

static double dist(const double *p, int a, int b) {
  const double dx = p[2 * a] - p[2 * b], dy = p[2 * a + 1] - p[2 * b + 1];
  return sqrt(dx * dx + dy * dy);
}
static double bound(double x) { /* simple_tag.py:103-108 */
  if (x < 0.9) return 0;
  if (x < 1.0) return (x - 0.9) * 10;
  const double e = exp(2 * x - 2);
  return e < 10 ? e : 10;
}

/* One world, one step.  pos [E][2], vel [A][2] in place; act [A][5]; obs: per-agent rows back to back;
 * rew [A] (after the shared-reward sum); collisions [A] = benchmark_data counts (may be NULL). */
void orc_step(const OrcSpec *s, double *pos, double *vel, const double *act, double *obs, double *rew,
              int32_t *collisions) {
  const int A = s->n_agents, L = s->n_landmarks, E = A + L;
  double f[ORC_MAX_E][2];
  for (int i = 0; i < A; ++i) { /* _set_action + apply_action_force */
    f[i][0] = (act[5 * i + 1] - act[5 * i + 2]) * s->accel[i];
    f[i][1] = (act[5 * i + 3] - act[5 * i + 4]) * s->accel[i];
  }
  for (int a = 0; a < E; ++a) /* apply_environment_force */
    for (int b = a + 1; b < E; ++b) {
      if (!s->collide[a] || !s->collide[b]) continue;
      const double dx = pos[2 * a] - pos[2 * b], dy = pos[2 * a + 1] - pos[2 * b + 1];
      const double d = sqrt(dx * dx + dy * dy);
      const double k = s->contact_margin;
      const double pen = logaddexp0(-(d - (s->size[a] + s->size[b])) / k) * k;
      const double fx = s->contact_force * dx / d * pen, fy = s->contact_force * dy / d * pen;
      if (a < A && s->movable[a]) { f[a][0] = fx + f[a][0]; f[a][1] = fy + f[a][1]; }
      if (b < A && s->movable[b]) { f[b][0] = -fx + f[b][0]; f[b][1] = -fy + f[b][1]; }
    }
  for (int i = 0; i < A; ++i) { /* integrate_state */
    if (!s->movable[i]) continue;
    double vx = vel[2 * i] * (1 - s->damping), vy = vel[2 * i + 1] * (1 - s->damping);
    vx += (f[i][0] / s->mass[i]) * s->dt;
    vy += (f[i][1] / s->mass[i]) * s->dt;
    if (s->max_speed[i] >= 0) {
      const double sp = sqrt(vx * vx + vy * vy);
      if (sp > s->max_speed[i]) { vx = vx / sp * s->max_speed[i]; vy = vy / sp * s->max_speed[i]; }
    }
    vel[2 * i] = vx; vel[2 * i + 1] = vy;
    pos[2 * i] += vx * s->dt; pos[2 * i + 1] += vy * s->dt;
  }
  /* observation / reward of every agent on the post-step state (environment.py:92-97) */
  double *o = obs;
  double r[ORC_MAX_E];
  for (int i = 0; i < A; ++i) {
    if (s->kind != ORC_SIMPLE || 1) { *o++ = vel[2 * i]; *o++ = vel[2 * i + 1]; }
    if (s->kind != ORC_SIMPLE) { *o++ = pos[2 * i]; *o++ = pos[2 * i + 1]; }
    for (int l = 0; l < L; ++l) { *o++ = pos[2 * (A + l)] - pos[2 * i]; *o++ = pos[2 * (A + l) + 1] - pos[2 * i + 1]; }
    if (s->kind != ORC_SIMPLE) {
      for (int j = 0; j < A; ++j) if (j != i) { *o++ = pos[2 * j] - pos[2 * i]; *o++ = pos[2 * j + 1] - pos[2 * i + 1]; }
      if (s->kind == ORC_SPREAD) for (int z = 0; z < s->dim_c * (A - 1); ++z) *o++ = 0;   /* silent agents' state.c */
      if (s->kind == ORC_TAG) for (int j = s->n_adversaries; j < A; ++j) if (j != i) { *o++ = vel[2 * j]; *o++ = vel[2 * j + 1]; }
    }
    int c = 0;
    if (s->kind == ORC_SIMPLE) {
      const double dx = pos[2 * i] - pos[2 * A], dy = pos[2 * i + 1] - pos[2 * A + 1];
      r[i] = -(dx * dx + dy * dy);
    } else if (s->kind == ORC_SPREAD) { /* simple_spread.py:72-82: recomputed per agent, like the reference */
      double rw = 0;
      for (int l = 0; l < L; ++l) {
        double m = INFINITY;
        for (int a = 0; a < A; ++a) { const double dd = dist(pos, a, A + l); if (dd < m) m = dd; }
        rw -= m;
      }
      if (s->collide[i])
        for (int a = 0; a < A; ++a) if (dist(pos, a, i) < s->size[a] + s->size[i]) { rw -= 1; ++c; }   /* includes a == i (Q1) */
      r[i] = rw;
    } else { /* simple_tag.py:84-129 */
      double rw = 0;
      if (i < s->n_adversaries) {
        if (s->collide[i])
          for (int g = s->n_adversaries; g < A; ++g)
            for (int v = 0; v < s->n_adversaries; ++v) if (dist(pos, g, v) < s->size[g] + s->size[v]) rw += 10;
        for (int g = s->n_adversaries; g < A; ++g) if (dist(pos, g, i) < s->size[g] + s->size[i]) ++c;
      } else {
        if (s->collide[i])
          for (int v = 0; v < s->n_adversaries; ++v) if (dist(pos, v, i) < s->size[v] + s->size[i]) rw -= 10;
        rw -= bound(fabs(pos[2 * i]));
        rw -= bound(fabs(pos[2 * i + 1]));
      }
      r[i] = rw;
    }
    if (collisions) collisions[i] = c;
  }
  if (s->collaborative) { /* environment.py:100-102 */
    double t = 0;
    for (int i = 0; i < A; ++i) t += r[i];
    for (int i = 0; i < A; ++i) rew[i] = t;
  } else {
    for (int i = 0; i < A; ++i) rew[i] = r[i];
  }
}

/* B independent worlds: pos [B][E][2], vel [B][A][2], act [B][A][5], obs [B][Dtot], rew [B][A], collisions [B][A] */
void orc_step_batch(const OrcSpec *s, int64_t B, double *pos, double *vel, const double *act, double *obs,
                    double *rew, int32_t *collisions, int threads) {
  const int A = s->n_agents, E = A + s->n_landmarks, D = orc_obs_total(s);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
  for (int64_t b = 0; b < B; ++b)
    orc_step(s, pos + b * 2 * E, vel + b * 2 * A, act + b * 5 * A, obs + b * D, rew + b * A,
             collisions ? collisions + b * A : NULL);
}

/* Throughput of the loop above: `threads` workers, each stepping its own world with uniform random one-hot
 * moves and a reset every `episode_len` steps for about `seconds`; returns env-steps/s (aggregate). */
static uint64_t xs(uint64_t *st) { uint64_t x = *st; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *st = x; }
static double u01(uint64_t *st) { return (double)(xs(st) >> 11) * (1.0 / 9007199254740992.0); }
double orc_bench(const OrcSpec *s, double seconds, int threads, int episode_len, double landmark_range) {
  const int A = s->n_agents, L = s->n_landmarks, E = A + L, D = orc_obs_total(s);
  double total = 0;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel num_threads(threads > 0 ? threads : 1) reduction(+ : total)
  {
    uint64_t st = 0x9E3779B97F4A7C15ull * (uint64_t)(1 + rand());
    double pos[2 * ORC_MAX_E], vel[2 * ORC_MAX_E], act[5 * ORC_MAX_E], rew[ORC_MAX_E];
    double *obs = (double *)malloc(sizeof(double) * (size_t)D);
    int64_t n = 0;
    for (;;) {
      if (episode_len == 0 ? n == 0 : n % episode_len == 0) {
        for (int e = 0; e < E; ++e) {
          const double r = e < A ? 1.0 : landmark_range;
          pos[2 * e] = (2 * u01(&st) - 1) * r; pos[2 * e + 1] = (2 * u01(&st) - 1) * r;
        }
        memset(vel, 0, sizeof(double) * 2 * (size_t)A);
      }
      memset(act, 0, sizeof(double) * 5 * (size_t)A);
      for (int i = 0; i < A; ++i) act[5 * i + (int)(xs(&st) % 5)] = 1.0;
      orc_step(s, pos, vel, act, obs, rew, NULL);
      ++n;
      if ((n & 255) == 0) {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) >= seconds) break;
      }
    }
    free(obs);
    total += (double)n;
  }
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return total / ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
}
