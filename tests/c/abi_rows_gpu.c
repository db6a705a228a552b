/* A C caller of the COMPOSABLE output stage (no Python, no torch): simple_spread (simple_spread.py:72-100) written as a row
 * program by hand -- 5 observation ops and 5 reward ops per agent, `enum MpeRowOp` of include/mpe_hip.h -- validated
 * (mpe_rows_validate) and stepped (mpe_step_rows) on 64 copies of the SURVEY.md A.3 known-answer world; the result is held
 * against the reference's recorded answer AND, bit for bit, against mpe_step's own kernel for that scenario.
 * Built (gcc, linked against libamdhip64) and run by tests/test_gpu_abi.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mpe_hip.h"

#define B 64
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 10; } } while (0)
#define W0(code, a0, a1, a2) ((int32_t)((code) | ((a0) << 8) | ((a1) << 16) | ((uint32_t)(a2) << 24)))
#define MINUS_ONE ((int32_t)0xbf800000)   /* the bits of -1.0f: word 2 of an op is its float operand */

int main(void) {
  static MpeScenarioDesc d;
  MpeBuffers b;
  memset(&b, 0, sizeof(b));
  d.kind = MPE_SCN_SPREAD;
  d.n_agents = 3; d.n_landmarks = 3; d.dim_c = 2; d.collaborative = 1;
  d.dt = 0.1f; d.damping = 0.25f; d.contact_force = 100.f; d.contact_margin = 1e-3f;
  for (int e = 0; e < 6; ++e) {
    d.size[e] = e < 3 ? 0.15f : 0.05f; d.mass[e] = 1.f; d.accel[e] = 5.f; d.max_speed[e] = -1.f;
    d.movable[e] = e < 3; d.collide[e] = e < 3;
  }
  if (mpe_fill_obs_layout(&d) != 54) return 1;
  /* ---- the program: agent i's row = [own vel | own pos | offsets to the 3 landmarks | offsets to the other agents | their
   * (silent) utterances: 4 zeros]; its reward = - sum over landmarks of the distance of the nearest agent - 1 per agent in
   * contact with it (itself included: SURVEY Q1), shared by the team (desc.collaborative) */
  static int32_t ops[30][4];
  MpeRowProgram p;
  memset(&p, 0, sizeof(p));
  int n = 0;
  for (int i = 0; i < 3; ++i) {
    p.obs_begin[i] = n;
    ops[n++][0] = W0(MPE_ROW_OBS_VEL, MPE_ROW_SELF, 0, 0);
    ops[n++][0] = W0(MPE_ROW_OBS_POS, MPE_ROW_SELF, 0, 0);
    ops[n++][0] = W0(MPE_ROW_OBS_REL_RANGE, 3, 3, 0);            /* entities 3, 4, 5: the landmarks */
    ops[n++][0] = W0(MPE_ROW_OBS_REL_RANGE, 0, 3, 1);            /* agents 0, 1, 2 without the observer (flag 1) */
    ops[n++][0] = W0(MPE_ROW_OBS_CONST_N, 0, 4, 0);              /* 4 times word 2 = 0.0 */
  }
  for (int i = 3; i <= MPE_ROWS_MAX_ENTITIES; ++i) p.obs_begin[i] = n;
  for (int i = 0; i < 3; ++i) {
    p.rew_begin[i] = n;
    ops[n++][0] = W0(MPE_ROW_R_ZERO, 0, 0, 0);
    ops[n++][0] = W0(MPE_ROW_R_ZERO, 0, 0, 1);
    ops[n][0] = W0(MPE_ROW_R_ADD_MIN_DIST_GRID, 0, 3, 0); ops[n][1] = 3 | (3 << 8); ops[n++][2] = MINUS_ONE;   /* agents 0-2 x landmarks 3-5 */
    ops[n][0] = W0(MPE_ROW_R_ADD_IF_HIT_GRID, 0, i, 0);   ops[n][1] = 3 | (1 << 8); ops[n++][2] = MINUS_ONE;   /* agents 0-2 x agent i */
    ops[n++][0] = W0(MPE_ROW_R_STORE, i, 0, 0);
  }
  for (int i = 3; i <= MPE_ROWS_MAX_ENTITIES; ++i) p.rew_begin[i] = n;
  if (n != 30) return 2;
  p.n_ops = n;
  p.n_vel = 3;
  int32_t *dops; void *dhdr;
  CHECK(hipMalloc((void **)&dops, sizeof(ops)));
  CHECK(hipMemcpy(dops, ops, sizeof(ops), hipMemcpyHostToDevice));
  CHECK(hipMalloc(&dhdr, MPE_ROWS_HEADER_BYTES));
  p.ops_device = dops;
  p.header_device = dhdr;
  MpeScenarioDesc g = d;                /* a user scenario has no kind of its own */
  g.kind = MPE_SCN_GENERIC;
  if (mpe_rows_validate(&g, &p, &ops[0][0]) != 0) { printf("validate: %s\n", mpe_last_error()); return 3; }

  const double init[12] = {0.0976270079, 0.4303787327, 0.2055267521, 0.0897663660, -0.1526904013, 0.2917882261,
                           -0.1248255775, 0.7835460016, 0.9273255210, -0.2331169623, 0.5834500762, 0.0577898395};
  const double obs0[18] = {0.6214080569, 0.0672186731, 0.1597678135, 0.4371006001, -0.2845933910, 0.3464454015,
                           0.7675577075, -0.6702175624, 0.4236822626, -0.3793107606, -0.0042410614, -0.3473342341,
                           -0.3245990205, -0.1020342412, 0, 0, 0, 0};
  const double rew = -8.1422486230;
  static float pos[6 * 2 * B], vel[3 * 2 * B], act[3 * B * 5], obs[2][54 * B], r[2][3 * B], pos_out[2][6 * 2 * B];
  for (int w = 0; w < B; ++w) {
    for (int e = 0; e < 6; ++e) { pos[(2 * e) * B + w] = (float)init[2 * e]; pos[(2 * e + 1) * B + w] = (float)init[2 * e + 1]; }
    for (int i = 0; i < 3; ++i) act[(i * B + w) * 5 + (i + 1)] = 1.f;
  }
  float *dp, *dv, *da, *dobs, *dr; unsigned char *dd;
  CHECK(hipMalloc((void **)&dp, sizeof(pos)));  CHECK(hipMalloc((void **)&dv, sizeof(vel)));
  CHECK(hipMalloc((void **)&da, sizeof(act)));  CHECK(hipMalloc((void **)&dobs, sizeof(obs[0])));
  CHECK(hipMalloc((void **)&dr, sizeof(r[0]))); CHECK(hipMalloc((void **)&dd, 3 * B));
  CHECK(hipMemcpy(da, act, sizeof(act), hipMemcpyHostToDevice));
  b.pos = dp; b.vel = dv; b.act = da; b.obs = dobs; b.rew = dr; b.done = dd;
  for (int which = 0; which < 2; ++which) {       /* 0: the program, 1: mpe_step's own simple_spread kernel */
    CHECK(hipMemcpy(dp, pos, sizeof(pos), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dv, vel, sizeof(vel), hipMemcpyHostToDevice));
    const int rc = which == 0 ? mpe_step_rows(&g, &b, &p, B, NULL) : mpe_step(&d, &b, B, NULL);
    if (rc) { printf("%s: %d %s\n", which == 0 ? "mpe_step_rows" : "mpe_step", rc, mpe_last_error()); return 4; }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(obs[which], dobs, sizeof(obs[0]), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r[which], dr, sizeof(r[0]), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(pos_out[which], dp, sizeof(pos), hipMemcpyDeviceToHost));
  }
  double worst = 0;
  for (int w = 0; w < B; ++w) {
    for (int k = 0; k < 18; ++k) { const double e = fabs(obs[0][w * 18 + k] - obs0[k]); if (e > worst) worst = e; }
    for (int i = 0; i < 3; ++i) { const double e = fabs(r[0][i * B + w] - rew) / 8.0; if (e > worst) worst = e; }
  }
  printf("row program: max err vs the reference KAT %.3e\n", worst);
  if (!(worst < 1e-5)) return 5;
  if (memcmp(obs[0], obs[1], sizeof(obs[0])) || memcmp(r[0], r[1], sizeof(r[0])) || memcmp(pos_out[0], pos_out[1], sizeof(pos))) {
    printf("the program's step differs from mpe_step's\n");
    return 6;
  }
  printf("row program == mpe_step's kernel, bit for bit\nok\n");
  return 0;
}
