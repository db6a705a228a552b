/* The step server from C, with nothing but the HIP runtime around it: T steps of simple_spread (3 agents, 3 landmarks) for 192
 * worlds -- the first from the SURVEY.md A.3 known-answer state, with a different one-hot move per step -- (a) as T mpe_step
 * launches and (b) COMMANDED to mpe_step_server_start: the moves of all T steps written and the doorbell rung T ahead
 * (mpe_step_server_ring), then one server launch on the same stream (every command precedes it: no second stream needed),
 * mpe_step_server_wait, and the per-workgroup flags read back.  (b) must equal (a) bit for bit, step by step -- rows,
 * rewards, dones in their own blocks, the state in HBM after the last step.  Built and run by tests/test_gpu_abi.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mpe_hip.h"

#define B 192
#define T 5
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 10; } } while (0)

int main(void) {
  static MpeScenarioDesc d;
  d.kind = MPE_SCN_SPREAD;            /* simple_spread.py:7-29 */
  d.n_agents = 3; d.n_landmarks = 3; d.dim_c = 2; d.collaborative = 1;
  d.dt = 0.1f; d.damping = 0.25f; d.contact_force = 100.f; d.contact_margin = 1e-3f;   /* core.py:94-99 */
  for (int e = 0; e < 6; ++e) {
    d.size[e] = e < 3 ? 0.15f : 0.05f; d.mass[e] = 1.f; d.accel[e] = 5.f; d.max_speed[e] = -1.f;
    d.movable[e] = e < 3; d.collide[e] = e < 3;
  }
  if (mpe_fill_obs_layout(&d) != 54) return 1;
  if (mpe_step_server_supported(&d, B) != 1) { printf("no server for this shape\n"); return 2; }
  const int n_flags = (int)mpe_step_server_flags(B);
  if (n_flags != 3) return 3;
  const double init[12] = {0.0976270079, 0.4303787327, 0.2055267521, 0.0897663660, -0.1526904013, 0.2917882261,
                           -0.1248255775, 0.7835460016, 0.9273255210, -0.2331169623, 0.5834500762, 0.0577898395};
  static float pos[6 * 2 * B], vel[3 * 2 * B], act[T][3 * B * 5];
  for (int w = 0; w < B; ++w) {
    for (int e = 0; e < 6; ++e) {      /* worlds differ a little, so that contacts differ */
      pos[(2 * e) * B + w] = (float)init[2 * e] + 0.001f * (float)(w % 17);
      pos[(2 * e + 1) * B + w] = (float)init[2 * e + 1] - 0.002f * (float)(w % 5);
    }
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < 3; ++i) act[t][(i * B + w) * 5 + ((i + t + w) % 5)] = 1.f;
  }
  const size_t obs_n = 54 * B, row_n = 3 * B;
  float *dp[2], *dv[2], *da, *dobs[2], *dr[2]; unsigned char *dd[2];
  for (int k = 0; k < 2; ++k) {       /* k = 0: launched, k = 1: served (T blocks of outputs each) */
    CHECK(hipMalloc((void **)&dp[k], sizeof(pos)));  CHECK(hipMalloc((void **)&dv[k], sizeof(vel)));
    CHECK(hipMalloc((void **)&dobs[k], T * obs_n * 4)); CHECK(hipMalloc((void **)&dr[k], T * row_n * 4));
    CHECK(hipMalloc((void **)&dd[k], T * row_n));
    CHECK(hipMemcpy(dp[k], pos, sizeof(pos), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dv[k], vel, sizeof(vel), hipMemcpyHostToDevice));
    CHECK(hipMemset(dd[k], 7, T * row_n));
  }
  CHECK(hipMalloc((void **)&da, sizeof(act)));
  CHECK(hipMemcpy(da, act, sizeof(act), hipMemcpyHostToDevice));
  /* (a) T launches, step t's outputs in block t */
  for (int t = 0; t < T; ++t) {
    MpeBuffers b;
    memset(&b, 0, sizeof(b));
    b.pos = dp[0]; b.vel = dv[0]; b.act = da + (size_t)t * 3 * B * 5;
    b.obs = dobs[0] + t * obs_n; b.rew = dr[0] + t * row_n; b.done = dd[0] + t * row_n;
    const int rc = mpe_step(&d, &b, B, NULL);
    if (rc) { printf("mpe_step: %d %s\n", rc, mpe_last_error()); return 4; }
  }
  /* (b) the same steps commanded to the server */
  uint64_t *door, *flag; uint32_t *status;
  CHECK(hipMalloc((void **)&door, 8)); CHECK(hipMalloc((void **)&flag, 8 * n_flags)); CHECK(hipMalloc((void **)&status, 4));
  CHECK(hipMemset(door, 0, 8)); CHECK(hipMemset(flag, 0, 8 * n_flags)); CHECK(hipMemset(status, 0, 4));
  MpeStepServer srv;
  memset(&srv, 0, sizeof(srv));
  srv.door = door; srv.flag = flag; srv.status = status; srv.act_ring = da; srv.ring = T; srv.slots = T; srv.timeout_us = 2000000;
  MpeBuffers bs;
  memset(&bs, 0, sizeof(bs));
  bs.pos = dp[1]; bs.vel = dv[1]; bs.obs = dobs[1]; bs.rew = dr[1]; bs.done = dd[1];
  int rc = mpe_step_server_ring(&srv, T, NULL);                      /* every command first ... */
  if (rc) { printf("ring: %d %s\n", rc, mpe_last_error()); return 5; }
  rc = mpe_step_server_start(&d, &bs, B, T, 0, 1.0f, 0, 0, 0, &srv, NULL);   /* ... then the launch: it never waits */
  if (rc) { printf("start: %d %s\n", rc, mpe_last_error()); return 6; }
  rc = mpe_step_server_wait(&srv, B, T, NULL);
  if (rc) { printf("wait: %d %s\n", rc, mpe_last_error()); return 7; }
  CHECK(hipDeviceSynchronize());
  uint64_t hflag[8]; uint32_t hstatus = 9;
  CHECK(hipMemcpy(hflag, flag, 8 * n_flags, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&hstatus, status, 4, hipMemcpyDeviceToHost));
  if (hstatus != 0) { printf("status %u\n", hstatus); return 8; }
  for (int k = 0; k < n_flags; ++k) if (hflag[k] != T) { printf("flag[%d] = %llu\n", k, (unsigned long long)hflag[k]); return 9; }
  static float o[2][T * 54 * B], r[2][T * 3 * B], p[2][6 * 2 * B], v[2][3 * 2 * B];
  static unsigned char dn[2][T * 3 * B];
  for (int k = 0; k < 2; ++k) {
    CHECK(hipMemcpy(o[k], dobs[k], sizeof(o[k]), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r[k], dr[k], sizeof(r[k]), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(dn[k], dd[k], sizeof(dn[k]), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(p[k], dp[k], sizeof(p[k]), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(v[k], dv[k], sizeof(v[k]), hipMemcpyDeviceToHost));
  }
  if (memcmp(o[0], o[1], sizeof(o[0])) || memcmp(r[0], r[1], sizeof(r[0])) || memcmp(dn[0], dn[1], sizeof(dn[0]))) {
    printf("served outputs differ from the launched steps'\n");
    return 11;
  }
  if (memcmp(p[0], p[1], sizeof(p[0])) || memcmp(v[0], v[1], sizeof(v[0]))) { printf("state differs\n"); return 12; }
  int moved = 0;
  for (int w = 0; w < B; ++w) moved += p[1][w] != pos[w];
  if (moved < B / 2 || dn[1][0] != 0) { printf("nothing happened?\n"); return 13; }
  printf("%d steps x %d worlds: served == launched bit for bit; flags %llu, status 0: ok\n", T, B, (unsigned long long)hflag[0]);
  return 0;
}
