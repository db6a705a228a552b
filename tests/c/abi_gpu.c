/* A C caller of libmpe_hip.so on a GPU, with nothing but the HIP runtime around it (no Python, no torch): the
 * SURVEY.md A.3 known-answer vector of simple_spread -- np.random.seed(0); env.reset(); step with one-hot moves
 * (+x, -x, +y) -- recorded from the reference, stepped for 64 copies of that world through mpe_step.
 * Built (gcc, linked against libamdhip64) and run by tests/test_gpu_abi.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mpe_hip.h"

#define B 64
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 10; } } while (0)

int main(void) {
  static MpeScenarioDesc d;
  MpeBuffers b;
  memset(&b, 0, sizeof(b));
  d.kind = MPE_SCN_SPREAD;            /* simple_spread.py:7-29 */
  d.n_agents = 3; d.n_landmarks = 3; d.dim_c = 2; d.collaborative = 1;
  d.dt = 0.1f; d.damping = 0.25f; d.contact_force = 100.f; d.contact_margin = 1e-3f;   /* core.py:94-99 */
  for (int e = 0; e < 6; ++e) {
    d.size[e] = e < 3 ? 0.15f : 0.05f; d.mass[e] = 1.f; d.accel[e] = 5.f; d.max_speed[e] = -1.f;
    d.movable[e] = e < 3; d.collide[e] = e < 3;
  }
  if (mpe_fill_obs_layout(&d) != 54) return 1;
  if (mpe_step_supported(&d) != 1) return 2;
  const double init[12] = {0.0976270079, 0.4303787327, 0.2055267521, 0.0897663660, -0.1526904013, 0.2917882261,
                           -0.1248255775, 0.7835460016, 0.9273255210, -0.2331169623, 0.5834500762, 0.0577898395};
  const double obs0[18] = {0.6214080569, 0.0672186731, 0.1597678135, 0.4371006001, -0.2845933910, 0.3464454015,
                           0.7675577075, -0.6702175624, 0.4236822626, -0.3793107606, -0.0042410614, -0.3473342341,
                           -0.3245990205, -0.1020342412, 0, 0, 0, 0};
  const double rew = -8.1422486230;
  static float pos[6 * 2 * B], vel[3 * 2 * B], act[3 * B * 5], obs[54 * B], r[3 * B];
  static unsigned char done[3 * B];
  for (int w = 0; w < B; ++w) {
    for (int e = 0; e < 6; ++e) { pos[(2 * e) * B + w] = (float)init[2 * e]; pos[(2 * e + 1) * B + w] = (float)init[2 * e + 1]; }
    for (int i = 0; i < 3; ++i) act[(i * B + w) * 5 + (i + 1)] = 1.f;   /* agent i moves with one-hot index i+1 */
  }
  float *dp, *dv, *da, *dobs, *dr; unsigned char *dd;
  CHECK(hipMalloc((void **)&dp, sizeof(pos)));  CHECK(hipMalloc((void **)&dv, sizeof(vel)));
  CHECK(hipMalloc((void **)&da, sizeof(act)));  CHECK(hipMalloc((void **)&dobs, sizeof(obs)));
  CHECK(hipMalloc((void **)&dr, sizeof(r)));    CHECK(hipMalloc((void **)&dd, sizeof(done)));
  CHECK(hipMemcpy(dp, pos, sizeof(pos), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dv, vel, sizeof(vel), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(da, act, sizeof(act), hipMemcpyHostToDevice));
  CHECK(hipMemset(dd, 7, sizeof(done)));
  b.pos = dp; b.vel = dv; b.act = da; b.obs = dobs; b.rew = dr; b.done = dd;
  const int rc = mpe_step(&d, &b, B, NULL);
  if (rc) { printf("mpe_step: %d %s\n", rc, mpe_last_error()); return 3; }
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(obs, dobs, sizeof(obs), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(r, dr, sizeof(r), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(done, dd, sizeof(done), hipMemcpyDeviceToHost));
  double worst = 0;
  for (int w = 0; w < B; ++w) {
    for (int k = 0; k < 18; ++k) { const double e = fabs(obs[w * 18 + k] - obs0[k]); if (e > worst) worst = e; }   /* obs_n[0] = rows [B][18] at offset 0 */
    for (int i = 0; i < 3; ++i) {
      const double e = fabs(r[i * B + w] - rew) / 8.0; if (e > worst) worst = e;
      if (done[i * B + w] != 0) return 4;
    }
  }
  printf("max err vs the reference KAT %.3e\n", worst);
  if (!(worst < 1e-5)) return 5;
  printf("ok\n");
  return 0;
}
