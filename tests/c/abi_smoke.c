/* A plain-C caller of libmpe_hip.so: proves include/mpe_hip.h is a C header (no C++/torch types cross the
 * boundary) and that the argument checking works without a GPU.  Built and run by tests/test_host_cpu.py. */
#include <stdio.h>
#include <string.h>

#include "mpe_hip.h"

int main(void) {
  static MpeScenarioDesc d; /* zero-initialised */
  MpeBuffers b;
  memset(&b, 0, sizeof(b));
  printf("abi %d desc %zu bufs %zu\n", mpe_abi_version(), mpe_sizeof_desc(), mpe_sizeof_buffers());
  if (mpe_abi_version() != MPE_ABI_VERSION) return 1;
  if (mpe_sizeof_desc() != sizeof(MpeScenarioDesc) || mpe_sizeof_buffers() != sizeof(MpeBuffers)) return 2;
  if (mpe_step(NULL, &b, 4, NULL) != MPE_EINVAL) return 3;
  if (strstr(mpe_last_error(), "desc is NULL") == NULL) return 4;
  d.kind = MPE_SCN_SPREAD;
  d.n_agents = 3;
  d.n_landmarks = 3;
  d.dim_c = 2;
  for (int e = 0; e < 6; ++e) { d.mass[e] = 1.0f; d.size[e] = e < 3 ? 0.15f : 0.05f; }
  if (mpe_fill_obs_layout(&d) != 54) return 5; /* 3 agents x 18 (simple_spread.py:84-100) */
  if (d.obs_off[1] != 18 || d.obs_off[3] != 54) return 6;
  if (mpe_step(&d, &b, 4, NULL) != MPE_EINVAL) return 7; /* pos is NULL */
  printf("ok\n");
  return 0;
}
