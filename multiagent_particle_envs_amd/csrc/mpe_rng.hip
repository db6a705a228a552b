// mpe_rng.hip -- device-side reset_world and synthetic random moves (Philox4x32-10, counter-based).
//
// reset_world (simple_spread.py:31-45, simple_tag.py:39-54, simple.py:24-39): for every selected
// world draw agent positions U[-1,1)^2, landmark positions U[-r,r)^2, zero the velocities.  The
// reference consumes NumPy's global MT19937; a counter-based generator keyed by
// (seed, world, episode, entity) gives every world its own reproducible stream with no state in
// HBM and no ordering between worlds (seed-exact parity resets are uploaded from the host instead).
#include "mpe_internal.h"

namespace mpe {

constexpr int kBlock = 256;

// one thread per (world, entity): all stores coalesced over the batch axis
__global__ void __launch_bounds__(kBlock)
k_reset(float *__restrict__ pos, float *__restrict__ vel, const uint8_t *__restrict__ mask, size_t B, int A, int E,
        float landmark_range, uint64_t seed, uint64_t episode, uint64_t world_offset, int32_t *__restrict__ choice,
        int n_choices, int pop0, int pop1, int pop2, int pop3) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  if (w >= B) return;
  if (mask && !mask[w]) return;
  float x, y;
  reset_draw(seed, world_offset + w, episode, e, e < A ? 1.0f : landmark_range, x, y);
  pos[(size_t)(2 * e) * B + w] = x;
  pos[(size_t)(2 * e + 1) * B + w] = y;
  if (e < A) {
    vel[(size_t)(2 * e) * B + w] = 0.f;
    vel[(size_t)(2 * e + 1) * B + w] = 0.f;
  }
  if (e == 0 && choice) {  // the np.random.choice draws of reset_world (goal landmark, ...)
    const int pop[MPE_MAX_CHOICES] = {pop0, pop1, pop2, pop3};
    for (int k = 0; k < n_choices; ++k) choice[(size_t)k * B + w] = choice_draw(seed, world_offset + w, episode, k, pop[k]);
  }
}

__global__ void __launch_bounds__(kBlock)
k_random_actions(float *__restrict__ act, int32_t *__restrict__ ids, size_t B, uint64_t seed, uint64_t step,
                 uint64_t world_offset) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (w >= B) return;
  const int m = action_draw(seed, world_offset + w, step, i);
  if (ids) ids[(size_t)i * B + w] = m;
  if (act) {
    float *row = act + ((size_t)i * B + w) * MPE_ACTION_DIM;
#pragma unroll
    for (int k = 0; k < MPE_ACTION_DIM; ++k) row[k] = (k == m) ? 1.f : 0.f;
  }
}

// Episode bookkeeping (new API, SURVEY 8 f1: the reference never ends an episode, environment.py:132-135): one
// thread per world bumps the world's step counter, ORs "horizon reached" into the A done rows the step kernel (zeros)
// or the done_callback wrote, and clears the counter of a finished world when its reset follows (auto-reset).
__global__ void __launch_bounds__(kBlock)
k_episode_tick(int32_t *__restrict__ episode_step, uint8_t *__restrict__ done, size_t B, int A, int max_steps,
               int clear_finished) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= B) return;
  const int c = episode_step[w] + 1;
  const bool over = max_steps > 0 && c >= max_steps;
  episode_step[w] = (over && clear_finished) ? 0 : c;
  if (over)
    for (int a = 0; a < A; ++a) done[(size_t)a * B + w] = 1;
}

int launch_episode_tick(int32_t *episode_step, uint8_t *done, int A, size_t B, int max_steps, int clear_finished,
                        hipStream_t stream) {
  hipLaunchKernelGGL(k_episode_tick, dim3((unsigned)((B + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, episode_step,
                     done, B, A, max_steps, clear_finished);
  return (int)hipGetLastError();
}

int launch_reset(int A, int L, const MpeBuffers &b, size_t B, const uint8_t *mask, float landmark_range,
                 uint64_t seed, uint64_t episode, uint64_t world_offset, int n_choices, const int32_t *pop,
                 hipStream_t stream) {
  const dim3 grid((unsigned)((B + kBlock - 1) / kBlock), (unsigned)(A + L));
  hipLaunchKernelGGL(k_reset, grid, dim3(kBlock), 0, stream, b.pos, b.vel, mask, B, A, A + L, landmark_range, seed,
                     episode, world_offset, n_choices > 0 ? b.choice : nullptr, n_choices, pop[0], pop[1], pop[2], pop[3]);
  return (int)hipGetLastError();
}

int launch_random_actions(float *act, int32_t *ids, int A, size_t B, uint64_t seed, uint64_t step,
                          uint64_t world_offset, hipStream_t stream) {
  const dim3 grid((unsigned)((B + kBlock - 1) / kBlock), (unsigned)A);
  hipLaunchKernelGGL(k_random_actions, grid, dim3(kBlock), 0, stream, act, ids, B, seed, step, world_offset);
  return (int)hipGetLastError();
}

}  // namespace mpe
