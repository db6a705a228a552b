// mpe_rng.hip -- device-side reset_world and synthetic random moves (Philox4x32-10, counter-based).
//
// reset_world (simple_spread.py:31-45, simple_tag.py:39-54, simple.py:24-39): for every selected
// world draw agent positions U[-1,1)^2, landmark positions U[-r,r)^2, zero the velocities.  The
// reference consumes NumPy's global MT19937; a counter-based generator keyed by
// (seed, world, episode, entity) gives every world its own reproducible stream with no state in
// HBM and no ordering between worlds (seed-exact parity resets are uploaded from the host instead).
#include "mpe_internal.h"
#include <cstring>

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

// Uniform random moves for T consecutive global steps (the moves a policy would hand over; bench / rollout
// workloads).  A wave owns 64 consecutive worlds of one agent QUAD at one step: one Philox block per lane
// yields the moves of agents 4q .. 4q+3 of that lane's world (action_draw's counter layout), and the
// one-hot rows [64 worlds][5] of an agent -- 1280 contiguous bytes of act [A][B][5] -- leave as five
// 256-byte wave stores, lane f of store k writing float 64k + f of the run (the move of world (64k+f)/5
// arrives by a wave shuffle).  Step s writes the s-th consecutive [A][B][5] / [A][B] tensor.
// (The same run as 80 16-byte pieces -- two store instructions per agent instead of five -- is SLOWER: 21.8 vs 20.4 us for the
// 98 MB block of 25 steps at 65 536 worlds, round 4 session 33: five whole-line 256-byte wave stores beat 1024 + 256.)
// (with `rs.enabled`: the launch ALSO is mpe_reset(mask = NULL) -- an episode boundary of a rollout that draws its next block of
//  moves there: the waves of the block's first step and first agent quad place their 64 worlds' entities, the draws k_reset makes,
//  one launch instead of two in front of the episode's first step)
struct ResetWithDraw {
  float *pos, *vel;
  int32_t *choice;
  int32_t enabled, E, n_choices, pop[MPE_MAX_CHOICES];
  float landmark_range;
  uint64_t episode;
};
__global__ void __launch_bounds__(kBlock)
k_random_actions(float *__restrict__ act, int32_t *__restrict__ ids, size_t B, int A, uint64_t seed, uint64_t step0,
                 uint64_t world_offset, const ResetWithDraw rs) {
  const int lane = threadIdx.x & (kWave - 1);
  const size_t w0 = ((size_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kWave;   // wave-uniform
  if (w0 >= B) return;
  const int q = blockIdx.y;
  const uint64_t step = step0 + blockIdx.z;
  const int nvalid = (B - w0) < (size_t)kWave ? (int)(B - w0) : kWave;
  const size_t w = w0 + (size_t)(lane < nvalid ? lane : nvalid - 1);
  const uint64_t gw = world_offset + w;
  if (rs.enabled && blockIdx.z == 0 && q == 0 && lane < nvalid) {   // (launch-uniform flag; block-uniform position)
    for (int e = 0; e < rs.E; ++e) {
      float x, y;
      reset_draw(seed, gw, rs.episode, e, e < A ? 1.0f : rs.landmark_range, x, y);
      rs.pos[(size_t)(2 * e) * B + w] = x;
      rs.pos[(size_t)(2 * e + 1) * B + w] = y;
      if (e < A) {
        rs.vel[(size_t)(2 * e) * B + w] = 0.f;
        rs.vel[(size_t)(2 * e + 1) * B + w] = 0.f;
      }
    }
    if (rs.choice)
      for (int k = 0; k < rs.n_choices; ++k) rs.choice[(size_t)k * B + w] = choice_draw(seed, gw, rs.episode, k, rs.pop[k]);
  }
  U4 c;
  c.x = (uint32_t)gw;
  c.y = (uint32_t)(gw >> 32) ^ (uint32_t)(step >> 32);
  c.z = (uint32_t)q;
  c.w = kStreamAction ^ (uint32_t)step;
  const U4 o = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t word[4] = {o.x, o.y, o.z, o.w};
  const size_t sblk = (size_t)blockIdx.z * (size_t)A * B;   // this step's tensor
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const int i = 4 * q + k4;
    if (i >= A) break;   // uniform
    const int m = (int)(((uint64_t)word[k4] * 5u) >> 32);
    if (ids && lane < nvalid) ids[sblk + (size_t)i * B + w] = m;
    if (act) {
      float *const run = act + (sblk + (size_t)i * B + w0) * MPE_ACTION_DIM;   // wave-uniform: 64 rows of 5 floats
#pragma unroll
      for (int k = 0; k < MPE_ACTION_DIM; ++k) {
        const int f = kWave * k + lane, wl = f / MPE_ACTION_DIM, comp = f - wl * MPE_ACTION_DIM;
        const int ms = __shfl(m, wl, kWave);
#ifdef MPE_ACT_BLOCK_NT      // A/B (round 5): nontemporal stores for the block of moves -- measured, not adopted (profiles/r5_act_block_nt_ab.txt)
        if (wl < nvalid) __builtin_nontemporal_store((comp == ms) ? 1.f : 0.f, &run[f]);
#else
        if (wl < nvalid) run[f] = (comp == ms) ? 1.f : 0.f;
#endif
      }
    }
  }
}

// Uniform random words of the agents that speak: one-hot rows into comm [A][B][dim_c] (communication scenarios'
// synthetic workload; the rows the fused rollout recomputes in-kernel).  One thread per (world, agent).
__global__ void __launch_bounds__(kBlock)
k_random_comm(float *__restrict__ comm, size_t B, int A, int dim_c, unsigned speakers, uint64_t seed, uint64_t step0,
              uint64_t world_offset) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (w >= B || !((speakers >> i) & 1u)) return;
  const int id = comm_draw(seed, world_offset + w, step0 + blockIdx.z, i, dim_c);
  float *row = comm + (((size_t)blockIdx.z * A + i) * B + w) * dim_c;   // step s: the s-th consecutive [A][B][dim_c] tensor
  for (int c = 0; c < dim_c; ++c) row[c] = c == id ? 1.f : 0.f;
}

int launch_random_comm(float *comm, int A, size_t B, int dim_c, unsigned speakers, uint64_t seed, uint64_t step0, int T,
                       uint64_t world_offset, hipStream_t stream) {
  const dim3 grid((unsigned)((B + kBlock - 1) / kBlock), (unsigned)A, (unsigned)T);
  hipLaunchKernelGGL(k_random_comm, grid, dim3(kBlock), 0, stream, comm, B, A, dim_c, speakers, seed, step0, world_offset);
  return (int)hipGetLastError();
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

// mpe_reset_rows with per-entity boxes (a row program's reset placement): the draws the in-kernel restarts of k_rows make
__global__ void __launch_bounds__(kBlock)
k_reset_box(float *__restrict__ pos, float *__restrict__ vel, const uint8_t *__restrict__ mask, size_t B, int A, int E,
            const ResetBoxes boxes, uint64_t seed, uint64_t episode, uint64_t world_offset, int32_t *__restrict__ choice,
            int n_choices, int pop0, int pop1, int pop2, int pop3) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  if (w >= B) return;
  if (mask && !mask[w]) return;
  float x, y;
  reset_draw_box(seed, world_offset + w, episode, e, boxes.box[e][0], boxes.box[e][1], boxes.box[e][2], boxes.box[e][3], x, y);
  pos[(size_t)(2 * e) * B + w] = x;
  pos[(size_t)(2 * e + 1) * B + w] = y;
  if (e < A) {
    vel[(size_t)(2 * e) * B + w] = 0.f;
    vel[(size_t)(2 * e + 1) * B + w] = 0.f;
  }
  if (e == 0 && choice) {
    const int pop[MPE_MAX_CHOICES] = {pop0, pop1, pop2, pop3};
    for (int k = 0; k < n_choices; ++k) choice[(size_t)k * B + w] = choice_draw(seed, world_offset + w, episode, k, pop[k]);
  }
}

int launch_reset_box(int A, int L, const MpeBuffers &b, size_t B, const uint8_t *mask, const ResetBoxes &boxes, uint64_t seed,
                     uint64_t episode, uint64_t world_offset, int n_choices, const int32_t *pop, hipStream_t stream) {
  const dim3 grid((unsigned)((B + kBlock - 1) / kBlock), (unsigned)(A + L));
  hipLaunchKernelGGL(k_reset_box, grid, dim3(kBlock), 0, stream, b.pos, b.vel, mask, B, A, A + L, boxes, seed, episode, world_offset,
                     n_choices > 0 ? b.choice : nullptr, n_choices, pop[0], pop[1], pop[2], pop[3]);
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

int launch_random_actions(float *act, int32_t *ids, int A, size_t B, uint64_t seed, uint64_t step0, int T,
                          uint64_t world_offset, hipStream_t stream) {
  const size_t per_block = (size_t)(kBlock / kWave) * kWave;
  const dim3 grid((unsigned)((B + per_block - 1) / per_block), (unsigned)((A + 3) / 4), (unsigned)T);
  ResetWithDraw rs;
  std::memset(&rs, 0, sizeof(rs));
  hipLaunchKernelGGL(k_random_actions, grid, dim3(kBlock), 0, stream, act, ids, B, A, seed, step0, world_offset, rs);
  return (int)hipGetLastError();
}

int launch_reset_random_actions(int A, int L, const MpeBuffers &b, size_t B, float landmark_range, uint64_t episode, int n_choices,
                                const int32_t *pop, float *act, int32_t *ids, uint64_t seed, uint64_t step0, int T,
                                uint64_t world_offset, hipStream_t stream) {
  const size_t per_block = (size_t)(kBlock / kWave) * kWave;
  const dim3 grid((unsigned)((B + per_block - 1) / per_block), (unsigned)((A + 3) / 4), (unsigned)T);
  ResetWithDraw rs;
  std::memset(&rs, 0, sizeof(rs));
  rs.pos = b.pos;
  rs.vel = b.vel;
  rs.choice = n_choices > 0 ? b.choice : nullptr;
  rs.enabled = 1;
  rs.E = A + L;
  rs.n_choices = n_choices;
  for (int k = 0; k < MPE_MAX_CHOICES; ++k) rs.pop[k] = pop[k];
  rs.landmark_range = landmark_range;
  rs.episode = episode;
  hipLaunchKernelGGL(k_random_actions, grid, dim3(kBlock), 0, stream, act, ids, B, A, seed, step0, world_offset, rs);
  return (int)hipGetLastError();
}

}  // namespace mpe
