// mpe_split.hip -- wave-per-agent / lane-per-world kernels: the fused step and the fused T-step rollout.
//
// Why: at the benchmark batch (65 536 worlds per GPU) a thread-per-world launch is 1024 waves --
// one per SIMD on 256 CUs -- and each lane walks ~2000 dependent VALU / transcendental
// instructions, so the step is latency-bound at ~10 us although its 27 MB would stream in ~4 us.
// Here a workgroup is (A + 1) waves x 64 lanes: lane = world, and the waves split the work of those
// 64 worlds by ROLE:
//   wave i < A   AGENT i: loads the world, its contacts and its integration (core.py:117-169), publishes
//                (pos_i, vel_i[, |pos_i - landmark_l|^2]) in LDS, and after ONE __syncthreads stores its
//                new state (behind the barrier: a sibling wave may still be loading the pre-step
//                positions before it) and assembles and stores agent i's observation rows;
//   wave A       REWARD: touches no global input at all -- behind the same barrier it reads what the
//                agents published and computes every agent's reward / done / benchmark_data once
//                (the reference recomputes the shared terms per agent: O(A^2 L) -> O(A L)).
// Every global access is a 256-byte coalesced wave access over the batch axis, the chip holds A + 1
// times as many waves as a thread-per-world launch to hide latency, and no wave walks both the
// observation and the reward chain.  Arithmetic per pair / per agent is the same code (mpe_device.h)
// in the same order as the thread-per-world kernels, so results are bit-identical to them.
//
// ROLL = true is the fused rollout (mpe_rollout_random): T steps in one launch, each agent wave keeps
// its agent's state in registers, moves are drawn in-kernel (Philox, identical to
// mpe_random_actions), resets happen in-kernel (identical to mpe_reset), and every step's
// obs/rew/done are still written -- to per-step trajectory blocks or over the same block.  The LDS
// exchange block is double-buffered by step parity: the reward wave may still be reading step t's
// block while the agents publish step t+1's; step t+2's publish is behind step t+1's barrier.
#include <type_traits>

#include "mpe_internal.h"

// ablation builds (tools/ab_build.sh): bit 0 skip the contact loop, bit 1 skip the reward stage, bit 2 skip the
// observation rows, bit 3 skip the state stores -- what each stage of the step costs
#ifndef MPE_SPLIT_ABLATE
#define MPE_SPLIT_ABLATE 0
#endif

// instrumented build (tools/phase_clock.py): lane 0 of every wave of the first workgroups stamps the shader clock at
// the phase boundaries of each step into the buffer passed as MpeBuffers.force: [workgroup < 4][role < 16][t < 32][8]
// uint64 -- where a step's cycles go.  Not part of the product build.
#ifdef MPE_PHASE_CLOCK
#define MPE_STAMP(k)                                                                                              \
  do {                                                                                                            \
    if (blockIdx.x < 4 && lane == 0 && b.force && t < 32)                                                         \
      reinterpret_cast<unsigned long long *>(b.force)[(((size_t)blockIdx.x * 16 + role) * 32 + t) * 8 + (k)] =     \
          __builtin_amdgcn_s_memtime();                                                                           \
  } while (0)
#else
#define MPE_STAMP(k) do { } while (0)
#endif

// instrumented build (libmpe_hip_span.so, tools/device_span.py): lane 0 of every wave stamps the device's constant-rate
// wall clock (s_memrealtime, 100 MHz, one counter for the whole chip) at its first instruction and at its last, into the
// buffer passed as MpeBuffers.force: [workgroup][role < 8][2] uint64.  min(start) over a launch's waves is when the launch
// began ON THE DEVICE; start-to-start of consecutive launches is the per-launch period of a dependent chain -- a time that
// needs neither the profiler's per-dispatch packets nor the host's events, and that the end stamps do not enter.
// The end stamp comes in two flavours, chosen per launch by bit 0 of the pointer: clear -- taken when the wave has ISSUED
// its last store (the stamp's own store overlaps the row stores: the launch is not lengthened; span is a lower bound);
// set -- taken after `s_waitcnt vmcnt(0)`, i.e. when the wave's stores are ACKNOWLEDGED (the true end of its memory work,
// but the stamp's store then adds one more round trip behind it: ~0.6 us on the launch).  Not part of the product build.
#ifdef MPE_DEVICE_SPAN
// (the start time is only READ at the top -- s_memrealtime into SGPRs, in flight next to the kernel's own scalar loads -- and
//  stored together with the end time by ONE 16-byte store at the wave's end: a store, or a wait for the clock, in front of
//  the wave's first loads would put a scalar round trip on the launch's critical path: measured +0.5-0.9 us on a 3 us launch)
#define MPE_SPAN_BEGIN() const unsigned long long span_t0_ = (unsigned long long)wall_clock64()
#define MPE_SPAN_END()                                                                                             \
  do {                                                                                                             \
    if (b_in.force && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 8) {                                           \
      const uintptr_t sp_ = reinterpret_cast<uintptr_t>(b_in.force);                                                \
      if (sp_ & 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
      ulonglong2 st_;                                                                                              \
      st_.x = span_t0_;                                                                                            \
      st_.y = (unsigned long long)wall_clock64();                                                                  \
      reinterpret_cast<ulonglong2 *>(sp_ & ~(uintptr_t)7)[(size_t)blockIdx.x * 8 + (threadIdx.x >> 6)] = st_;       \
    }                                                                                                              \
  } while (0)
#else
#define MPE_SPAN_BEGIN() do { } while (0)
#define MPE_SPAN_END() do { } while (0)
#endif

namespace mpe {

// DUAL roles in the fused rollout: every agent gets TWO waves -- a PHYSICS wave (World.step of step t+1: move, contacts,
// integration, publish) and a ROWS wave (observation rows of step t from what the physics waves published) -- next to the
// reward wave.  In the rollout a step of an agent is one ~600-instruction dependent chain (World.step -> publish ->
// barrier -> sibling reads -> row tile -> flush); at 16 384 worlds a CU holds one workgroup, i.e. ONE wave per SIMD issuing
// an instruction every 4-5 cycles with six LDS round trips in between: the SIMDs idle ~70 % of the time (round-2 PMC: a
// wave issues 36 % of its lifetime).  Interleaving the two halves inside one wave does not help (measured: the same
// instructions still issue serially, tag 1.38-1.42 vs 1.31 us per step); giving them to two waves does: step t's rows
// and step t+1's World.step depend on the same published state and not on each other, so they run side by side, and
// every workgroup barrier has twice the waves to hide its round trips under.  Same device functions, same values in
// the same order: bit-identical to the single-role loop (tests/test_gpu_rollout.py).
#ifndef MPE_SPLIT_DUAL
#define MPE_SPLIT_DUAL 1
#endif
// Who draws the next step's moves in the dual-role rollout (bit KIND set: every physics wave its own, at the head of its
// step; clear: the reward wave for everybody, as in the single-role rollout).  The reward wave's chain per step is
// [Philox ->] barrier -> reward; where the reward is heavy (simple_spread: A (A-1) / 2 strict-< tests, L minima and square
// roots) that chain is what the step waits for and the draw is better off in the physics waves; where it is light the
// physics waves are the longer chain and the draw stays with the reward wave.  Same-box A/B, us per rollout step
// (single-role | dual, reward wave draws | dual, physics waves draw; profiles/r3_ab_logs.txt session 8):
//   simple_spread N=3  4096 worlds 1.08 | 1.09 | 0.99      16 384 worlds 1.13 | 1.14 | 1.06      N=4, 16 384: 1.83 | 1.83 | 1.69
//   simple_adversary  16 384 worlds 0.79 | 0.58 | 0.64     simple_push 16 384: 0.93 | 0.65 | 0.76     simple 16 384: 0.43 | 0.37 | 0.46
//   simple_tag        16 384 worlds 1.31 | 1.01-1.15 | 1.09-1.10 (by box)
#ifndef MPE_SPLIT_DUAL_OWN_DRAW
#define MPE_SPLIT_DUAL_OWN_DRAW (1 << MPE_SCN_SPREAD)
#endif
template <int KIND>
constexpr bool dual_own_draw() { return ((MPE_SPLIT_DUAL_OWN_DRAW) >> KIND) & 1; }

template <int KIND>
constexpr bool dual_kind() {   // the kinds that have a dual-role rollout kernel
  return (MPE_SPLIT_DUAL) &&
         (KIND == MPE_SCN_SIMPLE || KIND == MPE_SCN_SPREAD || KIND == MPE_SCN_TAG || KIND == MPE_SCN_ADVERSARY || KIND == MPE_SCN_PUSH);
}

template <int KIND, int A, int L, int NADV>
struct SplitShape {
  static constexpr int E = A + L;
  // floats an agent publishes per world: pos, vel, then squared distances to the landmarks the reward needs
  static constexpr int XW = KIND == MPE_SCN_SPREAD ? 4 + L
                            : (KIND == MPE_SCN_SIMPLE || KIND == MPE_SCN_ADVERSARY || KIND == MPE_SCN_PUSH ||
                               KIND == MPE_SCN_SPEAKER_LISTENER || KIND == MPE_SCN_REFERENCE) ? 5
                            : KIND == MPE_SCN_WORLD_COMM ? 6 : 4;
  // World.dim_c of the communication scenarios (make_world of each)
  static constexpr int DC = KIND == MPE_SCN_SPEAKER_LISTENER ? 3 : KIND == MPE_SCN_REFERENCE ? 10
                            : (KIND == MPE_SCN_CRYPTO || KIND == MPE_SCN_WORLD_COMM) ? 4 : 2;
  // per-world picks this kernel reads (MpeBuffers.choice rows)
  static constexpr int NCH = (KIND == MPE_SCN_ADVERSARY || KIND == MPE_SCN_PUSH || KIND == MPE_SCN_SPEAKER_LISTENER) ? 1
                             : (KIND == MPE_SCN_REFERENCE || KIND == MPE_SCN_CRYPTO) ? 2 : 0;
  static constexpr int DMAX = KIND == MPE_SCN_SIMPLE   ? 2 + 2 * L
                              : KIND == MPE_SCN_SPREAD ? 4 + 2 * L + 4 * (A - 1)
                              : KIND == MPE_SCN_TAG    ? 4 + 2 * L + 2 * (A - 1) + 2 * (A - NADV)
                              : KIND == MPE_SCN_ADVERSARY ? 2 + 2 * L + 2 * (A - 1)
                              : KIND == MPE_SCN_PUSH   ? 7 + 5 * L + 2 * (A - 1)
                              : KIND == MPE_SCN_SPEAKER_LISTENER ? 2 + 2 * L + 3
                              : KIND == MPE_SCN_REFERENCE ? 2 + 2 * L + 3 + 10
                              : KIND == MPE_SCN_CRYPTO ? 8
                              : KIND == MPE_SCN_WORLD_COMM ? 4 + 2 * L + 2 * (A - 1) + 2 * (A - NADV) + 2 + 4
                                                       : 1;
  // rows a wave's LDS tile holds: 64, except in simple_world_comm's ROLLOUT (six agent waves, 34-float rows, two exchange
  // blocks: 72 KB per workgroup with 64-row tiles = two workgroups per CU).  With 32-row tiles (the wave's rows leave in
  // two halves) it is 45 KB: three per CU -- measured 10.8 vs 13.4 us per step; the single step is NOT faster that way
  // (19.4-19.7 vs 18.3-18.8 us) and keeps 64 rows.
  static constexpr int trows(bool roll) { return (KIND == MPE_SCN_WORLD_COMM && roll) ? 32 : kWave; }
  static constexpr int tile_floats(bool roll) { return trows(roll) * (DMAX | 1); }  // >= trows * tile_stride<D>() of every row width
  // A agent waves (+ A rows waves in the dual-role rollout) + the reward wave
  static constexpr int waves(bool dual) { return (dual ? 2 * A : A) + 1; }
  // rollout: + the moves of the next step, drawn by the reward wave for all agents (two buffers by step parity)
  static constexpr size_t lds_bytes(bool roll) {
    return sizeof(float) * ((roll ? 2 : 1) * A * XW * kWave + A * tile_floats(roll) + (roll ? 2 * A * kWave : 0));
  }
};

// Which agents speak in the communication scenarios (their make_world: `agent.silent = False`), bit a = agent a.
template <int KIND>
constexpr unsigned speakers_of() {
  return KIND == MPE_SCN_SPEAKER_LISTENER ? 0x1u : KIND == MPE_SCN_REFERENCE ? 0x3u : KIND == MPE_SCN_CRYPTO ? 0x7u
         : KIND == MPE_SCN_WORLD_COMM ? 0x1u : 0u;
}
// The utterance of one agent in one world at this step: a row of MpeBuffers.comm (what the caller passed), or -- in
// the fused rollout -- the one-hot of the word mpe_random_comm draws for (world, step, agent), recomputed where used.
// WM (word mode): 0 = the caller's row, 1 = drawn in the kernel (the rollout), 2 = the caller's row read at SYSTEM scope (the step
// server: another kernel wrote the utterance ring while this one runs -- no cache of ours may serve it)
constexpr int kWordRow = 0, kWordDrawn = 1, kWordRowSys = 2;
template <int WM>
struct Word {
  const float *row;
  int id;
  __device__ __forceinline__ float operator[](int c) const {
    if constexpr (WM == kWordDrawn) return c == id ? 1.f : 0.f;
    else if constexpr (WM == kWordRowSys)
      return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const int *>(row) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    else return row[c];
  }
};
template <int DC, int WM>
__device__ __forceinline__ Word<WM> word_of(const MpeBuffers &b, size_t B, size_t w0, unsigned ln, int j, uint64_t seed,
                                            uint64_t gw, uint64_t gt) {
  Word<WM> wd;
  wd.row = WM == kWordDrawn ? nullptr : b.comm + wave_off(((size_t)j * B + w0) * DC) + ln * DC;
  wd.id = WM == kWordDrawn ? comm_draw(seed, gw, gt, j, DC) : 0;
  return wd;
}

// ---- the reward wave: Scenario.reward / benchmark_data / done for all A agents of 64 worlds --------
// X is the exchange block the agent waves filled: X[(a * XW + c) * 64 + lane], c = 0,1 pos, 2,3 vel, 4.. d2.
// Cache policy of the 4- / 1-byte outputs (state, rewards, dones, counts): agent scope for simple_tag's SMALL launches (the
// kernels built with agent-scope rows), ordinary stores everywhere else.  Measured (profiles/r2_ab_logs.txt session 40): tag at
// 16 384 worlds 3.78 -> 3.47 us per step, 1.29 -> 1.25 per rollout step, at 4096 worlds 3.24 -> 3.18 -- but at 65 536 worlds
// 6.52 -> 6.95, and simple_spread loses at both 4096 (2.98 -> 3.11) and 65 536 worlds (5.55 -> 5.94).
template <int KIND, int RP>
constexpr int aux_policy() { return (KIND == MPE_SCN_TAG && RP == kRowsSc1) ? kRowsSc1 : kRowsPlain; }

template <int KIND, int A, int L, int NADV, bool ROLL, int AUX, int WM = (ROLL ? kWordDrawn : kWordRow)>
__device__ __forceinline__ void reward_wave(const NarrowDesc &d, const float *const sz /* pinned sz[] */, const MpeBuffers &b, const float *X, int lane,
                                            bool live, unsigned ln, size_t B, size_t w0 /* first world of the wave */,
                                            size_t ro /* uniform: row 0 of this step + w0 */, uint64_t seed, uint64_t gw,
                                            uint64_t gt, int goal_roll /* rollout: this world's pick 0 */,
                                            const float (&food_roll)[4] /* rollout, world_comm: food x0 y0 x1 y1 */) {
  constexpr int XW = SplitShape<KIND, A, L, NADV>::XW;
  if (KIND == MPE_SCN_SIMPLE) {  // simple.py:41-43: -|pos - landmark 0|^2
    if (live) {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        if (b.rew) store_aux<AUX>(b.rew + wave_off(ro + (size_t)a * B) + ln, -X[(a * XW + 4) * kWave + lane]);
        if (b.done) store_aux<AUX>(b.done + wave_off(ro + (size_t)a * B) + ln, 0);
      }
    }
  }
  if (KIND == MPE_SCN_SPREAD) {  // simple_spread.py:72-82, :47-63
    if (b.rew || b.info_rew) {
      float px[A], py[A];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        px[a] = X[(a * XW + 0) * kWave + lane];
        py[a] = X[(a * XW + 1) * kWave + lane];
      }
      // landmark term from the published SQUARED agent-landmark distances: min first, then one square
      // root (monotone); contact counts from the new positions
      float lm_term = 0.f, md = 0.f;
      int occupied = 0;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        float m2 = X[(0 * XW + 4 + l) * kWave + lane];
#pragma unroll
        for (int a = 1; a < A; ++a) m2 = fminf(m2, X[(a * XW + 4 + l) * kWave + lane]);
        const float m = fast_sqrt(m2);
        lm_term = lm_term - m;
        md = md + m;
        if (b.info_rew) occupied += sqrt_lt(m2, 0.1f) ? 1 : 0;  // benchmark_data only (uniform); the integer output takes the exact test
      }
      int cnt[A];
#pragma unroll
      for (int a = 0; a < A; ++a) cnt[a] = (sz[a] + sz[a] > 0.f) ? 1 : 0;  // the agent against itself (Q1): 0 < 2r
#pragma unroll
      for (int a = 0; a < A; ++a) {
#pragma unroll
        for (int c = a + 1; c < A; ++c) {
          const bool hit = sqrt_lt(sq2d(px[a] - px[c], py[a] - py[c]), sz[a] + sz[c]);
          cnt[a] += hit ? 1 : 0;
          cnt[c] += hit ? 1 : 0;
        }
      }
      float r[A];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const int c = ((d.collide >> a) & 1u) ? cnt[a] : 0;
        cnt[a] = c;
        float ra_ = lm_term;
#pragma unroll
        for (int s = 0; s < A; ++s) ra_ = ra_ - (c > s ? 1.f : 0.f);  // rew -= 1 per contact, in sequence
        r[a] = ra_;
      }
      // environment.py:100-102: reward = np.sum(reward_n) = r0 + (((0 + r1) + r2) + ...) for n < 9
      float rest = 0.f;
#pragma unroll
      for (int a = 1; a < A; ++a) rest += r[a];
      const float total = A > 1 ? r[0] + rest : r[0];
      if (live) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
          const size_t o = ro + (size_t)a * B;
          if (b.rew) store_aux<AUX>(b.rew + wave_off(o) + ln, d.collaborative ? total : r[a]);
          if (b.info_rew) {
            store_aux<AUX>(b.info_rew + wave_off(o) + ln, r[a]);
            store_aux<AUX>(b.info_collisions + wave_off(o) + ln, cnt[a]);
            store_aux<AUX>(b.info_min_dists + wave_off(o) + ln, md);
            store_aux<AUX>(b.info_occupied + wave_off(o) + ln, occupied);
          }
        }
      }
    }
    if (b.done && live) {
#pragma unroll
      for (int a = 0; a < A; ++a) store_aux<AUX>(b.done + wave_off(ro + (size_t)a * B) + ln, 0);
    }
  }
  if (KIND == MPE_SCN_TAG) {  // simple_tag.py:84-129, :57-66
    constexpr int NG = A - NADV;
    if (b.rew || b.info_collisions) {
      float px[A], py[A];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        px[a] = X[(a * XW + 0) * kWave + lane];
        py[a] = X[(a * XW + 1) * kWave + lane];
      }
      bool hit[NG][NADV];
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int v = 0; v < NADV; ++v)
          hit[g][v] = sqrt_lt(sq2d(px[NADV + g] - px[v], py[NADV + g] - py[v]), sz[NADV + g] + sz[v]);
      float adv_rew = 0.f;  // adversary_reward :115-129 -- same value for every adversary
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int v = 0; v < NADV; ++v) adv_rew += hit[g][v] ? 10.f : 0.f;
      if (live) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
          float r;
          int c = 0;
          if (a < NADV) {
            r = ((d.collide >> a) & 1u) ? adv_rew : 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) c += hit[g][a < NADV ? a : 0] ? 1 : 0;
          } else {  // agent_reward :89-113
            r = 0.f;
            if ((d.collide >> a) & 1u) {
#pragma unroll
              for (int v = 0; v < NADV; ++v) r -= hit[a >= NADV ? a - NADV : 0][v] ? 10.f : 0.f;
            }
            r -= tag_bound(fabsf(px[a]));
            r -= tag_bound(fabsf(py[a]));
          }
          const size_t o = ro + (size_t)a * B;
          if (b.rew) store_aux<AUX>(b.rew + wave_off(o) + ln, r);
          if (b.info_collisions) store_aux<AUX>(b.info_collisions + wave_off(o) + ln, c);  // benchmark_data :57-66
        }
      }
    }
    if (b.done && live) {
#pragma unroll
      for (int a = 0; a < A; ++a) store_aux<AUX>(b.done + wave_off(ro + (size_t)a * B) + ln, 0);
    }
  }
  if (KIND == MPE_SCN_ADVERSARY || KIND == MPE_SCN_PUSH) {
    // each agent published the SQUARED distance to this world's goal landmark (column 4)
    if (live) {
      float d2g[A];
#pragma unroll
      for (int a = 0; a < A; ++a) d2g[a] = X[(a * XW + 4) * kWave + lane];
      float m2 = d2g[NADV];   // the good agent nearest to the goal
#pragma unroll
      for (int a = NADV + 1; a < A; ++a) m2 = fminf(m2, d2g[a]);
      float adv_sum = 0.f;    // simple_adversary.py:88: sum over adversaries of their distance to the goal
#pragma unroll
      for (int a = 0; a < NADV; ++a) adv_sum = adv_sum + fast_sqrt(d2g[a]);
#pragma unroll
      for (int a = 0; a < A; ++a) {
        float r;
        if (KIND == MPE_SCN_ADVERSARY) r = a < NADV ? -d2g[a] : -fast_sqrt(m2) + adv_sum;            // :113 / :100-107
        else                           r = a < NADV ? fast_sqrt(m2) - fast_sqrt(d2g[a]) : -fast_sqrt(d2g[a]);  // simple_push.py:68-76 / :64-66
        const size_t o = ro + (size_t)a * B;
        if (b.rew) store_aux<AUX>(b.rew + wave_off(o) + ln, r);
        if (b.done) store_aux<AUX>(b.done + wave_off(o) + ln, 0);
      }
    }
  }
  if constexpr (KIND == MPE_SCN_SPEAKER_LISTENER) {  // simple_speaker_listener.py:63-67: -|listener - goal landmark|^2 for both
    if (live) {
      const float r = -X[(1 * XW + 4) * kWave + lane];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const size_t o = ro + (size_t)a * B;
        if (b.rew) store_aux<AUX>(b.rew + wave_off(o) + ln, d.collaborative ? r + r : r);
        if (b.done) store_aux<AUX>(b.done + wave_off(o) + ln, 0);
      }
    }
  }
  if constexpr (KIND == MPE_SCN_REFERENCE) {  // simple_reference.py:57-61: agent i wants the OTHER agent at i's goal landmark
    if (live) {
      const float r0 = -X[(1 * XW + 4) * kWave + lane];   // |pos_1 - landmark goal_0|^2, published by agent 1
      const float r1 = -X[(0 * XW + 4) * kWave + lane];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const size_t o = ro + (size_t)a * B;
        if (b.rew) store_aux<AUX>(b.rew + wave_off(o) + ln, d.collaborative ? r0 + r1 : (a == 0 ? r0 : r1));
        if (b.done) store_aux<AUX>(b.done + wave_off(o) + ln, 0);
      }
    }
  }
  if constexpr (KIND == MPE_SCN_CRYPTO) {  // simple_crypto.py:97-124: squared error of what Bob / Eve say against the goal one-hot
    constexpr int DC = SplitShape<KIND, A, L, NADV>::DC;
    if (live) {
      const int g = ROLL ? goal_roll : (b.choice + wave_off(w0))[ln];
      float err[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {   // a = 0: Eve (adversary), a = 1: Bob (good listener)
        const Word<WM> c = word_of<DC, WM>(b, B, w0, ln, a, seed, gw, gt);
        float e = 0.f;
        bool silent = true;
#pragma unroll
        for (int k = 0; k < DC; ++k) {
          const float v = c[k], dv = v - (k == g ? 1.f : 0.f);
          silent = silent && (v == 0.f);
          e = e + dv * dv;
        }
        err[a] = silent ? 0.f : e;   // `continue` on an all-zero utterance (:107, :112, :122)
      }
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const float r = a == 0 ? 0.f - err[0] : (0.f + err[0]) + (0.f - err[1]);
        const size_t o = ro + (size_t)a * B;
        if (b.rew) store_aux<AUX>(b.rew + wave_off(o) + ln, r);
        if (b.done) store_aux<AUX>(b.done + wave_off(o) + ln, 0);
      }
    }
  }
  if constexpr (KIND == MPE_SCN_WORLD_COMM) {  // simple_world_comm.py:143-203
    constexpr int NG = A - NADV;
    float px[A], py[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      px[a] = X[(a * XW + 0) * kWave + lane];
      py[a] = X[(a * XW + 1) * kWave + lane];
    }
    bool hit[NG][NADV];
    float dmin[NADV];   // adversary v: distance to the nearest good agent
#pragma unroll
    for (int v = 0; v < NADV; ++v) dmin[v] = INFINITY;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int v = 0; v < NADV; ++v) {
        const float d2 = sq2d(px[NADV + g] - px[v], py[NADV + g] - py[v]);
        hit[g][v] = sqrt_lt(d2, sz[NADV + g] + sz[v]);
        dmin[v] = fminf(dmin[v], d2);
      }
    // food = landmarks 1, 2 (world.landmarks = [obstacle] + food + forests)
    float fx[2], fy[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {   // (the rollout's in-kernel resets move the landmarks: HBM holds them only at the end)
      fx[f] = ROLL ? food_roll[2 * f] : (b.pos + wave_off((size_t)(2 * (A + 1 + f)) * B + w0))[ln];
      fy[f] = ROLL ? food_roll[2 * f + 1] : (b.pos + wave_off((size_t)(2 * (A + 1 + f) + 1) * B + w0))[ln];
    }
    if (live) {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        float r = 0.f;
        if (a < NADV) {   // adversary_reward :188-203 (shape = True)
          r = r - 0.1f * fast_sqrt(dmin[a < NADV ? a : 0]);
          if ((d.collide >> a) & 1u) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
              for (int v = 0; v < NADV; ++v) r = r + (hit[g][v] ? 5.f : 0.f);
          }
        } else {          // agent_reward :156-186
          if ((d.collide >> a) & 1u) {
#pragma unroll
            for (int v = 0; v < NADV; ++v) r = r - (hit[a >= NADV ? a - NADV : 0][v] ? 5.f : 0.f);
          }
          r = r - 2.f * tag_bound(fabsf(px[a]));
          r = r - 2.f * tag_bound(fabsf(py[a]));
          float m2 = INFINITY;
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const float d2 = sq2d(px[a] - fx[f], py[a] - fy[f]);
            r = r + (sqrt_lt(d2, sz[a] + sz[A + 1 + f]) ? 2.f : 0.f);
            m2 = fminf(m2, d2);
          }
          r = r + 0.05f * fast_sqrt(m2);
        }
        const size_t o = ro + (size_t)a * B;
        if (b.rew) store_aux<AUX>(b.rew + wave_off(o) + ln, r);
        if (b.done) store_aux<AUX>(b.done + wave_off(o) + ln, 0);
        if (b.info_collisions) {   // benchmark_data, simple_world_comm.py:115-124: an adversary's contacts with good agents, 0 for the others
          int c = 0;
          if (a < NADV) {
#pragma unroll
            for (int g = 0; g < NG; ++g) c += hit[g][a < NADV ? a : 0] ? 1 : 0;
          }
          store_aux<AUX>(b.info_collisions + wave_off(o) + ln, c);
        }
      }
    }
  }
}

template <int AUX>
__device__ __forceinline__ void store_state(const MpeBuffers &b, size_t B, int i, size_t w0, unsigned ln, float mx,
                                            float my, float mvx, float mvy) {
  store_aux<AUX>(b.pos + wave_off((size_t)(2 * i) * B + w0) + ln, mx);
  store_aux<AUX>(b.pos + wave_off((size_t)(2 * i + 1) * B + w0) + ln, my);
  store_aux<AUX>(b.vel + wave_off((size_t)(2 * i) * B + w0) + ln, mvx);
  store_aux<AUX>(b.vel + wave_off((size_t)(2 * i + 1) * B + w0) + ln, mvy);
}

// ---- the step server (SERVE: a ROLL instantiation that is COMMANDED step by step) -----------------------------------------
// The per-step launch is the reference's contract (one env.step per policy decision, environment.py:80-104), and a dependent
// launch costs 1.0-2.0 us of launch-to-launch gap on top of the kernel's span (profiles/r6_device_span_*.txt).  The server is
// ONE resident launch that executes the SAME steps on command: step g (global step number) runs when the doorbell word says
// `door > g` -- rung (door += n) by mpe_step_server_ring, a one-thread launch ordered on the CALLER's stream behind whatever produced
// that step's moves -- reads its one-hot moves from tensor g % ring of the caller's move ring with system-scope loads
// (another kernel wrote them while this one runs: no cache of ours may serve them), writes that step's rows / rewards /
// dones / state WRITE-THROUGH (sc1: visible to other kernels when acknowledged, not at kernel end) into output block
// g % slots, and publishes per workgroup `flag[wg] = g + 1` once every wave's stores of step g are acknowledged.  State
// stays in registers between steps and episodes restart inside the launch (the rollout's in-kernel reset), so a launch
// serves up to T steps; a wave that waits longer than `timeout_ticks` of the 100 MHz wall clock for its next command sets
// `status` and leaves (a server never outlives its commander).  Same device functions in the same order as the per-step
// kernel: bit-identical to T x {mpe_reset at the boundaries; mpe_step} (tests/test_gpu_server.py).
//
// Completion without a second barrier: the stores of step g are known acknowledged where the wave next waits for memory
// anyway -- `s_waitcnt vmcnt(0)` in front of step g+1's barrier, behind which the reward wave publishes g+1.  A server that
// is AHEAD of its commander (closed loop: the next doorbell rings only after the consumer saw this step) must not wait for
// that barrier: a wave that finds its next doorbell unrung drains its stores and counts itself in LDS; the reward wave,
// spinning on the same doorbell, publishes as soon as all agent waves are counted.
struct ServeArgs {
  unsigned long long *door;     // commanded steps (absolute count): step g may run when *door > g
  unsigned long long *flag;     // [grid] completed steps per workgroup (absolute count)
  unsigned int *status;         // != 0: a wave gave up waiting (timeout)
  const float *act_ring;        // `ring` consecutive [A][B][5] move tensors; step g reads tensor g % ring
  const float *comm_ring;       // communication scenarios: `ring` consecutive [A][B][dim_c] utterance tensors, likewise
  int32_t ring, slots;          // output block of step g: g % slots (blocks of obs_off[A] * B floats / A * B entries)
  unsigned long long timeout_ticks;
};
__device__ __forceinline__ unsigned long long door_load(const unsigned long long *p) {
  // SYSTEM scope (sc0 sc1): an agent-scope (sc1) load is served by this XCD's L2, which keeps the line it fetched while the
  // word still held the old count -- at 4096 worlds (a working set that never evicts it) a served step took 33 us that way
  const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int KIND, int A, int L, int NADV, bool ROLL, int RP /* row-store policy: kRowsNt / kRowsSc1 (mpe_device.h) */,
          bool DUALP = false /* the dual-role rollout: physics + rows waves per agent */,
          bool SERVE = false /* the step server: ROLL commanded step by step (above) */>
__global__ void __launch_bounds__((SplitShape<KIND, A, L, NADV>::waves(DUALP) * kWave))
k_split(float *const g_pos, float *const g_vel, const float *const g_act, const int32_t *const g_ids, const size_t B,
        const int g_wpw, const int g_observe_only, const unsigned g_movable, const NarrowDesc d, const MpeBuffers b_in,
        const RollArgs ra, const ServeArgs sv) {
  static_assert(!SERVE || (ROLL && RP == kRowsSc1), "the step server: a rollout instantiation, write-through stores");
  // the agents' utterances: the caller's rows (a launched step), drawn in the kernel (the rollout), or the caller's rows of THIS step
  // out of the utterance ring, read at system scope (the step server)
  constexpr int WM = SERVE ? kWordRowSys : (ROLL ? kWordDrawn : kWordRow);
  // The leading scalar arguments (13 dwords) repeat what the first global loads of a wave need -- the state and
  // action pointers, the batch size, the worlds per workgroup, the movable mask -- so that the CP can PRELOAD them
  // into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count, _build.py): the wave's loads leave without a
  // scalar round trip to the kernarg segment in front of them (three dependent ones before: the sizes, the
  // MpeBuffers block, the per-agent constants).  Everything else still comes from the structs behind them.
  using S = SplitShape<KIND, A, L, NADV>;
  MPE_SPAN_BEGIN();
  MpeBuffers b = b_in;
  b.pos = g_pos;
  b.vel = g_vel;
  b.act = g_act;
  b.ids = g_ids;
  constexpr int E = A + L, XW = S::XW;
  constexpr int AUX = SERVE ? kRowsSc1 : aux_policy<KIND, RP>();   // (a served step's state / rewards / dones leave write-through)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (kWave - 1);
  // uniform role of this wave: [0, A) the agent waves (in the dual-role rollout: the PHYSICS waves), then -- dual-role
  // rollout only -- [A, 2A) the ROWS waves, and last the reward wave
  constexpr bool DUAL = DUALP && ROLL;
  constexpr bool OWN_DRAW = DUAL && dual_own_draw<KIND>();   // the physics waves draw their own moves
  constexpr int NAW = DUAL ? 2 * A : A;   // agent-side waves
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool is_agent = role < NAW;
  const bool is_rows = DUAL && role >= A && role < NAW;
  const int i = is_agent ? (role >= A ? role - A : role) : 0;
  // worlds of this workgroup: ra.wpw (64, 32 or 16) consecutive ones, lane = world; with fewer than 64 the upper
  // lanes idle, which buys more workgroups -- more waves per SIMD to hide latency -- when the batch is small
  const int wpw = g_wpw;
  const size_t w0 = (size_t)blockIdx.x * (size_t)wpw;
  if (w0 >= B) return;  // workgroup-uniform
#ifdef MPE_STRESS_DELAY_WAVE
  // test build only (libmpe_hip_stress.so, tests/test_gpu_race.py): hold one agent wave back for ~30 us before
  // its first load, so that every sibling wave has long finished its World.step when this one starts -- any
  // ordering bug between the state stores and the sibling loads then shows as a wrong contact force
  if (role == (MPE_STRESS_DELAY_WAVE) % A)
    for (int k = 0; k < 10; ++k) __builtin_amdgcn_s_sleep(127);   // 10 x 8128 clocks
#endif
  const int nvalid = (B - w0) < (size_t)wpw ? (int)(B - w0) : wpw;
  const bool live = lane < nvalid;
  // dead lanes of a ragged last wave shadow its last live world (their stores are masked).  Every
  // global address below is "wave-uniform base (SGPRs) + ln": scalar-base addressing, no per-lane
  // 64-bit arithmetic.
  const unsigned ln = (unsigned)(live ? lane : nvalid - 1) & 63u;
  const size_t w = w0 + (size_t)ln;
  float *const xch = smem;
  float *const tile = smem + (ROLL ? 2 : 1) * A * XW * kWave + i * S::tile_floats(ROLL);
  // rollout: mv[(t & 1)][a][lane] = the move of agent a at step t.  The agent waves of a world would each run the
  // SAME Philox block (one block serves four agents) at the head of their per-step dependency chain; instead the
  // reward wave, which idles until the barrier, draws the NEXT step's moves for everybody while the agents step.
  int *const mv = reinterpret_cast<int *>(smem + (ROLL ? 2 : 1) * A * XW * kWave + A * S::tile_floats(ROLL));

  const int T = ROLL ? ra.T : 1;
  const size_t obs_stride = (ra.trajectory || SERVE) ? (size_t)d.obs_off[A] * B : 0;
  const size_t row_stride = (ra.trajectory || SERVE) ? (size_t)A * B : 0;
  // served steps: output block g % slots and move tensor g % ring of global step g, as counters (a 64-bit modulo per step
  // costs ~130 instructions); `seen` = this wave's last look at the doorbell; cnt[parity] = agent waves drained while idle
  int blk0 = 0, mvt0 = 0;
  unsigned long long seen = 0;
  int *const idle_cnt = reinterpret_cast<int *>(smem + SplitShape<KIND, A, L, NADV>::lds_bytes(true) / sizeof(float));
  if constexpr (SERVE) {
    blk0 = (int)(ra.step0 % (uint64_t)sv.slots);
    mvt0 = (int)(ra.step0 % (uint64_t)sv.ring);
    if (threadIdx.x < 2) idle_cnt[threadIdx.x] = 0;
    __syncthreads();
  }
  auto block_of = [&](const int t) { return SERVE ? (blk0 + t) % sv.slots : t; };   // (32-bit; t < T)
  auto with_words = [&](const int t) {      // the buffers with `comm` = the utterances of step t (served: tensor g mod ring)
    MpeBuffers q = b;
    if constexpr (SERVE)
      q.comm = sv.comm_ring ? sv.comm_ring + (size_t)((mvt0 + t) % sv.ring) * ((size_t)A * B * SplitShape<KIND, A, L, NADV>::DC) : nullptr;
    return q;
  };
  // wait_door: until step g = step0 + t is commanded.  -> false: gave up (timeout): the wave leaves the kernel.
  // `publisher`: the reward wave -- while idle it publishes the previous step as soon as every agent wave has drained.
  auto wait_door = [&](const int t, const bool publisher) -> bool {
    if constexpr (!SERVE) return true;
    const unsigned long long g = ra.step0 + (unsigned long long)t;
    if (seen > g) return true;
    seen = door_load(sv.door);
    if (seen > g) return true;
    // ahead of the commander: this wave's stores of the steps so far are acknowledged before anybody is told so
    drain_stores();
    bool told = t == 0;      // (nothing of this launch to publish before its first step)
    if (!publisher && !told && lane == 0) __hip_atomic_fetch_add(&idle_cnt[t & 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long t_begin = (unsigned long long)wall_clock64();
    for (;;) {
      if (publisher && !told) {
        const int n = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&idle_cnt[t & 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (n >= NAW) {   // every agent-side wave drained: steps < g of this workgroup are complete
          if (lane == 0) __hip_atomic_store(sv.flag + blockIdx.x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          told = true;
        }
      }
      __builtin_amdgcn_s_sleep(4);
      seen = door_load(sv.door);
      if (seen > g) return true;
      if ((unsigned long long)wall_clock64() - t_begin > sv.timeout_ticks) {
        if (lane == 0) __hip_atomic_store(sv.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    }
  };

  if (!is_agent) {
    // ---- the reward wave --------------------------------------------------------------------------
    // (what the reward needs besides the agents' published state follows the in-kernel resets of the rollout with the
    //  same countdown the agent waves run: simple_crypto's goal pick, simple_world_comm's food positions)
    constexpr bool TRACK = ROLL && (KIND == MPE_SCN_CRYPTO || KIND == MPE_SCN_WORLD_COMM);
    const uint64_t gw_r = ra.world_offset + w;
    int goal_r = (ROLL && KIND == MPE_SCN_CRYPTO) ? (b.choice + wave_off(w0))[ln] : 0;
    float food[4] = {0.f, 0.f, 0.f, 0.f};
    if (ROLL && KIND == MPE_SCN_WORLD_COMM) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        food[2 * f] = (b.pos + wave_off((size_t)(2 * (A + 1 + f)) * B + w0))[ln];
        food[2 * f + 1] = (b.pos + wave_off((size_t)(2 * (A + 1 + f) + 1) * B + w0))[ln];
      }
    }
    float sz[E];   // every entity's size, in SGPRs before the barrier (the reward's strict-< tests sit behind it)
#pragma unroll
    for (int e = 0; e < E; ++e) sz[e] = d.size[e];
    pin_s(sz);
    int cd = -1;
    uint64_t ep_r = 0;
    if (TRACK && ra.episode_len > 0) {
      const uint64_t len = (uint64_t)ra.episode_len, r = ra.step0 % len;
      cd = r == 0 ? 0 : (int)(len - r);
      ep_r = ra.step0 / len + (r ? 1 : 0);
    }
    for (int t = 0; t < T; ++t) {
      MPE_STAMP(0);
      if (SERVE && !wait_door(t, true)) return;
      if (TRACK && cd >= 0) {
        if (cd == 0) {
          if (KIND == MPE_SCN_CRYPTO) goal_r = choice_draw(ra.seed, gw_r, ep_r, 0, d.choice_pop[0]);
          if (KIND == MPE_SCN_WORLD_COMM) {
#pragma unroll
            for (int f = 0; f < 2; ++f) reset_draw(ra.seed, gw_r, ep_r, A + 1 + f, ra.landmark_range, food[2 * f], food[2 * f + 1]);
          }
          ++ep_r;
          cd = ra.episode_len - 1;
        } else {
          --cd;
        }
      }
      if (ROLL && !SERVE && !OWN_DRAW && t + 1 < T) {   // the moves of step t + 1: read by the agent waves behind this step's barrier
        const uint64_t gt1 = ra.step0 + (uint64_t)t + 1;
#pragma unroll
        for (int q = 0; q < (A + 3) / 4; ++q) {
          const U4 o = action_block(ra.seed, gw_r, gt1, q);
          const uint32_t word[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (4 * q + k < A) mv[(((t + 1) & 1) * A + 4 * q + k) * kWave + lane] = move_of(word[k]);
        }
      }
      const float *const X = xch + (ROLL ? (t & 1) * A * XW * kWave : 0);
      MPE_STAMP(1);
      if (SERVE) drain_stores();   // this wave's outputs of step t - 1 are acknowledged ...
      __syncthreads();
      if (SERVE) {                 // ... and so are every agent wave's (they drain in front of this barrier too): step t - 1 is complete
        if (t > 0 && lane == 0) __hip_atomic_store(sv.flag + blockIdx.x, ra.step0 + (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (lane == 0) idle_cnt[t & 1] = 0;      // (next used by step t + 2's waiters, who are behind step t + 1's barrier)
      }
      MPE_STAMP(2);
      if (!(MPE_SPLIT_ABLATE & 2))
      reward_wave<KIND, A, L, NADV, ROLL, AUX, WM>(d, sz, with_words(t), X, lane, live, ln, B, w0, (size_t)block_of(t) * row_stride + w0,
                                                   ra.seed, gw_r, ra.step0 + (uint64_t)t, goal_r, food);
      MPE_STAMP(3);
    }
    if (SERVE) {   // the last step: every wave drains, then the workgroup says so
      drain_stores();
      __syncthreads();
      if (lane == 0) __hip_atomic_store(sv.flag + blockIdx.x, ra.step0 + (unsigned long long)T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    MPE_SPAN_END();
    return;
  }

  // ---- agent wave i ---------------------------------------------------------------------------------
  // The loads of the world go first: their addresses come from the preloaded arguments alone, so they are in flight
  // while the scalar loads of this agent's constants (below) and of the remaining buffer pointers are still outstanding.
  float px[E], py[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    px[e] = (b.pos + wave_off((size_t)(2 * e) * B + w0))[ln];
    py[e] = (b.pos + wave_off((size_t)(2 * e + 1) * B + w0))[ln];
  }
  float mvx = (b.vel + wave_off((size_t)(2 * i) * B + w0))[ln];
  float mvy = (b.vel + wave_off((size_t)(2 * i + 1) * B + w0))[ln];
  const bool movable_i = (g_movable >> i) & 1u;
  const bool step_world = ROLL || !g_observe_only;   // mpe_observe: outputs of the current state only
  // the move of this step as the caller handed it over: the raw one-hot row differences / the id, scaled further down
  // (a one-hot row and an id are both decoded to an exact -1 / 0 / +1 pair, so "decode, then scale" is the same product)
  float ux0 = 0.f, uy0 = 0.f;
  if (!ROLL && step_world && movable_i && (b.act || b.ids)) fetch_action_wave(b, B, i, w0, ln, 1.0f, ux0, uy0);

  // this agent's constants: one scalar load each at a wave-uniform index, and every entity's size in the same batch
  // (pinned: left to itself the compiler sinks each d.size[j] load into its iteration of the contact loop, a scalar
  //  round trip apiece on the step's critical path)
  float kc[9 + E];
  kc[0] = d.size[i]; kc[1] = d.inv_mass[i]; kc[2] = d.accel[i]; kc[3] = d.max_speed[i];
  kc[4] = d.dt; kc[5] = d.damp; kc[6] = d.cforce; kc[7] = d.cmargin; kc[8] = d.cmargin_inv;
#pragma unroll
  for (int e = 0; e < E; ++e) kc[9 + e] = d.size[e];
  pin_s(kc);
  const float size_i = kc[0], mass_i = kc[1], accel_i = kc[2], maxspd_i = kc[3];
  const float k_dt = kc[4], k_damp = kc[5], k_cforce = kc[6], k_cmargin = kc[7], k_cmargin_inv = kc[8];
  const float *const size_e = kc + 9;
  const int obs_off_i = d.obs_off[i];
  const bool collide_i = (d.collide >> i) & 1u;
  float mx = 0.f, my = 0.f;
#pragma unroll
  for (int a = 0; a < A; ++a)
    if (a == i) { mx = px[a]; my = py[a]; }

  const uint64_t gw = ra.world_offset + w;  // global world number (RNG streams)
  constexpr bool HAS_GOAL = KIND == MPE_SCN_ADVERSARY || KIND == MPE_SCN_PUSH;
  constexpr int NCH = S::NCH, DC = S::DC;
  // this world's picks of reset_world (np.random.choice: goal landmark, key ...)
  int goal = NCH >= 1 ? (b.choice + wave_off(w0))[ln] : 0;
  int pick1 = NCH >= 2 ? (b.choice + wave_off(B + w0))[ln] : 0;

  // resets fall on global steps that are multiples of episode_len: one 64-bit divide up front, then a
  // countdown (a per-step 64-bit modulo costs ~130 instructions on this ISA)
  int countdown = -1;
  uint64_t ep = 0;
  if (ROLL && ra.episode_len > 0) {
    const uint64_t len = (uint64_t)ra.episode_len, r = ra.step0 % len;
    countdown = r == 0 ? 0 : (int)(len - r);
    ep = ra.step0 / len + (r ? 1 : 0);
  }

  // ---- one step of agent i, in the pieces the two loop shapes below are made of ---------------------------------
  float gx = 0.f, gy = 0.f;   // this world's goal landmark (publish -> rows)
  [[maybe_unused]] bool did_reset = false;   // served steps: an in-launch reset moved the landmarks / picks -> behind_barrier stores them
  // step_forces: [in-kernel reset,] this step's move, the action force and the contacts with every other entity in
  // ascending order (Q9) -- everything of World.step (core.py:117-155) up to the force on agent i
  auto step_forces = [&](const int t, float &fx, float &fy) {
    float ux, uy;
    const uint64_t gt = ra.step0 + (uint64_t)t;   // global step: indexes the move / word streams of the rollout
    if (ROLL) {
      const bool reset_now = countdown == 0;
      if (countdown >= 0) countdown = reset_now ? ra.episode_len - 1 : countdown - 1;
      if (reset_now) {  // reset_world, as mpe_reset does it for episode ep = gt / episode_len
#pragma unroll
        for (int e = 0; e < E; ++e) reset_draw(ra.seed, gw, ep, e, e < A ? 1.0f : ra.landmark_range, px[e], py[e]);
#pragma unroll
        for (int a = 0; a < A; ++a)
          if (a == i) { mx = px[a]; my = py[a]; }
        mvx = 0.f;
        mvy = 0.f;
        if (NCH >= 1) goal = choice_draw(ra.seed, gw, ep, 0, d.choice_pop[0]);
        if (NCH >= 2) pick1 = choice_draw(ra.seed, gw, ep, 1, d.choice_pop[1]);
        ++ep;
        if (SERVE) did_reset = true;
      }
      // the one-hot row mpe_random_actions would write: drawn here at the first step, by the reward wave afterwards
      // (dual-role rollout of the kinds in MPE_SPLIT_DUAL_OWN_DRAW: always drawn here)
      if constexpr (SERVE) {
        // the caller's one-hot row of this step (environment.py:174-181: u = (a1 - a2, a3 - a4) * sensitivity), floats 1..4 of a
        // 20-byte row as one 16-byte SYSTEM-scope load (sc0 sc1: the tensor was written by another kernel while this one runs)
        const float *const mt = sv.act_ring + (size_t)((mvt0 + t) % sv.ring) * ((size_t)A * B * MPE_ACTION_DIM);
        // (inline asm: __builtin_amdgcn_raw_buffer_load_b128(..., aux = 17) compiles to a ONE-dword load on this toolchain; the
        //  wait is part of the statement -- the compiler does not count asm loads -- and nothing of this wave's step can start
        //  before its move anyway)
        typedef float vf4 __attribute__((ext_vector_type(4)));
        const float *const mp = mt + wave_off(((size_t)i * B + w0) * MPE_ACTION_DIM) + (ln * MPE_ACTION_DIM + 1);
        vf4 m4;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(m4) : "v"(mp) : "memory");
        if (movable_i) {
          ux = (m4.x - m4.y) * accel_i;
          uy = (m4.z - m4.w) * accel_i;
        } else {
          ux = 0.f;
          uy = 0.f;
        }
      } else {
      const int m = (t == 0 || OWN_DRAW) ? action_draw(ra.seed, gw, gt, i) : mv[((t & 1) * A + i) * kWave + lane];
      ux = ((m == 1 ? 1.f : 0.f) - (m == 2 ? 1.f : 0.f)) * accel_i;
      uy = ((m == 3 ? 1.f : 0.f) - (m == 4 ? 1.f : 0.f)) * accel_i;
      }
    } else if (step_world && movable_i) {
      if (b.act || b.ids) { ux = ux0 * accel_i; uy = uy0 * accel_i; }
      else fetch_action_wave(b, B, i, w0, ln, accel_i, ux, uy);   // pre-decoded Action.u (mpe_world_step's callers)
    } else {
      ux = 0.f;
      uy = 0.f;
    }
    fx = ux + 0.f;
    fy = uy + 0.f;
    if (movable_i && step_world && collide_i && !(MPE_SPLIT_ABLATE & 1)) {
#pragma unroll
      for (int j = 0; j < E; ++j) {
        if (j == i) continue;                        // uniform
        if (!((d.collide >> j) & 1u)) continue;      // uniform
        float cx, cy;
        contact_force(mx - px[j], my - py[j], size_i + size_e[j], k_cforce, k_cmargin, k_cmargin_inv, cx, cy);
        fx = cx + fx;
        fy = cy + fy;
      }
    }
  };
  // step_integrate: integrate_state for agent i (core.py:158-169)
  auto step_integrate = [&](const int t, const float fx, const float fy) {
    if (movable_i && step_world) {
      integrate_one(mx, my, mvx, mvy, fx, fy, mass_i, maxspd_i, k_damp, k_dt);
#ifdef MPE_STRESS_STORE_BEFORE_BARRIER   // negative control of tests/test_gpu_race.py: the ordering that races
      if (live && (!ROLL || t == T - 1)) store_state<AUX>(b, B, i, w0, ln, mx, my, mvx, mvy);
#endif
    }
#ifdef MPE_PHASE_CLOCK
    asm volatile("" ::"v"(mx), "v"(my), "v"(mvx), "v"(mvy));   // World.step has finished here
#endif
  };
  // publish: this agent's new state (and what the reward wave needs of it) into the exchange block of step t's parity
  auto publish = [&](const int t) {
    float *const X = xch + (ROLL ? (t & 1) * A * XW * kWave : 0);
    X[(i * XW + 0) * kWave + lane] = mx;
    X[(i * XW + 1) * kWave + lane] = my;
    X[(i * XW + 2) * kWave + lane] = mvx;
    X[(i * XW + 3) * kWave + lane] = mvy;
    if (KIND == MPE_SCN_SPREAD) {
#pragma unroll
      for (int l = 0; l < L; ++l) X[(i * XW + 4 + l) * kWave + lane] = sq2d(mx - px[A + l], my - py[A + l]);
    }
    if (KIND == MPE_SCN_SIMPLE) X[(i * XW + 4) * kWave + lane] = sq2d(mx - px[A], my - py[A]);
    if (HAS_GOAL || KIND == MPE_SCN_SPEAKER_LISTENER) {
      goal_pos<A, L>(px, py, goal, gx, gy);
      X[(i * XW + 4) * kWave + lane] = sq2d(mx - gx, my - gy);
    }
    if constexpr (KIND == MPE_SCN_REFERENCE) {   // what the OTHER agent's reward needs: my distance to its goal landmark
      goal_pos<A, L>(px, py, i == 0 ? pick1 : goal, gx, gy);
      X[(i * XW + 4) * kWave + lane] = sq2d(mx - gx, my - gy);
    }
    if constexpr (KIND == MPE_SCN_WORLD_COMM) {  // am I inside forest 0 / forest 1 (landmarks 3, 4)?  strict dist < size + size
#pragma unroll
      for (int f = 0; f < 2; ++f)
        X[(i * XW + 4 + f) * kWave + lane] =
            sqrt_lt(sq2d(mx - px[A + 3 + f], my - py[A + 3 + f]), size_i + size_e[A + 3 + f]) ? 1.f : -1.f;
    }
  };
  // behind_barrier: the state store and the siblings' new positions
  auto behind_barrier = [&](const int t) {
    // The new state goes back to HBM only BEHIND the barrier: every sibling wave loaded this agent's
    // pre-step position at kernel entry and consumed it in its contact loop, which lies before its own
    // arrival at this barrier -- so no wave can observe a post-step position in World.step, whatever
    // the dispatch order or timing of the waves (core.py:117-131: forces from the pre-step positions).
#ifndef MPE_STRESS_STORE_BEFORE_BARRIER
    if (movable_i && step_world && live && (!ROLL || SERVE || t == T - 1) && !(MPE_SPLIT_ABLATE & 8))
      store_state<AUX>(b, B, i, w0, ln, mx, my, mvx, mvy);   // (a served step leaves its state in HBM like a launched one)
#endif
    if constexpr (SERVE) {
      if (did_reset && live) {   // what mpe_reset would have left in HBM in front of this step: landmarks, picks, an immovable agent
#pragma unroll
        for (int l = 0; l < L; ++l)
          if (l % A == i) {
            store_aux<AUX>(b.pos + wave_off((size_t)(2 * (A + l)) * B + w0) + ln, px[A + l]);
            store_aux<AUX>(b.pos + wave_off((size_t)(2 * (A + l) + 1) * B + w0) + ln, py[A + l]);
          }
        if (NCH >= 1 && i == 0) {
          store_aux<AUX>(b.choice + wave_off(w0) + ln, goal);
          if (NCH >= 2) store_aux<AUX>(b.choice + wave_off(B + w0) + ln, pick1);
        }
        if (!movable_i) store_state<AUX>(b, B, i, w0, ln, mx, my, mvx, mvy);
      }
      did_reset = false;
    }
    if constexpr (KIND == MPE_SCN_ADVERSARY && !ROLL) {
      // benchmark_data (simple_adversary.py:57-67): the squared distance to the goal landmark (an adversary's datum, the
      // last of a good agent's) and to every landmark (a good agent's first L) -- each agent wave writes its own
      if (live && b.info_rew) {
        store_aux<AUX>(b.info_rew + wave_off((size_t)i * B + w0) + ln, sq2d(mx - gx, my - gy));
        if (b.info_min_dists) {
#pragma unroll
          for (int l = 0; l < L; ++l)
            store_aux<AUX>(b.info_min_dists + wave_off(((size_t)l * A + i) * B + w0) + ln, sq2d(mx - px[A + l], my - py[A + l]));
        }
      }
    }
    const float *const X = xch + (ROLL ? (t & 1) * A * XW * kWave : 0);
#pragma unroll
    for (int a = 0; a < A; ++a) {
      if (a == i) { px[a] = mx; py[a] = my; continue; }  // uniform
      px[a] = X[(a * XW + 0) * kWave + lane];
      py[a] = X[(a * XW + 1) * kWave + lane];
    }
#ifdef MPE_PHASE_CLOCK
#pragma unroll
    for (int a = 0; a < A; ++a) asm volatile("" ::"v"(px[a]), "v"(py[a]));   // the siblings' positions have arrived
#endif
  };
  // rows: agent i's observation row of step t -- the 64 rows of the wave assembled in its LDS tile, streamed out
  auto rows = [&](const int t) {
    if (MPE_SPLIT_ABLATE & 4) return;   // (no observation rows)
    const float *const X = xch + (ROLL ? (t & 1) * A * XW * kWave : 0);
    const uint64_t gt = ra.step0 + (uint64_t)t;   // global step (the word stream of the rollout)
    float *const obs_t = b.obs + (size_t)block_of(t) * obs_stride;
    [[maybe_unused]] const MpeBuffers bw = with_words(t);
    if (KIND == MPE_SCN_SIMPLE) {  // simple.py:45-50
      constexpr int D = 2 + 2 * L;
      {
        RowPairs<D> r(tile, lane);
        r.put(0, mvx, mvy);
#pragma unroll
        for (int l = 0; l < L; ++l) r.put(2 + 2 * l, px[A + l] - mx, py[A + l] - my);
      }
      flush_rows<D, true, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
    }
    if (KIND == MPE_SCN_SPREAD) {  // simple_spread.py:84-100
      constexpr int D = 4 + 2 * L + 4 * (A - 1);
      {
        RowPairs<D> r(tile, lane);
        r.put(0, mvx, mvy);
        r.put(2, mx, my);
#pragma unroll
        for (int l = 0; l < L; ++l) r.put(4 + 2 * l, px[A + l] - mx, py[A + l] - my);
        int k = 4 + 2 * L;  // uniform running column
#pragma unroll
        for (int j = 0; j < A; ++j) {
          if (j == i) continue;
          r.put(k, px[j] - mx, py[j] - my);
          k += 2;
        }
#pragma unroll
        for (int z = 0; z < 2 * (A - 1); z += 2) r.put(k + z, 0.f, 0.f);  // silent agents' comm
      }
      flush_rows<D, true, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
    }
    if (KIND == MPE_SCN_TAG) {  // simple_tag.py:131-147
      constexpr int NG = A - NADV;
      constexpr int DA = 4 + 2 * L + 2 * (A - 1) + 2 * NG, DG = DA - 2;
      const bool adv = i < NADV;
      auto row = [&](auto dsel) {  // one observation row of width D (adversaries DA, good agents DG)
        constexpr int D = decltype(dsel)::value;
        {
          RowPairs<D> r(tile, lane);
          r.put(0, mvx, mvy);
          r.put(2, mx, my);
#pragma unroll
          for (int l = 0; l < L; ++l) r.put(4 + 2 * l, px[A + l] - mx, py[A + l] - my);
          int k = 4 + 2 * L;
#pragma unroll
          for (int j = 0; j < A; ++j) {
            if (j == i) continue;
            r.put(k, px[j] - mx, py[j] - my);
            k += 2;
          }
#pragma unroll
          for (int j = NADV; j < A; ++j) {
            if (j == i) continue;
            r.put(k, X[(j * XW + 2) * kWave + lane], X[(j * XW + 3) * kWave + lane]);
            k += 2;
          }
        }
        flush_rows<D, true, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
      };
      if (adv) row(std::integral_constant<int, DA>{});
      else     row(std::integral_constant<int, DG>{});
    }
    if (KIND == MPE_SCN_ADVERSARY) {  // simple_adversary.py:121-139
      constexpr int DG = 2 + 2 * L + 2 * (A - 1), DA = DG - 2;
      const bool adv = i < NADV;
      auto row = [&](auto dsel, auto good) {
        constexpr int D = decltype(dsel)::value;
        {
          RowPairs<D> r(tile, lane);
          int k = 0;
          if (decltype(good)::value) { r.put(k, gx - mx, gy - my); k += 2; }
#pragma unroll
          for (int l = 0; l < L; ++l) { r.put(k, px[A + l] - mx, py[A + l] - my); k += 2; }
#pragma unroll
          for (int j = 0; j < A; ++j) {
            if (j == i) continue;
            r.put(k, px[j] - mx, py[j] - my);
            k += 2;
          }
        }
        flush_rows<D, true, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
      };
      if (adv) row(std::integral_constant<int, DA>{}, std::false_type{});
      else     row(std::integral_constant<int, DG>{}, std::true_type{});
    }
    if constexpr (KIND == MPE_SCN_PUSH) {  // simple_push.py:78-96
      constexpr int DG = 7 + 5 * L + 2 * (A - 1), DA = 2 + 2 * L + 2 * (A - 1);
      const bool adv = i < NADV;
      auto row = [&](auto dsel, auto good) {
        constexpr int D = decltype(dsel)::value, RS = tile_stride<D>();
        constexpr bool GOOD = decltype(good)::value;
        int k = 0;
        if (GOOD || (RS & 1)) { put1<RS>(tile, lane, k, mvx); put1<RS>(tile, lane, k + 1, mvy); }
        else put2<RS>(tile, lane, k, mvx, mvy);
        k += 2;
        if (GOOD) {   // goal, own colour (0.25 grey, +0.5 on channel goal+1), later the landmark colours
          put1<RS>(tile, lane, k, gx - mx); put1<RS>(tile, lane, k + 1, gy - my);
          k += 2;
#pragma unroll
          for (int c = 0; c < 3; ++c) put1<RS>(tile, lane, k + c, (goal + 1 == c) ? 0.75f : 0.25f);
          k += 3;
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
          if ((RS & 1) || (k & 1)) { put1<RS>(tile, lane, k, px[A + l] - mx); put1<RS>(tile, lane, k + 1, py[A + l] - my); }
          else put2<RS>(tile, lane, k, px[A + l] - mx, py[A + l] - my);
          k += 2;
        }
        if (GOOD) {
#pragma unroll
          for (int l = 0; l < L; ++l)
#pragma unroll
            for (int c = 0; c < 3; ++c) put1<RS>(tile, lane, k + 3 * l + c, (l + 1 == c) ? (float)(0.1 + 0.8) : 0.1f);
          k += 3 * L;
        }
#pragma unroll
        for (int j = 0; j < A; ++j) {
          if (j == i) continue;
          if ((RS & 1) || (k & 1)) { put1<RS>(tile, lane, k, px[j] - mx); put1<RS>(tile, lane, k + 1, py[j] - my); }
          else put2<RS>(tile, lane, k, px[j] - mx, py[j] - my);
          k += 2;
        }
        flush_rows<D, false, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
      };
      if (adv) row(std::integral_constant<int, DA>{}, std::false_type{});
      else     row(std::integral_constant<int, DG>{}, std::true_type{});
    }
    if constexpr (KIND == MPE_SCN_SPEAKER_LISTENER) {  // simple_speaker_listener.py:69-92
      if (i == 0) {   // speaker: the goal landmark's colour (0.65 on channel goal, 0.15 elsewhere)
        constexpr int D = 3, RS = tile_stride<D>();
#pragma unroll
        for (int c = 0; c < 3; ++c) put1<RS>(tile, lane, c, goal == c ? 0.65f : 0.15f);
        flush_rows<D, false, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
      } else {        // listener: vel, landmarks, what the speaker says
        constexpr int D = 2 + 2 * L + DC, RS = tile_stride<D>();
        put1<RS>(tile, lane, 0, mvx); put1<RS>(tile, lane, 1, mvy);
#pragma unroll
        for (int l = 0; l < L; ++l) { put1<RS>(tile, lane, 2 + 2 * l, px[A + l] - mx); put1<RS>(tile, lane, 3 + 2 * l, py[A + l] - my); }
        const Word<WM> c0 = word_of<DC, WM>(bw, B, w0, ln, 0, ra.seed, gw, gt);
#pragma unroll
        for (int c = 0; c < DC; ++c) put1<RS>(tile, lane, 2 + 2 * L + c, c0[c]);
        flush_rows<D, false, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
      }
    }
    if constexpr (KIND == MPE_SCN_REFERENCE) {  // simple_reference.py:63-83
      constexpr int D = 2 + 2 * L + 3 + DC, RS = tile_stride<D>();
      const int mine = i == 0 ? goal : pick1;   // agent.goal_b
      put1<RS>(tile, lane, 0, mvx); put1<RS>(tile, lane, 1, mvy);
#pragma unroll
      for (int l = 0; l < L; ++l) { put1<RS>(tile, lane, 2 + 2 * l, px[A + l] - mx); put1<RS>(tile, lane, 3 + 2 * l, py[A + l] - my); }
#pragma unroll
      for (int c = 0; c < 3; ++c) put1<RS>(tile, lane, 2 + 2 * L + c, mine == c ? 0.75f : 0.25f);   // goal_b.color
      const Word<WM> co = word_of<DC, WM>(bw, B, w0, ln, 1 - i, ra.seed, gw, gt);
#pragma unroll
      for (int c = 0; c < DC; ++c) put1<RS>(tile, lane, 5 + 2 * L + c, co[c]);
      flush_rows<D, false, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
    }
    if constexpr (KIND == MPE_SCN_CRYPTO) {  // simple_crypto.py:127-169 (goal = pick 0, key = pick 1; colours are one-hots of width dim_c)
      const Word<WM> cs = word_of<DC, WM>(bw, B, w0, ln, 2, ra.seed, gw, gt);   // the speaker's utterance
      static_assert((DC & 1) == 0, "crypto rows are written as pairs");
      if (i == 0) {          // Eve: what the speaker says
        constexpr int D = DC;
        RowPairs<D> r(tile, lane);
#pragma unroll
        for (int c = 0; c < DC; c += 2) r.put(c, cs[c], cs[c + 1]);
        flush_rows<D, true, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
      } else {
        constexpr int D = 2 * DC;
        RowPairs<D> r(tile, lane);
        const int first = i == 1 ? pick1 : goal;   // Bob: key, utterance;  Alice: goal colour, key
#pragma unroll
        for (int c = 0; c < DC; c += 2) r.put(c, first == c ? 1.f : 0.f, first == c + 1 ? 1.f : 0.f);
#pragma unroll
        for (int c = 0; c < DC; c += 2) {
          if (i == 1) r.put(DC + c, cs[c], cs[c + 1]);
          else        r.put(DC + c, pick1 == c ? 1.f : 0.f, pick1 == c + 1 ? 1.f : 0.f);
        }
        flush_rows<D, true, RP>(tile, obs_t + B * obs_off_i + w0 * D, nvalid, lane, d.vec4);
      }
    }
    if constexpr (KIND == MPE_SCN_WORLD_COMM) {  // simple_world_comm.py:231-289
      constexpr int NG = A - NADV;
      constexpr int DA = 4 + 2 * L + 2 * (A - 1) + 2 * NG + 2 + DC, DGd = 4 + 2 * L + 2 * (A - 1) + 2 + 2 * (NG - 1);
      const bool f1 = X[(i * XW + 4) * kWave + lane] > 0.f, f2 = X[(i * XW + 5) * kWave + lane] > 0.f;
      bool vis[A];   // may agent i see agent j: same forest, or both in the open; the leader sees everybody (:253)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        const bool o1 = X[(j * XW + 4) * kWave + lane] > 0.f, o2 = X[(j * XW + 5) * kWave + lane] > 0.f;
        vis[j] = i == 0 || (f1 && o1) || (f2 && o2) || (!f1 && !o1 && !f2 && !o2);
      }
      // (column-by-column 4-byte writes: building these rows from (x, y) pairs -- RowPairs, as the other scenarios do --
      //  measured 1.5 us SLOWER here, 20.0 vs 18.6 us at B = 65536, same box)
      auto row = [&](auto dsel, auto advt) {
        constexpr int D = decltype(dsel)::value, RS = tile_stride<D>();
        constexpr bool ADV = decltype(advt)::value;
        constexpr bool HALF = S::trows(ROLL) == 32;   // 32-row tile: rows of worlds 0..31, then of worlds 32..63
#pragma unroll
        for (int h = 0; h < (HALF ? 2 : 1); ++h) {
        const int r = HALF ? (lane & 31) : lane;
        if (!HALF || (lane >> 5) == h) {
        int k = 0;
        put1<RS>(tile, r, 0, mvx); put1<RS>(tile, r, 1, mvy); put1<RS>(tile, r, 2, mx); put1<RS>(tile, r, 3, my);
        k = 4;
#pragma unroll
        for (int l = 0; l < L; ++l) { put1<RS>(tile, r, k, px[A + l] - mx); put1<RS>(tile, r, k + 1, py[A + l] - my); k += 2; }
#pragma unroll
        for (int j = 0; j < A; ++j) {
          if (j == i) continue;
          put1<RS>(tile, r, k, vis[j] ? px[j] - mx : 0.f);
          put1<RS>(tile, r, k + 1, vis[j] ? py[j] - my : 0.f);
          k += 2;
        }
        if (!ADV) { put1<RS>(tile, r, k, f1 ? 1.f : -1.f); put1<RS>(tile, r, k + 1, f2 ? 1.f : -1.f); k += 2; }
#pragma unroll
        for (int j = NADV; j < A; ++j) {
          if (j == i) continue;
          put1<RS>(tile, r, k, vis[j] ? X[(j * XW + 2) * kWave + lane] : 0.f);
          put1<RS>(tile, r, k + 1, vis[j] ? X[(j * XW + 3) * kWave + lane] : 0.f);
          k += 2;
        }
        if (ADV) {
          put1<RS>(tile, r, k, f1 ? 1.f : -1.f); put1<RS>(tile, r, k + 1, f2 ? 1.f : -1.f); k += 2;
          const Word<WM> cl = word_of<DC, WM>(bw, B, w0, ln, 0, ra.seed, gw, gt);   // world.agents[0].state.c
#pragma unroll
          for (int c = 0; c < DC; ++c) put1<RS>(tile, r, k + c, cl[c]);
        }
        }
        const int nv = HALF ? min(max(nvalid - 32 * h, 0), 32) : nvalid;   // uniform
        if (nv > 0) flush_rows<D, false, RP>(tile, obs_t + B * obs_off_i + (w0 + (HALF ? 32 * h : 0)) * D, nv, lane, d.vec4);
        }
      };
      if (i < NADV) row(std::integral_constant<int, DA>{}, std::true_type{});
      else          row(std::integral_constant<int, DGd>{}, std::false_type{});
    }
  };
  if constexpr (DUAL) {
    if (is_rows) {
      // ---- ROWS wave of agent i: behind barrier t, step t's state of every agent out of the exchange block -> rows ------
      // (the landmarks and the per-world picks follow the in-kernel resets with the physics waves' countdown)
      for (int t = 0; t < T; ++t) {
        MPE_STAMP(0);
        const bool reset_now = countdown == 0;
        if (countdown >= 0) countdown = reset_now ? ra.episode_len - 1 : countdown - 1;
        if (reset_now) {
#pragma unroll
          for (int l = 0; l < L; ++l) reset_draw(ra.seed, gw, ep, A + l, ra.landmark_range, px[A + l], py[A + l]);
          if (NCH >= 1) goal = choice_draw(ra.seed, gw, ep, 0, d.choice_pop[0]);
          if (NCH >= 2) pick1 = choice_draw(ra.seed, gw, ep, 1, d.choice_pop[1]);
          ++ep;
        }
        if (HAS_GOAL || KIND == MPE_SCN_SPEAKER_LISTENER) goal_pos<A, L>(px, py, goal, gx, gy);
        MPE_STAMP(2);
        __syncthreads();
        MPE_STAMP(3);
        const float *const X = xch + (t & 1) * A * XW * kWave;
        mx = X[(i * XW + 0) * kWave + lane];
        my = X[(i * XW + 1) * kWave + lane];
        mvx = X[(i * XW + 2) * kWave + lane];
        mvy = X[(i * XW + 3) * kWave + lane];
#pragma unroll
        for (int a = 0; a < A; ++a) {
          if (a == i) { px[a] = mx; py[a] = my; continue; }  // uniform
          px[a] = X[(a * XW + 0) * kWave + lane];
          py[a] = X[(a * XW + 1) * kWave + lane];
        }
#ifdef MPE_PHASE_CLOCK
#pragma unroll
        for (int a = 0; a < A; ++a) asm volatile("" ::"v"(px[a]), "v"(py[a]));
#endif
        MPE_STAMP(4);
        rows(t);
        MPE_STAMP(5);
        if constexpr (SERVE) {
          // served: this wave has nothing to do before the next barrier but wait -- it drains its rows and counts itself (the
          // reward wave publishes step t from its idle loop once the physics waves, idle too, have done the same)
          drain_stores();
          if (lane == 0) __hip_atomic_fetch_add(&idle_cnt[(t + 1) & 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      if (SERVE) __syncthreads();   // (the last step's completion: the reward wave publishes behind this barrier)
      MPE_SPAN_END();
      return;
    }
    // ---- PHYSICS wave of agent i: World.step of step t+1 behind barrier t, from the siblings' state of step t -----------
    float fx, fy;
    { [[maybe_unused]] const int t = 0; MPE_STAMP(0); }
    if (SERVE && !wait_door(0, false)) return;
    step_forces(0, fx, fy);
    step_integrate(0, fx, fy);
    { [[maybe_unused]] const int t = 0; MPE_STAMP(1); }
    publish(0);
    for (int t = 0; t < T; ++t) {
      MPE_STAMP(2);
      if (SERVE) drain_stores();   // step t - 1's state is acknowledged
      __syncthreads();
      MPE_STAMP(3);
      behind_barrier(t);
      MPE_STAMP(4);
      if (t + 1 < T) {
        if (SERVE && !wait_door(t + 1, false)) return;
        step_forces(t + 1, fx, fy);
        step_integrate(t + 1, fx, fy);
        MPE_STAMP(5);
        publish(t + 1);
      }
    }
    if (SERVE) {
      drain_stores();
      __syncthreads();
      MPE_SPAN_END();
      return;
    }
  } else {
    for (int t = 0; t < T; ++t) {
      float fx, fy;
      MPE_STAMP(0);
      if (SERVE && !wait_door(t, false)) return;
      step_forces(t, fx, fy);
      step_integrate(t, fx, fy);
      MPE_STAMP(1);
      publish(t);
      MPE_STAMP(2);
      if (SERVE) drain_stores();   // step t - 1's rows and state are acknowledged: the reward wave says so behind this barrier
      __syncthreads();
      MPE_STAMP(3);
      behind_barrier(t);
      MPE_STAMP(4);
      rows(t);
      MPE_STAMP(5);   // this step's rows are on their way
    }
    if (SERVE) {
      drain_stores();
      __syncthreads();
      MPE_SPAN_END();
      return;
    }
  }
  if (ROLL && NCH >= 1 && ra.episode_len > 0 && live && i == 0) {   // the picks of the last in-kernel reset
    store_aux<AUX>(b.choice + wave_off(w0) + ln, goal);
    if (NCH >= 2) store_aux<AUX>(b.choice + wave_off(B + w0) + ln, pick1);
  }
  if (ROLL && ((speakers_of<KIND>() >> i) & 1u) && live && T > 0) {
    // update_agent_state (core.py:171-177): what agent i said at the last step is its comm state afterwards
    float *const mine = const_cast<float *>(b.comm) + wave_off(((size_t)i * B + w0) * DC) + ln * DC;
    const int id = comm_draw(ra.seed, gw, ra.step0 + (uint64_t)(T - 1), i, DC);
#pragma unroll
    for (int c = 0; c < DC; ++c) mine[c] = c == id ? 1.f : 0.f;
  }
  if (ROLL && ra.episode_len > 0 && live) {
    // in-kernel resets moved the landmarks: hand their positions back (wave i writes landmarks l = i mod A)
#pragma unroll
    for (int l = 0; l < L; ++l)
      if (l % A == i) {
        (b.pos + wave_off((size_t)(2 * (A + l)) * B + w0))[ln] = px[A + l];
        (b.pos + wave_off((size_t)(2 * (A + l) + 1) * B + w0))[ln] = py[A + l];
      }
    if (!movable_i) {  // an immovable agent is never integrated, but a reset did place it
      store_aux<AUX>(b.pos + wave_off((size_t)(2 * i) * B + w0) + ln, mx);
      store_aux<AUX>(b.pos + wave_off((size_t)(2 * i + 1) * B + w0) + ln, my);
      store_aux<AUX>(b.vel + wave_off((size_t)(2 * i) * B + w0) + ln, mvx);
      store_aux<AUX>(b.vel + wave_off((size_t)(2 * i + 1) * B + w0) + ln, mvy);
    }
  }
  MPE_SPAN_END();
}

// ---- dispatch -----------------------------------------------------------------------------------

constexpr size_t kRowsNtFromBytes = 8u << 20, kRollNtFromBytes = 12u << 20;   // bytes of rows per launch / per rollout step
using SplitFn = void (*)(float *, float *, const float *, const int32_t *, const size_t, const int, const int, const unsigned,
                         const NarrowDesc, const MpeBuffers, const RollArgs, const ServeArgs);
struct SplitEntry {
  int kind, A, L, nadv;
  SplitFn step, step_small, roll, roll_small;   // rows stored nontemporal; *_small: at agent scope (mpe_device.h)
  SplitFn roll_dual, roll_dual_small;           // the dual-role rollout (small batches), or nullptr
  SplitFn serve, serve_dual;                    // the step server (scenarios without utterances), single- and dual-role, or nullptr
  size_t lds_step, lds_roll;
};
template <int KIND, int A, int L, int NADV>
constexpr SplitFn serve_fn() { return k_split<KIND, A, L, NADV, true, kRowsSc1, false, true>; }
template <int KIND, int A, int L, int NADV>
constexpr SplitFn serve_dual_fn() {
  if constexpr (dual_kind<KIND>() && SplitShape<KIND, A, L, NADV>::waves(true) * kWave <= 1024)
    return k_split<KIND, A, L, NADV, true, kRowsSc1, true, true>;
  else return nullptr;
}
template <int KIND, int A, int L, int NADV, int RP>
constexpr SplitFn dual_fn() {
  if constexpr (dual_kind<KIND>() && SplitShape<KIND, A, L, NADV>::waves(true) * kWave <= 1024) return k_split<KIND, A, L, NADV, true, RP, true>;
  else return nullptr;
}
#define MPE_SPLIT_ENTRY(KIND, A, L, NADV)                                                                         \
  { KIND, A, L, NADV, k_split<KIND, A, L, NADV, false, kRowsNt>, k_split<KIND, A, L, NADV, false, kRowsSc1>,       \
    k_split<KIND, A, L, NADV, true, kRowsNt>, k_split<KIND, A, L, NADV, true, kRowsSc1>,                           \
    dual_fn<KIND, A, L, NADV, kRowsNt>(), dual_fn<KIND, A, L, NADV, kRowsSc1>(), serve_fn<KIND, A, L, NADV>(),     \
    serve_dual_fn<KIND, A, L, NADV>(),                                                                             \
    SplitShape<KIND, A, L, NADV>::lds_bytes(false), SplitShape<KIND, A, L, NADV>::lds_bytes(true) }

static const SplitEntry kSplitTable[] = {
    MPE_SPLIT_ENTRY(MPE_SCN_SIMPLE, 1, 1, 0),
    MPE_SPLIT_ENTRY(MPE_SCN_SPREAD, 1, 1, 0), MPE_SPLIT_ENTRY(MPE_SCN_SPREAD, 2, 2, 0),
    MPE_SPLIT_ENTRY(MPE_SCN_SPREAD, 3, 3, 0), MPE_SPLIT_ENTRY(MPE_SCN_SPREAD, 4, 4, 0),
    MPE_SPLIT_ENTRY(MPE_SCN_SPREAD, 5, 5, 0), MPE_SPLIT_ENTRY(MPE_SCN_SPREAD, 6, 6, 0),
    MPE_SPLIT_ENTRY(MPE_SCN_TAG, 4, 2, 3), MPE_SPLIT_ENTRY(MPE_SCN_TAG, 2, 1, 1),
    MPE_SPLIT_ENTRY(MPE_SCN_TAG, 6, 3, 4),
    MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 3, 2, 1), MPE_SPLIT_ENTRY(MPE_SCN_PUSH, 2, 2, 1),
    MPE_SPLIT_ENTRY(MPE_SCN_SPEAKER_LISTENER, 2, 3, 0), MPE_SPLIT_ENTRY(MPE_SCN_REFERENCE, 2, 3, 0),
    MPE_SPLIT_ENTRY(MPE_SCN_CRYPTO, 3, 2, 1), MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 6, 5, 4),
    // (Round 3 listed 19 more entries here -- simple_adversary at 2-6 agents x 1-2 adversaries, simple_world_comm at 1-3 prey x
    //  2-5 predators: 360 instantiations, a 2.5 min build.  Their callbacks are written for any team size, and so are their
    //  row programs: shapes without an entry step as World.step + mpe_rows (rowspec.builtin_program), every team size up to
    //  64 entities, no instantiation per shape.  A/B of the two on the removed shapes: profiles/r4_team_sizes_ab.txt,
    //  measured with the -DMPE_SPLIT_TEAM_GRID build, which puts the entries back.)
#ifdef MPE_SPLIT_TEAM_GRID
    MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 2, 1, 1), MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 3, 2, 2),
    MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 4, 3, 1), MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 4, 3, 2),
    MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 5, 4, 1), MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 5, 4, 2),
    MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 6, 5, 1), MPE_SPLIT_ENTRY(MPE_SCN_ADVERSARY, 6, 5, 2),
    MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 3, 5, 2), MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 4, 5, 3),
    MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 5, 5, 4), MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 6, 5, 5),
    MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 4, 5, 2), MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 5, 5, 3),
    MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 7, 5, 5),
    MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 5, 5, 2), MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 6, 5, 3),
    MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 7, 5, 4), MPE_SPLIT_ENTRY(MPE_SCN_WORLD_COMM, 8, 5, 5),
#endif
};

static const SplitEntry *find_split(int kind, int A, int L, int nadv) {
  for (const SplitEntry &e : kSplitTable)
    if (e.kind == kind && e.A == A && e.L == L && (kind < MPE_SCN_TAG || e.nadv == nadv)) return &e;
  return nullptr;
}

bool split_supports(int kind, int A, int L, int nadv) { return find_split(kind, A, L, nadv) != nullptr; }

int launch_split(bool roll, int kind, int A, int L, int nadv, const NarrowDesc &d, const MpeBuffers &b, size_t B,
                 const RollArgs &ra, hipStream_t stream) {
  const SplitEntry *e = find_split(kind, A, L, nadv);
  if (!e) return MPE_EUNSUPPORTED;
  // worlds per workgroup: full waves.  Narrower workgroups (more of them, idle upper lanes) were tried for the
  // small-batch configs, hoping for more waves per SIMD to hide latency: they lose --
  RollArgs r2 = ra;
#ifdef MPE_SPLIT_WPW
  r2.wpw = MPE_SPLIT_WPW;
#else
  r2.wpw = kWave;   // measured (tools/ab_run.sh, MPE_SPLIT_WPW builds): 64 wins at every batch -- tag B=16384 4.1 / 4.5 / 5.7 us at 64 / 32 / 16
#endif
  const unsigned grid = (unsigned)((B + r2.wpw - 1) / r2.wpw);
  // row-store policy (mpe_device.h): nontemporal from ~8 MB of rows per launch, agent scope below (measured: spread N=3 at
  // 65 536 worlds (14 MB) 5.54 vs 5.70 us, simple_reference (11 MB) 5.05 vs 5.21, simple_adversary (7 MB) 4.67 vs 4.63,
  // simple_tag at 16 384 worlds (4 MB) 3.98 vs 3.75, spread N=3 at 4096 worlds (0.9 MB) 3.10 vs 2.98)
  // The rollouts want the agent-scope form up to larger steps (nt LOSES there: simple_adversary 1.47 -> 1.73 us per step with nt,
  // 1.39 with sc1; simple_reference 1.97 / 2.29 / 1.88) and nt only for the big rows of simple_world_comm (50 MB per step:
  // 10.8 plain / 10.7 nt / 16.5 sc1).
  const size_t row_bytes = (size_t)d.obs_off[A] * sizeof(float) * B;
  const bool small = row_bytes < (roll ? kRollNtFromBytes : kRowsNtFromBytes);
  SplitFn fn = roll ? (small ? e->roll_small : e->roll) : small ? e->step_small : e->step;
  int waves = A + 1;
  // The dual-role rollout (a physics and a rows wave per agent) where the chip is under-filled: up to 1.5 workgroups per
  // CU the extra waves find idle SIMD time (simple_tag at 16 384 worlds 1.31 -> 1.01 us per step); from ~4 per CU on the
  // CUs are full either way and the second barrier partner only costs (simple_tag at 65 536 worlds 3.54 -> 3.80).
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
    (void)hipGetLastError();
    return 256;
  }();
#ifdef MPE_SPLIT_DUAL_MAX_WG_PER_CU_X2
  const unsigned dual_max = (unsigned)n_cu * (MPE_SPLIT_DUAL_MAX_WG_PER_CU_X2) / 2;
#else
  const unsigned dual_max = (unsigned)n_cu * 3 / 2;
#endif
  SplitFn dual = small ? e->roll_dual_small : e->roll_dual;
  if (roll && dual && grid <= dual_max) {
    fn = dual;
    waves = 2 * A + 1;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(waves * kWave), roll ? e->lds_roll : e->lds_step, stream,
                     b.pos, b.vel, b.act, b.ids, B, (int)r2.wpw, (int)r2.observe_only, (unsigned)d.movable, d, b, r2, ServeArgs{});
  return (int)hipGetLastError();
}

bool serve_supports(int kind, int A, int L, int nadv) {
  const SplitEntry *e = find_split(kind, A, L, nadv);
  return e && e->serve;
}
unsigned serve_grid(size_t B) { return (unsigned)((B + kWave - 1) / kWave); }

// The step server's launch: T commanded steps (mpe_step_server_start).  The grid is the per-step kernel's -- one workgroup per
// 64 worlds -- and every workgroup must be RESIDENT for the launch to make progress on its own terms (a workgroup that waits
// for a doorbell holds its CU slots): the caller (mpe_abi.hip) refuses grids beyond what the occupancy query admits.
int launch_split_serve(int kind, int A, int L, int nadv, const NarrowDesc &d, const MpeBuffers &b, size_t B, const RollArgs &ra,
                       const ServeHandles &h, hipStream_t stream) {
  const SplitEntry *e = find_split(kind, A, L, nadv);
  if (!e || !e->serve) return MPE_EUNSUPPORTED;
  RollArgs r2 = ra;
  r2.wpw = kWave;
  ServeArgs sv;
  sv.door = reinterpret_cast<unsigned long long *>(h.door);
  sv.flag = reinterpret_cast<unsigned long long *>(h.flag);
  sv.status = h.status;
  sv.act_ring = h.act_ring;
  sv.comm_ring = h.comm_ring;
  sv.ring = h.ring;
  sv.slots = h.slots;
  sv.timeout_ticks = h.timeout_ticks;
  const size_t lds = e->lds_roll + 16;   // + the two idle counters
  const unsigned grid = serve_grid(B);
  int per_cu = 0, dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return MPE_EUNSUPPORTED;
  }
  // the dual-role server (a physics and a rows wave per agent, as the dual-role rollout) where the chip is under-filled
  SplitFn fn = e->serve;
  int waves = A + 1;
#ifndef MPE_SERVE_NO_DUAL
  if (e->serve_dual && grid <= (unsigned)n_cu * 3 / 2) {
    fn = e->serve_dual;
    waves = 2 * A + 1;
  }
#endif
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(fn), waves * kWave, lds) != hipSuccess) {
    (void)hipGetLastError();
    return MPE_EUNSUPPORTED;
  }
  // (one fewer per CU than the API's answer: the hardware admits one fewer at some SGPR counts, MI355X_MICROARCH.md)
  // (a launch whose commands all precede it never waits: its workgroups may come and go as any kernel's)
  if (!h.ahead && (size_t)grid > (size_t)n_cu * (size_t)(per_cu > 1 ? per_cu - 1 : per_cu)) return MPE_ESERVER_TOO_LARGE;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(waves * kWave), lds, stream, b.pos, b.vel, b.act, b.ids, B, (int)r2.wpw, 0,
                     (unsigned)d.movable, d, b, r2, sv);
  return (int)hipGetLastError();
}

// the commander's side of the step server: one-thread / one-thread-per-workgroup launches on the CALLER's stream
__global__ void k_serve_ring(unsigned long long *const door, const unsigned long long n) {
  // (stream order put this launch behind whatever produced the commanded steps' moves; their writes were released when that
  //  launch ended.  The count is RELATIVE -- n more steps -- so a captured ring replays as "n more" whatever the step number.)
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_fetch_add(door, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(256)
k_serve_wait(const unsigned long long *const flag, const unsigned n_flags, const unsigned long long completed,
             unsigned int *const status, const unsigned long long timeout_ticks) {
  // ends when every workgroup of the server has published `completed` steps (or at the timeout): launches behind it on this
  // stream read those steps' outputs, which were written through and acknowledged before their flag was stored
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_flags) return;
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  while (__hip_atomic_load(flag + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < completed) {
    __builtin_amdgcn_s_sleep(8);
    if ((unsigned long long)wall_clock64() - t0 > timeout_ticks) {
      __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
}
int launch_serve_ring(uint64_t *door, uint64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(k_serve_ring, dim3(1), dim3(64), 0, stream, reinterpret_cast<unsigned long long *>(door), (unsigned long long)n);
  return (int)hipGetLastError();
}
int launch_serve_wait(const uint64_t *flag, unsigned n_flags, uint64_t completed, uint32_t *status, uint64_t timeout_ticks,
                      hipStream_t stream) {
  hipLaunchKernelGGL(k_serve_wait, dim3((n_flags + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const unsigned long long *>(flag),
                     n_flags, (unsigned long long)completed, status, (unsigned long long)timeout_ticks);
  return (int)hipGetLastError();
}

}  // namespace mpe
