// mpe_rows.hip -- the composable output stage: Scenario.observation / reward of a USER scenario as a small program, so that
// a scenario nobody wrote a kernel for still steps in ONE launch (mpe_step_rows) instead of the generic path's hundred-odd
// torch launches -- interpreted op by op, or, for a program that stays the same, compiled in (the same body with the
// program as constants: MPE_ROWS_STATIC below).
//
// What a reference observation is made of (simple_spread.py:84-100, simple_tag.py:131-147, simple_adversary.py:121-139,
// simple_push.py:78-96, simple_speaker_listener.py:69-92, simple_reference.py:63-83, simple_crypto.py:127-169,
// simple_world_comm.py:231-289): a concatenation of a handful of SEGMENT kinds -- own velocity / position, the offset to
// an entity (every landmark, the other agents of a team, the per-world goal), another agent's velocity or utterance, a
// one-hot / colour of a per-world pick, constants, and simple_world_comm's forest visibility.  And a reference reward is
// an ordered sum of a few TERM kinds -- (minimum) distances, strict-< contact tests, the boundary penalty, squared
// utterance errors -- in an order that fixes the rounding.  A program is that list, 16 bytes per op, built on the host
// from ObsSpec / RewardSpec objects (multiagent_particle_envs_amd/rowspec.py).
//
// Shape of the kernel (E = A + L <= 64): a workgroup is 64 worlds x W waves (W <= min(A, 16)), lane = world.
//   all waves    stage the worlds' state in LDS once -- pos [E][2][64], vel [n_vel][2][64]: coalesced 256-byte loads; every
//                later access is `lds[row * 64 + lane]` with a wave-uniform row (the program counter is wave-uniform:
//                all 64 lanes interpret the same op), i.e. conflict-free and scalar-addressed
//   wave w       agents w, w + W, ...: [World.step of the agent (mpe_step_rows)], its observation program -- each op appends
//                its columns to the wave's LDS tile ([64][D] row-major = the output segment), the tile leaves as contiguous
//                16-byte stores -- and its reward program (two accumulators, one value register, eight slots); behind one
//                barrier the shared-reward sum (environment.py:100-102) in the reference's order.
// Interpreted: the ops and the per-entity tables (sizes, masses, row offsets, program ranges) sit in device memory and are
// read by scalar loads, one op ahead.  (Staging them in LDS per workgroup, and two waves per agent, were both measured
// slower: profiles/r4_ab_logs.txt.)  The interpreter costs 3-4x the instructions of a fused kernel (1350 scalar + 560 vector
// per wave against 165 + 158 for simple_spread): 13.4 us per launch at 65 536 worlds where the fused kernel takes 5.5.
// Compiled in: 5.8 us.
// Arithmetic is the device functions of mpe_device.h (sq2d, sqrt_lt, fast_sqrt, tag_bound) in program order: a built-in
// scenario written as a program reproduces its fused kernel bit for bit, interpreted and compiled (tests/test_rowspec.py).
#include "mpe_internal.h"

// instrumented build (-DMPE_ROWS_CLOCK, tools/rows_clock.py): lane 0 of every wave of the first 8 workgroups stamps the
// shader clock at the phase boundaries into MpeBuffers.force: [workgroup < 8][wave < 16][8] uint64.  Not in the product build.
#ifdef MPE_ROWS_CLOCK
#define MPE_RSTAMP(k)                                                                                           \
  do {                                                                                                          \
    if (b.force && blockIdx.x < 8 && (threadIdx.x & 63) == 0)                                                   \
      reinterpret_cast<unsigned long long *>(b.force)[(((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8) + (k)] = \
          __builtin_amdgcn_s_memtime();                                                                         \
  } while (0)
#else
#define MPE_RSTAMP(k) do { } while (0)
#endif

// test builds (libmpe_hip_stress.so / _stress_racy.so, tests/test_gpu_race.py): at each phase boundary one wave of every workgroup --
// a different one per boundary -- is held back ~30 us, so that its siblings have long moved on: any missing barrier between the
// phases (staged state / World.step / the new state taking its place / row tiles in the same LDS / the next rollout step) then
// shows as a changed result.  Not in the product build.
#ifdef MPE_STRESS_DELAY_WAVE
#define MPE_ROWS_STRESS(k)                                                              \
  do {                                                                                  \
    if (wave == ((MPE_STRESS_DELAY_WAVE) + (k)) % NW)                                   \
      for (int s_ = 0; s_ < 10; ++s_) __builtin_amdgcn_s_sleep(127);                    \
  } while (0)
#else
#define MPE_ROWS_STRESS(k) do { } while (0)
#endif

namespace mpe {

namespace {

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

// (nt: nontemporal 16-byte stores for launches whose rows are 8 MB or more -- a launch-uniform branch, not an instantiation: the
//  kernels, and the images of compiled programs, exist once instead of twice)
__device__ __forceinline__ void flush_tile(const float *tile, float *__restrict__ g, int D, int nvalid, int lane, bool vec4, bool nt) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (vec4 && nvalid == kWave) {          // a full wave of worlds, aligned segment: 16 D float4 pieces, straight copies
    const int nq = 16 * D;
    if (nt) {
      for (int q = lane; q < nq; q += kWave) store_row4<kRowsNt>(g + 4 * q, *reinterpret_cast<const float4 *>(tile + 4 * q));
    } else {
      for (int q = lane; q < nq; q += kWave) store_row4<kRowsPlain>(g + 4 * q, *reinterpret_cast<const float4 *>(tile + 4 * q));
    }
  } else {
    const int nfl = nvalid * D;
    for (int j = lane; j < nfl; j += kWave) g[j] = tile[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The tables live in DEVICE memory (uploaded by this 64-thread kernel whenever their content changes: they arrive as its
// kernel argument, captured at launch time) -- as kernel arguments of k_rows their ~30 cache lines would each be a scalar
// load from the host-coherent kernarg segment.
__global__ void __launch_bounds__(64) k_rows_header(const RowTables t, RowTables *__restrict__ dst) {
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&t);
  uint32_t *out = reinterpret_cast<uint32_t *>(dst);
  for (unsigned k = threadIdx.x; k < sizeof(RowTables) / 4; k += 64) out[k] = src[k];
}

#define MPE_TAB(field) ((int)(offsetof(RowTables, field) / 4))

// ---- a program COMPILED IN (mpe_rows_static_source -> hipcc --genco -> mpe_rows_load_image) ---------------------------------
// The same body, instantiated with STATIC = true, reads its dims, tables and ops from constexpr arrays a generated header
// defines, and walks its agents through literal indices: after inlining, every op code, entity index, column and LDS
// address is a compile-time constant -- the interpreter folds away into straight-line code (the switch, the scalar loads,
// the loop control), the LDS reads of a program batch up, and what is left is the arithmetic in the same order: the
// interpreted and the compiled step agree bit for bit (tests/test_rowspec.py).
#ifdef MPE_ROWS_STATIC
template <bool PHYS> constexpr int static_waves() { return PHYS ? MPE_ROWS_STATIC_WAVES_STEP : MPE_ROWS_STATIC_WAVES_ROWS; }
template <bool PHYS> constexpr int static_lds_floats() { return (PHYS ? MPE_ROWS_STATIC_LDS_STEP : MPE_ROWS_STATIC_LDS_ROWS) / 4; }
__device__ __forceinline__ constexpr RowDims static_dims() { constexpr RowDims d = MPE_ROWS_STATIC_DIMS; return d; }
__device__ __forceinline__ uint32_t static_tab(int k) { constexpr uint32_t T[] = MPE_ROWS_STATIC_TABLES; return T[k]; }
__device__ __forceinline__ int4 static_op(int pc) {
  constexpr int32_t O[][4] = MPE_ROWS_STATIC_OPS;
  return make_int4(O[pc][0], O[pc][1], O[pc][2], O[pc][3]);
}
#else
template <bool PHYS> constexpr int static_waves() { return 1; }
template <bool PHYS> constexpr int static_lds_floats() { return 4; }
__device__ __forceinline__ constexpr RowDims static_dims() { return RowDims{}; }
__device__ __forceinline__ uint32_t static_tab(int) { return 0u; }
__device__ __forceinline__ int4 static_op(int) { return make_int4(0, 0, 0, 0); }
#endif

// agents W, W + NW, ... < A of wave `wave`, each through a literal index (the compiled form); f is always_inline
template <int I, int NW, int A, class F>
__device__ __forceinline__ void static_agents_from(F &f) {
  if constexpr (I < A) {
    f(I);
    static_agents_from<I + NW, NW, A>(f);
  }
}
template <int W, int NW, int A, class F>
__device__ __forceinline__ void static_agents_of_wave(int wave, F &f) {
  if constexpr (W < NW) {
    if (wave == W) static_agents_from<W, NW, A>(f);
    else static_agents_of_wave<W + 1, NW, A>(wave, f);
  }
}
template <bool STATIC, bool PHYS, class F>
__device__ __forceinline__ void agents_of_wave(int first, int stride, int A, F &f) {
  if constexpr (STATIC) static_agents_of_wave<0, static_waves<PHYS>(), static_dims().n_agents>(first, f);
  else for (int i = first; i < A; i += stride) f(i);
}

// EP2: the instantiation behind mpe_step_rows_episode (episodes end inside the launch: a second pass of the observation programs
// for workgroups with a finished world).  Its own instantiation because the loop around the programs costs the plain step
// 8 % interpreted and up to 38 % compiled in (session r4s27) when it is merely present.
// ROLL: the instantiation behind mpe_rollout_rows -- T steps per launch with the state resident in LDS, the moves drawn in the
// kernel (action_draw: the rows mpe_random_actions_block would write), a reset_world of every world at the episode boundaries
// (mpe_reset's draws), every step's rows / rewards / dones into its own trajectory block or over the same one.  (Its own
// instantiation for the reason EP2 is one.)  ROLL + EP2 (mpe_rollout_rows_episode): the episodes end where the done programs or
// the horizon say, per world, inside the rollout -- every step is what mpe_step_rows_episode does.
template <bool PHYS, bool STATIC, bool EP2, bool ROLL = false>
__device__ __forceinline__ void rows_body(const MpeBuffers &b, const RowEpisode &ep, const RowDims &h_arg,
                                          const uint32_t *__restrict__ const tables, const int32_t vec4_nt, const int32_t split_arg,
                                          const uint32_t *__restrict__ const ops_g, const size_t B, const RollArgs &ra = RollArgs{}) {
  const int32_t vec4 = vec4_nt & 1;      // bit 0: 16-byte row stores are possible; bit 1: make them nontemporal
  const bool nt = (vec4_nt & 2) != 0;
  static_assert(!ROLL || PHYS, "a rollout steps the world");
  // LDS: a launch parameter for the interpreter; a compiled program knows its size (no 64 KB opt-in for module kernels needed)
  float *smem;
  if constexpr (STATIC) {
    __shared__ __attribute__((aligned(16))) float smem_static[static_lds_floats<PHYS>()];
    smem = smem_static;
  } else {
    extern __shared__ __attribute__((aligned(16))) float smem_dynamic[];
    smem = smem_dynamic;
  }
  const RowDims h = STATIC ? static_dims() : h_arg;
  const int32_t split = STATIC ? 0 : split_arg;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = uni((int)(threadIdx.x >> 6));
  const int NW = STATIC ? static_waves<PHYS>() : uni((int)(blockDim.x >> 6));          // waves of the workgroup
  // roles: with `split` (two waves per agent: the launch has 2 RW waves) waves [0, RW) run World.step and the observation
  // programs of agents w, w + RW, ..., waves [RW, 2 RW) the reward programs of the same agents at the same time; without it
  // every wave does both.  (Measured slower at every shape tried -- spread N=3 15.3 vs 13.1 us: the launch is not bound by
  // one wave's chain -- and switched off in launch_rows; kept as the A/B.)
  const int RW = split ? NW / 2 : NW;
  const bool is_rows = !split || wave < RW, is_rew = !split || wave >= RW;
  const int A = h.n_agents, E = h.n_entities, NV = h.n_vel, DC = h.dim_c;
  const int rwave = is_rew ? (split ? wave - RW : wave) : A;      // (a rows-only wave owns no reward program: its loops are empty)
  const size_t w0 = (size_t)blockIdx.x * kWave;
  if (w0 >= B) return;
  const int nvalid = (B - w0) < (size_t)kWave ? (int)(B - w0) : kWave;
  const bool live = lane < nvalid;
  const unsigned ln = (unsigned)(live ? lane : nvalid - 1) & 63u;

  MPE_RSTAMP(0);
  MPE_ROWS_STRESS(0);
  // the per-entity tables and the ops: device memory read with scalar loads (uniform addresses) -- staging them in LDS per
  // workgroup measured 7 % slower at 65 536 worlds (one more dependent load + barrier in front of every wave) and halves the
  // waves of the largest programs --, or constants of the image (STATIC)
  const int4 *const ops_d = reinterpret_cast<const int4 *>(ops_g);
  auto OP = [&](int pc) { if constexpr (STATIC) return static_op(pc); else return ops_d[pc]; };
  auto TU = [&](int k) { if constexpr (STATIC) return static_tab(k); else return tables[k]; };
  auto TI = [&](int base, int k) { return (int)TU(base + k); };
  auto TF = [&](int base, int k) { return __builtin_bit_cast(float, TU(base + k)); };
  float *const S_pos = reinterpret_cast<float *>(smem);   // [E][2][64]
  float *const S_vel = S_pos + 2 * E * kWave;             // [NV][2][64]
  float *const S_rew = S_vel + 2 * NV * kWave;            // [A][64]    rewards before the shared sum
  float *const S_slot = S_rew + A * kWave;                // [NW][8][64] the reward programs' value slots, per wave
  int *const S_pick = reinterpret_cast<int *>(S_slot + (size_t)NW * kRowSlots * kWave);   // [4][64]  the per-world picks
  // the post-step state of World.step ([A][4][64], PHYS only) and the waves' row tiles ([W][64 * Dmax]) share one region: the
  // new state has moved into S_pos / S_vel (behind a barrier) before the first tile column is written
  int *const S_word = S_pick + kRowPicks * kWave;      // [A][64] (step layouts) the word each speaking agent says this step (ROLL)
  // (traced programs) values several agents' rewards share: [n_shared][64], computed once per world ahead of the observation programs
  float *const S_shared = reinterpret_cast<float *>(S_word + (PHYS ? A * kWave : 0));
  float *const S_new = S_shared + (size_t)h.n_shared * kWave;
  float *const tiles = S_new;

  // ---- episode bookkeeping (mpe_episode_finish): count the step, find the worlds that finished, leave if none did ----------
  bool fin = false;
  int cnt_now = 0;       // (mode 2: the worlds' step counts, read with the state; every wave keeps the same copy)
  if constexpr (EP2) cnt_now = (ep.episode_step + wave_off(w0))[ln];
  if (ep.enabled == 1) {
    const int cnt = (ep.episode_step + wave_off(w0))[ln] + 1;
    const bool horizon = ep.max_steps > 0 && cnt >= ep.max_steps;
    fin = horizon;
    for (int a = 0; a < A; ++a) fin = fin || (b.done + wave_off((size_t)a * B + w0))[ln] != 0;
    fin = fin && live;
    // every wave of the workgroup reads the same counters and flags and reaches the same verdict; nobody writes them
    // before everybody has read them (the barrier), so either all waves leave here or none does
    const bool any = __builtin_amdgcn_ballot_w64(fin) != 0;
    __syncthreads();
    if (wave == 0 && live) {
      (ep.episode_step + wave_off(w0))[ln] = fin ? 0 : cnt;
      if (horizon)
        for (int a = 0; a < A; ++a) (b.done + wave_off((size_t)a * B + w0))[ln] = 1;
    }
    if (!any) return;
  }
  const uint64_t gw = ep.world_offset + w0 + ln;
  // reset_world's placement of entity e: the program's own boxes (MpeRowProgram.reset_boxes), else agents on [-1,1)^2 and landmarks
  // on [-range, range)^2 (launch-uniform choice; a constant in a compiled program)
  auto place = [&](uint64_t seed, uint64_t episode, int e, float range, float &x, float &y) {
    if (h.reset_boxes)
      reset_draw_box(seed, gw, episode, e, TF(MPE_TAB(reset_box), 4 * e), TF(MPE_TAB(reset_box), 4 * e + 1),
                     TF(MPE_TAB(reset_box), 4 * e + 2), TF(MPE_TAB(reset_box), 4 * e + 3), x, y);
    else
      reset_draw(seed, gw, episode, e, e < A ? 1.0f : range, x, y);
  };

  // ---- (World.step) this wave's first agent's move leaves for HBM together with the state loads: one memory round trip, not two
  float act_x = 0.f, act_y = 0.f;
  if constexpr (PHYS && !ROLL) {
    if (is_rows && wave < A && ((h.movable >> wave) & 1ull)) fetch_action_wave(b, B, wave, w0, ln, 1.0f, act_x, act_y);   // raw: scaled by accel below
  }
  // ---- stage the state (finished worlds: reset_world first -- the draws of mpe_reset for (seed, world, episode)) -----------
  // two entities per turn: their eight loads are in flight together before the first LDS write waits for any of them
  for (int e0 = wave; e0 < E; e0 += 2 * NW) {
    const int e1 = e0 + NW;
    const bool two = e1 < E;
    float x[2], y[2], vx[2] = {0.f, 0.f}, vy[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = k ? e1 : e0;
      if (k && !two) break;      // uniform
      x[k] = (b.pos + wave_off((size_t)(2 * e) * B + w0))[ln];
      y[k] = (b.pos + wave_off((size_t)(2 * e + 1) * B + w0))[ln];
      if (e < NV) {
        vx[k] = (b.vel + wave_off((size_t)(2 * e) * B + w0))[ln];
        vy[k] = (b.vel + wave_off((size_t)(2 * e + 1) * B + w0))[ln];
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = k ? e1 : e0;
      if (k && !two) break;
      if (fin) {
        place(ep.seed, ep.episode, e, ep.landmark_range, x[k], y[k]);
        (b.pos + wave_off((size_t)(2 * e) * B + w0))[ln] = x[k];
        (b.pos + wave_off((size_t)(2 * e + 1) * B + w0))[ln] = y[k];
        if (e < NV && e < A) {
          vx[k] = vy[k] = 0.f;
          (b.vel + wave_off((size_t)(2 * e) * B + w0))[ln] = 0.f;
          (b.vel + wave_off((size_t)(2 * e + 1) * B + w0))[ln] = 0.f;
        }
      }
      S_pos[(2 * e) * kWave + lane] = x[k];
      S_pos[(2 * e + 1) * kWave + lane] = y[k];
      if (e < NV) {
        S_vel[(2 * e) * kWave + lane] = vx[k];
        S_vel[(2 * e + 1) * kWave + lane] = vy[k];
      }
    }
  }
  for (int k = wave; k < h.n_picks; k += NW) {
    int g = (b.choice + wave_off((size_t)k * B + w0))[ln];
    if (fin) {
      g = choice_draw(ep.seed, gw, ep.episode, k, ep.choice_pop[k]);
      (b.choice + wave_off((size_t)k * B + w0))[ln] = g;
    }
    S_pick[k * kWave + lane] = g;
  }
  if (ep.enabled && DC > 0 && b.comm) {      // every reset_world zeroes the utterances of the worlds it resets
    for (int a = wave; a < A; a += NW)
      if (fin)
        for (int c = 0; c < DC; ++c) (const_cast<float *>(b.comm) + wave_off(((size_t)a * B + w0) * DC))[ln * DC + c] = 0.f;
  }
  MPE_RSTAMP(1);      // state loads issued and parked in LDS
  __syncthreads();
  MPE_RSTAMP(2);

  auto P = [&](int e, int c) { return S_pos[(2 * e + c) * kWave + lane]; };
  auto V = [&](int e, int c) { return e < NV ? S_vel[(2 * e + c) * kWave + lane] : 0.f; };
  auto pick = [&](int k) { return S_pick[k * kWave + lane]; };
  auto shared = [&](int k) { return S_shared[k * kWave + lane]; };
  auto word = [&](int j, int c) {
    if constexpr (ROLL) return (!fin && ((ep.speakers >> j) & 1u)) ? (S_word[j * kWave + lane] == c ? 1.f : 0.f) : 0.f;   // drawn this step / silent / restarted
    else return (fin || !b.comm) ? 0.f : (b.comm + wave_off(((size_t)j * B + w0) * DC))[ln * DC + c];
  };
  // inside region r (a landmark, e.g. a forest of simple_world_comm.py:231-261): strict |e - region| < size_e + size_region
  auto in_region = [&](int e, int r) {
    const int f = h.region_entity[r];
    return sqrt_lt(sq2d(P(e, 0) - P(f, 0), P(e, 1) - P(f, 1)), TF(MPE_TAB(size), e) + TF(MPE_TAB(size), f));
  };

  // ---- (ROLL) T steps: the output pointers of the current step, the reset at an episode boundary, then the step itself ---------
  MpeBuffers bo = b;
  const int T_ = ROLL ? ra.T : 1;
  // (one step as a lambda: a `return` below ends the step -- the launch, when there is only one)
  auto one_step = [&](const int t) __attribute__((always_inline)) {
  const uint64_t gstep = ra.step0 + (uint64_t)t;
  const int cnt2 = cnt_now + 1;
  const uint64_t episode_now = ep.episode + (ROLL ? (uint64_t)t : 0u);      // (one episode number per step, as the host counts them)
  if constexpr (ROLL && EP2) fin = false;
  if constexpr (ROLL) {
    if (ra.trajectory && t > 0) {
      bo.obs += (size_t)TI(MPE_TAB(obs_off), A) * B;
      if (bo.rew) bo.rew += (size_t)A * B;
      if (bo.done) bo.done += (size_t)A * B;
    }
    if (ra.episode_len > 0 && gstep % (uint64_t)ra.episode_len == 0) {      // (uniform) reset_world of every world
      const uint64_t episode = gstep / (uint64_t)ra.episode_len;
      for (int e = wave; e < E; e += NW) {
        float x, y;
        place(ra.seed, episode, e, ra.landmark_range, x, y);
        S_pos[(2 * e) * kWave + lane] = x;
        S_pos[(2 * e + 1) * kWave + lane] = y;
        if (live) {
          (b.pos + wave_off((size_t)(2 * e) * B + w0))[ln] = x;
          (b.pos + wave_off((size_t)(2 * e + 1) * B + w0))[ln] = y;
        }
        if (e < NV && e < A) {
          S_vel[(2 * e) * kWave + lane] = 0.f;
          S_vel[(2 * e + 1) * kWave + lane] = 0.f;
          if (live && !((h.movable >> e) & 1ull)) {      // (movable agents' velocities leave with this step's state)
            (b.vel + wave_off((size_t)(2 * e) * B + w0))[ln] = 0.f;
            (b.vel + wave_off((size_t)(2 * e + 1) * B + w0))[ln] = 0.f;
          }
        }
      }
      for (int k = wave; k < h.n_picks; k += NW) {
        const int g = choice_draw(ra.seed, gw, episode, k, ep.choice_pop[k]);
        if (live) (b.choice + wave_off((size_t)k * B + w0))[ln] = g;
        S_pick[k * kWave + lane] = g;
      }
      __syncthreads();
    }
  }
  MPE_ROWS_STRESS(1 + t);
  if constexpr (ROLL) {      // the words of this step (mpe_random_comm's draws); read behind World.step's barriers
    if (ep.speakers != 0u)
      for (int a = wave; a < A; a += NW)
        if ((ep.speakers >> a) & 1u) S_word[a * kWave + lane] = comm_draw(ra.seed, gw, gstep, a, DC);
  }
  if constexpr (PHYS) {
    // ---- World.step (core.py:117-169) by the agent waves: action force, contacts with every other entity in ascending order
    // (Q9) from the PRE-step positions, integration -- the device functions and the order of the step kernels, so the state
    // that leaves here is theirs to the bit.  New state -> S_new; behind the barrier it replaces the staged one and goes
    // back to HBM (no wave still reads pre-step positions then).
    {
      auto phys_agent = [&](const int i) __attribute__((always_inline)) {
        float mx = P(i, 0), my = P(i, 1), mvx = V(i, 0), mvy = V(i, 1);
        if ((h.movable >> i) & 1ull) {
          float ux, uy;
          if constexpr (ROLL) {    // the one-hot row mpe_random_actions_block would write for (seed, world, step, agent) -- or the caller's
            if (ra.act_seq) {
              fetch_action_seq(ra.act_seq, t, A, B, i, w0 + ln, TF(MPE_TAB(accel), i), ux, uy);
            } else {
              const int m = action_draw(ra.seed, gw, gstep, i);
              ux = ((m == 1 ? 1.f : 0.f) - (m == 2 ? 1.f : 0.f)) * TF(MPE_TAB(accel), i);
              uy = ((m == 3 ? 1.f : 0.f) - (m == 4 ? 1.f : 0.f)) * TF(MPE_TAB(accel), i);
            }
          } else if (i == wave) {  // prefetched at kernel entry; a one-hot row / an id decodes to exact -1 / 0 / +1: scale now
            const bool raw = b.act || b.ids;
            ux = raw ? act_x * TF(MPE_TAB(accel), i) : act_x;
            uy = raw ? act_y * TF(MPE_TAB(accel), i) : act_y;
          } else {
            fetch_action_wave(b, B, i, w0, ln, TF(MPE_TAB(accel), i), ux, uy);
          }
          float fx = ux + 0.f, fy = uy + 0.f;
          if ((h.collide >> i) & 1ull) {
            for (int j = 0; j < E; ++j) {
              if (j == i || !((h.collide >> j) & 1ull)) continue;      // uniform
              float cx, cy;
              contact_force(mx - P(j, 0), my - P(j, 1), TF(MPE_TAB(size), i) + TF(MPE_TAB(size), j), h.cforce, h.cmargin, h.cmargin_inv, cx, cy);
              fx = cx + fx;
              fy = cy + fy;
            }
          }
          integrate_one(mx, my, mvx, mvy, fx, fy, TF(MPE_TAB(inv_mass), i), TF(MPE_TAB(max_speed), i), h.damp, h.dt);
        }
        S_new[(4 * i + 0) * kWave + lane] = mx;
        S_new[(4 * i + 1) * kWave + lane] = my;
        S_new[(4 * i + 2) * kWave + lane] = mvx;
        S_new[(4 * i + 3) * kWave + lane] = mvy;
      };
      agents_of_wave<STATIC, PHYS>(is_rows ? wave : A, RW, A, phys_agent);
    }
    MPE_RSTAMP(3);    // World.step of this wave's agents computed
#ifndef MPE_STRESS_STORE_BEFORE_BARRIER      // (the negative control of tests/test_gpu_race.py: without this barrier a wave that is done
    __syncthreads();                         //  replaces the staged positions while a late sibling still computes contacts from them)
#endif
    for (int i = wave; i < A; i += NW) {
      const float mx = S_new[(4 * i + 0) * kWave + lane], my = S_new[(4 * i + 1) * kWave + lane];
      const float mvx = S_new[(4 * i + 2) * kWave + lane], mvy = S_new[(4 * i + 3) * kWave + lane];
      S_pos[(2 * i) * kWave + lane] = mx;
      S_pos[(2 * i + 1) * kWave + lane] = my;
      S_vel[(2 * i) * kWave + lane] = mvx;
      S_vel[(2 * i + 1) * kWave + lane] = mvy;
      if (live && ((h.movable >> i) & 1ull)) {
        (b.pos + wave_off((size_t)(2 * i) * B + w0))[ln] = mx;
        (b.pos + wave_off((size_t)(2 * i + 1) * B + w0))[ln] = my;
        (b.vel + wave_off((size_t)(2 * i) * B + w0))[ln] = mvx;
        (b.vel + wave_off((size_t)(2 * i + 1) * B + w0))[ln] = mvy;
      }
    }
    __syncthreads();
  }
  MPE_RSTAMP(4);      // the post-step state is in LDS
  MPE_ROWS_STRESS(2 + t);

  // Two passes at most: the second only in mode 2 (mpe_step_rows_episode) and only in a workgroup where a world finished --
  // its worlds restarted, the observation programs run once more on the new episode's first state.  (One copy of the code.)
  constexpr int kPasses = EP2 ? 2 : 1;
  if constexpr (ROLL && !EP2) {
    // without a trajectory every step's outputs land on the same block: only the last step's survive, so only the last step
    // computes and stores them (the state moves on either way; with EP2 the done programs decide the restarts: every step runs)
    if (!ra.trajectory && t != T_ - 1) return;
  }
#ifdef MPE_ROWS_TRACED
  // ---- (traced programs) what several agents' rewards share, once per world: the tasks dealt to the waves, parked in LDS; the
  // observation programs run before anybody reads them (a barrier in front of the reward programs)
  if (h.n_shared > 0 && ep.enabled != 1 && bo.rew) {
    auto shared_task = [&](const int k) __attribute__((always_inline)) { traced_shared(k, S_shared + k * kWave + lane, P, V, word, pick); };
    if constexpr (STATIC) static_agents_of_wave<0, static_waves<PHYS>(), static_dims().n_shared>(wave, shared_task);
    else for (int k = wave; k < h.n_shared; k += NW) shared_task(k);
  }
#endif
#pragma nounroll
  for (int pass = 0; pass < kPasses; ++pass) {
  // ---- observation programs of this wave's agents ------------------------------------------------------------------------
  {
    float *const tile = tiles + (size_t)wave * kWave * h.d_max;
    auto obs_agent = [&](const int i) __attribute__((always_inline)) {
      const int D = TI(MPE_TAB(obs_off), i + 1) - TI(MPE_TAB(obs_off), i);
      if (D == 0) return;
      const float mx = P(i, 0), my = P(i, 1);
      // who is inside which region: bit (e * 2 + r), for the visibility rule (same region, or both in the open; agents in
      // h.all_seeing see everybody -- the leader of simple_world_comm.py:253)
      unsigned long long inmask = 0;
      if (h.n_regions > 0) {
        for (int e = 0; e < A; ++e)
          for (int r = 0; r < h.n_regions; ++r) inmask |= in_region(e, r) ? (1ull << (2 * e + r)) : 0ull;
      }
      auto visible = [&](int j) {
        const unsigned mi = (unsigned)(inmask >> (2 * i)) & 3u, mj = (unsigned)(inmask >> (2 * j)) & 3u;
        return ((h.all_seeing >> i) & 1u) || (mi & mj) != 0u || (mi == 0u && mj == 0u);
      };
      float *const row = tile + lane * D;
      int col = 0;
      const int pc0 = TI(MPE_TAB(obs_begin), i), pc1 = TI(MPE_TAB(obs_begin), i + 1);
      auto obs_op = [&](const int4 op) __attribute__((always_inline)) {
        const int code = uni(op.x & 0xff), a0 = uni((op.x >> 8) & 0xff), a1 = uni((op.x >> 16) & 0xff);
        const int e = a0 == kRowSelf ? i : a0;
        switch (code) {
          case ROW_OBS_VEL: row[col] = V(e, 0); row[col + 1] = V(e, 1); col += 2; break;
          case ROW_OBS_POS: row[col] = P(e, 0); row[col + 1] = P(e, 1); col += 2; break;
          case ROW_OBS_REL: row[col] = P(e, 0) - mx; row[col + 1] = P(e, 1) - my; col += 2; break;
          case ROW_OBS_REL_PICK: {       // the entity a per-world pick names (agent.goal_a = np.random.choice(world.landmarks))
            const int g = uni(op.y) + pick(a1);       // per lane; the LDS address keeps its lane column: conflict-free
            row[col] = P(g, 0) - mx; row[col + 1] = P(g, 1) - my; col += 2;
            break;
          }
          case ROW_OBS_COMM:             // agent e's utterance (AgentState.c): a1 floats
            for (int c = 0; c < a1; ++c) row[col + c] = word(e, c);
            col += a1;
            break;
          case ROW_OBS_CONST: row[col] = unif(__builtin_bit_cast(float, op.z)); col += 1; break;
          case ROW_OBS_ONEHOT: {         // a1 floats: hi where (pick a0 + offset == column), lo elsewhere (colours, keys)
            const int g = pick(a0) + uni(op.y);
            const float lo = unif(__builtin_bit_cast(float, op.z)), hi = unif(__builtin_bit_cast(float, op.w));
            for (int c = 0; c < a1; ++c) row[col + c] = g == c ? hi : lo;
            col += a1;
            break;
          }
          case ROW_OBS_REL_VIS: {
            const bool s = visible(e);
            row[col] = s ? P(e, 0) - mx : 0.f; row[col + 1] = s ? P(e, 1) - my : 0.f; col += 2;
            break;
          }
          case ROW_OBS_VEL_VIS: {
            const bool s = visible(e);
            row[col] = s ? V(e, 0) : 0.f; row[col + 1] = s ? V(e, 1) : 0.f; col += 2;
            break;
          }
          // (inmask holds the agents' bits -- the visibility rule is about agents; any other entity is tested where it is asked for)
          case ROW_OBS_IN_REGION: row[col] = (e < A ? ((inmask >> (2 * e + a1)) & 1ull) != 0ull : in_region(e, a1)) ? 1.f : -1.f; col += 1; break;
          // ---- range forms: one decode, the entities of a run in an inner loop (two of them in flight per LDS round trip) ----
          case ROW_OBS_REL_RANGE: case ROW_OBS_VEL_RANGE: case ROW_OBS_REL_VIS_RANGE: case ROW_OBS_VEL_VIS_RANGE: {
            const int skip = (uni(op.x >> 24) & 1) ? i : -1;
            const bool rel = code == ROW_OBS_REL_RANGE || code == ROW_OBS_REL_VIS_RANGE;
            const bool vis = code == ROW_OBS_REL_VIS_RANGE || code == ROW_OBS_VEL_VIS_RANGE;
            const float *const src = rel ? S_pos : S_vel;
            const int end = a0 + a1;
            for (int q = a0; q < end; q += 4) {
              // four entities' coordinates leave for LDS together (one round trip for the group, not one per entity); slots
              // past the run re-read its first entity and are dropped
              float xs[4], ys[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int qq = q + k < end ? q + k : q;
                const bool have = rel || qq < NV;
                const int qs = have ? qq : 0;
                const float x = src[(2 * qs) * kWave + lane], y = src[(2 * qs + 1) * kWave + lane];
                xs[k] = have ? x : 0.f;
                ys[k] = have ? y : 0.f;
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int qq = q + k;
                if (qq < end && qq != skip) {      // uniform
                  float x = rel ? xs[k] - mx : xs[k], y = rel ? ys[k] - my : ys[k];
                  if (vis) { const bool s_ = visible(qq); x = s_ ? x : 0.f; y = s_ ? y : 0.f; }
                  row[col] = x; row[col + 1] = y; col += 2;
                }
              }
            }
            break;
          }
          case ROW_OBS_CONST_N: {
            const float cv = unif(__builtin_bit_cast(float, op.z));
            for (int c = 0; c < a1; ++c) row[col + c] = cv;
            col += a1;
            break;
          }
#ifdef MPE_ROWS_TRACED
          // a reference-style file's observation(agent, world), traced (symtrace.py): straight-line code over the staged state
          case ROW_OBS_CODE: traced_obs(i, row + col, P, V, word, pick); col += uni(op.y); break;
#endif
          default: break;
        }
      };
      if constexpr (STATIC) {      // every op a constant: the loop unrolls, the switch folds
#pragma unroll
        for (int pc = pc0; pc < pc1; ++pc) obs_op(OP(pc));
      } else {
        int4 nxt = pc0 < pc1 ? OP(pc0) : make_int4(0, 0, 0, 0);
        for (int pc = pc0; pc < pc1; ++pc) {
          const int4 op = nxt;
          if (pc + 1 < pc1) nxt = OP(pc + 1);        // in flight while this op executes
          obs_op(op);
        }
      }
      flush_tile(tile, bo.obs + B * (size_t)TI(MPE_TAB(obs_off), i) + w0 * (size_t)D, D, nvalid, lane, vec4 != 0, nt);
    };
    agents_of_wave<STATIC, PHYS>(is_rows ? wave : A, RW, A, obs_agent);
  }
  MPE_RSTAMP(5);      // observation rows stored
  if (ep.enabled == 1 || pass == 1) return;      // (mpe_episode_finish, or the restarted worlds' rows: rewards and dones are the step's)
#ifndef MPE_STRESS_NO_SHARED_BARRIER      // (the negative control of tests/test_gpu_race.py: without it a late wave's shared values are read before they exist)
  if (h.n_shared > 0 && bo.rew) __syncthreads();      // (launch-uniform) the shared values are complete
#endif

  // ---- reward and done programs of this wave's agents ----------------------------------------------------------------------
  const bool has_done = TI(MPE_TAB(done_begin), A) != TI(MPE_TAB(done_begin), 0);
  bool dn_wave = false;      // some agent of this wave is done (per world)
  if (bo.rew || (bo.done && has_done)) {
    float *const slot = S_slot + (size_t)wave * kRowSlots * kWave;      // this wave's eight value slots
    // min over the run first .. first + n - 1 of |p[q] - o|^2 (flip: |o - p[q]|^2), first to last; four positions per LDS round trip
    auto min_d2_run = [&](int first, int n, float ox, float oy, bool flip) {
      const int end = first + n;
      float m = 0.f;
      for (int q = first; q < end; q += 4) {
        float xs[4], ys[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int qq = q + k < end ? q + k : q;
          xs[k] = P(qq, 0); ys[k] = P(qq, 1);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (q + k < end) {      // uniform
            const float d2 = flip ? sq2d(ox - xs[k], oy - ys[k]) : sq2d(xs[k] - ox, ys[k] - oy);
            m = (q + k == first) ? d2 : fminf(m, d2);
          }
      }
      return m;
    };
    auto rew_agent = [&](const int i) __attribute__((always_inline)) {
      float acc[2] = {0.f, 0.f}, v = 0.f;
      bool dn = false;
      const int pc0 = TI(MPE_TAB(rew_begin), i), pc1 = TI(MPE_TAB(rew_begin), i + 1);
      auto rew_op = [&](const int4 op) __attribute__((always_inline)) {
        const int code = uni(op.x & 0xff), a0 = uni((op.x >> 8) & 0xff), a1 = uni((op.x >> 16) & 0xff), a2 = uni((op.x >> 24) & 0xff);
        const float f = unif(__builtin_bit_cast(float, op.z));
        switch (code) {
          case ROW_R_D2: v = sq2d(P(a0, 0) - P(a1, 0), P(a0, 1) - P(a1, 1)); break;
          case ROW_R_MIN_D2: v = fminf(v, sq2d(P(a0, 0) - P(a1, 0), P(a0, 1) - P(a1, 1))); break;
          case ROW_R_D2_PICK: case ROW_R_MIN_D2_PICK: {
            const int g = uni(op.y) + pick(a1);
            const float d2 = sq2d(P(a0, 0) - P(g, 0), P(a0, 1) - P(g, 1));
            v = code == ROW_R_D2_PICK ? d2 : fminf(v, d2);
            break;
          }
          case ROW_R_SQRT: v = fast_sqrt(v); break;
          case ROW_R_BOUND: v = tag_bound(fabsf(P(a0, a1))); break;     // simple_tag.py:103-108 on coordinate a1 of entity a0
          case ROW_R_COMM_ERR: {        // simple_crypto.py:97-124: squared error of agent a0's utterance against the one-hot of pick a1; 0 when silent
            const int g = pick(a1);
            float e = 0.f;
            bool silent = true;
            for (int c = 0; c < DC; ++c) {
              const float x = word(a0, c), dv = x - (c == g ? 1.f : 0.f);
              silent = silent && (x == 0.f);
              e = e + dv * dv;
            }
            v = silent ? 0.f : e;
            break;
          }
          case ROW_R_COMM_SUM: {        // sum of agent a0's utterance (a chatter penalty)
            float e = 0.f;
            for (int c = 0; c < DC; ++c) e = e + word(a0, c);
            v = e;
            break;
          }
          case ROW_R_CONST: v = f; break;
          case ROW_R_SAVE: slot[a0 * kWave + lane] = v; break;
          case ROW_R_LOAD: v = slot[a0 * kWave + lane]; break;
          case ROW_R_ZERO: acc[a2 & 1] = 0.f; break;
          case ROW_R_ADD: {             // acc += coef * v   (coef = +-1: exactly acc +- v)
            const float t = f * v;
            if (a2 & 1) acc[1] = acc[1] + t; else acc[0] = acc[0] + t;
            break;
          }
          case ROW_R_ADD_IF_HIT: {      // strict contact test, the reference's `if self.is_collision(a, b): rew += coef`
            const bool hit = sqrt_lt(sq2d(P(a0, 0) - P(a1, 0), P(a0, 1) - P(a1, 1)), TF(MPE_TAB(size), a0) + TF(MPE_TAB(size), a1));
            const float t = hit ? f : 0.f;
            if (a2 & 1) acc[1] = acc[1] + t; else acc[0] = acc[0] + t;
            break;
          }
          case ROW_R_ADD_ACC: acc[0] = acc[0] + acc[1]; break;
          case ROW_R_STORE: S_rew[a0 * kWave + lane] = acc[0]; break;
          // ---- done programs: tests on the value machine, OR-ed (environment.py:132-135: the scenario's done callback) ----------
          case ROW_R_ABS_POS: v = fabsf(P(a0, a1)); break;
          case ROW_R_DONE_IF_GT: dn = dn || v > f; break;
          case ROW_R_DONE_IF_LT: dn = dn || v < f; break;
          case ROW_R_DONE_IF_HIT:
            dn = dn || sqrt_lt(sq2d(P(a0, 0) - P(a1, 0), P(a0, 1) - P(a1, 1)), TF(MPE_TAB(size), a0) + TF(MPE_TAB(size), a1));
            break;
#ifdef MPE_ROWS_TRACED
          case ROW_R_CODE: acc[0] = traced_rew(i, P, V, word, pick, shared); break;
          case ROW_R_DONE_CODE: dn = dn || traced_done(i, P, V, word, pick); break;
#endif
          // ---- range forms ---------------------------------------------------------------------------------------------------
          case ROW_R_MIN_D2_RANGE:        // min over agents a0 .. a0 + n - 1 of |a - p[a1]|^2, first to last
            v = min_d2_run(a0, uni(op.y), P(a1, 0), P(a1, 1), false);
            break;
          case ROW_R_MIN_D2_TO_RANGE:     // min over targets a1 .. a1 + n - 1 of |p[a0] - target|^2
            v = min_d2_run(a1, uni(op.y), P(a0, 0), P(a0, 1), true);
            break;
          case ROW_R_ADD_IF_HIT_GRID: {   // every pair of two entity runs: the same constant per contact, any order gives the same float
            const int na = uni(op.y) & 255, nb = (uni(op.y) >> 8) & 255;
            float t = acc[a2 & 1];
            for (int qa = a0; qa < a0 + na; ++qa) {
              const float ax = P(qa, 0), ay = P(qa, 1), sa = TF(MPE_TAB(size), qa);
              const int end = a1 + nb;
              for (int qb = a1; qb < end; qb += 4) {
                float xs[4], ys[4], ss[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const int qq = qb + k < end ? qb + k : qb;
                  xs[k] = P(qq, 0); ys[k] = P(qq, 1); ss[k] = TF(MPE_TAB(size), qq);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  if (qb + k < end) {
                    const bool hit = sqrt_lt(sq2d(ax - xs[k], ay - ys[k]), sa + ss[k]);
                    t = t + (hit ? f : 0.f);
                  }
              }
            }
            if (a2 & 1) acc[1] = t; else acc[0] = t;
            break;
          }
          case ROW_R_ADD_MIN_DIST_GRID: {  // per target b: the distance of the nearest of a run of entities, accumulated in target order
            const int na = uni(op.y) & 255, nb = (uni(op.y) >> 8) & 255;
            float t = acc[a2 & 1];
            for (int qb = a1; qb < a1 + nb; ++qb) {
              v = fast_sqrt(min_d2_run(a0, na, P(qb, 0), P(qb, 1), false));
              t = t + f * v;
            }
            if (a2 & 1) acc[1] = t; else acc[0] = t;
            break;
          }
          default: break;
        }
      };
      if constexpr (STATIC) {
#pragma unroll
        for (int pc = pc0; pc < pc1; ++pc) rew_op(OP(pc));
      } else {
        int4 nxt = pc0 < pc1 ? OP(pc0) : make_int4(0, 0, 0, 0);
        for (int pc = pc0; pc < pc1; ++pc) {
          const int4 op = nxt;
          if (pc + 1 < pc1) nxt = OP(pc + 1);
          rew_op(op);
        }
      }
      // the agent's done program, on a fresh machine
      const int dc0 = TI(MPE_TAB(done_begin), i), dc1 = TI(MPE_TAB(done_begin), i + 1);
      if (dc0 < dc1) {
        acc[0] = acc[1] = v = 0.f;
        if constexpr (STATIC) {
#pragma unroll
          for (int pc = dc0; pc < dc1; ++pc) rew_op(OP(pc));
        } else {
          for (int pc = dc0; pc < dc1; ++pc) rew_op(OP(pc));
        }
      }
      if (bo.done && live) (bo.done + wave_off((size_t)i * B + w0))[ln] = dn ? 1 : 0;
      dn_wave = dn_wave || dn;
    };
    agents_of_wave<STATIC, PHYS>(rwave, RW, A, rew_agent);
    MPE_RSTAMP(6);    // reward programs done
    if (h.collaborative) __syncthreads();      // (uniform: a kernel argument)
    // environment.py:100-102: every agent gets np.sum(reward_n) = r0 + (((0 + r1) + r2) + ...) for n < 9
    float total = 0.f;
    if (h.collaborative) {
      float rest = 0.f;
      for (int a = 1; a < A; ++a) rest += S_rew[a * kWave + lane];
      total = A > 1 ? S_rew[lane] + rest : S_rew[lane];
    }
    if (live && bo.rew)
      for (int i = rwave; i < A; i += RW) (bo.rew + wave_off((size_t)i * B + w0))[ln] = h.collaborative ? total : S_rew[i * kWave + lane];
  } else if (bo.done && live) {
    for (int i = rwave; i < A; i += RW) (bo.done + wave_off((size_t)i * B + w0))[ln] = 0;
  }
  if constexpr (ROLL) {
    if (t == T_ - 1 && ep.speakers != 0u && b.comm && live)      // the agents' comm state after the last step = their last words
      for (int a = wave; a < A; a += NW)
        if ((ep.speakers >> a) & 1u)
          for (int c = 0; c < DC; ++c)
            (const_cast<float *>(b.comm) + wave_off(((size_t)a * B + w0) * DC))[ln * DC + c] = S_word[a * kWave + lane] == c ? 1.f : 0.f;
  }
  if constexpr (!EP2) return;

  // ---- mode 2: the episode ends inside the launch (mpe_step_rows_episode = this step + mpe_episode_finish) -----------------
  // every wave publishes "one of my agents is done", everybody reads everybody's: the same verdict in every wave
  int *const S_dw = reinterpret_cast<int *>(S_slot);      // [NW][64], the first row of each wave's slots (free by now)
  S_dw[(size_t)wave * kRowSlots * kWave + lane] = dn_wave ? 1 : 0;
  __syncthreads();      // (also: every done row of this step is stored and acknowledged before wave 0 may overwrite it below)
  bool any_done = false;
  for (int w = 0; w < NW; ++w) any_done = any_done || S_dw[(size_t)w * kRowSlots * kWave + lane] != 0;
  const bool horizon = ep.max_steps > 0 && cnt2 >= ep.max_steps;
  fin = (horizon || any_done) && live;
  cnt_now = fin ? 0 : cnt2;
  if (wave == 0 && live) {
    (ep.episode_step + wave_off(w0))[ln] = cnt_now;
    if (horizon)
      for (int a = 0; a < A; ++a) (bo.done + wave_off((size_t)a * B + w0))[ln] = 1;
  }
  if (__builtin_amdgcn_ballot_w64(fin) == 0) return;      // nothing finished among these 64 worlds: the usual case
  // reset_world for the finished worlds -- the draws of mpe_reset for (seed, world, episode) -- into HBM and the staged state
  for (int e = wave; e < E; e += NW) {
    if (fin) {
      float x, y;
      place(ep.seed, episode_now, e, ep.landmark_range, x, y);
      (b.pos + wave_off((size_t)(2 * e) * B + w0))[ln] = x;
      (b.pos + wave_off((size_t)(2 * e + 1) * B + w0))[ln] = y;
      S_pos[(2 * e) * kWave + lane] = x;
      S_pos[(2 * e + 1) * kWave + lane] = y;
      if (e < NV && e < A) {
        (b.vel + wave_off((size_t)(2 * e) * B + w0))[ln] = 0.f;
        (b.vel + wave_off((size_t)(2 * e + 1) * B + w0))[ln] = 0.f;
        S_vel[(2 * e) * kWave + lane] = 0.f;
        S_vel[(2 * e + 1) * kWave + lane] = 0.f;
      }
    }
  }
  for (int k = wave; k < h.n_picks; k += NW) {
    if (fin) {
      const int g = choice_draw(ep.seed, gw, episode_now, k, ep.choice_pop[k]);
      (b.choice + wave_off((size_t)k * B + w0))[ln] = g;
      S_pick[k * kWave + lane] = g;
    }
  }
  if (DC > 0 && b.comm) {
    for (int a = wave; a < A; a += NW)
      if (fin)
        for (int c = 0; c < DC; ++c) (const_cast<float *>(b.comm) + wave_off(((size_t)a * B + w0) * DC))[ln * DC + c] = 0.f;
  }
  __syncthreads();
  }      // (second pass: the observation programs on the restarted worlds' state -- `fin` lanes read their utterances as zero)
  };     // one_step
#pragma nounroll
  for (int t = 0; t < T_; ++t) {
    one_step(t);
    if constexpr (ROLL) __syncthreads();      // the next step reuses S_rew, S_word, the slots and the tiles
  }
}

#ifndef MPE_ROWS_STATIC
template <bool PHYS, bool EP2>
__global__ void __launch_bounds__(1024) k_rows(const MpeBuffers b, const RowEpisode ep, const RowDims h,
                                                 const uint32_t *__restrict__ const tables, const int32_t vec4_nt, const int32_t split,
                                                 const uint32_t *__restrict__ const ops_g, const size_t B) {
  rows_body<PHYS, false, EP2>(b, ep, h, tables, vec4_nt, split, ops_g, B);
}
template <bool EP2>
__global__ void __launch_bounds__(1024) k_rows_roll(const MpeBuffers b, const RowEpisode ep, const RowDims h,
                                                      const uint32_t *__restrict__ const tables, const int32_t vec4_nt,
                                                      const uint32_t *__restrict__ const ops_g, const size_t B, const RollArgs ra) {
  rows_body<true, false, EP2, true>(b, ep, h, tables, vec4_nt, 0, ops_g, B, ra);
}
#endif

}  // namespace

#ifdef MPE_ROWS_STATIC
// the five entry points of a compiled program: <name>_{s,r,e,l,m} = step, rows only, step with the episode end inside, T-step
// rollout, T-step rollout with the episode ends inside
#define MPE_ROWS_CAT2(a, b) a##b
#define MPE_ROWS_CAT(a, b) MPE_ROWS_CAT2(a, b)
#define MPE_ROWS_STATIC_KERNEL(suffix, PHYS, EP2)                                                                             \
  extern "C" __global__ void __launch_bounds__(static_waves<PHYS>() * kWave)                                                   \
      __attribute__((amdgpu_waves_per_eu(PHYS ? MPE_ROWS_STATIC_OCC_STEP : MPE_ROWS_STATIC_OCC_ROWS)))                         \
      MPE_ROWS_CAT(MPE_ROWS_STATIC_NAME, suffix)(const MpeBuffers b, const RowEpisode ep, const int32_t vec4_nt,              \
                                                 const size_t B) {                                                             \
    rows_body<PHYS, true, EP2>(b, ep, RowDims{}, nullptr, vec4_nt, 0, nullptr, B);                                             \
  }
MPE_ROWS_STATIC_KERNEL(_s, true, false)
MPE_ROWS_STATIC_KERNEL(_r, false, false)
MPE_ROWS_STATIC_KERNEL(_e, true, true)
#define MPE_ROWS_STATIC_ROLL_KERNEL(suffix, EP2)                                                                              \
  extern "C" __global__ void __launch_bounds__(static_waves<true>() * kWave)                                                   \
      __attribute__((amdgpu_waves_per_eu(MPE_ROWS_STATIC_OCC_STEP)))                                                           \
      MPE_ROWS_CAT(MPE_ROWS_STATIC_NAME, suffix)(const MpeBuffers b, const RowEpisode ep, const int32_t vec4_nt,              \
                                                 const size_t B, const RollArgs ra) {                                          \
    rows_body<true, true, EP2, true>(b, ep, RowDims{}, nullptr, vec4_nt, 0, nullptr, B, ra);                                   \
  }
MPE_ROWS_STATIC_ROLL_KERNEL(_l, false)
MPE_ROWS_STATIC_ROLL_KERNEL(_m, true)
#else

int launch_rows_header(const RowTables &t, void *dst, hipStream_t stream) {
  hipLaunchKernelGGL(k_rows_header, dim3(1), dim3(64), 0, stream, t, reinterpret_cast<RowTables *>(dst));
  return (int)hipGetLastError();
}

// LDS: the staged state + reward scratch, and one [64][d_max] tile + eight slots per wave -- as many waves as fit (each
// wave takes its share of the agents in turn), at most one per agent and 16; past 4 waves stay within 64 KB (two workgroups
// per CU).  One function for the launch and for the generator of compiled programs (the wave count is a constant there).
int rows_geometry(const RowDims &h, bool phys, int *waves, size_t *lds_bytes, int max_waves) {
  constexpr size_t kLdsCap = 160 * 1024;
  const size_t fixed = sizeof(float) * (size_t)(2 * h.n_entities + 2 * h.n_vel + h.n_agents + kRowPicks + (phys ? h.n_agents : 0) + h.n_shared) * kWave;
  const size_t new_state = phys ? sizeof(float) * (size_t)(4 * h.n_agents) * kWave : 0;      // shares the tiles' region
  const size_t tile = sizeof(float) * (size_t)kWave * (size_t)h.d_max, slots = sizeof(float) * (size_t)kWave * kRowSlots;
  auto need = [&](int w) { return fixed + (size_t)w * slots + ((size_t)w * tile > new_state ? (size_t)w * tile : new_state); };
  if (need(1) > kLdsCap) return MPE_EUNSUPPORTED;
  int W = h.n_agents < kRowMaxObsWaves ? h.n_agents : kRowMaxObsWaves;
  if (max_waves > 0 && W > max_waves) W = max_waves;
  while (W > 1 && need(W) > (W > 4 ? 64u * 1024u : kLdsCap)) --W;
  *waves = W;
  *lds_bytes = need(W);
  return 0;
}

static bool rows_nontemporal(const RowDims &h, const RowTables &host, int vec4, const RowEpisode &ep, size_t B) {
  const size_t row_bytes = (size_t)host.obs_off[h.n_agents] * sizeof(float) * B;
  return row_bytes >= (8u << 20) && vec4 && ep.enabled != 1;
}

int launch_rows(const MpeBuffers &b, const RowDims &h, const RowTables &host, const void *tables_device, bool phys, int vec4,
                const RowEpisode &ep, const int32_t *ops_device, size_t B, hipStream_t stream, const RollArgs *roll) {
  int W = 0;
  size_t lds = 0;
  // (episode mode, mpe_episode_finish, with ONE wave per 64 worlds -- most workgroups only read their flags and leave -- costs
  //  the same 3.3 us after a step as with W waves: that launch sits on the dependent-launch floor either way, and the step at
  //  which every world reaches the horizon wants the W waves: profiles/r4_finish_cost.txt)
  if (int rc = rows_geometry(h, phys, &W, &lds, 0)) return rc;
  // (two waves per agent, rows || reward -- `split` -- measured slower, see rows_body; the kernel keeps the switch for the A/B)
  const unsigned grid = (unsigned)((B + kWave - 1) / kWave);
  const bool nt = rows_nontemporal(h, host, vec4, ep, B);
  const bool ep2 = ep.enabled == 2;      // (phys by construction: mpe_step_rows_episode)
  auto fn = ep2 ? k_rows<true, true> : phys ? k_rows<true, false> : k_rows<false, false>;
  const int32_t vec4_nt = (vec4 ? 1 : 0) | (nt ? 2 : 0);
  if (lds > 64 * 1024) {
    const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return (int)rc;
  }
  if (roll) {
    auto fr = ep.enabled == 2 ? k_rows_roll<true> : k_rows_roll<false>;
    if (lds > 64 * 1024) {
      const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(fr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (rc != hipSuccess) return (int)rc;
    }
    hipLaunchKernelGGL(fr, dim3(grid), dim3(W * kWave), lds, stream, b, ep, h, reinterpret_cast<const uint32_t *>(tables_device),
                       vec4_nt, reinterpret_cast<const uint32_t *>(ops_device), B, *roll);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(W * kWave), lds, stream, b, ep, h, reinterpret_cast<const uint32_t *>(tables_device),
                     vec4_nt, (int32_t)0, reinterpret_cast<const uint32_t *>(ops_device), B);
  return (int)hipGetLastError();
}

// a compiled program (hipModule functions in the order s, r, e, l, m): same geometry, no tables, no ops
int launch_rows_image(void *const fns[5], const MpeBuffers &b, const RowDims &h, const RowTables &host, bool phys, int vec4,
                      const RowEpisode &ep, size_t B, hipStream_t stream, const RollArgs *roll) {
  int W = 0;
  size_t lds = 0;
  if (int rc = rows_geometry(h, phys, &W, &lds, 0)) return rc;
  const bool nt = rows_nontemporal(h, host, vec4, ep, B);
  hipFunction_t fn = static_cast<hipFunction_t>(fns[roll ? (ep.enabled == 2 ? 4 : 3) : ep.enabled == 2 ? 2 : phys ? 0 : 1]);
  MpeBuffers b_ = b;
  RowEpisode ep_ = ep;
  int32_t vec4_ = (vec4 ? 1 : 0) | (nt ? 2 : 0);
  size_t B_ = B;
  RollArgs ra_;
  if (roll) ra_ = *roll;
  void *args[] = {&b_, &ep_, &vec4_, &B_, &ra_};      // (the trailing RollArgs only exists in the rollout entry points)
  const unsigned grid = (unsigned)((B + kWave - 1) / kWave);
  return (int)hipModuleLaunchKernel(fn, grid, 1, 1, (unsigned)(W * kWave), 1, 1, 0u, stream, args, nullptr);      // (its LDS is static)
}
#endif

}  // namespace mpe
