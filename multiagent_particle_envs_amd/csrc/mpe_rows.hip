// mpe_rows.hip -- the composable output stage: Scenario.observation / reward of a USER scenario as a small program the
// kernel interprets (mpe_rows), so that a scenario nobody wrote a kernel for still steps in two launches
// (mpe_world_step + mpe_rows) instead of the generic path's hundred-odd torch launches.
//
// What a reference observation is made of (simple_spread.py:84-100, simple_tag.py:131-147, simple_adversary.py:121-139,
// simple_push.py:78-96, simple_speaker_listener.py:69-92, simple_reference.py:63-83, simple_crypto.py:127-169,
// simple_world_comm.py:231-289): a concatenation of a handful of SEGMENT kinds -- own velocity / position, the offset to
// an entity (every landmark, the other agents of a team, the per-world goal), another agent's velocity or utterance, a
// one-hot / colour of a per-world pick, constants, and simple_world_comm's forest visibility.  And a reference reward is
// an ordered sum of a few TERM kinds -- (minimum) distances, strict-< contact tests, the boundary penalty, squared
// utterance errors -- in an order that fixes the rounding.  A program is that list, 16 bytes per op, built on the host
// from ObsSpec / RewardSpec objects (multiagent_particle_envs_amd/rowspec.py); no JIT, no code generation.
//
// Shape of the kernel (E = A + L <= 64): a workgroup is 64 worlds x W waves (W <= min(A, 16)), lane = world.
//   all waves    stage the worlds' state in LDS once -- pos [E][2][64], vel [n_vel][2][64]: coalesced 256-byte loads; every
//                later access is `lds[row * 64 + lane]` with a wave-uniform row (the program counter is wave-uniform:
//                all 64 lanes interpret the same op), i.e. conflict-free and scalar-addressed
//   wave w       agents w, w + W, ...: [World.step of the agent (mpe_step_rows)], its observation program -- each op appends
//                its columns to the wave's LDS tile ([64][D] row-major = the output segment), the tile leaves as contiguous
//                16-byte stores -- and its reward program (two accumulators, one value register, eight slots); behind one
//                barrier the shared-reward sum (environment.py:100-102) in the reference's order.  The next op is fetched
//                (one scalar 16-byte load) while the current one executes: an op costs its own LDS round trip, not two.
// Arithmetic is the device functions of mpe_device.h (sq2d, sqrt_lt, fast_sqrt, tag_bound) in program order: a built-in
// scenario written as a program reproduces its fused kernel bit for bit (tests/test_gpu_rowspec.py).
#include "mpe_internal.h"

namespace mpe {

namespace {

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

template <bool NT>
__device__ __forceinline__ void flush_tile(const float *tile, float *__restrict__ g, int D, int nvalid, int lane, bool vec4) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (vec4 && nvalid == kWave) {          // a full wave of worlds, aligned segment: 16 D float4 pieces, straight copies
    const int nq = 16 * D;
    for (int q = lane; q < nq; q += kWave)
      store_row4<NT ? kRowsNt : kRowsPlain>(g + 4 * q, *reinterpret_cast<const float4 *>(tile + 4 * q));
  } else {
    const int nfl = nvalid * D;
    for (int j = lane; j < nfl; j += kWave) g[j] = tile[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <bool NT, bool PHYS>
__global__ void __launch_bounds__(1024) k_rows(const MpeBuffers b, const RowHeader h, const RowPhys ph, const RowEpisode ep,
                                                 const int4 *__restrict__ const ops, const size_t B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = uni((int)(threadIdx.x >> 6));
  const int NW = uni((int)(blockDim.x >> 6));          // waves of the workgroup: wave w takes agents w, w + NW, ...
  const int A = h.n_agents, E = h.n_entities, NV = h.n_vel, DC = h.dim_c;
  const size_t w0 = (size_t)blockIdx.x * kWave;
  if (w0 >= B) return;
  const int nvalid = (B - w0) < (size_t)kWave ? (int)(B - w0) : kWave;
  const bool live = lane < nvalid;
  const unsigned ln = (unsigned)(live ? lane : nvalid - 1) & 63u;

  float *const S_pos = smem;                              // [E][2][64]
  float *const S_vel = S_pos + 2 * E * kWave;             // [NV][2][64]
  float *const S_rew = S_vel + 2 * NV * kWave;            // [A][64]    rewards before the shared sum
  float *const S_slot = S_rew + A * kWave;                // [NW][8][64] the reward programs' value slots, per wave
  int *const S_pick = reinterpret_cast<int *>(S_slot + (size_t)NW * kRowSlots * kWave);   // [4][64]  the per-world picks
  float *const S_new = reinterpret_cast<float *>(S_pick + kRowPicks * kWave);   // [A][4][64] post-step state (PHYS only)
  float *const tiles = S_new + (PHYS ? 4 * A * kWave : 0);   // [W][64 * Dmax]

  // ---- episode bookkeeping (mpe_episode_finish): count the step, find the worlds that finished, leave if none did ----------
  bool fin = false;
  if (ep.enabled) {
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

  // ---- stage the state (finished worlds: reset_world first -- the draws of mpe_reset for (seed, world, episode)) -----------
  for (int e = wave; e < E; e += NW) {
    float x = (b.pos + wave_off((size_t)(2 * e) * B + w0))[ln], y = (b.pos + wave_off((size_t)(2 * e + 1) * B + w0))[ln];
    if (fin) {
      reset_draw(ep.seed, gw, ep.episode, e, e < A ? 1.0f : ep.landmark_range, x, y);
      (b.pos + wave_off((size_t)(2 * e) * B + w0))[ln] = x;
      (b.pos + wave_off((size_t)(2 * e + 1) * B + w0))[ln] = y;
    }
    S_pos[(2 * e) * kWave + lane] = x;
    S_pos[(2 * e + 1) * kWave + lane] = y;
    if (e < NV) {
      float vx = (b.vel + wave_off((size_t)(2 * e) * B + w0))[ln], vy = (b.vel + wave_off((size_t)(2 * e + 1) * B + w0))[ln];
      if (fin && e < A) {
        vx = vy = 0.f;
        (b.vel + wave_off((size_t)(2 * e) * B + w0))[ln] = 0.f;
        (b.vel + wave_off((size_t)(2 * e + 1) * B + w0))[ln] = 0.f;
      }
      S_vel[(2 * e) * kWave + lane] = vx;
      S_vel[(2 * e + 1) * kWave + lane] = vy;
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
  __syncthreads();

  auto P = [&](int e, int c) { return S_pos[(2 * e + c) * kWave + lane]; };
  auto V = [&](int e, int c) { return e < NV ? S_vel[(2 * e + c) * kWave + lane] : 0.f; };
  auto pick = [&](int k) { return S_pick[k * kWave + lane]; };
  auto word = [&](int j, int c) { return (fin || !b.comm) ? 0.f : (b.comm + wave_off(((size_t)j * B + w0) * DC))[ln * DC + c]; };
  // inside region r (a landmark, e.g. a forest of simple_world_comm.py:231-261): strict |e - region| < size_e + size_region
  auto in_region = [&](int e, int r) {
    const int f = h.region_entity[r];
    return sqrt_lt(sq2d(P(e, 0) - P(f, 0), P(e, 1) - P(f, 1)), h.size[e] + h.size[f]);
  };

  if constexpr (PHYS) {
    // ---- World.step (core.py:117-169) by the agent waves: action force, contacts with every other entity in ascending order
    // (Q9) from the PRE-step positions, integration -- the device functions and the order of the step kernels, so the state
    // that leaves here is theirs to the bit.  New state -> S_new; behind the barrier it replaces the staged one and goes
    // back to HBM (no wave still reads pre-step positions then).
    {
      for (int i = wave; i < A; i += NW) {
        float mx = P(i, 0), my = P(i, 1), mvx = V(i, 0), mvy = V(i, 1);
        if ((ph.movable >> i) & 1ull) {
          float ux, uy;
          fetch_action_wave(b, B, i, w0, ln, ph.accel[i], ux, uy);
          float fx = ux + 0.f, fy = uy + 0.f;
          if ((ph.collide >> i) & 1ull) {
            for (int j = 0; j < E; ++j) {
              if (j == i || !((ph.collide >> j) & 1ull)) continue;      // uniform
              float cx, cy;
              contact_force(mx - P(j, 0), my - P(j, 1), h.size[i] + h.size[j], ph.cforce, ph.cmargin, ph.cmargin_inv, cx, cy);
              fx = cx + fx;
              fy = cy + fy;
            }
          }
          integrate_one(mx, my, mvx, mvy, fx, fy, ph.inv_mass[i], ph.max_speed[i], ph.damp, ph.dt);
        }
        S_new[(4 * i + 0) * kWave + lane] = mx;
        S_new[(4 * i + 1) * kWave + lane] = my;
        S_new[(4 * i + 2) * kWave + lane] = mvx;
        S_new[(4 * i + 3) * kWave + lane] = mvy;
      }
    }
    __syncthreads();
    for (int i = wave; i < A; i += NW) {
      const float mx = S_new[(4 * i + 0) * kWave + lane], my = S_new[(4 * i + 1) * kWave + lane];
      const float mvx = S_new[(4 * i + 2) * kWave + lane], mvy = S_new[(4 * i + 3) * kWave + lane];
      S_pos[(2 * i) * kWave + lane] = mx;
      S_pos[(2 * i + 1) * kWave + lane] = my;
      S_vel[(2 * i) * kWave + lane] = mvx;
      S_vel[(2 * i + 1) * kWave + lane] = mvy;
      if (live && ((ph.movable >> i) & 1ull)) {
        (b.pos + wave_off((size_t)(2 * i) * B + w0))[ln] = mx;
        (b.pos + wave_off((size_t)(2 * i + 1) * B + w0))[ln] = my;
        (b.vel + wave_off((size_t)(2 * i) * B + w0))[ln] = mvx;
        (b.vel + wave_off((size_t)(2 * i + 1) * B + w0))[ln] = mvy;
      }
    }
    __syncthreads();
  }

  // ---- observation programs of this wave's agents ------------------------------------------------------------------------
  {
    float *const tile = tiles + (size_t)wave * kWave * h.d_max;
    for (int i = wave; i < A; i += NW) {
      const int D = h.obs_off[i + 1] - h.obs_off[i];
      if (D == 0) continue;
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
      const int pc0 = h.obs_begin[i], pc1 = h.obs_begin[i + 1];
      int4 nxt = pc0 < pc1 ? ops[pc0] : make_int4(0, 0, 0, 0);
      for (int pc = pc0; pc < pc1; ++pc) {
        const int4 op = nxt;
        if (pc + 1 < pc1) nxt = ops[pc + 1];        // in flight while this op executes
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
          case ROW_OBS_IN_REGION: row[col] = ((inmask >> (2 * e + a1)) & 1ull) ? 1.f : -1.f; col += 1; break;
          default: break;
        }
      }
      flush_tile<NT>(tile, b.obs + B * (size_t)h.obs_off[i] + w0 * (size_t)D, D, nvalid, lane, h.vec4 != 0);
    }
  }
  if (ep.enabled) return;      // (mpe_episode_finish: rewards and dones belong to the step that just ran)

  // ---- reward programs of this wave's agents -----------------------------------------------------------------------------
  if (b.rew) {
    float *const slot = S_slot + (size_t)wave * kRowSlots * kWave;      // this wave's eight value slots
    for (int i = wave; i < A; i += NW) {
      float acc[2] = {0.f, 0.f}, v = 0.f;
      const int pc0 = h.rew_begin[i], pc1 = h.rew_begin[i + 1];
      int4 nxt = pc0 < pc1 ? ops[pc0] : make_int4(0, 0, 0, 0);
      for (int pc = pc0; pc < pc1; ++pc) {
        const int4 op = nxt;
        if (pc + 1 < pc1) nxt = ops[pc + 1];
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
            const bool hit = sqrt_lt(sq2d(P(a0, 0) - P(a1, 0), P(a0, 1) - P(a1, 1)), h.size[a0] + h.size[a1]);
            const float t = hit ? f : 0.f;
            if (a2 & 1) acc[1] = acc[1] + t; else acc[0] = acc[0] + t;
            break;
          }
          case ROW_R_ADD_ACC: acc[0] = acc[0] + acc[1]; break;
          case ROW_R_STORE: S_rew[a0 * kWave + lane] = acc[0]; break;
          default: break;
        }
      }
    }
    if (h.collaborative) __syncthreads();      // (uniform: a kernel argument)
    // environment.py:100-102: every agent gets np.sum(reward_n) = r0 + (((0 + r1) + r2) + ...) for n < 9
    float total = 0.f;
    if (h.collaborative) {
      float rest = 0.f;
      for (int a = 1; a < A; ++a) rest += S_rew[a * kWave + lane];
      total = A > 1 ? S_rew[lane] + rest : S_rew[lane];
    }
    if (live)
      for (int i = wave; i < A; i += NW) (b.rew + wave_off((size_t)i * B + w0))[ln] = h.collaborative ? total : S_rew[i * kWave + lane];
  }
  if (b.done && live)
    for (int i = wave; i < A; i += NW) (b.done + wave_off((size_t)i * B + w0))[ln] = 0;
}

}  // namespace

int launch_rows(const MpeBuffers &b, const RowHeader &h, const RowPhys &ph, const RowEpisode &ep, const int32_t *ops_device, size_t B,
                hipStream_t stream) {
  // LDS: the staged state + reward scratch, and one [64][d_max] tile per observation wave -- as many waves as fit (each
  // wave takes its share of the agents in turn), at most one per agent and 15 (+ the reward wave = 1024 threads)
  constexpr size_t kLdsCap = 160 * 1024;
  const size_t fixed = sizeof(float) * (size_t)(2 * h.n_entities + 2 * h.n_vel + h.n_agents + kRowPicks +
                                                (ph.enabled ? 4 * h.n_agents : 0)) * kWave;
  const size_t tile = sizeof(float) * (size_t)kWave * ((size_t)h.d_max + kRowSlots);      // a wave's row tile + its eight slots
  if (fixed + tile > kLdsCap) return MPE_EUNSUPPORTED;
  int W = h.n_agents < kRowMaxObsWaves ? h.n_agents : kRowMaxObsWaves;
  while (W > 1 && fixed + (size_t)W * tile > (W > 4 ? 64u * 1024u : kLdsCap)) --W;   // (past 4 waves, stay within 64 KB: two workgroups per CU)
  const size_t lds = fixed + (size_t)W * tile;
  const unsigned grid = (unsigned)((B + kWave - 1) / kWave);
  const size_t row_bytes = (size_t)h.obs_off[h.n_agents] * sizeof(float) * B;
  const int4 *ops = reinterpret_cast<const int4 *>(ops_device);
  const bool nt = row_bytes >= (8u << 20) && h.vec4 && !ep.enabled;
  auto fn = ph.enabled ? (nt ? k_rows<true, true> : k_rows<false, true>) : (nt ? k_rows<true, false> : k_rows<false, false>);
  if (lds > 64 * 1024) {
    const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (rc != hipSuccess) return (int)rc;
  }
  hipLaunchKernelGGL(fn, dim3(grid), dim3(W * kWave), lds, stream, b, h, ph, ep, ops, B);
  return (int)hipGetLastError();
}

}  // namespace mpe
