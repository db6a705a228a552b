// mpe_narrow.hip -- thread-per-world kernels (small entity counts; the whole world lives in VGPRs).
//
// One lane owns one world: its E positions and A velocities sit in registers, every global
// load/store of the SoA state is a 256-byte coalesced wave access over the batch axis, and the
// five reference phases (action decode, action force, pairwise contact force, integrate,
// observation/reward assembly -- environment.py:80-104, core.py:117-196, scenarios/*.py) run
// back to back without touching HBM in between.  Row-major observation rows leave through the
// wave-private LDS transpose in mpe_device.h.  No MFMA: there is no contraction in this path.
#include "mpe_internal.h"

namespace mpe {

constexpr int kBlock = 256;

template <int KIND, int A, int L, int NADV>
struct ObsTile {  // LDS floats one wave needs for its widest observation row
  static constexpr int DC = 2;  // World.dim_c of simple_spread / simple_tag
  static constexpr int D =
      KIND == MPE_SCN_SIMPLE   ? 2 + 2 * L
      : KIND == MPE_SCN_SPREAD ? 4 + 2 * L + 2 * (A - 1) + DC * (A - 1)
      : KIND == MPE_SCN_TAG    ? 4 + 2 * L + 2 * (A - 1) + 2 * (A - NADV)
      : KIND == MPE_SCN_ADVERSARY ? 2 + 2 * L + 2 * (A - 1)
      : KIND == MPE_SCN_PUSH   ? 7 + 5 * L + 2 * (A - 1)
                               : 1;
  static constexpr int floats = kWave * (D | 1);  // >= kWave * tile_stride<D>()
};

// ---- scenario output stages: Scenario.observation / reward / benchmark_data per world ----------
template <int A, int L>
__device__ __forceinline__ void out_simple(const NarrowDesc &d, const MpeBuffers &b, size_t B, size_t w,
                                           size_t w0, int nvalid, int lane, bool live, float *tile,
                                           const float (&px)[A + L], const float (&py)[A + L],
                                           const float (&vx)[A], const float (&vy)[A]) {
  // simple.py:45-50 observation = [vel, landmark - pos ...]; :41-43 reward = -|pos - lm0|^2
  constexpr int D = 2 + 2 * L;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    float row[D];
    row[0] = vx[i];
    row[1] = vy[i];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      row[2 + 2 * l] = px[A + l] - px[i];
      row[3 + 2 * l] = py[A + l] - py[i];
    }
    store_rows<D>(tile, row, b.obs + B * d.obs_off[i] + w0 * D, nvalid, lane, d.vec4);
    if (b.rew && live) {
      const float dx = px[i] - px[A], dy = py[i] - py[A];
      const float sx = dx * dx, sy = dy * dy;
      b.rew[i * B + w] = -(sx + sy);
    }
    if (b.done && live) b.done[i * B + w] = 0;
  }
}

template <int A, int L>
__device__ __forceinline__ void out_spread(const NarrowDesc &d, const MpeBuffers &b, size_t B, size_t w,
                                           size_t w0, int nvalid, int lane, bool live, float *tile,
                                           const float (&px)[A + L], const float (&py)[A + L],
                                           const float (&vx)[A], const float (&vy)[A]) {
  constexpr int DC = 2;
  constexpr int D = 4 + 2 * L + 2 * (A - 1) + DC * (A - 1);
  // observation (simple_spread.py:84-100): [vel, pos, landmarks rel., other agents rel., comm zeros]
#pragma unroll
  for (int i = 0; i < A; ++i) {
    float row[D];
    row[0] = vx[i];
    row[1] = vy[i];
    row[2] = px[i];
    row[3] = py[i];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      row[4 + 2 * l] = px[A + l] - px[i];
      row[5 + 2 * l] = py[A + l] - py[i];
    }
    int k = 4 + 2 * L;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      if (j == i) continue;
      row[k++] = px[j] - px[i];
      row[k++] = py[j] - py[i];
    }
#pragma unroll
    for (int z = 0; z < DC * (A - 1); ++z) row[k + z] = 0.f;  // silent agents' state.c (core.py:173-174)
    store_rows<D>(tile, row, b.obs + B * d.obs_off[i] + w0 * D, nvalid, lane, d.vec4);
  }
  if (!b.rew && !b.info_rew) return;
  // reward (simple_spread.py:72-82), computed once per world instead of once per agent:
  //   landmark term  -sum_l min_a |a - l|   is the same for every agent
  //   collision term -#{a : |a - i| < r_a + r_i}  (counts i itself)
  float lm_term = 0.f, md = 0.f;
  int occupied = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    // min over agents on squared distances, then ONE square root (sqrt is monotone: min(dists) of the
    // reference is the distance of the nearest agent)
    float m2 = sq2d(px[0] - px[A + l], py[0] - py[A + l]);
#pragma unroll
    for (int a = 1; a < A; ++a) m2 = fminf(m2, sq2d(px[a] - px[A + l], py[a] - py[A + l]));
    const float m = fast_sqrt(m2);
    lm_term = lm_term - m;
    md = md + m;
    occupied += sqrt_lt(m2, 0.1f) ? 1 : 0;  // the integer output takes the exact test
  }
  float r[A];
  int cnt[A];
#pragma unroll
  for (int i = 0; i < A; ++i) {
    int c = 0;
    if ((d.collide >> i) & 1u) {
#pragma unroll
      for (int a = 0; a < A; ++a)
        c += sqrt_lt(sq2d(px[a] - px[i], py[a] - py[i]), d.size[a] + d.size[i]) ? 1 : 0;
    }
    cnt[i] = c;
    float ri = lm_term;
#pragma unroll
    for (int s = 0; s < A; ++s) ri = ri - (c > s ? 1.f : 0.f);  // rew -= 1 per contact, in sequence
    r[i] = ri;
  }
  // environment.py:100-102: reward = np.sum(reward_n) = r0 + (((0 + r1) + r2) + ...) for n < 9
  float rest = 0.f;
#pragma unroll
  for (int i = 1; i < A; ++i) rest += r[i];
  const float total = A > 1 ? r[0] + rest : r[0];
  if (!live) return;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    if (b.rew) b.rew[i * B + w] = d.collaborative ? total : r[i];
    if (b.done) b.done[i * B + w] = 0;
    if (b.info_rew) {  // benchmark_data (simple_spread.py:47-63)
      b.info_rew[i * B + w] = r[i];
      b.info_collisions[i * B + w] = cnt[i];
      b.info_min_dists[i * B + w] = md;
      b.info_occupied[i * B + w] = occupied;
    }
  }
}

template <int A, int L, int NADV>
__device__ __forceinline__ void out_tag(const NarrowDesc &d, const MpeBuffers &b, size_t B, size_t w,
                                        size_t w0, int nvalid, int lane, bool live, float *tile,
                                        const float (&px)[A + L], const float (&py)[A + L],
                                        const float (&vx)[A], const float (&vy)[A]) {
  constexpr int NG = A - NADV;
  // observation (simple_tag.py:131-147): [vel, pos, landmarks rel., others rel., good others' vel]
#pragma unroll
  for (int i = 0; i < A; ++i) {
    constexpr int DMAX = 4 + 2 * L + 2 * (A - 1) + 2 * NG;
    const bool adv = i < NADV;
    float row[DMAX];
    row[0] = vx[i];
    row[1] = vy[i];
    row[2] = px[i];
    row[3] = py[i];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      row[4 + 2 * l] = px[A + l] - px[i];
      row[5 + 2 * l] = py[A + l] - py[i];
    }
    int k = 4 + 2 * L;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      if (j == i) continue;
      row[k++] = px[j] - px[i];
      row[k++] = py[j] - py[i];
    }
#pragma unroll
    for (int j = NADV; j < A; ++j) {
      if (j == i) continue;
      row[k++] = vx[j];
      row[k++] = vy[j];
    }
    if (adv) {
      store_rows<DMAX>(tile, row, b.obs + B * d.obs_off[i] + w0 * DMAX, nvalid, lane, d.vec4);
    } else {
      constexpr int DG = DMAX - 2;
      float rg[DG];
#pragma unroll
      for (int c = 0; c < DG; ++c) rg[c] = row[c];
      store_rows<DG>(tile, rg, b.obs + B * d.obs_off[i] + w0 * DG, nvalid, lane, d.vec4);
    }
  }
  if (!b.rew && !b.info_collisions) return;
  // reward (simple_tag.py:84-129): is_collision(good, adversary) is a strict dist < r_g + r_v
  bool hit[NG][NADV];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int v = 0; v < NADV; ++v)
      hit[g][v] = sqrt_lt(sq2d(px[NADV + g] - px[v], py[NADV + g] - py[v]), d.size[NADV + g] + d.size[v]);
  float adv_rew = 0.f;  // adversary_reward :115-129 -- same value for every adversary
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int v = 0; v < NADV; ++v) adv_rew += hit[g][v] ? 10.f : 0.f;
  if (!live) return;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    float r;
    int c = 0;
    if (i < NADV) {
      r = ((d.collide >> i) & 1u) ? adv_rew : 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) c += hit[g][i] ? 1 : 0;
    } else {  // agent_reward :89-113
      r = 0.f;
      if ((d.collide >> i) & 1u) {
#pragma unroll
        for (int v = 0; v < NADV; ++v) r -= hit[i - NADV][v] ? 10.f : 0.f;
      }
      r -= tag_bound(fabsf(px[i]));
      r -= tag_bound(fabsf(py[i]));
    }
    if (b.rew) b.rew[i * B + w] = r;
    if (b.done) b.done[i * B + w] = 0;
    if (b.info_collisions) b.info_collisions[i * B + w] = c;  // benchmark_data :57-66
  }
}

// simple_adversary.py: observation :121-139, reward :76-118 (shaped).  g = this world's goal landmark.
template <int A, int L, int NADV>
__device__ __forceinline__ void out_adversary(const NarrowDesc &d, const MpeBuffers &b, size_t B, size_t w,
                                              size_t w0, int nvalid, int lane, bool live, float *tile,
                                              const float (&px)[A + L], const float (&py)[A + L]) {
  const int g = b.choice[w];
  float gx, gy;
  goal_pos<A, L>(px, py, g, gx, gy);
#pragma unroll
  for (int i = 0; i < A; ++i) {
    constexpr int DG = 2 + 2 * L + 2 * (A - 1), DA = DG - 2;
    float row[DG];
    int k = 0;
    if (i >= NADV) { row[k++] = gx - px[i]; row[k++] = gy - py[i]; }
#pragma unroll
    for (int l = 0; l < L; ++l) { row[k++] = px[A + l] - px[i]; row[k++] = py[A + l] - py[i]; }
#pragma unroll
    for (int j = 0; j < A; ++j) {
      if (j == i) continue;
      row[k++] = px[j] - px[i];
      row[k++] = py[j] - py[i];
    }
    if (i >= NADV) {
      store_rows<DG>(tile, row, b.obs + B * d.obs_off[i] + w0 * DG, nvalid, lane, d.vec4);
    } else {
      float ra[DA];
#pragma unroll
      for (int c = 0; c < DA; ++c) ra[c] = row[c];
      store_rows<DA>(tile, ra, b.obs + B * d.obs_off[i] + w0 * DA, nvalid, lane, d.vec4);
    }
  }
  if (!b.rew && !b.done) return;
  float d2g[A];
#pragma unroll
  for (int a = 0; a < A; ++a) d2g[a] = sq2d(px[a] - gx, py[a] - gy);
  float adv_rew = 0.f;   // sum over adversaries of their distance to the goal (:88)
#pragma unroll
  for (int a = 0; a < NADV; ++a) adv_rew = adv_rew + fast_sqrt(d2g[a]);
  float m2 = d2g[NADV];  // nearest good agent (:100-101)
#pragma unroll
  for (int a = NADV + 1; a < A; ++a) m2 = fminf(m2, d2g[a]);
  const float good_rew = -fast_sqrt(m2) + adv_rew;
  if (!live) return;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    if (b.rew) b.rew[i * B + w] = i < NADV ? -d2g[i] : good_rew;   // adversary: -|pos - goal|^2 (:113)
    if (b.done) b.done[i * B + w] = 0;
  }
}

// simple_push.py: observation :78-96, reward :60-76.
template <int A, int L, int NADV>
__device__ __forceinline__ void out_push(const NarrowDesc &d, const MpeBuffers &b, size_t B, size_t w,
                                         size_t w0, int nvalid, int lane, bool live, float *tile,
                                         const float (&px)[A + L], const float (&py)[A + L],
                                         const float (&vx)[A], const float (&vy)[A]) {
  const int g = b.choice[w];
  float gx, gy;
  goal_pos<A, L>(px, py, g, gx, gy);
#pragma unroll
  for (int i = 0; i < A; ++i) {
    constexpr int DG = 7 + 5 * L + 2 * (A - 1), DA = 2 + 2 * L + 2 * (A - 1);
    if (i >= NADV) {
      float row[DG];
      int k = 0;
      row[k++] = vx[i]; row[k++] = vy[i];
      row[k++] = gx - px[i]; row[k++] = gy - py[i];
#pragma unroll
      for (int c = 0; c < 3; ++c) row[k++] = (g + 1 == c) ? 0.75f : 0.25f;          // agent.color (:44-49)
#pragma unroll
      for (int l = 0; l < L; ++l) { row[k++] = px[A + l] - px[i]; row[k++] = py[A + l] - py[i]; }
#pragma unroll
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int c = 0; c < 3; ++c) row[k++] = (l + 1 == c) ? (float)(0.1 + 0.8) : 0.1f;   // landmark.color (:36-38)
#pragma unroll
      for (int j = 0; j < A; ++j) {
        if (j == i) continue;
        row[k++] = px[j] - px[i];
        row[k++] = py[j] - py[i];
      }
      store_rows<DG>(tile, row, b.obs + B * d.obs_off[i] + w0 * DG, nvalid, lane, d.vec4);
    } else {
      float row[DA];
      int k = 0;
      row[k++] = vx[i]; row[k++] = vy[i];
#pragma unroll
      for (int l = 0; l < L; ++l) { row[k++] = px[A + l] - px[i]; row[k++] = py[A + l] - py[i]; }
#pragma unroll
      for (int j = 0; j < A; ++j) {
        if (j == i) continue;
        row[k++] = px[j] - px[i];
        row[k++] = py[j] - py[i];
      }
      store_rows<DA>(tile, row, b.obs + B * d.obs_off[i] + w0 * DA, nvalid, lane, d.vec4);
    }
  }
  if (!b.rew && !b.done) return;
  float dg[A];
#pragma unroll
  for (int a = 0; a < A; ++a) dg[a] = fast_sqrt(sq2d(px[a] - gx, py[a] - gy));
  float m = dg[NADV];   // nearest good agent to the goal (:70-71)
#pragma unroll
  for (int a = NADV + 1; a < A; ++a) m = fminf(m, dg[a]);
  if (!live) return;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    if (b.rew) b.rew[i * B + w] = i < NADV ? m - dg[i] : -dg[i];
    if (b.done) b.done[i * B + w] = 0;
  }
}

// World.apply_environment_force (core.py:143-155): a < c over ALL entities; per entity the
// contributions arrive as action first, then partners in ascending order (Q9); pairs with both
// sides immovable are evaluated by the reference but applied to nobody (Q8) -- skipped here.
template <int A, int L>
__device__ __forceinline__ void pair_forces(const NarrowDesc &d, const float (&px)[A + L],
                                            const float (&py)[A + L], float (&fx)[A], float (&fy)[A]) {
  constexpr int E = A + L;
#pragma unroll
  for (int a = 0; a < E; ++a) {
#pragma unroll
    for (int c = a + 1; c < E; ++c) {
      const bool both = ((d.collide >> a) & (d.collide >> c) & 1u) != 0;
      const bool ma = a < A && ((d.movable >> a) & 1u);
      const bool mc = c < A && ((d.movable >> c) & 1u);
      if (both && (ma || mc)) {  // uniform: kernarg bits
        float gx, gy;
        contact_force(px[a] - px[c], py[a] - py[c], d.size[a] + d.size[c], d.cforce, d.cmargin, d.cmargin_inv, gx, gy);
        if (ma) { constexpr int z = 0; const int ia = a < A ? a : z; fx[ia] = gx + fx[ia]; fy[ia] = gy + fy[ia]; }
        if (mc) { constexpr int z = 0; const int ic = c < A ? c : z; fx[ic] = -gx + fx[ic]; fy[ic] = -gy + fy[ic]; }
      }
    }
  }
}

// ---- the fused kernel -----------------------------------------------------------------------
template <int KIND, int A, int L, int NADV, bool PHYS, bool OUT>
__global__ void __launch_bounds__(kBlock)
k_narrow(const NarrowDesc d, const MpeBuffers b, const size_t B) {
  constexpr int E = A + L;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const size_t w0 = (size_t)blockIdx.x * blockDim.x + (size_t)wave * kWave;
  if (w0 >= B) return;  // wave-uniform
  const int nvalid = (B - w0) < (size_t)kWave ? (int)(B - w0) : kWave;
  const bool live = lane < nvalid;
  const size_t w = live ? w0 + lane : B - 1;  // dead lanes shadow the last world, stores are masked

  float px[E], py[E], vx[A], vy[A];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    px[e] = b.pos[(size_t)(2 * e) * B + w];
    py[e] = b.pos[(size_t)(2 * e + 1) * B + w];
  }
#pragma unroll
  for (int i = 0; i < A; ++i) {
    vx[i] = b.vel[(size_t)(2 * i) * B + w];
    vy[i] = b.vel[(size_t)(2 * i + 1) * B + w];
  }

  if (PHYS) {
    // _set_action + apply_action_force (environment.py:144-181, core.py:134-140)
    float fx[A], fy[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
      float ux, uy;
      fetch_action(b, B, i, w, d.accel[i], ux, uy);
      fx[i] = ux + 0.f;
      fy[i] = uy + 0.f;
    }
    // apply_environment_force (core.py:143-155): a < b over all entities, partners ascending (Q9)
    pair_forces<A, L>(d, px, py, fx, fy);
    // integrate_state (core.py:158-169); update_agent_state (:171-177) zeroes comm of silent agents,
    // which the output stage emits as constant zeros.
#pragma unroll
    for (int i = 0; i < A; ++i) {
      if ((d.movable >> i) & 1u) {
        integrate_one(px[i], py[i], vx[i], vy[i], fx[i], fy[i], d.inv_mass[i], d.max_speed[i], d.damp, d.dt);
        if (live) {
          b.pos[(size_t)(2 * i) * B + w] = px[i];
          b.pos[(size_t)(2 * i + 1) * B + w] = py[i];
          b.vel[(size_t)(2 * i) * B + w] = vx[i];
          b.vel[(size_t)(2 * i + 1) * B + w] = vy[i];
        }
      }
    }
  }

  if constexpr (OUT) {
    float *tile = smem + wave * ObsTile<KIND, A, L, NADV>::floats;
    if constexpr (KIND == MPE_SCN_SIMPLE) out_simple<A, L>(d, b, B, w, w0, nvalid, lane, live, tile, px, py, vx, vy);
    if constexpr (KIND == MPE_SCN_SPREAD) out_spread<A, L>(d, b, B, w, w0, nvalid, lane, live, tile, px, py, vx, vy);
    if constexpr (KIND == MPE_SCN_TAG) out_tag<A, L, NADV>(d, b, B, w, w0, nvalid, lane, live, tile, px, py, vx, vy);
    if constexpr (KIND == MPE_SCN_ADVERSARY) out_adversary<A, L, NADV>(d, b, B, w, w0, nvalid, lane, live, tile, px, py);
    if constexpr (KIND == MPE_SCN_PUSH) out_push<A, L, NADV>(d, b, B, w, w0, nvalid, lane, live, tile, px, py, vx, vy);
  }
}

// ---- phase-level kernels (one reference function each; force lives in HBM between them) ---------
template <int A, int L, int PHASE>
__global__ void __launch_bounds__(kBlock)
k_phase(const NarrowDesc d, const MpeBuffers b, const size_t B) {
  constexpr int E = A + L;
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= B) return;
  if (PHASE == 0) {  // apply_action_force
#pragma unroll
    for (int i = 0; i < A; ++i) {
      float ux, uy;
      fetch_action(b, B, i, w, d.accel[i], ux, uy);
      b.force[(size_t)(2 * i) * B + w] = ux + 0.f;
      b.force[(size_t)(2 * i + 1) * B + w] = uy + 0.f;
    }
  } else if (PHASE == 1) {  // apply_environment_force / get_collision_force
    float px[E], py[E], fx[A], fy[A];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      px[e] = b.pos[(size_t)(2 * e) * B + w];
      py[e] = b.pos[(size_t)(2 * e + 1) * B + w];
    }
#pragma unroll
    for (int i = 0; i < A; ++i) {
      fx[i] = b.force[(size_t)(2 * i) * B + w];
      fy[i] = b.force[(size_t)(2 * i + 1) * B + w];
    }
    pair_forces<A, L>(d, px, py, fx, fy);
#pragma unroll
    for (int i = 0; i < A; ++i) {
      b.force[(size_t)(2 * i) * B + w] = fx[i];
      b.force[(size_t)(2 * i + 1) * B + w] = fy[i];
    }
  } else {  // integrate_state
#pragma unroll
    for (int i = 0; i < A; ++i) {
      if (!((d.movable >> i) & 1u)) continue;
      float px = b.pos[(size_t)(2 * i) * B + w], py = b.pos[(size_t)(2 * i + 1) * B + w];
      float vx = b.vel[(size_t)(2 * i) * B + w], vy = b.vel[(size_t)(2 * i + 1) * B + w];
      integrate_one(px, py, vx, vy, b.force[(size_t)(2 * i) * B + w], b.force[(size_t)(2 * i + 1) * B + w],
                    d.inv_mass[i], d.max_speed[i], d.damp, d.dt);
      b.pos[(size_t)(2 * i) * B + w] = px;
      b.pos[(size_t)(2 * i + 1) * B + w] = py;
      b.vel[(size_t)(2 * i) * B + w] = vx;
      b.vel[(size_t)(2 * i + 1) * B + w] = vy;
    }
  }
}

// ---- dispatch ---------------------------------------------------------------------------------
using NarrowFn = void (*)(const NarrowDesc, const MpeBuffers, const size_t);
struct NarrowEntry {
  int kind, A, L, nadv;
  NarrowFn step, observe;
  int lds_floats_per_wave;
};

#define MPE_SCN_ENTRY(KIND, A, L, NADV)                                              \
  { KIND, A, L, NADV, k_narrow<KIND, A, L, NADV, true, true>,                        \
    k_narrow<KIND, A, L, NADV, false, true>, ObsTile<KIND, A, L, NADV>::floats }
#define MPE_GEN_ENTRY(A, L) \
  { MPE_SCN_GENERIC, A, L, 0, k_narrow<MPE_SCN_GENERIC, A, L, 0, true, false>, nullptr, 0 }

static const NarrowEntry kNarrowTable[] = {
    MPE_SCN_ENTRY(MPE_SCN_SIMPLE, 1, 1, 0),
    MPE_SCN_ENTRY(MPE_SCN_SPREAD, 1, 1, 0), MPE_SCN_ENTRY(MPE_SCN_SPREAD, 2, 2, 0),
    MPE_SCN_ENTRY(MPE_SCN_SPREAD, 3, 3, 0), MPE_SCN_ENTRY(MPE_SCN_SPREAD, 4, 4, 0),
    MPE_SCN_ENTRY(MPE_SCN_SPREAD, 5, 5, 0), MPE_SCN_ENTRY(MPE_SCN_SPREAD, 6, 6, 0),
    MPE_SCN_ENTRY(MPE_SCN_TAG, 4, 2, 3), MPE_SCN_ENTRY(MPE_SCN_TAG, 2, 1, 1),
    MPE_SCN_ENTRY(MPE_SCN_TAG, 6, 3, 4),
    MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 3, 2, 1), MPE_SCN_ENTRY(MPE_SCN_PUSH, 2, 2, 1),
    // (simple_adversary at the other team sizes of round 3's grid: with the k_split entries, in the -DMPE_SPLIT_TEAM_GRID A/B build
    //  only -- those shapes step through their row programs, mpe_split.hip's table says why)
#ifdef MPE_SPLIT_TEAM_GRID
    MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 2, 1, 1), MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 3, 2, 2),
    MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 4, 3, 1), MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 4, 3, 2),
    MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 5, 4, 1), MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 5, 4, 2),
    MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 6, 5, 1), MPE_SCN_ENTRY(MPE_SCN_ADVERSARY, 6, 5, 2),
#endif
    MPE_GEN_ENTRY(1, 0), MPE_GEN_ENTRY(1, 1), MPE_GEN_ENTRY(1, 2), MPE_GEN_ENTRY(1, 3),
    MPE_GEN_ENTRY(2, 0), MPE_GEN_ENTRY(2, 1), MPE_GEN_ENTRY(2, 2), MPE_GEN_ENTRY(2, 3), MPE_GEN_ENTRY(2, 4),
    MPE_GEN_ENTRY(3, 0), MPE_GEN_ENTRY(3, 1), MPE_GEN_ENTRY(3, 2), MPE_GEN_ENTRY(3, 3), MPE_GEN_ENTRY(3, 4),
    MPE_GEN_ENTRY(4, 0), MPE_GEN_ENTRY(4, 1), MPE_GEN_ENTRY(4, 2), MPE_GEN_ENTRY(4, 3), MPE_GEN_ENTRY(4, 4),
    MPE_GEN_ENTRY(5, 0), MPE_GEN_ENTRY(5, 2), MPE_GEN_ENTRY(5, 5), MPE_GEN_ENTRY(6, 0), MPE_GEN_ENTRY(6, 3),
    MPE_GEN_ENTRY(6, 6),
};

static const NarrowEntry *find_narrow(int kind, int A, int L, int nadv) {
  for (const NarrowEntry &e : kNarrowTable)
    if (e.kind == kind && e.A == A && e.L == L && (kind < MPE_SCN_TAG || e.nadv == nadv)) return &e;
  return nullptr;
}

bool narrow_supports(int kind, int A, int L, int nadv) { return find_narrow(kind, A, L, nadv) != nullptr; }

int launch_narrow(NarrowOp op, int kind, int A, int L, int nadv, const NarrowDesc &d, const MpeBuffers &b,
                  size_t B, hipStream_t stream) {
  const NarrowEntry *e = find_narrow(kind, A, L, nadv);
  if (!e) return MPE_EUNSUPPORTED;
  NarrowFn fn = nullptr;
  size_t lds = 0;
  switch (op) {
    case NarrowOp::Step: fn = e->step; lds = (size_t)e->lds_floats_per_wave * (kBlock / kWave) * sizeof(float); break;
    case NarrowOp::Observe: fn = e->observe; lds = (size_t)e->lds_floats_per_wave * (kBlock / kWave) * sizeof(float); break;
  }
  if (!fn) return MPE_EUNSUPPORTED;
  const unsigned grid = (unsigned)((B + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), lds, stream, d, b, B);
  return (int)hipGetLastError();
}

// phase kernels exist for the generic-physics shapes (A,L) of the table above
template <int PHASE>
static NarrowFn phase_fn(int A, int L) {
#define MPE_PH(a, l) if (A == a && L == l) return k_phase<a, l, PHASE>;
  MPE_PH(1, 0) MPE_PH(1, 1) MPE_PH(2, 0) MPE_PH(2, 2) MPE_PH(3, 0) MPE_PH(3, 3) MPE_PH(4, 2) MPE_PH(4, 4)
  MPE_PH(5, 5) MPE_PH(6, 6) MPE_PH(6, 3)
#undef MPE_PH
  return nullptr;
}

int launch_phase(int phase, int A, int L, const NarrowDesc &d, const MpeBuffers &b, size_t B,
                 hipStream_t stream) {
  NarrowFn fn = phase == 0 ? phase_fn<0>(A, L) : phase == 1 ? phase_fn<1>(A, L) : phase_fn<2>(A, L);
  if (!fn) return MPE_EUNSUPPORTED;
  const unsigned grid = (unsigned)((B + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, stream, d, b, B);
  return (int)hipGetLastError();
}

}  // namespace mpe
