// mpe_wide.hip -- wave-per-world kernel for large entity counts (simple_spread N=64: 128 entities).
//
// One WAVE owns one world; lane = agent (agents lane, lane+64, ... when A > 64).  The world's
// positions and velocities are staged once in a wave-private LDS block (2 KiB at N=64) and every
// phase of the step runs out of it with no workgroup barrier at all -- the only synchronisation is
// the wave's own program order plus LDS fences.  A 256-thread workgroup is four independent waves
// (four consecutive worlds) that merely share the per-entity constant table.
//
// Why a wave and not a workgroup per world (the first version): with a workgroup per world every
// phase ends in __syncthreads and all co-resident workgroups run their phases in lock-step, so the
// physics (latency-bound, ~40 us at B=4096) and the observation stores (~90 us) added up.  A wave
// per world has no barriers, four to eight worlds per SIMD slide against each other, and the whole
// batch is resident at once: the kernel is the store stream plus one world's latency.
//
//   contacts   lane i, two passes over the collidable entities (ascending): pass 1 marks
//              the partners close enough to exert a non-zero force (squared distance under
//              (r_i + r_j + 20k)^2: beyond it the fp32 soft-plus term is exactly 0) in a 64-bit mask,
//              pass 2 evaluates only those.  ~4 of 63 partners at N=64.
//   obs        98 % of the HBM bytes (A rows x D floats = 98 KiB per world at N=64).  Rows are
//              computed straight from LDS in output order: consecutive lanes write consecutive
//              16-byte pieces of a row, a full 1 KiB per wave store.
//   reward     lane l: min over agents of the SQUARED distance to landmark l (sqrt is monotone: one
//              sqrt per landmark afterwards), lane i: contact count of agent i with the exact
//              sqrt_lt test; sums by wave64 shuffle reductions.
// State loads are 4-byte accesses 4*B bytes apart (the SoA layout is batch-innermost), so a 128-byte
// line serves 32 neighbouring worlds: the world -> workgroup map keeps each such group of worlds on
// ONE XCD (blockIdx % 8 selects the XCD), so that the line is fetched into one L2, not eight.
// Per pair the arithmetic is mpe_device.h's, as in the small-N kernels; sums over landmarks / agents
// are reduction trees (documented in DESIGN.md 4).

#include <cstring>

#include "mpe_internal.h"

// tuning knobs of k_duo (A/B builds override them with -D): share of the rows wave 0 emits, and an off switch
#ifndef MPE_DUO_SPLIT_NUM
#define MPE_DUO_SPLIT_NUM 5
#define MPE_DUO_SPLIT_DEN 8
#endif
#ifndef MPE_DUO_ENABLE
#define MPE_DUO_ENABLE 1
#endif
#ifndef MPE_MULTI_ABLATE   // k_multi ablation builds: bit 0 skip the reward, bit 1 the contact loop, bit 2 the rows
#define MPE_MULTI_ABLATE 0
#endif
#ifndef MPE_DUO_MIN_N
#define MPE_DUO_MIN_N 32   // k_duo serves max(A, L) above this (below: k_multi, several worlds per wave)
#endif
#ifndef MPE_DUO_ROLL
#define MPE_DUO_ROLL 1    // 0: the rollout of 33..64-agent spread stays on k_wave<ROLL>
#endif
#ifndef MPE_MULTI_ROLL_MAX_N
#define MPE_MULTI_ROLL_MAX_N 24   // the rollout runs on k_multi<ROLL> up to this max(A, L), on k_wave<ROLL> above (measured)
#endif
#ifndef MPE_DUO_G
#define MPE_DUO_G 4   // worlds per workgroup of k_duo (1, 2, 4 or 8)
#endif
#ifndef MPE_DUO_LDS_PAD
#define MPE_DUO_LDS_PAD 0   // A/B builds: unused dynamic LDS bytes per k_duo workgroup (caps the workgroups resident per CU)
#endif
// ablation builds (tools/ab_build.sh): bit 0 skip the reward, bit 1 skip the contact loop, bit 2 skip the row stores
#ifndef MPE_DUO_ABLATE
#define MPE_DUO_ABLATE 0
#endif
namespace mpe {

namespace {

constexpr int kWavesPerWg = 4;
constexpr int kMovable = 1, kCollide = 2;
// Beyond dist_min + kFarX * contact_margin the fp32 soft-plus term is exactly zero:
// x = (dist_min - d)/k < -16.64 => 1 + exp(x) rounds to 1 => log = 0 => penetration = 0 => force +-0.
constexpr float kFarX = 20.0f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
// order LDS traffic between lanes of this wave (no workgroup barrier anywhere in this file)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
// floats of one step's observation block of simple_tag (all agents' rows of all worlds)
__host__ __device__ inline size_t obs_stride_tag(int A, int L, int nadv, size_t B) {
  const int NG = A - nadv, DA = 4 + 2 * L + 2 * (A - 1) + 2 * NG;
  return B * ((size_t)nadv * DA + (size_t)NG * (DA - 2));
}

// LDS carve-up.  Shared by the workgroup (constants, staged once): sizeq, crank/csz, agent constants.
// Private to each wave: the world.  Positions sit in OBSERVATION order
//   Q = [ landmark 0 .. L-1 | agent 0 .. A-1 ]
// so that the observation row of agent i is  Q[idx + (idx >= L+i)] - Q[L+i]  for idx = 0 .. L+A-2.
constexpr size_t kMaxLds = 160 * 1024;   // LDS of one gfx950 CU
struct Carve {
  size_t sizeq, crank, csz, aconst, shared_bytes;  // shared block offsets
  size_t q, v, u, qn, cpw, wave_bytes;             // per-wave block offsets (relative to the wave's base)
};
__host__ __device__ inline Carve carve(int A, int L) {
  const int E = A + L;
  Carve c;
  size_t o = 0;
  c.sizeq = o;  o += align16(sizeof(float) * E);       // Entity.size by Q index
  c.crank = o;  o += align16(sizeof(int) * E);         // entity e -> its rank among the collidable entities (ascending e), or -1
  c.csz = o;    o += align16(sizeof(float) * E);       // Entity.size of the collidable entities, by rank
  c.aconst = o; o += align16(sizeof(float4) * A);      // per agent: (1/mass, max_speed, sensitivity, flags)
  c.shared_bytes = o;
  o = 0;
  c.q = o;  o += align16(sizeof(float2) * E);
  c.v = o;  o += align16(sizeof(float2) * A);
  c.u = o;  o += align16(sizeof(float2) * A);          // action force; later the contact counts
  c.qn = o; o += align16(sizeof(float2) * (A > kWave ? A : 0));  // new positions while A > 64 agents integrate in batches
  c.cpw = o; o += align16(sizeof(float2) * E);         // positions of the collidable entities, by rank (the contact partner list)
  c.wave_bytes = o;
  return c;
}

// One world's observation rows (simple_spread.py:84-100), row by row:
//   row i = [vel_i | pos_i | landmark_l - pos_i ... | pos_j - pos_i (j != i, ascending) ... | zeros]
//         = [vel_i | pos_i | Q[idx + (idx >= L+i)] - Q[L+i]  for idx = 0 .. L+A-2 | zeros]
// The row index is wave-uniform, so the row base address, pos_i, vel_i and the self-skip threshold
// live in scalar registers / broadcast LDS reads, and a lane's share of the row -- pieces
// lane, lane+64, ... of VEC floats -- costs two LDS reads, four subtractions and one store per piece
// with no per-lane address arithmetic: consecutive lanes write consecutive pieces, 1 KiB (VEC = 4) per
// wave store.  Wave-pieces that lie entirely in the zero tail skip the LDS reads.
template <int VEC>
__device__ __forceinline__ void emit_rows(const float2 *Q, const float2 *V, int A, int L, int D,
                                          float *obs_w, size_t rowlen, int lane, int i_begin = 0, int i_end = -1) {
  constexpr int PP = VEC / 2;       // (x, y) pairs per piece
  const int E = A + L;
  const int P = D / VEC;            // pieces per row
  const int kpz = 2 + L + (A - 1);  // first all-zero pair
  const int K = (P + kWave - 1) / kWave;
  if (i_end < 0) i_end = A;
  float2 me = Q[L + min(i_begin, A - 1)], vel = V[min(i_begin, A - 1)];
  for (int i = i_begin; i < i_end; ++i) {
    const int thr = L + i;
    float *const row = obs_w + (size_t)i * rowlen;  // wave-uniform
    const int inext = min(i + 1, A - 1);
    const float2 me_n = Q[L + inext], vel_n = V[inext];  // next row's operands: in flight under this row
    for (int k = 0; k < K; ++k) {
      const int p = lane + kWave * k;
      float o[VEC];
      if (PP * kWave * k >= kpz) {  // uniform: the whole wave-piece is zero tail
#pragma unroll
        for (int m = 0; m < VEC; ++m) o[m] = 0.f;
      } else {
        const bool tail = PP * kWave * (k + 1) > kpz;  // uniform: some lanes reach the zero tail
#pragma unroll
        for (int h = 0; h < PP; ++h) {
          const int kp = PP * p + h;
          const int idx = kp - 2;
          int sidx = idx + (idx >= thr ? 1 : 0);
          if (k == 0) sidx = max(sidx, 0);
          if (tail) sidx = min(sidx, E - 1);
          const float2 pj = Q[sidx];
          float2 v = make_float2(pj.x - me.x, pj.y - me.y);
          if (tail && kp >= kpz) v = make_float2(0.f, 0.f);
          o[2 * h] = v.x;
          o[2 * h + 1] = v.y;
        }
        if (k == 0) {  // the row's header
          if (VEC == 4) {
            if (p == 0) { o[0] = vel.x; o[1] = vel.y; o[2] = me.x; o[3] = me.y; }
          } else {
            if (p == 0) { o[0] = vel.x; o[1] = vel.y; }
            if (p == 1) { o[0] = me.x; o[1] = me.y; }
          }
        }
      }
      if (p < P) {
        if (VEC == 4) *reinterpret_cast<float4 *>(row + (unsigned)(4 * p)) = make_float4(o[0], o[1], o[2], o[3]);
        else          *reinterpret_cast<float2 *>(row + (unsigned)(2 * p)) = make_float2(o[0], o[1]);
      }
    }
    me = me_n;
    vel = vel_n;
  }
}

// The same rows when a row has at most 128 16-byte pieces (D <= 512: every spread shape up to N = 85).
// What a lane reads for its pieces does not depend on the row except for the self-skip: piece p covers
// pairs kp = 2p, 2p+1, i.e. Q[idx] or Q[idx+1] for idx = kp-2.  Both candidates of both pairs of both
// pieces are fetched ONCE per world into registers (16 VGPRs); a row is then, per pair, one compare
// against the uniform threshold L+i, two selects and two subtractions -- no LDS traffic and no address
// arithmetic in the row loop beyond the broadcast read of (pos_i, vel_i) for the next row.
template <int RP /* row-store policy (mpe_device.h): kRowsNt when a row is a whole number of lines, else kRowsPlain */>
__device__ __forceinline__ void emit_rows_fast(const float2 *Q, const float2 *V, int A, int L, int D,
                                               float *obs_w, size_t rowlen, int lane, int i_begin = 0, int i_end = -1) {
  if (i_end < 0) i_end = A;
  const int E = A + L;
  const int P = D >> 2;             // 16-byte pieces per row, <= 128
  const int kpz = 2 + L + (A - 1);  // first all-zero pair
  const bool two = P > kWave;       // uniform: rows longer than one wave store
  float2 ca[2][2], cb[2][2];
  int idx[2][2];
  bool live[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kp = 2 * (lane + kWave * k) + h;
      idx[k][h] = kp - 2;
      live[k][h] = kp < kpz;
      ca[k][h] = Q[min(max(kp - 2, 0), E - 1)];
      cb[k][h] = Q[min(max(kp - 1, 0), E - 1)];
    }
  const bool st0 = lane < P, st1 = lane + kWave < P;
  const bool tail0 = 2 * kWave > kpz;  // uniform: the first wave store already reaches the zero tail
  float2 me = Q[L + min(i_begin, A - 1)], vel = V[min(i_begin, A - 1)];
  for (int i = i_begin; i < i_end; ++i) {
    const int thr = L + i;
    float *const row = obs_w + (size_t)i * rowlen;  // wave-uniform
    const int inext = min(i + 1, A - 1);
    const float2 me_n = Q[L + inext], vel_n = V[inext];  // next row's operands: in flight under this row
    {
      float4 o;
      const float2 s0 = idx[0][0] >= thr ? cb[0][0] : ca[0][0];
      const float2 s1 = idx[0][1] >= thr ? cb[0][1] : ca[0][1];
      o.x = s0.x - me.x; o.y = s0.y - me.y;
      o.z = s1.x - me.x; o.w = s1.y - me.y;
      if (tail0) {
        if (!live[0][0]) { o.x = 0.f; o.y = 0.f; }
        if (!live[0][1]) { o.z = 0.f; o.w = 0.f; }
      }
      if (lane == 0) o = make_float4(vel.x, vel.y, me.x, me.y);  // the row's header
      if (st0) store_row4<RP>(row + (unsigned)(4 * lane), o);
    }
    if (two) {
      float4 o;
      const float2 s0 = idx[1][0] >= thr ? cb[1][0] : ca[1][0];
      const float2 s1 = idx[1][1] >= thr ? cb[1][1] : ca[1][1];
      o.x = s0.x - me.x; o.y = s0.y - me.y;
      o.z = s1.x - me.x; o.w = s1.y - me.y;
      if (!live[1][0]) { o.x = 0.f; o.y = 0.f; }
      if (!live[1][1]) { o.z = 0.f; o.w = 0.f; }
      if (st1) store_row4<RP>(row + (unsigned)(4 * (lane + kWave)), o);
    }
    me = me_n;
    vel = vel_n;
  }
}

// One world's simple_tag observation rows (simple_tag.py:131-147), any team sizes:
//   row i = [vel_i | pos_i | landmark_l - pos_i ... | pos_j - pos_i (j != i, ascending) ... | vel_g of the good
//            agents g != i, ascending]                 -- adversaries i < nadv see NG velocities, good agents NG - 1
// i.e. spread's row with the zero tail replaced by raw velocities, and ragged widths DA / DA - 2: agent i's
// [B][D_i] block starts at float offset B * (i * DA) for adversaries, B * (nadv * DA + (i - nadv) * DG) for the
// good agents.  8-byte pieces (D_i is even, never both widths a multiple of 4): lane -> pair kp, 512 B per wave store.
__device__ __forceinline__ void emit_rows_tag(const float2 *Q, const float2 *V, int A, int L, int nadv, float *obs,
                                              size_t B, size_t w, int lane) {
  const int NG = A - nadv, DA = 4 + 2 * L + 2 * (A - 1) + 2 * NG, DG = DA - 2;
  const int kpz = 2 + L + (A - 1);   // first velocity pair
  for (int i = 0; i < A; ++i) {
    const bool adv = i < nadv;       // uniform
    const int Di = adv ? DA : DG;
    const size_t off = adv ? (size_t)i * DA : (size_t)nadv * DA + (size_t)(i - nadv) * DG;
    float *const row = obs + B * off + w * (size_t)Di;   // wave-uniform
    const float2 me = Q[L + i], vel = V[i];
    const int thr = L + i, P = Di >> 1;
    for (int kp = lane; kp < P; kp += kWave) {
      float2 o;
      if (kp >= kpz) {
        int g = kp - kpz;
        if (!adv) g += (g >= i - nadv) ? 1 : 0;
        o = V[nadv + g];
      } else if (kp >= 2) {
        const int idx = kp - 2;
        const float2 pj = Q[idx + (idx >= thr ? 1 : 0)];
        o = make_float2(pj.x - me.x, pj.y - me.y);
      } else {
        o = kp == 0 ? vel : me;
      }
      *reinterpret_cast<float2 *>(row + 2 * kp) = o;
    }
  }
}

// Pass 1 of the contact phase for up to 32 partners CPW[kb .. kb+n): bit 31-j of the result is set when
// partner kb+j is within reach (squared distance under (r_i + r_j + far)^2).  Partner data come from
// broadcast LDS reads at compile-time offsets of a uniform base (no address arithmetic); the mask is
// accumulated with one shift-or per partner.  UNI: all collidable entities share one size, so the
// squared reach is a per-lane constant.
template <bool UNI>
__device__ __forceinline__ unsigned near_mask32(const float2 *CPW, const float *csz, int kb, int n, float2 me,
                                                float rfar, float reach2_u) {
  unsigned near = 0u;
#pragma unroll 8
  for (int j = 0; j < n; ++j) {
    const float2 pj = CPW[kb + j];
    float r2 = reach2_u;
    if (!UNI) {
      const float r = rfar + csz[kb + j];
      r2 = r * r;
    }
    const bool hit = sq2d(me.x - pj.x, me.y - pj.y) < r2;
    near = (near << 1) | (hit ? 1u : 0u);
  }
  return near << (32 - n);
}

// sqrt_lt with the bounds of the guard band precomputed (m2 = m*m, lo = m2 (1 - 4e-7), hi = m2 (1 + 4e-7))
__device__ __forceinline__ bool sqrt_lt_pre(float s2, float m, float lo, float hi, bool has_band) {
  const bool below = s2 < lo;
  const bool above = s2 > hi && has_band;
  bool r = below;
  if (!below && !above) r = sqrtf(s2) < m;
  return r;
}

// ROLL: the fused T-step rollout (mpe_rollout_random) -- the world stays in LDS across the steps, moves and resets
// are drawn in-kernel (Philox, as mpe_random_actions / mpe_reset draw them), every step's rows / rewards are still
// written.  A wave runs ahead into step t+1's contact phase while its step-t row stores drain, so the VALU-bound
// phases that a single step cannot hide (every wave owns exactly one world at B = 4096) disappear under the stores.
template <bool PHYS, bool OUT, bool ROLL>
__global__ void __launch_bounds__(kWavesPerWg *kWave)
k_wave(const WideDesc d, const MpeBuffers b, const size_t B, const unsigned n_groups_padded, const RollArgs ra) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int A = d.A, L = d.L, E = A + L;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Carve cv = carve(A, L);
  float *const sizeq = reinterpret_cast<float *>(smem + cv.sizeq);
  int *const crank = reinterpret_cast<int *>(smem + cv.crank);
  float *const csz = reinterpret_cast<float *>(smem + cv.csz);
  float4 *const aconst = reinterpret_cast<float4 *>(smem + cv.aconst);
  char *const wbase = smem + cv.shared_bytes + (size_t)wave * cv.wave_bytes;
  float2 *const Q = reinterpret_cast<float2 *>(wbase + cv.q);
  float2 *const V = reinterpret_cast<float2 *>(wbase + cv.v);
  float2 *const U = reinterpret_cast<float2 *>(wbase + cv.u);
  int *const CNT = reinterpret_cast<int *>(wbase + cv.u);
  float2 *const QN = A > kWave ? reinterpret_cast<float2 *>(wbase + cv.qn) : Q + L;

  // ---- per-entity constants: staged once per workgroup, the only __syncthreads of the kernel ------
  // (identical agents, non-colliding landmarks -- the shipped spread at any N: straight from the kernel arguments, no
  //  global load in front of the first world; k_multi, same change: 24.5 vs 25.3 us at N=16, 30.2 vs 34.2 at N=8 B=65536)
  const float *tab = b.entity_table;  // [6][E]: size, mass, accel, max_speed, movable, collide
  int nC = 0;
  bool agents_only = true, one_csize = true, one_asize = true;
  if (d.homo) {
    const bool coll = d.a_flags & kCollide;
    for (int e = tid; e < E; e += blockDim.x) {
      sizeq[e < A ? L + e : e - A] = e < A ? d.a_size : 0.f;   // (landmark sizes are not read: they do not collide)
      crank[e] = (e < A && coll) ? e : -1;
      if (e < A) csz[e] = d.a_size;
    }
    for (int i = tid; i < A; i += blockDim.x)
      aconst[i] = make_float4(d.a_inv_mass, d.a_max_speed, d.a_accel, __int_as_float(d.a_flags));
    nC = coll ? A : 0;
    agents_only = coll;
  } else {
  for (int e = tid; e < E; e += blockDim.x) sizeq[e < A ? L + e : e - A] = tab[0 * E + e];
  for (int i = tid; i < A; i += blockDim.x) {
    const int fl = (tab[4 * E + i] != 0.f ? kMovable : 0) | (tab[5 * E + i] != 0.f ? kCollide : 0);
    aconst[i] = make_float4(1.0f / tab[1 * E + i], tab[3 * E + i], tab[2 * E + i], __int_as_float(fl));
  }
  if (tid < kWave) {  // rank the collidable entities in ascending entity order (wave 0: ballot + prefix count)
    int n = 0;
    for (int e0 = 0; e0 < E; e0 += kWave) {
      const int e = e0 + lane;
      const bool c = e < E && tab[5 * E + e] != 0.f;
      const unsigned long long m = __ballot(c);
      if (e < E) {
        const int k = n + __popcll(m & ((1ull << lane) - 1ull));
        crank[e] = c ? k : -1;
        if (c) csz[k] = tab[0 * E + e];
      }
      n += __popcll(m);
    }
  }
  // uniform facts every wave derives from the table itself (E/64 loads): the number of collidable
  // entities, whether they are exactly the agents (then the partner list IS the agent block of Q),
  // whether the collidable entities / the agents all have one size (thresholds hoist out of the loops)
  {
    float first_c = 0.f;
    bool have_c = false;
    const float first_a = tab[0];
    for (int e0 = 0; e0 < E; e0 += kWave) {
      const int e = e0 + lane;
      const bool in = e < E;
      const bool c = in && tab[5 * E + e] != 0.f;
      const float sz = in ? tab[0 * E + e] : 0.f;
      const unsigned long long m = __ballot(c);
      nC += __popcll(m);
      if (m && !have_c) {
        first_c = __shfl(sz, __ffsll((long long)m) - 1, kWave);
        have_c = true;
      }
      agents_only = agents_only && !__any(in && (c != (e < A)));
      one_csize = one_csize && !__any(c && sz != first_c);
      one_asize = one_asize && !__any(in && e < A && sz != first_a);
    }
  }
  }
  nC = __builtin_amdgcn_readfirstlane(nC);
  float2 *const CPW = agents_only ? Q + L : reinterpret_cast<float2 *>(wbase + cv.cpw);
  __syncthreads();

  const float far = kFarX * d.cmargin;
  const int D = d.D;

  // ---- persistent loop over groups of kWavesPerWg consecutive worlds ------------------------------
  // x -> group: XCD = x % 8 (hardware round-robin), slot = x / 8;  eight consecutive slots of one XCD
  // own eight consecutive groups (32 worlds = one 128-byte line of every state row).
  for (unsigned x = blockIdx.x; x < n_groups_padded; x += gridDim.x) {
    const unsigned xcd = x & 7u, slot = x >> 3;
    const unsigned g = ((slot >> 3) << 6) | (xcd << 3) | (slot & 7u);
    const size_t w = (size_t)g * kWavesPerWg + wave;
    if (w >= B) continue;  // wave-uniform

    // ---- stage the world -----------------------------------------------------------------------
    for (int e = lane; e < E; e += kWave) {
      const float2 p = make_float2(b.pos[(size_t)(2 * e) * B + w], b.pos[(size_t)(2 * e + 1) * B + w]);
      Q[e < A ? L + e : e - A] = p;
      if (PHYS && !agents_only) {
        const int r = crank[e];
        if (r >= 0) CPW[r] = p;
      }
    }
    for (int i = lane; i < A; i += kWave) {
      V[i] = make_float2(b.vel[(size_t)(2 * i) * B + w], b.vel[(size_t)(2 * i + 1) * B + w]);
      if (PHYS && !ROLL) {  // decode actions (environment.py:144-181)
        float ux, uy;
        fetch_action(b, B, i, w, aconst[i].z, ux, uy);
        U[i] = make_float2(ux + 0.f, uy + 0.f);
      }
    }
    wave_sync();

    // ---- the steps of this world (one, unless this is the rollout kernel) ---------------------------
    const uint64_t gw = ra.world_offset + w;   // global world number (RNG streams)
    const int T = ROLL ? ra.T : 1;
    const size_t obs_stride = (ROLL && ra.trajectory) ? (size_t)A * D * B : 0;
    const size_t row_stride = (ROLL && ra.trajectory) ? (size_t)A * B : 0;
    int countdown = -1;    // resets fall on global steps that are multiples of episode_len
    uint64_t ep = 0;
    if (ROLL && ra.episode_len > 0) {
      const uint64_t len = (uint64_t)ra.episode_len, r = ra.step0 % len;
      countdown = r == 0 ? 0 : (int)(len - r);
      ep = ra.step0 / len + (r ? 1 : 0);
    }
    for (int t = 0; t < T; ++t) {
    if (ROLL) {
      const uint64_t gt = ra.step0 + (uint64_t)t;
      const bool reset_now = countdown == 0;
      if (countdown >= 0) countdown = reset_now ? ra.episode_len - 1 : countdown - 1;
      if (reset_now) {  // reset_world, as mpe_reset draws it for episode ep
        for (int e = lane; e < E; e += kWave) {
          float x, y;
          reset_draw(ra.seed, gw, ep, e, e < A ? 1.0f : ra.landmark_range, x, y);
          Q[e < A ? L + e : e - A] = make_float2(x, y);
        }
        for (int i = lane; i < A; i += kWave) V[i] = make_float2(0.f, 0.f);
        ++ep;
      }
      for (int i = lane; i < A; i += kWave) {   // the one-hot row mpe_random_actions would write, decoded -- or the caller's (act_seq)
        const float sens = aconst[i].z;
        float ux, uy;
        if (ra.act_seq) {
          fetch_action_seq(ra.act_seq, t, A, B, i, w, sens, ux, uy);
        } else {
          const int m = action_draw(ra.seed, gw, gt, i);
          ux = ((m == 1 ? 1.f : 0.f) - (m == 2 ? 1.f : 0.f)) * sens;
          uy = ((m == 3 ? 1.f : 0.f) - (m == 4 ? 1.f : 0.f)) * sens;
        }
        U[i] = make_float2(ux + 0.f, uy + 0.f);
      }
      wave_sync();
      if (!agents_only) {  // the partner list follows the positions
        for (int e = lane; e < E; e += kWave) {
          const int r = crank[e];
          if (r >= 0) CPW[r] = Q[e < A ? L + e : e - A];
        }
        wave_sync();
      }
    }

    if (PHYS) {
      // ---- pairwise contact force (core.py:143-155,180-196) + integrate (core.py:158-169) -----------
      for (int i0 = 0; i0 < A; i0 += kWave) {
        const int i = i0 + lane;
        const bool have = i < A;
        const float4 ac = aconst[have ? i : 0];
        const int fi = have ? __float_as_int(ac.w) : 0;
        const float2 me = Q[L + (have ? i : 0)];
        const float ri = sizeq[L + (have ? i : 0)];
        const float2 u = U[have ? i : 0];
        float ax = u.x, ay = u.y;  // action force first, then the partners in ascending order (Q9)
        const bool pushes = (fi & kCollide) && (fi & kMovable);
        const int self_rank = pushes ? crank[i] : -1;
        const float rfar = ri + far;
        const float reach_u = rfar + csz[0];
        for (int kb = 0; kb < nC; kb += 32) {
          const int n = min(nC - kb, 32);
          unsigned near = one_csize ? near_mask32<true>(CPW, csz, kb, n, me, rfar, reach_u * reach_u)
                                    : near_mask32<false>(CPW, csz, kb, n, me, rfar, 0.f);
          if (self_rank >= kb && self_rank < kb + 32) near &= ~(0x80000000u >> (self_rank - kb));  // not against itself
          if (!pushes) near = 0u;
          while (near) {  // pass 2: those only, ascending (Q9)
            const int j = __clz((int)near);
            near &= ~(0x80000000u >> j);
            const float2 pj = CPW[kb + j];
            float gx, gy;
            contact_force(me.x - pj.x, me.y - pj.y, ri + csz[kb + j], d.cforce, d.cmargin, d.cmargin_inv, gx, gy);
            ax = gx + ax;
            ay = gy + ay;
          }
        }
        if (have && (fi & kMovable)) {
          float2 p = me, v = V[i];
          integrate_one(p.x, p.y, v.x, v.y, ax, ay, ac.x, ac.y, d.damp, d.dt);
          QN[i] = p;
          V[i] = v;
          if (!ROLL) {   // (the rollout writes the state back once, after its last step)
            b.pos[(size_t)(2 * i) * B + w] = p.x;
            b.pos[(size_t)(2 * i + 1) * B + w] = p.y;
            b.vel[(size_t)(2 * i) * B + w] = v.x;
            b.vel[(size_t)(2 * i + 1) * B + w] = v.y;
          }
        } else if (have && A > kWave) {
          QN[i] = me;
        }
      }
      wave_sync();
      if (A > kWave) {  // batches integrated against the OLD positions; publish the new ones now
        for (int i = lane; i < A; i += kWave) Q[L + i] = QN[i];
        wave_sync();
      }
    }

    if (OUT && d.kind == MPE_SCN_TAG) {
      // ---- simple_tag, any team sizes: rows (:131-147), rewards (:84-129), benchmark_data (:57-66) ------------
      const int nadv = d.nadv;
      emit_rows_tag(Q, V, A, L, nadv, b.obs + (size_t)t * (ROLL && ra.trajectory ? obs_stride_tag(A, L, nadv, B) : 0), B, w, lane);
      if (b.rew || b.info_collisions) {
        float hits = 0.f;   // all (good, adversary) contacts of the world, counted on the adversary lanes
        for (int t0 = 0; t0 < A; t0 += kWave) {
          const int tt = t0 + lane;
          const bool hi = tt < A, is_adv = tt < nadv;
          const float2 pi = Q[L + (hi ? tt : 0)];
          const float ri = sizeq[L + (hi ? tt : 0)];
          int c = 0;
          for (int j = 0; j < A; ++j) {   // uniform j; a lane counts the agents of the OTHER team it touches
            const float2 pj = Q[L + j];
            const bool other = (j < nadv) != is_adv;
            c += (other && sqrt_lt(sq2d(pj.x - pi.x, pj.y - pi.y), sizeq[L + j] + ri)) ? 1 : 0;
          }
          if (hi) CNT[tt] = c;
          hits += (hi && is_adv) ? (float)c : 0.f;
        }
        hits = wave_sum(hits);
        const float adv_rew = 10.f * hits;   // adversary_reward :115-129: +10 per contact, the same for every adversary
        wave_sync();
        for (int i = lane; i < A; i += kWave) {
          const int c = CNT[i];
          const bool coll = __float_as_int(aconst[i].w) & kCollide;
          float r;
          if (i < nadv) {
            r = coll ? adv_rew : 0.f;
          } else {                            // agent_reward :89-113
            const float2 p = Q[L + i];
            r = coll ? 0.f - 10.f * (float)c : 0.f;
            r -= tag_bound(fabsf(p.x));
            r -= tag_bound(fabsf(p.y));
          }
          const size_t o = (size_t)t * row_stride + (size_t)i * B + w;
          if (b.rew) b.rew[o] = r;
          if (b.done) b.done[o] = 0;
          if (b.info_collisions) b.info_collisions[o] = i < nadv ? c : 0;   // benchmark_data :57-66
        }
      } else if (b.done) {
        for (int i = lane; i < A; i += kWave) b.done[(size_t)t * row_stride + (size_t)i * B + w] = 0;
      }
    }
    if (OUT && d.kind == MPE_SCN_SPREAD) {
      // ---- observation rows (simple_spread.py:84-100): issued first, they drain while the reward is computed
      // agent i's rows form one [B][D] block (obs_n[i] of the drop-in API): the rows of one world are B*D floats apart
      const size_t rowlen = (size_t)B * D;
      float *const obs_w = b.obs + (size_t)t * obs_stride + w * (size_t)D;
      const bool vec4 = (D & 3) == 0 && (reinterpret_cast<uintptr_t>(b.obs) & 15) == 0 && ((B * (size_t)D) & 3) == 0 &&
                        ((obs_stride & 3) == 0);
      if (vec4 && D <= 8 * kWave)
        if (d.rows_nt) emit_rows_fast<kRowsNt>(Q, V, A, L, D, obs_w, rowlen, lane);
        else           emit_rows_fast<kRowsPlain>(Q, V, A, L, D, obs_w, rowlen, lane);
      else if (vec4)
        emit_rows<4>(Q, V, A, L, D, obs_w, rowlen, lane);
      else
        emit_rows<2>(Q, V, A, L, D, obs_w, rowlen, lane);  // D is even (checked on the host)

      // ---- reward (simple_spread.py:72-82) + benchmark_data (:47-63) --------------------------------
      if (b.rew || b.info_rew) {
        float neg = 0.f, csum = 0.f;
        int occ = 0;
        const int nb = (max(A, L) + kWave - 1) / kWave;
        for (int k0 = 0; k0 < nb; ++k0) {
          const int t = k0 * kWave + lane;
          const bool hl = t < L, hi = t < A;
          const float2 pl = Q[hl ? t : 0];          // landmark t
          const float2 pi = Q[L + (hi ? t : 0)];    // agent t
          const float ri = sizeq[L + (hi ? t : 0)];
          float m2 = INFINITY;
          int c = 0;
          if (one_asize) {
            // every agent has the same size: the guard band of the strict `<` test is a per-lane constant
            const float m = ri + ri, mm = m * m, lo = mm * 0.9999996f, hi_ = mm * 1.0000004f;
            const bool has_band = mm > 1e-30f;
            bool band = false;   // did any pair land in the guard band of the strict `<` (1e-6 wide: almost never)
#pragma unroll 8
            for (int a = 0; a < A; ++a) {  // uniform a: broadcast reads at compile-time offsets; branch-free body
              const float2 pa = Q[L + a];
              m2 = fminf(m2, sq2d(pa.x - pl.x, pa.y - pl.y));
              const float da = sq2d(pa.x - pi.x, pa.y - pi.y);
              const bool below = da < lo;
              c += below ? 1 : 0;                                  // includes a == t
              band = band || (!below && !(da > hi_ && has_band));
            }
            if (band) {  // recount this lane's row with the exact test
              c = 0;
              for (int a = 0; a < A; ++a) {
                const float2 pa = Q[L + a];
                c += sqrt_lt_pre(sq2d(pa.x - pi.x, pa.y - pi.y), m, lo, hi_, has_band) ? 1 : 0;
              }
            }
          } else {
#pragma unroll 4
            for (int a = 0; a < A; ++a) {
              const float2 pa = Q[L + a];
              m2 = fminf(m2, sq2d(pa.x - pl.x, pa.y - pl.y));
              c += sqrt_lt(sq2d(pa.x - pi.x, pa.y - pi.y), sizeq[L + a] + ri) ? 1 : 0;
            }
          }
          if (hl) {
            neg = neg - fast_sqrt(m2);
            occ += sqrt_lt(m2, 0.1f) ? 1 : 0;
          }
          if (hi) {
            c = (__float_as_int(aconst[t].w) & kCollide) ? c : 0;
            CNT[t] = c;
            csum += (float)c;
          }
        }
        neg = wave_sum(neg);
        occ = wave_sum_i(occ);
        csum = wave_sum(csum);
        const float tot = (float)A * neg - csum;  // environment.py:100-102: sum over agents of (neg - count_i)
        wave_sync();
        for (int i = lane; i < A; i += kWave) {
          const int c = CNT[i];
          const float r = neg - (float)c;
          const size_t o = (size_t)t * row_stride + (size_t)i * B + w;
          if (b.rew) b.rew[o] = d.collaborative ? tot : r;
          if (b.done) b.done[o] = 0;
          if (b.info_rew) {
            b.info_rew[o] = r;
            b.info_collisions[o] = c;
            b.info_min_dists[o] = -neg;
            b.info_occupied[o] = occ;
          }
        }
      } else if (b.done) {
        for (int i = lane; i < A; i += kWave) b.done[(size_t)t * row_stride + (size_t)i * B + w] = 0;
      }
    }
    wave_sync();  // every LDS read of this step is done before the next step / world overwrites the block
    }  // steps
    if (ROLL) {   // hand the state back: every entity (in-kernel resets move the landmarks too)
      for (int e = lane; e < E; e += kWave) {
        const float2 p = Q[e < A ? L + e : e - A];
        b.pos[(size_t)(2 * e) * B + w] = p.x;
        b.pos[(size_t)(2 * e + 1) * B + w] = p.y;
      }
      for (int i = lane; i < A; i += kWave) {
        b.vel[(size_t)(2 * i) * B + w] = V[i].x;
        b.vel[(size_t)(2 * i + 1) * B + w] = V[i].y;
      }
      wave_sync();
    }
  }
}

// ---- two waves per world, G worlds per workgroup: the headline large-N shape (simple_spread, 32 < N <= 64) ------
// What k_wave costs at B = 4096, N = 64 (88 us; ablations in DESIGN.md 6): the rows stream at the HBM write rate
// (60-73 us), but before them every world pays ~13 us of STATE I/O that nothing overlaps -- with a wave per world,
// lane = agent, every 4-byte state / action access of a wave touches 64 different 128-byte lines (the SoA layout is
// batch-innermost), ~1000 L2 requests per world, 4.2 M per launch: more requests than the 403 MB of rows.  And the
// reward (~1600 VALU per wave) of the last waves is exposed at the end.  Here
//   * a workgroup owns G consecutive worlds and stages their state COOPERATIVELY: thread -> (row, world), so the G
//     worlds of a row are G*4 contiguous bytes of one line -- ~5x fewer L2 requests at G = 4; the new state leaves
//     the same way (staged in LDS, stored by rows);
//   * each world has two waves with different ROLES: wave 2g runs World.step (lane = agent) and, after the
//     workgroup's second barrier, emits observation rows [0, split); wave 2g+1 stores the new state, computes the
//     reward / benchmark_data and emits rows [split, A) -- the reward's arithmetic overlaps other waves' row stores.
// Scenario constants are kernel arguments (identical agents, landmarks do not collide): no constant table.  Per-pair
// arithmetic, accumulation order (action first, partners ascending: Q9) and the reward's reduction trees are
// k_wave's: results are bit-identical to it (tests/test_gpu_parity.py).
constexpr int kDuoSplitNum = MPE_DUO_SPLIT_NUM, kDuoSplitDen = MPE_DUO_SPLIT_DEN;   // wave 2g emits rows [0, A * num / den)

struct DuoCarve { size_t q, v, u, araw, slot_bytes; };
__host__ __device__ inline DuoCarve duo_carve(int A, int L) {
  DuoCarve c;
  size_t o = 0;
  c.q = o;    o += align16(sizeof(float2) * (A + L));   // positions, observation order [landmarks | agents]
  c.v = o;    o += align16(sizeof(float2) * A);
  c.u = o;    o += align16(sizeof(float2) * A);         // decoded action force
  c.araw = o; o += align16(sizeof(float) * MPE_ACTION_DIM * A);   // the world's raw action rows [A][5]
  c.slot_bytes = o;
  return c;
}

// Reward (simple_spread.py:72-82) + benchmark_data (:47-63) of one world by one wave, identical agents: lane = landmark
// `lane` and agent `lane`; o = this lane's element of the [A][B] output rows (+ the step's block in a trajectory).
__device__ __forceinline__ void duo_reward(const WideDesc &d, const MpeBuffers &b, const float2 *Q, int A, int L, size_t B,
                                           size_t w, size_t o, int lane, bool collide) {
  if (b.rew || b.info_rew) {
    // reward (simple_spread.py:72-82) + benchmark_data (:47-63): lane = landmark `lane` and agent `lane`
    const bool hl = lane < L, hi = lane < A;
    const float2 pl = Q[hl ? lane : 0];
    const float2 pi = Q[L + (hi ? lane : 0)];
    const float m = d.a_size + d.a_size, mm = m * m, lo = mm * 0.9999996f, hi_ = mm * 1.0000004f;
    const bool has_band = mm > 1e-30f;
    float m2 = INFINITY;
    int c = 0;
    bool band = false;   // did any pair land in the guard band of the strict `<` (1e-6 wide: almost never)
#pragma unroll 8
    for (int a = 0; a < A; ++a) {  // uniform a: broadcast reads at compile-time offsets; branch-free body
      const float2 pa = Q[L + a];
      m2 = fminf(m2, sq2d(pa.x - pl.x, pa.y - pl.y));
      const float da = sq2d(pa.x - pi.x, pa.y - pi.y);
      const bool below = da < lo;
      c += below ? 1 : 0;                                  // includes a == lane
      band = band || (!below && !(da > hi_ && has_band));
    }
    if (band) {  // recount this lane's row with the exact test
      c = 0;
      for (int a = 0; a < A; ++a) {
        const float2 pa = Q[L + a];
        c += sqrt_lt_pre(sq2d(pa.x - pi.x, pa.y - pi.y), m, lo, hi_, has_band) ? 1 : 0;
      }
    }
    float neg = hl ? 0.f - fast_sqrt(m2) : 0.f;
    int occ = (hl && sqrt_lt(m2, 0.1f)) ? 1 : 0;
    c = (hi && collide) ? c : 0;
    float csum = (float)c;
    neg = wave_sum(neg);
    occ = wave_sum_i(occ);
    csum = wave_sum(csum);
    const float tot = (float)A * neg - csum;  // environment.py:100-102: sum over agents of (neg - count_i)
    if (hi) {
      const float r = neg - (float)c;
      if (b.rew) b.rew[o] = d.collaborative ? tot : r;
      if (b.done) b.done[o] = 0;
      if (b.info_rew) {
        b.info_rew[o] = r;
        b.info_collisions[o] = c;
        b.info_min_dists[o] = -neg;
        b.info_occupied[o] = occ;
      }
    }
  } else if (b.done && lane < A) {
    b.done[o] = 0;
  }
}

// (register budget: the occupancy caps that were tried -- amdgpu_waves_per_eu forcing 8 or 4 waves per SIMD at 71 VGPRs:
//  85.6-86.5 / 87.5-88.3 vs 84.5-85.0 us; capping at 6 or 4 at today's 46 VGPRs: 76.8-80.3 / 76.3-83.0 vs 76.1-80.2 --
//  are inside the run-to-run spread of the row stream)
template <int G>
__global__ void __launch_bounds__(2 * G * kWave)
k_duo(const WideDesc d, const MpeBuffers b, const size_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 2 * G * kWave;
  const int A = d.A, L = d.L, E = A + L, D = d.D;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 1, role = wave & 1;
  const DuoCarve cv = duo_carve(A, L);
  auto Qs = [&](int sl) { return reinterpret_cast<float2 *>(smem + (size_t)sl * cv.slot_bytes + cv.q); };
  auto Vs = [&](int sl) { return reinterpret_cast<float2 *>(smem + (size_t)sl * cv.slot_bytes + cv.v); };
  auto Us = [&](int sl) { return reinterpret_cast<float2 *>(smem + (size_t)sl * cv.slot_bytes + cv.u); };
  auto Rs = [&](int sl) { return reinterpret_cast<float *>(smem + (size_t)sl * cv.slot_bytes + cv.araw); };
  // worlds of this workgroup: G consecutive ones; XCD = blockIdx % 8 (hardware round-robin), and the 32 / G groups
  // that share a 128-byte line of every state row (32 consecutive worlds) sit on ONE XCD
  constexpr unsigned GPL = 32 / G;
  const unsigned x = blockIdx.x, xcd = x & 7u, slot = x >> 3;
  const size_t w0 = (size_t)(((slot / GPL) * 8u + xcd) * GPL + (slot % GPL)) * G;
  if (w0 >= B) return;   // workgroup-uniform
  const int nvalid = (B - w0) < (size_t)G ? (int)(B - w0) : G;
  const bool movable = d.a_flags & kMovable, collide = d.a_flags & kCollide;
#ifdef MPE_STRESS_DELAY_WAVE   // test build (libmpe_hip_stress.so): one wave of every workgroup starts ~30 us late
  if (wave == 1)
    for (int k = 0; k < 10; ++k) __builtin_amdgcn_s_sleep(127);
#endif

  // ---- cooperative stage: thread -> (row, world slot); the slots of a row are adjacent lanes and adjacent bytes ----
  {
    const int gg = tid % G, r0 = tid / G;
    if (gg < nvalid) {
      const size_t wg = w0 + gg;
      float *const Qf = reinterpret_cast<float *>(Qs(gg));
      for (int r = r0; r < 2 * E; r += NT / G) {
        const int e = r >> 1;
        Qf[2 * (e < A ? L + e : e - A) + (r & 1)] = b.pos[(size_t)r * B + wg];
      }
      float *const Vf = reinterpret_cast<float *>(Vs(gg));
      for (int r = r0; r < 2 * A; r += NT / G) Vf[r] = b.vel[(size_t)r * B + wg];
      if (b.ids) {
        for (int i = r0; i < A; i += NT / G) {
          float ux, uy;
          decode_id(b.ids[(size_t)i * B + wg], d.a_accel, ux, uy);
          Us(gg)[i] = make_float2(ux + 0.f, uy + 0.f);
        }
      } else if (b.u) {
        float *const Uf = reinterpret_cast<float *>(Us(gg));
        for (int r = r0; r < 2 * A; r += NT / G) Uf[r] = b.u[(size_t)r * B + wg] + 0.f;
      }
    }
    if (b.act) {   // agent i's rows of the G worlds are 5 G contiguous floats of act [A][B][5]
      constexpr int RW = MPE_ACTION_DIM * G;
      for (int t = tid; t < A * RW; t += NT) {
        const int i = t / RW, k = t - i * RW, sl = k / MPE_ACTION_DIM, c = k - sl * MPE_ACTION_DIM;
        if (sl < nvalid) Rs(sl)[i * MPE_ACTION_DIM + c] = b.act[((size_t)i * B + w0) * MPE_ACTION_DIM + k];
      }
    }
  }
  __syncthreads();

  const bool wok = g < nvalid;
  const size_t w = w0 + (wok ? g : 0);
  float2 *const Q = Qs(g), *const V = Vs(g);
  const int split = (A * kDuoSplitNum) / kDuoSplitDen;

  if (role == 0 && wok && movable) {
    // ---- World.step (core.py:117-169), lane = agent ------------------------------------------------------------
    const bool have = lane < A;
    const int i = have ? lane : 0;
    float2 me = Q[L + i], v = V[i];
    float ux, uy;
    if (b.act) {
      decode_row(Rs(g) + i * MPE_ACTION_DIM, d.a_accel, ux, uy);
      ux = ux + 0.f;
      uy = uy + 0.f;
    } else {
      const float2 u = Us(g)[i];
      ux = u.x;
      uy = u.y;
    }
    float ax = ux, ay = uy;   // action force first, then the partners in ascending order (Q9)
    if (collide && !(MPE_DUO_ABLATE & 2)) {
      const float ri = d.a_size, rfar = ri + kFarX * d.cmargin, reach = rfar + ri;
      const float2 *const CPW = Q + L;
      for (int kb = 0; kb < A; kb += 32) {
        const int n = min(A - kb, 32);
        unsigned near = near_mask32<true>(CPW, nullptr, kb, n, me, rfar, reach * reach);
        if (i >= kb && i < kb + 32) near &= ~(0x80000000u >> (i - kb));   // not against itself
        if (!have) near = 0u;
        while (near) {   // pass 2: the partners within reach only, ascending
          const int j = __clz((int)near);
          near &= ~(0x80000000u >> j);
          const float2 pj = CPW[kb + j];
          float gx, gy;
          contact_force(me.x - pj.x, me.y - pj.y, ri + ri, d.cforce, d.cmargin, d.cmargin_inv, gx, gy);
          ax = gx + ax;
          ay = gy + ay;
        }
      }
    }
    integrate_one(me.x, me.y, v.x, v.y, ax, ay, d.a_inv_mass, d.a_max_speed, d.damp, d.dt);
    wave_sync();   // every lane has read the old positions
    if (have) {
      Q[L + i] = me;
      V[i] = v;
    }
  }
  __syncthreads();

  // rows of D = 6 N floats: 16-byte pieces when N is even, 8-byte pieces when N is odd (rows then start 8 bytes off)
  const bool rows16 = (D & 3) == 0 && ((B * (size_t)D) & 3) == 0;
  if (role == 0) {
    if (wok && !(MPE_DUO_ABLATE & 4)) {
      // (nontemporal rows.  Agent scope (sc1) wins by 0.5-2.5 us when ONE buffer is written launch after launch (profiles/
      //  r3_ab_logs.txt sessions 24-25: 66.6-67.0 vs 67.0-69.2 us fast buffers, 79-82 vs 82-84 slow ones) and loses 0.7 us under
      //  the env's ping-pong of two output sets (session 30: 68.0-69.1 vs 67.3-67.8) -- the protocol that counts)
      if (rows16 && d.rows_nt) emit_rows_fast<kRowsNt>(Q, V, A, L, D, b.obs + w * (size_t)D, (size_t)B * D, lane, 0, split);
      else if (rows16) emit_rows_fast<kRowsPlain>(Q, V, A, L, D, b.obs + w * (size_t)D, (size_t)B * D, lane, 0, split);
      else        emit_rows<2>(Q, V, A, L, D, b.obs + w * (size_t)D, (size_t)B * D, lane, 0, split);
    }
    return;
  }

  // ---- odd waves: the new state back to HBM by rows (thread -> (row, world slot)), reward, the other rows --------
  if (movable) {
    const int t1 = g * kWave + lane;   // 0 .. 64 G - 1 over the G odd waves
    const int gg = t1 % G;
    if (gg < nvalid) {
      const float *const Qa = reinterpret_cast<const float *>(Qs(gg) + L);
      const float *const Vf = reinterpret_cast<const float *>(Vs(gg));
      for (int r = t1 / G; r < 2 * A; r += kWave) {
        b.pos[(size_t)r * B + w0 + gg] = Qa[r];
        b.vel[(size_t)r * B + w0 + gg] = Vf[r];
      }
    }
  }
  if (!wok) return;
  if (!(MPE_DUO_ABLATE & 1)) duo_reward(d, b, Q, A, L, B, w, (size_t)lane * B + w, lane, collide);
  if (!(MPE_DUO_ABLATE & 4)) {
    if (rows16 && d.rows_nt) emit_rows_fast<kRowsNt>(Q, V, A, L, D, b.obs + w * (size_t)D, (size_t)B * D, lane, split, A);
    else if (rows16) emit_rows_fast<kRowsPlain>(Q, V, A, L, D, b.obs + w * (size_t)D, (size_t)B * D, lane, split, A);
    else        emit_rows<2>(Q, V, A, L, D, b.obs + w * (size_t)D, (size_t)B * D, lane, split, A);
  }
}

// ---- the fused T-step rollout on k_duo's plan (mpe_rollout_random, spread 32 < N <= 64, identical agents) ---------
// The worlds of the workgroup stay in LDS for all T steps, so the per-step state I/O of the stepwise kernel is gone;
// positions and velocities are DOUBLE-buffered by step parity: step t's World.step (wave 2g) reads P(t-1) from buffer
// t & 1 and writes P(t) into the other one.  One workgroup barrier per step closes step t's World.step; behind it wave
// 2g emits rows [0, split) of step t and goes straight on to step t+1's moves and contacts -- which only read P(t) and
// write the buffer nobody reads any more -- while wave 2g+1 computes step t's reward and emits the remaining rows.  In-kernel resets (mpe_reset's draws) overwrite the
// buffer the previous step's rows are read from: they sit behind one more barrier, on reset steps only.  Moves are
// drawn per agent lane (action_draw: the rows mpe_random_actions would write).  Bit-identical to
// T x { [mpe_reset]; mpe_random_actions; mpe_step } (tests/test_gpu_rollout.py).
struct DuoRollCarve { size_t q_bytes, v_bytes, v0, slot_bytes; };   // per slot: Q buffer 0, Q buffer 1, V buffer 0, V buffer 1
__host__ __device__ inline DuoRollCarve duo_roll_carve(int A, int L) {
  DuoRollCarve c;
  c.q_bytes = align16(sizeof(float2) * (A + L));
  c.v_bytes = align16(sizeof(float2) * A);
  c.v0 = 2 * c.q_bytes;
  c.slot_bytes = 2 * c.q_bytes + 2 * c.v_bytes;
  return c;
}

// (128 VGPRs, no scratch; forcing the budget of 5 / 6 / 8 waves per SIMD measured 71.1-72.3 / 72.6-73.2 / 74.5-75.7 vs
//  70.5-72.3 us per step in the same box)
template <int G>
__global__ void __launch_bounds__(2 * G * kWave)
k_duo_roll(const WideDesc d, const MpeBuffers b, const size_t B, const RollArgs ra) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 2 * G * kWave;
  const int A = d.A, L = d.L, E = A + L, D = d.D;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 1, role = wave & 1;
  const DuoRollCarve cv = duo_roll_carve(A, L);
  // (offsets by arithmetic, not by indexing an array with the step parity: that would live in scratch memory)
  auto Qs = [&](int sl, int k) { return reinterpret_cast<float2 *>(smem + (size_t)sl * cv.slot_bytes + (size_t)k * cv.q_bytes); };
  auto Vs = [&](int sl, int k) { return reinterpret_cast<float2 *>(smem + (size_t)sl * cv.slot_bytes + cv.v0 + (size_t)k * cv.v_bytes); };
  constexpr unsigned GPL = 32 / G;
  const unsigned x = blockIdx.x, xcd = x & 7u, slot = x >> 3;
  const size_t w0 = (size_t)(((slot / GPL) * 8u + xcd) * GPL + (slot % GPL)) * G;
  if (w0 >= B) return;   // workgroup-uniform
  const int nvalid = (B - w0) < (size_t)G ? (int)(B - w0) : G;
  const bool movable = d.a_flags & kMovable, collide = d.a_flags & kCollide;

  // ---- stage P(-1) into buffer 0 (landmarks into both buffers), cooperatively: thread -> (row, world slot) ----------
  {
    const int gg = tid % G, r0 = tid / G;
    if (gg < nvalid) {
      const size_t wg = w0 + gg;
      float *const Q0 = reinterpret_cast<float *>(Qs(gg, 0)), *const Q1 = reinterpret_cast<float *>(Qs(gg, 1));
      for (int r = r0; r < 2 * E; r += NT / G) {
        const int e = r >> 1, qi = 2 * (e < A ? L + e : e - A) + (r & 1);
        const float val = b.pos[(size_t)r * B + wg];
        Q0[qi] = val;
        if (e >= A) Q1[qi] = val;
      }
      float *const V0 = reinterpret_cast<float *>(Vs(gg, 0));
      for (int r = r0; r < 2 * A; r += NT / G) V0[r] = b.vel[(size_t)r * B + wg];
    }
  }
  __syncthreads();

  const bool wok = g < nvalid;
  const size_t w = w0 + (wok ? g : 0);
  const uint64_t gw = ra.world_offset + w;   // global world number (RNG streams)
  const int split = (A * kDuoSplitNum) / kDuoSplitDen;
  const int T = ra.T;
  const size_t obs_stride = ra.trajectory ? (size_t)A * D * B : 0;
  const size_t row_stride = ra.trajectory ? (size_t)A * B : 0;
  const bool rows16 = (D & 3) == 0 && ((B * (size_t)D) & 3) == 0 && ((obs_stride & 3) == 0);
  int countdown = -1;    // resets fall on global steps that are multiples of episode_len
  uint64_t ep = 0;
  if (ra.episode_len > 0) {
    const uint64_t len = (uint64_t)ra.episode_len, r = ra.step0 % len;
    countdown = r == 0 ? 0 : (int)(len - r);
    ep = ra.step0 / len + (r ? 1 : 0);
  }

  for (int t = 0; t < T; ++t) {
#ifdef MPE_STRESS_DELAY_WAVE   // test build: every step one wave lags ~7 us -- alternately a row/reward wave and a physics wave --
    if (wave == 1 + (t & 1))   // so that its partners run ahead into the next step's buffer as far as the barriers allow
      for (int k = 0; k < 2; ++k) __builtin_amdgcn_s_sleep(127);
#endif
    const int cur = t & 1, nxt = cur ^ 1;            // P(t-1) in buffer cur, P(t) goes to buffer nxt
    const uint64_t gt = ra.step0 + (uint64_t)t;
    const bool reset_now = countdown == 0;           // uniform over the grid
    if (countdown >= 0) countdown = reset_now ? ra.episode_len - 1 : countdown - 1;
    if (reset_now) {
      __syncthreads();   // the rows of step t-1 (read from buffer cur by both waves) are out
      if (role == 0 && wok) {   // reset_world, as mpe_reset draws it for episode ep
        if (lane < A) {
          float px, py;
          reset_draw(ra.seed, gw, ep, lane, 1.0f, px, py);
          Qs(g, cur)[L + lane] = make_float2(px, py);
          Vs(g, cur)[lane] = make_float2(0.f, 0.f);
        }
        if (lane < L) {
          float px, py;
          reset_draw(ra.seed, gw, ep, A + lane, ra.landmark_range, px, py);
          Qs(g, cur)[lane] = make_float2(px, py);
          Qs(g, nxt)[lane] = make_float2(px, py);
        }
        wave_sync();
      }
      ++ep;
    }
    if (role == 0 && wok) {
      // ---- World.step (core.py:117-169), lane = agent: P(t-1) -> P(t) -----------------------------------------
      const float2 *const Qc = Qs(g, cur);
      const bool have = lane < A;
      const int i = have ? lane : 0;
      float2 me = Qc[L + i], v = Vs(g, cur)[i];
      if (movable) {
        float ux, uy;      // the one-hot row mpe_random_actions would write, decoded -- or the caller's (act_seq)
        if (ra.act_seq) {
          fetch_action_seq(ra.act_seq, t, A, B, i, w, d.a_accel, ux, uy);
        } else {
          const int m = action_draw(ra.seed, gw, gt, i);
          ux = ((m == 1 ? 1.f : 0.f) - (m == 2 ? 1.f : 0.f)) * d.a_accel;
          uy = ((m == 3 ? 1.f : 0.f) - (m == 4 ? 1.f : 0.f)) * d.a_accel;
        }
        float ax = ux + 0.f, ay = uy + 0.f;   // action force first, then the partners in ascending order (Q9)
        if (collide) {
          const float ri = d.a_size, rfar = ri + kFarX * d.cmargin, reach = rfar + ri;
          const float2 *const CPW = Qc + L;
          for (int kb = 0; kb < A; kb += 32) {
            const int n = min(A - kb, 32);
            unsigned near = near_mask32<true>(CPW, nullptr, kb, n, me, rfar, reach * reach);
            if (i >= kb && i < kb + 32) near &= ~(0x80000000u >> (i - kb));   // not against itself
            if (!have) near = 0u;
            while (near) {   // pass 2: the partners within reach only, ascending
              const int j = __clz((int)near);
              near &= ~(0x80000000u >> j);
              const float2 pj = CPW[kb + j];
              float gx, gy;
              contact_force(me.x - pj.x, me.y - pj.y, ri + ri, d.cforce, d.cmargin, d.cmargin_inv, gx, gy);
              ax = gx + ax;
              ay = gy + ay;
            }
          }
        }
        integrate_one(me.x, me.y, v.x, v.y, ax, ay, d.a_inv_mass, d.a_max_speed, d.damp, d.dt);
      }
      if (have) {   // the other buffer: nobody reads it before the barrier below
        Qs(g, nxt)[L + i] = me;
        Vs(g, nxt)[i] = v;
      }
    }
    __syncthreads();   // P(t) is complete; every read of buffer nxt's previous contents (rows of step t-2 ...) is long over

    const float2 *const Q = Qs(g, nxt), *const V = Vs(g, nxt);
    float *const obs_w = b.obs + (size_t)t * obs_stride + w * (size_t)D;
    if (role == 0) {
      if (wok) {
        if (rows16 && d.rows_nt) emit_rows_fast<kRowsNt>(Q, V, A, L, D, obs_w, (size_t)B * D, lane, 0, split);
        else if (rows16) emit_rows_fast<kRowsPlain>(Q, V, A, L, D, obs_w, (size_t)B * D, lane, 0, split);
        else        emit_rows<2>(Q, V, A, L, D, obs_w, (size_t)B * D, lane, 0, split);
      }
    } else if (wok) {
      duo_reward(d, b, Q, A, L, B, w, (size_t)t * row_stride + (size_t)lane * B + w, lane, collide);
      if (rows16 && d.rows_nt) emit_rows_fast<kRowsNt>(Q, V, A, L, D, obs_w, (size_t)B * D, lane, split, A);
      else if (rows16) emit_rows_fast<kRowsPlain>(Q, V, A, L, D, obs_w, (size_t)B * D, lane, split, A);
      else        emit_rows<2>(Q, V, A, L, D, obs_w, (size_t)B * D, lane, split, A);
    }
  }

  // ---- hand the state back: P(T-1) sits in buffer T & 1; every entity (in-kernel resets move the landmarks too) -------
  __syncthreads();
  {
    const int fin = T & 1;
    const int gg = tid % G, r0 = tid / G;
    if (gg < nvalid) {
      const size_t wg = w0 + gg;
      const float *const Qf = reinterpret_cast<const float *>(Qs(gg, fin));
      const float *const Vf = reinterpret_cast<const float *>(Vs(gg, fin));
      for (int r = r0; r < 2 * E; r += NT / G) {
        const int e = r >> 1;
        b.pos[(size_t)r * B + wg] = Qf[2 * (e < A ? L + e : e - A) + (r & 1)];
      }
      for (int r = r0; r < 2 * A; r += NT / G) b.vel[(size_t)r * B + wg] = Vf[r];
    }
  }
}

// ---- several worlds per wave: the mid-size regime (7 <= A, L <= 32) ------------------------------------------
// With a world per wave, a 16-agent world leaves three quarters of the lanes idle in every phase and its 384-byte
// rows fill a sixth of a wave store.  Here a wave owns WPW = 64 / AP consecutive worlds (AP = 8, 16 or 32 lanes per
// world slot, the power of two covering max(A, L)): lane = (slot, a) is agent a -- and landmark a -- of world slot.
// The phases are k_wave's, with per-slot LDS blocks instead of wave-uniform ones and SEGMENTED shuffle reductions
// (xor offsets below AP); row stores are flat over (slot, piece), so the WPW adjacent rows of one agent index form
// one contiguous WPW * D * 4-byte run.  Same per-pair arithmetic, same accumulation order (action, then partners
// ascending) as every other kernel.
// ROLL: the fused T-step rollout on the same plan (mpe_rollout_random for these sizes): the wave's worlds stay in
// their LDS blocks for all T steps, moves and resets are drawn in-kernel exactly as k_wave<ROLL> draws them, every
// step's rows / rewards are written, the state goes back to HBM once.  (k_wave<ROLL> served these sizes before: a
// world per wave at N = 8 is 113 us per step at B = 65 536 against 30 us for a launch per step on this kernel.)
template <bool PHYS, bool OUT, bool ROLL>
__global__ void __launch_bounds__(kWavesPerWg *kWave)
k_multi(const WideDesc d, const MpeBuffers b, const size_t B, const unsigned n_groups_padded, const int AP, const RollArgs ra) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int A = d.A, L = d.L, E = A + L;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int WPW = kWave / AP;
  const int slot = lane / AP, a = lane - slot * AP;
  const Carve cv = carve(A, L);
  float *const sizeq = reinterpret_cast<float *>(smem + cv.sizeq);
  int *const crank = reinterpret_cast<int *>(smem + cv.crank);
  float *const csz = reinterpret_cast<float *>(smem + cv.csz);
  float4 *const aconst = reinterpret_cast<float4 *>(smem + cv.aconst);
  // per wave: WPW blocks of {Q [E], V [A], U [A], CPW [E]} float2
  const int blk = 2 * E + 2 * A;   // float2 per slot
  float2 *const wbase = reinterpret_cast<float2 *>(smem + cv.shared_bytes) + (size_t)wave * WPW * blk;
  auto Qs = [&](int sl) { return wbase + sl * blk; };
  auto Vs = [&](int sl) { return wbase + sl * blk + E; };
  auto Us = [&](int sl) { return wbase + sl * blk + E + A; };
  auto Cs = [&](int sl) { return wbase + sl * blk + E + 2 * A; };
  float2 *const Q = Qs(slot), *const V = Vs(slot), *const U = Us(slot);

  // ---- per-entity constants (as k_wave): from the kernel arguments when the agents are identical and no landmark
  //      collides (the shipped spread: no global load in front of the first world), else from the device table ---------
  const float *tab = b.entity_table;
  int nC = 0;
  bool agents_only = true;
  if (d.homo) {
    const bool coll = d.a_flags & kCollide;
    for (int e = tid; e < E; e += blockDim.x) {
      sizeq[e < A ? L + e : e - A] = e < A ? d.a_size : 0.f;   // (landmark sizes are not read: they do not collide)
      crank[e] = (e < A && coll) ? e : -1;
      if (e < A) csz[e] = d.a_size;
    }
    for (int i = tid; i < A; i += blockDim.x)
      aconst[i] = make_float4(d.a_inv_mass, d.a_max_speed, d.a_accel, __int_as_float(d.a_flags));
    nC = coll ? A : 0;
    agents_only = coll;
  } else {
    for (int e = tid; e < E; e += blockDim.x) sizeq[e < A ? L + e : e - A] = tab[0 * E + e];
    for (int i = tid; i < A; i += blockDim.x) {
      const int fl = (tab[4 * E + i] != 0.f ? kMovable : 0) | (tab[5 * E + i] != 0.f ? kCollide : 0);
      aconst[i] = make_float4(1.0f / tab[1 * E + i], tab[3 * E + i], tab[2 * E + i], __int_as_float(fl));
    }
    if (tid < kWave) {   // E <= 64: one pass
      const bool c = lane < E && tab[5 * E + lane] != 0.f;
      const unsigned long long m = __ballot(c);
      if (lane < E) {
        const int kq = __popcll(m & ((1ull << lane) - 1ull));
        crank[lane] = c ? kq : -1;
        if (c) csz[kq] = tab[0 * E + lane];
      }
    }
    {
      const bool in = lane < E;
      const bool c = in && tab[5 * E + lane] != 0.f;
      nC = __popcll(__ballot(c));
      agents_only = !__any(in && (c != (lane < A)));
    }
  }
  nC = __builtin_amdgcn_readfirstlane(nC);
  float2 *const CPW = agents_only ? Q + L : Cs(slot);
  __syncthreads();

  const float far = kFarX * d.cmargin;
  const int D = d.D;
  const int T = ROLL ? ra.T : 1;
  const size_t obs_stride = (ROLL && ra.trajectory) ? (size_t)A * D * B : 0;   // per-step output blocks of a trajectory
  const size_t row_stride = (ROLL && ra.trajectory) ? (size_t)A * B : 0;
  const bool vec4 = OUT && (D & 3) == 0 && (reinterpret_cast<uintptr_t>(b.obs) & 15) == 0 && ((B * (size_t)D) & 3) == 0 &&
                    (obs_stride & 3) == 0;
  const int P = vec4 ? D >> 2 : D >> 1;       // pieces (16 or 8 bytes) per row
  const int npc = WPW * P;                    // pieces per row index over the wave's slots
  const int kpz = 2 + L + (A - 1);

  const size_t per_wg = (size_t)kWavesPerWg * WPW;
  for (unsigned x = blockIdx.x; x < n_groups_padded; x += gridDim.x) {
    const unsigned xcd = x & 7u, sl8 = x >> 3;
    const unsigned g = ((sl8 >> 3) << 6) | (xcd << 3) | (sl8 & 7u);
    const size_t wb = (size_t)g * per_wg + (size_t)wave * WPW;   // first world of this wave
    if (wb >= B) continue;                                       // wave-uniform
    const size_t w = wb + slot;
    const bool ok = w < B;                                       // this lane's slot holds a world

    // ---- stage -------------------------------------------------------------------------------------------
    if (ok) {
      if (a < A) {
        const float2 p = make_float2(b.pos[(size_t)(2 * a) * B + w], b.pos[(size_t)(2 * a + 1) * B + w]);
        Q[L + a] = p;
        if (PHYS && !agents_only) { const int r = crank[a]; if (r >= 0) CPW[r] = p; }
        V[a] = make_float2(b.vel[(size_t)(2 * a) * B + w], b.vel[(size_t)(2 * a + 1) * B + w]);
        if (PHYS && !ROLL) {
          float ux, uy;
          fetch_action(b, B, a, w, aconst[a].z, ux, uy);
          U[a] = make_float2(ux + 0.f, uy + 0.f);
        }
      }
      if (a < L) {
        const int e = A + a;
        const float2 p = make_float2(b.pos[(size_t)(2 * e) * B + w], b.pos[(size_t)(2 * e + 1) * B + w]);
        Q[a] = p;
        if (PHYS && !agents_only) { const int r = crank[e]; if (r >= 0) CPW[r] = p; }
      }
    }
    wave_sync();

    const bool have = ok && a < A;
    const uint64_t gw = ra.world_offset + w;   // global world number (RNG streams)
    int countdown = -1;    // resets fall on global steps that are multiples of episode_len (as in k_wave<ROLL>)
    uint64_t ep = 0;
    if (ROLL && ra.episode_len > 0) {
      const uint64_t len = (uint64_t)ra.episode_len, r = ra.step0 % len;
      countdown = r == 0 ? 0 : (int)(len - r);
      ep = ra.step0 / len + (r ? 1 : 0);
    }
    for (int t = 0; t < T; ++t) {
    if (ROLL) {
      const uint64_t gt = ra.step0 + (uint64_t)t;
      const bool reset_now = countdown == 0;   // wave-uniform
      if (countdown >= 0) countdown = reset_now ? ra.episode_len - 1 : countdown - 1;
      if (reset_now) {  // reset_world, as mpe_reset draws it for episode ep
        if (ok && a < A) {
          float x, y;
          reset_draw(ra.seed, gw, ep, a, 1.0f, x, y);
          Q[L + a] = make_float2(x, y);
          V[a] = make_float2(0.f, 0.f);
        }
        if (ok && a < L) {
          float x, y;
          reset_draw(ra.seed, gw, ep, A + a, ra.landmark_range, x, y);
          Q[a] = make_float2(x, y);
        }
        ++ep;
      }
      if (have) {   // the one-hot row mpe_random_actions would write, decoded -- or the caller's (act_seq)
        const float sens = aconst[a].z;
        float ux, uy;
        if (ra.act_seq) {
          fetch_action_seq(ra.act_seq, t, A, B, a, w, sens, ux, uy);
        } else {
          const int m = action_draw(ra.seed, gw, gt, a);
          ux = ((m == 1 ? 1.f : 0.f) - (m == 2 ? 1.f : 0.f)) * sens;
          uy = ((m == 3 ? 1.f : 0.f) - (m == 4 ? 1.f : 0.f)) * sens;
        }
        U[a] = make_float2(ux + 0.f, uy + 0.f);
      }
      wave_sync();
      if (!agents_only) {  // the partner list follows the positions
        if (ok && a < A) { const int r = crank[a]; if (r >= 0) CPW[r] = Q[L + a]; }
        if (ok && a < L) { const int r = crank[A + a]; if (r >= 0) CPW[r] = Q[a]; }
        wave_sync();
      }
    }
    if (PHYS) {
      // ---- contacts + integrate (core.py:143-169), lane = (slot, agent a) -------------------------------------
      const float4 ac = aconst[a < A ? a : 0];
      const int fi = have ? __float_as_int(ac.w) : 0;
      const float2 me = Q[L + (a < A ? a : 0)];
      const float ri = sizeq[L + (a < A ? a : 0)];
      const float2 u = U[a < A ? a : 0];
      float ax = u.x, ay = u.y;
      const bool pushes = (fi & kCollide) && (fi & kMovable);
      const int self_rank = pushes ? crank[a] : -1;
      const float rfar = ri + far;
      for (int kb = 0; kb < nC && !(MPE_MULTI_ABLATE & 2); kb += 32) {
        const int n = min(nC - kb, 32);
        unsigned near = near_mask32<false>(CPW, csz, kb, n, me, rfar, 0.f);
        if (self_rank >= kb && self_rank < kb + 32) near &= ~(0x80000000u >> (self_rank - kb));
        if (!pushes) near = 0u;
        while (near) {
          const int j = __clz((int)near);
          near &= ~(0x80000000u >> j);
          const float2 pj = CPW[kb + j];
          float gx, gy;
          contact_force(me.x - pj.x, me.y - pj.y, ri + csz[kb + j], d.cforce, d.cmargin, d.cmargin_inv, gx, gy);
          ax = gx + ax;
          ay = gy + ay;
        }
      }
      wave_sync();   // every lane has read the old positions
      if (have && (fi & kMovable)) {
        float2 p = me, v = V[a];
        integrate_one(p.x, p.y, v.x, v.y, ax, ay, ac.x, ac.y, d.damp, d.dt);
        Q[L + a] = p;
        V[a] = v;
        if (!ROLL) {   // (the rollout writes the state back once, after its last step)
          b.pos[(size_t)(2 * a) * B + w] = p.x;
          b.pos[(size_t)(2 * a + 1) * B + w] = p.y;
          b.vel[(size_t)(2 * a) * B + w] = v.x;
          b.vel[(size_t)(2 * a + 1) * B + w] = v.y;
        }
      }
      wave_sync();
    }

    if (OUT) {
      // ---- observation rows: for each agent index i, the rows of the wave's WPW worlds are adjacent in HBM --------
      const int nvalid = (B - wb) < (size_t)WPW ? (int)(B - wb) : WPW;   // slots that hold a world
      if (MPE_MULTI_ABLATE & 4) {
      } else if (vec4 && npc <= 3 * kWave) {
        // fast form: this lane's (slot, piece) for its up to three pieces per row index, and both candidate sources
        // of both pairs of each piece (without / with the self-skip), are row-invariant: fetched once per batch of
        // worlds.  A row then costs one broadcast-ish read of (pos_i, vel_i) per piece, selects, subtractions, a store.
        int psl[3], ppc[3];
        bool pok[3];
        float2 ca[3][2], cb[3][2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int idx = lane + kWave * k;
          psl[k] = idx / P;
          ppc[k] = idx - psl[k] * P;
          pok[k] = idx < npc && psl[k] < nvalid;
          if (!pok[k]) psl[k] = 0;
          const float2 *const Qx = Qs(psl[k]);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int id = 2 * ppc[k] + h - 2;
            ca[k][h] = Qx[min(max(id, 0), E - 1)];
            cb[k][h] = Qx[min(max(id + 1, 0), E - 1)];
          }
        }
        for (int i = 0; i < A; ++i) {
          float *const rows = b.obs + (size_t)t * obs_stride + ((size_t)i * B + wb) * D;   // wave-uniform: WPW rows of D floats, contiguous
          const int thr = L + i;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (kWave * k >= npc) break;   // uniform
            const float2 mi = Qs(psl[k])[L + i];
            const float2 vi = Vs(psl[k])[i];
            float o[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int kp = 2 * ppc[k] + h, id = kp - 2;
              const float2 pj = id >= thr ? cb[k][h] : ca[k][h];
              float2 v = make_float2(pj.x - mi.x, pj.y - mi.y);
              v = kp >= kpz ? make_float2(0.f, 0.f) : v;
              v = kp == 0 ? vi : v;
              v = kp == 1 ? mi : v;
              o[2 * h] = v.x;
              o[2 * h + 1] = v.y;
            }
            if (pok[k]) {
              if (d.rows_nt) store_row4<kRowsNt>(rows + 4 * (lane + kWave * k), make_float4(o[0], o[1], o[2], o[3]));
              else           store_row4<kRowsPlain>(rows + 4 * (lane + kWave * k), make_float4(o[0], o[1], o[2], o[3]));
            }
          }
        }
      } else
      for (int i = 0; i < A; ++i) {   // generic form (8-byte pieces, or more than three pieces per lane)
        float *const rows = b.obs + (size_t)t * obs_stride + ((size_t)i * B + wb) * D;   // wave-uniform: WPW rows of D floats, contiguous
        for (int idx = lane; idx < npc; idx += kWave) {
          const int sl = idx / P, pc = idx - sl * P;
          if (sl >= nvalid) continue;
          const float2 *const Qx = Qs(sl);
          const float2 mi = Qx[L + i];
          if (vec4) {
            float o[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int kp = 2 * pc + h, id = kp - 2;
              const float2 pj = Qx[min(max(id + (id >= L + i ? 1 : 0), 0), E - 1)];
              float2 v = make_float2(pj.x - mi.x, pj.y - mi.y);
              if (kp >= kpz) v = make_float2(0.f, 0.f);
              if (kp == 0) v = Vs(sl)[i];
              if (kp == 1) v = mi;
              o[2 * h] = v.x;
              o[2 * h + 1] = v.y;
            }
            *reinterpret_cast<float4 *>(rows + 4 * idx) = make_float4(o[0], o[1], o[2], o[3]);
          } else {
            const int kp = pc, id = kp - 2;
            const float2 pj = Qx[min(max(id + (id >= L + i ? 1 : 0), 0), E - 1)];
            float2 v = make_float2(pj.x - mi.x, pj.y - mi.y);
            if (kp >= kpz) v = make_float2(0.f, 0.f);
            if (kp == 0) v = Vs(sl)[i];
            if (kp == 1) v = mi;
            *reinterpret_cast<float2 *>(rows + 2 * idx) = v;
          }
        }
      }

      // ---- reward (simple_spread.py:72-82) + benchmark_data: lane (slot, a) = landmark a and agent a ----------------
      if ((b.rew || b.info_rew) && !(MPE_MULTI_ABLATE & 1)) {
        const bool hl = ok && a < L, hi = have;
        const float2 pl = Q[a < L ? a : 0];
        const float2 pi = Q[L + (a < A ? a : 0)];
        const float ri = sizeq[L + (a < A ? a : 0)];
        float m2 = INFINITY;
        int c = 0;
        for (int j = 0; j < A; ++j) {
          const float2 pa = Q[L + j];
          m2 = fminf(m2, sq2d(pa.x - pl.x, pa.y - pl.y));
          c += sqrt_lt(sq2d(pa.x - pi.x, pa.y - pi.y), sizeq[L + j] + ri) ? 1 : 0;   // includes j == a
        }
        float neg = hl ? -fast_sqrt(m2) : 0.f;
        int occ = (hl && sqrt_lt(m2, 0.1f)) ? 1 : 0;
        c = (hi && (__float_as_int(aconst[a < A ? a : 0].w) & kCollide)) ? c : 0;
        float csum = (float)c;
        for (int o = AP >> 1; o > 0; o >>= 1) {   // segmented: xor offsets stay inside the slot
          neg += __shfl_xor(neg, o, kWave);
          occ += __shfl_xor(occ, o, kWave);
          csum += __shfl_xor(csum, o, kWave);
        }
        if (hi) {
          const float tot = (float)A * neg - csum;   // environment.py:100-102
          const float r = neg - (float)c;
          const size_t o = (size_t)t * row_stride + (size_t)a * B + w;
          if (b.rew) b.rew[o] = d.collaborative ? tot : r;
          if (b.done) b.done[o] = 0;
          if (b.info_rew) {
            b.info_rew[o] = r;
            b.info_collisions[o] = c;
            b.info_min_dists[o] = -neg;
            b.info_occupied[o] = occ;
          }
        }
      } else if (b.done && have) {
        b.done[(size_t)t * row_stride + (size_t)a * B + w] = 0;
      }
    }
    wave_sync();   // every LDS read of this step is done before the next step / batch of worlds overwrites the blocks
    }  // steps
    if (ROLL) {   // hand the state back: every entity (in-kernel resets move the landmarks too)
      if (have) {
        const float2 p = Q[L + a], v = V[a];
        b.pos[(size_t)(2 * a) * B + w] = p.x;
        b.pos[(size_t)(2 * a + 1) * B + w] = p.y;
        b.vel[(size_t)(2 * a) * B + w] = v.x;
        b.vel[(size_t)(2 * a + 1) * B + w] = v.y;
      }
      if (ok && a < L) {
        const float2 p = Q[a];
        b.pos[(size_t)(2 * (A + a)) * B + w] = p.x;
        b.pos[(size_t)(2 * (A + a) + 1) * B + w] = p.y;
      }
      wave_sync();
    }
  }
}

}  // namespace

// k_duo serves the fused spread step of 33..64 identical agents (rows of at most 512 floats)
static bool duo_eligible(const WideDesc &d, const MpeBuffers &b, size_t B, bool phys, bool out, bool roll) {
  const int amax = d.A > d.L ? d.A : d.L;
  return phys && out && !roll && d.kind == MPE_SCN_SPREAD && d.homo && d.dim_c == 2 && amax > MPE_DUO_MIN_N && d.A <= kWave &&
         d.L <= kWave && (d.D & 1) == 0 && d.D <= 8 * kWave && (reinterpret_cast<uintptr_t>(b.obs) & 15) == 0 && MPE_DUO_ENABLE;
}

bool wide_supports(const WideDesc &d, bool out) {
  if (out && ((d.kind != MPE_SCN_SPREAD && d.kind != MPE_SCN_TAG) || d.dim_c != 2 || (d.D & 1))) return false;
  const Carve cv = carve(d.A, d.L);
  return cv.shared_bytes + kWavesPerWg * cv.wave_bytes <= kMaxLds;
}

int launch_wide(bool phys, bool out, const WideDesc &d_in, const MpeBuffers &b, size_t B, hipStream_t stream,
                const RollArgs *roll) {
  WideDesc d = d_in;
  // rows may go out nontemporal when no 128-byte line is shared between two wave stores (mpe_device.h): a row (k_wave /
  // k_duo) resp. the rows of a wave's worlds for one agent index (k_multi, set below) is a whole number of lines
  const bool lines = out && d.kind == MPE_SCN_SPREAD && (reinterpret_cast<uintptr_t>(b.obs) & 127) == 0 &&
                     (((size_t)B * d.D * sizeof(float)) & 127) == 0;
  d.rows_nt = lines && ((d.D * sizeof(float)) & 127) == 0;
  if (out && d.kind != MPE_SCN_SPREAD && d.kind != MPE_SCN_TAG) return MPE_EUNSUPPORTED;
  if (out && (d.dim_c != 2 || (d.D & 1))) return MPE_EUNSUPPORTED;
  const Carve cv = carve(d.A, d.L);
  const size_t lds = cv.shared_bytes + kWavesPerWg * cv.wave_bytes;
  if (lds > kMaxLds) return MPE_EUNSUPPORTED;
  static const int n_cu = [] {   // (a C++11 magic static: concurrent first calls from several host threads are fine)
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
    (void)hipGetLastError();
    return 256;
  }();
  // groups of kWavesPerWg worlds, padded to whole 64-group blocks so that the XCD-aware permutation
  // inside the kernel is a bijection; persistent grid of up to 8 workgroups per CU (a multiple of 64,
  // hence of 8: a workgroup's later iterations stay on its XCD's share of the worlds)
  const size_t groups = (B + kWavesPerWg - 1) / kWavesPerWg;
  const size_t padded = (groups + 63) / 64 * 64;
  if (padded > 0xffffffffull) return MPE_EUNSUPPORTED;
  constexpr int bpc = 8;
  size_t cap = (size_t)n_cu * (size_t)bpc / 64 * 64;
  if (cap < 64) cap = 64;
  const dim3 grid((unsigned)(padded < cap ? padded : cap)), block(kWavesPerWg * kWave);
  const unsigned np = (unsigned)padded;
  RollArgs ra;
  std::memset(&ra, 0, sizeof(ra));
  const int amax = d.A > d.L ? d.A : d.L;
  if (roll && duo_eligible(d, b, B, phys, out, false) && MPE_DUO_ROLL) {
    constexpr int G = MPE_DUO_G;
    const size_t padded_w = (B + 255) / 256 * 256;
    if (padded_w <= 0x7fffffffull) {
      const size_t dlds = (size_t)G * duo_roll_carve(d.A, d.L).slot_bytes;
      hipLaunchKernelGGL(k_duo_roll<G>, dim3((unsigned)(padded_w / G)), dim3(2 * G * kWave), dlds, stream, d, b, B, *roll);
      return (int)hipGetLastError();
    }
  }
  if (duo_eligible(d, b, B, phys, out, roll != nullptr)) {
    // two waves per world (k_duo): one workgroup per world, worlds padded to whole 256-world blocks of the XCD map
    constexpr int G = MPE_DUO_G;
    const size_t padded_w = (B + 255) / 256 * 256;   // whole 256-world blocks of the XCD map
    if (padded_w <= 0x7fffffffull) {
      const size_t dlds = (size_t)G * duo_carve(d.A, d.L).slot_bytes + (MPE_DUO_LDS_PAD);
      hipLaunchKernelGGL(k_duo<G>, dim3((unsigned)(padded_w / G)), dim3(2 * G * kWave), dlds, stream, d, b, B);
      return (int)hipGetLastError();
    }
  }
  if ((!roll || (phys && out && amax <= MPE_MULTI_ROLL_MAX_N)) && amax <= 32 && d.A + d.L <= kWave && !(out && d.kind != MPE_SCN_SPREAD)) {
    // several worlds per wave (k_multi): AP lanes per world slot
    const int AP = amax <= 8 ? 8 : amax <= 16 ? 16 : 32, WPW = kWave / AP;
    d.rows_nt = lines && (((size_t)WPW * d.D * sizeof(float)) & 127) == 0;
    const size_t mlds = cv.shared_bytes + (size_t)kWavesPerWg * WPW * (2 * (d.A + d.L) + 2 * d.A) * sizeof(float2);
    if (mlds <= 64 * 1024) {
      const size_t mg = (B + (size_t)kWavesPerWg * WPW - 1) / ((size_t)kWavesPerWg * WPW);
      const size_t mp = (mg + 63) / 64 * 64;
      const dim3 mgrid((unsigned)(mp < cap ? mp : cap));
      if (roll) hipLaunchKernelGGL((k_multi<true, true, true>), mgrid, block, mlds, stream, d, b, B, (unsigned)mp, AP, *roll);
      else if (phys && out) hipLaunchKernelGGL((k_multi<true, true, false>), mgrid, block, mlds, stream, d, b, B, (unsigned)mp, AP, ra);
      else if (phys) hipLaunchKernelGGL((k_multi<true, false, false>), mgrid, block, mlds, stream, d, b, B, (unsigned)mp, AP, ra);
      else hipLaunchKernelGGL((k_multi<false, true, false>), mgrid, block, mlds, stream, d, b, B, (unsigned)mp, AP, ra);
      return (int)hipGetLastError();
    }
  }
  if (lds > 64 * 1024) {  // beyond the default dynamic-LDS window: ask for it (a workgroup may own all 160 KiB of a gfx950 CU)
    const void *fn = roll ? (const void *)k_wave<true, true, true>
                          : phys && out ? (const void *)k_wave<true, true, false>
                                        : phys ? (const void *)k_wave<true, false, false> : (const void *)k_wave<false, true, false>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return MPE_EUNSUPPORTED;
    }
  }
  if (roll) {
    if (!(phys && out)) return MPE_EUNSUPPORTED;
    hipLaunchKernelGGL((k_wave<true, true, true>), grid, block, lds, stream, d, b, B, np, *roll);
  } else if (phys && out) hipLaunchKernelGGL((k_wave<true, true, false>), grid, block, lds, stream, d, b, B, np, ra);
  else if (phys) hipLaunchKernelGGL((k_wave<true, false, false>), grid, block, lds, stream, d, b, B, np, ra);
  else hipLaunchKernelGGL((k_wave<false, true, false>), grid, block, lds, stream, d, b, B, np, ra);
  return (int)hipGetLastError();
}

}  // namespace mpe
