// mpe_wide.hip -- workgroup-per-world kernel for large entity counts (simple_spread N=64: 128 entities).
//
// One workgroup owns one world.  The world's positions/velocities are staged once in LDS
// ((E + A) float2 -- 1.5 KiB at N=64, a sliver of the CU's 160 KiB, so many worlds are resident per
// CU), the O(E^2) contact loop and the O(A*L + A^2) distance loops of the reward run out of LDS
// with the pair index spread over the lanes, min-distance / collision-count reductions are
// wave64 shuffles and ballots, and the observation rows -- 98 % of this kernel's HBM bytes
// (A rows of D floats: 98 KiB per world at N=64) -- leave as 16-byte stores that are contiguous
// along each row.  Same arithmetic and operation order per pair as mpe_narrow.hip / the reference
// (core.py:143-196, simple_spread.py:47-100); partial sums are regrouped (documented in DESIGN.md).
#include "mpe_internal.h"

namespace mpe {

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
  return v;
}
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

struct WideLds {  // carve-up of the dynamic LDS block (all offsets multiples of 16 bytes)
  float2 *pos, *vel, *u, *part;
  float *size, *mass, *maxspd, *lmin;
  int *flags, *cnt;
  float *red;
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline size_t wide_lds_bytes(int A, int L, int Q) {
  const int E = A + L;
  size_t n = 0;
  n += align16(sizeof(float2) * E);      // pos
  n += align16(sizeof(float2) * A);      // vel
  n += align16(sizeof(float2) * A);      // u
  n += align16(sizeof(float2) * A * Q);  // partial forces
  n += align16(sizeof(float) * E);       // size
  n += align16(sizeof(float) * A);       // mass
  n += align16(sizeof(float) * A);       // max_speed
  n += align16(sizeof(float) * (L > 0 ? L : 1));  // per-landmark min distance
  n += align16(sizeof(int) * E);         // flags
  n += align16(sizeof(int) * A);         // counts
  n += align16(sizeof(float) * 8);       // reduction results
  return n;
}

__device__ inline WideLds carve(char *base, int A, int L, int Q) {
  const int E = A + L;
  WideLds s;
  size_t o = 0;
  s.pos = reinterpret_cast<float2 *>(base + o); o += align16(sizeof(float2) * E);
  s.vel = reinterpret_cast<float2 *>(base + o); o += align16(sizeof(float2) * A);
  s.u = reinterpret_cast<float2 *>(base + o); o += align16(sizeof(float2) * A);
  s.part = reinterpret_cast<float2 *>(base + o); o += align16(sizeof(float2) * A * Q);
  s.size = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * E);
  s.mass = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * A);
  s.maxspd = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * A);
  s.lmin = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * (L > 0 ? L : 1));
  s.flags = reinterpret_cast<int *>(base + o); o += align16(sizeof(int) * E);
  s.cnt = reinterpret_cast<int *>(base + o); o += align16(sizeof(int) * A);
  s.red = reinterpret_cast<float *>(base + o);
  return s;
}

constexpr int kMovable = 1, kCollide = 2;

// observation element pair kp (floats 2kp, 2kp+1) of agent i's row -- simple_spread.py:84-100:
// [vel_i | pos_i | landmark_l - pos_i ... | pos_j - pos_i (j != i, ascending) ... | zeros]
__device__ __forceinline__ float2 spread_pair(const WideLds &s, int A, int L, int i, int kp, float2 me) {
  if (kp >= 2 + L + (A - 1)) return make_float2(0.f, 0.f);
  if (kp == 0) return s.vel[i];
  if (kp == 1) return me;
  int src;
  if (kp < 2 + L) src = A + (kp - 2);
  else { const int jj = kp - 2 - L; src = jj + (jj >= i ? 1 : 0); }
  const float2 p = s.pos[src];
  return make_float2(p.x - me.x, p.y - me.y);
}

template <bool PHYS, bool OUT>
__global__ void k_wide(const WideDesc d, const MpeBuffers b, const size_t B, const int Q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int A = d.A, L = d.L, E = A + L;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6, nwaves = nthr >> 6;
  const size_t w = blockIdx.x;
  const WideLds s = carve(smem, A, L, Q);
  const float *tab = b.entity_table;  // [6][E]: size, mass, accel, max_speed, movable, collide

  // ---- stage the world in LDS; decode actions (environment.py:144-181) -------------------------
  for (int e = tid; e < E; e += nthr) {
    s.pos[e] = make_float2(b.pos[(size_t)(2 * e) * B + w], b.pos[(size_t)(2 * e + 1) * B + w]);
    s.size[e] = tab[0 * E + e];
    s.flags[e] = (tab[4 * E + e] != 0.f ? kMovable : 0) | (tab[5 * E + e] != 0.f ? kCollide : 0);
  }
  for (int i = tid; i < A; i += nthr) {
    s.vel[i] = make_float2(b.vel[(size_t)(2 * i) * B + w], b.vel[(size_t)(2 * i + 1) * B + w]);
    s.mass[i] = 1.0f / tab[1 * E + i];  // inverse mass
    s.maxspd[i] = tab[3 * E + i];
    if (PHYS) {
      float ux, uy;
      const float sens = tab[2 * E + i];
      fetch_action(b, B, i, w, sens, ux, uy);
      s.u[i] = make_float2(ux + 0.f, uy + 0.f);
    }
  }
  __syncthreads();

  if (PHYS) {
    // ---- pairwise contact force (core.py:143-155,180-196): agent i x partner chunk q ------------
    const int CS = (E + Q - 1) / Q;
    for (int item = tid; item < A * Q; item += nthr) {
      const int i = item % A, q = item / A;
      float px_ = 0.f, py_ = 0.f;
      const int fi = s.flags[i];
      if ((fi & kCollide) && (fi & kMovable)) {
        const float2 me = s.pos[i];
        const float ri = s.size[i];
        const int j1 = min(E, (q + 1) * CS);
        for (int j = q * CS; j < j1; ++j) {
          if (j == i || !(s.flags[j] & kCollide)) continue;
          const float2 pj = s.pos[j];
          const float dmin = j > i ? ri + s.size[j] : s.size[j] + ri;
          float gx, gy;
          contact_force(me.x - pj.x, me.y - pj.y, dmin, d.cforce, d.cmargin, d.cmargin_inv, gx, gy);
          px_ = gx + px_;
          py_ = gy + py_;
        }
      }
      s.part[q * A + i] = make_float2(px_, py_);
    }
    __syncthreads();
    // ---- integrate (core.py:158-169) ----------------------------------------------------------
    for (int i = tid; i < A; i += nthr) {
      if (!(s.flags[i] & kMovable)) continue;
      float2 f = s.u[i];
      for (int q = 0; q < Q; ++q) {
        const float2 p = s.part[q * A + i];
        f.x = p.x + f.x;
        f.y = p.y + f.y;
      }
      float2 p = s.pos[i], v = s.vel[i];
      integrate_one(p.x, p.y, v.x, v.y, f.x, f.y, s.mass[i], s.maxspd[i], d.damp, d.dt);
      s.pos[i] = p;
      s.vel[i] = v;
      b.pos[(size_t)(2 * i) * B + w] = p.x;
      b.pos[(size_t)(2 * i + 1) * B + w] = p.y;
      b.vel[(size_t)(2 * i) * B + w] = v.x;
      b.vel[(size_t)(2 * i + 1) * B + w] = v.y;
    }
    __syncthreads();
  }

  if (!OUT) return;

  // ---- reward (simple_spread.py:72-82): per-landmark min over agents, per-agent contact count ---
  if (b.rew || b.info_rew) {
    for (int l = wave; l < L; l += nwaves) {
      const float2 pl = s.pos[A + l];
      float m = INFINITY;
      for (int a = lane; a < A; a += kWave) {
        const float2 pa = s.pos[a];
        m = fminf(m, dist2d(pa.x - pl.x, pa.y - pl.y));
      }
      m = wave_min(m);
      if (lane == 0) s.lmin[l] = m;
    }
    for (int i = wave; i < A; i += nwaves) {
      const float2 pi = s.pos[i];
      const float ri = s.size[i];
      int c = 0;
      if (s.flags[i] & kCollide) {
        for (int a0 = 0; a0 < A; a0 += kWave) {
          const int a = a0 + lane;
          bool hit = false;
          if (a < A) {
            const float2 pa = s.pos[a];
            hit = dist2d(pa.x - pi.x, pa.y - pi.y) < s.size[a] + ri;
          }
          c += __popcll(__ballot(hit));
        }
      }
      if (lane == 0) s.cnt[i] = c;
    }
    __syncthreads();
    if (wave == 0) {
      float neg = 0.f;
      int occ = 0;
      for (int l = lane; l < L; l += kWave) {
        const float m = s.lmin[l];
        neg = neg - m;
        occ += (m < 0.1f) ? 1 : 0;
      }
      neg = wave_sum(neg);
      occ = wave_sum_i(occ);
      float tot = 0.f;
      for (int i = lane; i < A; i += kWave) {
        float r = neg;
        const int c = s.cnt[i];
        for (int k = 0; k < c; ++k) r = r - 1.f;
        tot += r;
      }
      tot = wave_sum(tot);
      if (lane == 0) {
        s.red[0] = neg;
        s.red[1] = tot;
        s.red[2] = __int_as_float(occ);
      }
    }
    __syncthreads();
    const float neg = s.red[0], tot = s.red[1];
    const int occ = __float_as_int(s.red[2]);
    for (int i = tid; i < A; i += nthr) {
      float r = neg;
      const int c = s.cnt[i];
      for (int k = 0; k < c; ++k) r = r - 1.f;
      if (b.rew) b.rew[(size_t)i * B + w] = d.collaborative ? tot : r;
      if (b.done) b.done[(size_t)i * B + w] = 0;
      if (b.info_rew) {
        b.info_rew[(size_t)i * B + w] = r;
        b.info_collisions[(size_t)i * B + w] = c;
        b.info_min_dists[(size_t)i * B + w] = -neg;
        b.info_occupied[(size_t)i * B + w] = occ;
      }
    }
  } else if (b.done) {
    for (int i = tid; i < A; i += nthr) b.done[(size_t)i * B + w] = 0;
  }

  // ---- observation rows (simple_spread.py:84-100) ---------------------------------------------
  const int D = d.D;
  if ((D & 3) == 0 && (reinterpret_cast<uintptr_t>(b.obs) & 15) == 0) {
    const int Dq = D >> 2;
    const float inv = 1.0f / (float)Dq;
    const int total = A * Dq;
    for (int idx = tid; idx < total; idx += nthr) {
      const int i = (int)(((float)idx + 0.5f) * inv);  // exact: |err| << 0.5/Dq for idx < 2^20
      const int q4 = idx - i * Dq;
      const float2 me = s.pos[i];
      const float2 a0 = spread_pair(s, A, L, i, 2 * q4, me);
      const float2 a1 = spread_pair(s, A, L, i, 2 * q4 + 1, me);
      float4 *g = reinterpret_cast<float4 *>(b.obs + ((size_t)i * B + w) * D) + q4;
      *g = make_float4(a0.x, a0.y, a1.x, a1.y);
    }
  } else {
    const int Dp = D >> 1;  // D is even when dim_c is (checked on the host)
    const float inv = 1.0f / (float)Dp;
    const int total = A * Dp;
    for (int idx = tid; idx < total; idx += nthr) {
      const int i = (int)(((float)idx + 0.5f) * inv);
      const int kp = idx - i * Dp;
      const float2 v = spread_pair(s, A, L, i, kp, s.pos[i]);
      float *g = b.obs + ((size_t)i * B + w) * D + 2 * kp;
      g[0] = v.x;
      g[1] = v.y;
    }
  }
}

int launch_wide(bool phys, bool out, const WideDesc &d, const MpeBuffers &b, size_t B, hipStream_t stream) {
  if (out && d.kind != MPE_SCN_SPREAD) return MPE_EUNSUPPORTED;
  if (out && (d.dim_c != 2 || (d.D & 1))) return MPE_EUNSUPPORTED;
  const int A = d.A;
  const int nthr = A >= 48 ? 256 : A >= 24 ? 128 : 64;
  const int Q = nthr / A > 0 ? nthr / A : 1;
  const size_t lds = wide_lds_bytes(d.A, d.L, Q);
  if (lds > 160 * 1024) return MPE_EUNSUPPORTED;
  const dim3 grid((unsigned)B), block(nthr);
  if (phys && out) hipLaunchKernelGGL((k_wide<true, true>), grid, block, lds, stream, d, b, B, Q);
  else if (phys) hipLaunchKernelGGL((k_wide<true, false>), grid, block, lds, stream, d, b, B, Q);
  else hipLaunchKernelGGL((k_wide<false, true>), grid, block, lds, stream, d, b, B, Q);
  return (int)hipGetLastError();
}

}  // namespace mpe
