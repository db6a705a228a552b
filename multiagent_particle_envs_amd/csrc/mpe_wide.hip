// mpe_wide.hip -- workgroup-per-world kernel for large entity counts (simple_spread N=64: 128 entities).
//
// One workgroup owns one world.  The world's positions/velocities are staged once in LDS
// ((E + A) float2 -- 1.5 KiB at N=64, a sliver of the CU's 160 KiB, so many worlds are resident per
// CU) and everything else runs out of LDS:
//   contacts   thread = (agent i, partner chunk q).  Pass 1 builds a bitmask of the partners that can
//              exert a non-zero force (squared distance under (r_i + r_j + 0.02)^2: beyond that the
//              fp32 soft-plus term is exactly 0), pass 2 evaluates only those, in ascending order.
//              At N=64 that is ~1-5 of 32 partners per thread instead of all of them.
//   reward     per-landmark min over agents on SQUARED distances (sqrt is monotone: one correctly
//              rounded sqrt per landmark afterwards gives the same value as min over sqrt's), per-agent
//              contact counts with the exact sqrt_lt test, partials combined through LDS, the sums by
//              wave64 shuffle reductions.
//   obs        98 % of the kernel's HBM bytes (A rows x D floats: 98 KiB per world at N=64).  Rows are
//              computed straight from LDS in output order: consecutive lanes write consecutive 16-byte
//              pieces of a row, consecutive waves consecutive KiB, so each row (1536 B) is one short
//              sequential burst -- the order HBM rewards (see the note at the store loop).
// Per pair the arithmetic is mpe_device.h's, as in the thread-per-world kernels; partial sums are
// regrouped by partner chunk / reduction tree (documented in DESIGN.md 4).
#include <cstdlib>

#include "mpe_internal.h"

namespace mpe {

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
  float *size, *inv_mass, *maxspd, *lmin, *lpart;
  int *flags, *cnt, *cpart;
  float *red;
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }

// partial-result slots: Q partner chunks per agent (forces), Qr agent chunks per landmark (min
// distance), Qc agent chunks per agent (contact counts) -- all "threads / items", at least 1
struct WideSplit { int Q, Qr, Qc; };
__host__ __device__ inline WideSplit wide_split(int A, int L, int nthr) {
  WideSplit s;
  s.Q = imax(1, nthr / A);
  s.Qr = imax(1, nthr / imax(L, 1));
  s.Qc = imax(1, nthr / A);
  return s;
}

__host__ __device__ inline size_t wide_lds_bytes(int A, int L, int nthr) {
  const int E = A + L;
  const WideSplit sp = wide_split(A, L, nthr);
  size_t n = 0;
  n += align16(sizeof(float2) * E);            // pos
  n += align16(sizeof(float2) * A);            // vel
  n += align16(sizeof(float2) * A);            // u
  n += align16(sizeof(float2) * A * sp.Q);     // partial forces
  n += align16(sizeof(float) * E);             // size
  n += align16(sizeof(float) * A);             // inverse mass
  n += align16(sizeof(float) * A);             // max_speed
  n += align16(sizeof(float) * imax(L, 1));    // per-landmark min distance
  n += align16(sizeof(float) * imax(L, 1) * sp.Qr);  // partial min of squared distances
  n += align16(sizeof(int) * E);               // flags
  n += align16(sizeof(int) * A);               // counts
  n += align16(sizeof(int) * A * sp.Qc);       // partial counts
  n += align16(sizeof(float) * 8);             // reduction results
  return n;
}

__device__ inline WideLds carve(char *base, int A, int L, int nthr) {
  const int E = A + L;
  const WideSplit sp = wide_split(A, L, nthr);
  WideLds s;
  size_t o = 0;
  s.pos = reinterpret_cast<float2 *>(base + o); o += align16(sizeof(float2) * E);
  s.vel = reinterpret_cast<float2 *>(base + o); o += align16(sizeof(float2) * A);
  s.u = reinterpret_cast<float2 *>(base + o); o += align16(sizeof(float2) * A);
  s.part = reinterpret_cast<float2 *>(base + o); o += align16(sizeof(float2) * A * sp.Q);
  s.size = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * E);
  s.inv_mass = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * A);
  s.maxspd = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * A);
  s.lmin = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * imax(L, 1));
  s.lpart = reinterpret_cast<float *>(base + o); o += align16(sizeof(float) * imax(L, 1) * sp.Qr);
  s.flags = reinterpret_cast<int *>(base + o); o += align16(sizeof(int) * E);
  s.cnt = reinterpret_cast<int *>(base + o); o += align16(sizeof(int) * A);
  s.cpart = reinterpret_cast<int *>(base + o); o += align16(sizeof(int) * A * sp.Qc);
  s.red = reinterpret_cast<float *>(base + o);
  return s;
}

constexpr int kMovable = 1, kCollide = 2;
constexpr int kPF = 2;  // entities prefetched per thread => E <= kPF * blockDim
// Beyond dist_min + kFarMargin the fp32 soft-plus term is exactly zero: x = (dist_min - d)/k < -16.64
// => 1 + exp(x) rounds to 1 => log = 0 => penetration = 0 => force = +-0 (k = 1e-3 => 0.01664; the
// margin is scaled by k/1e-3 on the host side of the comparison below).
constexpr float kFarX = 20.0f;  // in units of contact_margin

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

__device__ __forceinline__ bool getenv_world_major(const WideDesc &d) { return d.obs_world_major != 0; }

template <bool PHYS, bool OUT>
__global__ void k_wide(const WideDesc d, const MpeBuffers b, const size_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int A = d.A, L = d.L, E = A + L;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const WideLds s = carve(smem, A, L, nthr);
  const WideSplit sp = wide_split(A, L, nthr);
  const float *tab = b.entity_table;  // [6][E]: size, mass, accel, max_speed, movable, collide

  // per-entity constants: staged once per workgroup (the workgroup then walks over several worlds)
  for (int e = tid; e < E; e += nthr) {
    s.size[e] = tab[0 * E + e];
    s.flags[e] = (tab[4 * E + e] != 0.f ? kMovable : 0) | (tab[5 * E + e] != 0.f ? kCollide : 0);
  }
  for (int i = tid; i < A; i += nthr) {
    s.inv_mass[i] = 1.0f / tab[1 * E + i];
    s.maxspd[i] = tab[3 * E + i];
  }

  // Persistent workgroup: worlds w = blockIdx.x, + gridDim.x, ...  The next world's state and action
  // rows are fetched into registers (entity e = tid + k*nthr, k < kPF) while this world's
  // observation rows are being stored; stores are fire-and-forget, so a world's 98 KiB drain while
  // the next world's contacts are computed.
  float2 npos[kPF], nvel[kPF], nu[kPF];
  float sens[kPF];
#pragma unroll
  for (int k = 0; k < kPF; ++k) {
    const int e = tid + k * nthr;
    sens[k] = (PHYS && e < A) ? tab[2 * E + e] : 0.f;
    npos[k] = nvel[k] = nu[k] = make_float2(0.f, 0.f);
  }
  auto prefetch = [&](size_t wn) {
#pragma unroll
    for (int k = 0; k < kPF; ++k) {
      const int e = tid + k * nthr;
      if (e < E) npos[k] = make_float2(b.pos[(size_t)(2 * e) * B + wn], b.pos[(size_t)(2 * e + 1) * B + wn]);
      if (e < A) {
        nvel[k] = make_float2(b.vel[(size_t)(2 * e) * B + wn], b.vel[(size_t)(2 * e + 1) * B + wn]);
        if (PHYS) {  // decode actions (environment.py:144-181)
          float ux, uy;
          fetch_action(b, B, e, wn, sens[k], ux, uy);
          nu[k] = make_float2(ux + 0.f, uy + 0.f);
        }
      }
    }
  };
  size_t w = blockIdx.x;
  if (w < B) prefetch(w);
  if (d.stagger > 0) {  // experiment: break the phase lockstep of co-resident workgroups
    const int gen = (int)(blockIdx.x >> 8) & 7;   // dispatch round on a 256-CU part
    for (int k = 0; k < gen * d.stagger; ++k) __builtin_amdgcn_s_sleep(127);
  }

  for (; w < B; w += gridDim.x) {
  // ---- commit the prefetched world to LDS ------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < kPF; ++k) {
    const int e = tid + k * nthr;
    if (e < E) s.pos[e] = npos[k];
    if (e < A) {
      s.vel[e] = nvel[k];
      if (PHYS) s.u[e] = nu[k];
    }
  }
  __syncthreads();

  if (PHYS) {
    // ---- pairwise contact force (core.py:143-155,180-196): agent i x partner chunk q ------------
    const int Q = sp.Q;
    const int CS = (E + Q - 1) / Q;
    const float far = kFarX * d.cmargin;
    for (int item = tid; item < A * Q; item += nthr) {
      const int i = item % A, q = item / A;
      float ax = 0.f, ay = 0.f;
      const int fi = s.flags[i];
      if ((fi & kCollide) && (fi & kMovable)) {
        const float2 me = s.pos[i];
        const float ri = s.size[i];
        const int j1 = min(E, (q + 1) * CS);
        for (int jb = q * CS; jb < j1; jb += 64) {
          const int je = min(j1, jb + 64);
          unsigned long long near = 0ull;
          for (int j = jb; j < je; ++j) {  // pass 1: who is close enough to push at all
            if (j == i || !(s.flags[j] & kCollide)) continue;
            const float2 pj = s.pos[j];
            const float reach = ri + s.size[j] + far;
            if (sq2d(me.x - pj.x, me.y - pj.y) < reach * reach) near |= 1ull << (j - jb);
          }
          while (near) {  // pass 2: those only, ascending (Q9)
            const int j = jb + __ffsll((long long)near) - 1;
            near &= near - 1;
            const float2 pj = s.pos[j];
            float gx, gy;
            contact_force(me.x - pj.x, me.y - pj.y, ri + s.size[j], d.cforce, d.cmargin, d.cmargin_inv, gx, gy);
            ax = gx + ax;
            ay = gy + ay;
          }
        }
      }
      s.part[q * A + i] = make_float2(ax, ay);
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
      integrate_one(p.x, p.y, v.x, v.y, f.x, f.y, s.inv_mass[i], s.maxspd[i], d.damp, d.dt);
      s.pos[i] = p;
      s.vel[i] = v;
      b.pos[(size_t)(2 * i) * B + w] = p.x;
      b.pos[(size_t)(2 * i + 1) * B + w] = p.y;
      b.vel[(size_t)(2 * i) * B + w] = v.x;
      b.vel[(size_t)(2 * i + 1) * B + w] = v.y;
    }
    __syncthreads();
  }

  if (OUT) {
  // ---- reward (simple_spread.py:72-82) -----------------------------------------------------------
  if (b.rew || b.info_rew) {
    {  // partial min over an agent chunk of the squared distance to landmark l
      const int Qr = sp.Qr, CA = (A + Qr - 1) / Qr;
      for (int item = tid; item < L * Qr; item += nthr) {
        const int l = item % L, q = item / L;
        const float2 pl = s.pos[A + l];
        float m2 = INFINITY;
        const int a1 = min(A, (q + 1) * CA);
        for (int a = q * CA; a < a1; ++a) {
          const float2 pa = s.pos[a];
          m2 = fminf(m2, sq2d(pa.x - pl.x, pa.y - pl.y));
        }
        s.lpart[q * L + l] = m2;
      }
    }
    {  // partial contact count of agent i against an agent chunk (includes i itself)
      const int Qc = sp.Qc, CA = (A + Qc - 1) / Qc;
      for (int item = tid; item < A * Qc; item += nthr) {
        const int i = item % A, q = item / A;
        const float2 pi = s.pos[i];
        const float ri = s.size[i];
        int c = 0;
        const int a1 = min(A, (q + 1) * CA);
        for (int a = q * CA; a < a1; ++a) {
          const float2 pa = s.pos[a];
          c += sqrt_lt(sq2d(pa.x - pi.x, pa.y - pi.y), s.size[a] + ri) ? 1 : 0;
        }
        s.cpart[q * A + i] = c;
      }
    }
    __syncthreads();
    for (int l = tid; l < L; l += nthr) {
      float m2 = s.lpart[l];
      for (int q = 1; q < sp.Qr; ++q) m2 = fminf(m2, s.lpart[q * L + l]);
      s.lmin[l] = sqrtf(m2);  // == min over agents of the rounded distances
    }
    for (int i = tid; i < A; i += nthr) {
      int c = 0;
      for (int q = 0; q < sp.Qc; ++q) c += s.cpart[q * A + i];
      s.cnt[i] = (s.flags[i] & kCollide) ? c : 0;
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

  // next world's loads go out now and complete under the store loop below
  if (w + gridDim.x < B) prefetch(w + gridDim.x);

  // ---- observation rows (simple_spread.py:84-100) ---------------------------------------------
  const int D = d.D;
  // agent-major (default): agent i's rows form one [B][D] block => rows of one world are B*D apart;
  // world-major: one world's A rows are contiguous ([B][A][D])
  const bool world_major = getenv_world_major(d);
  const size_t rowlen = world_major ? (size_t)D : (size_t)B * D;
  float *const obs_w = b.obs + (world_major ? w * (size_t)A * D : w * (size_t)D);
  if ((D & 3) == 0 && (reinterpret_cast<uintptr_t>(b.obs) & 15) == 0) {
    // Flat order: consecutive lanes write consecutive 16-byte pieces of a row and consecutive waves
    // consecutive KiB, rows in order -- every row is completed within one sweep of the workgroup.
    // (Measured on MI355X, N=64, B=4096: 83 us for the 403 MB of rows vs 118 us when the same bytes
    // are written region by region with cheaper per-thread operands: HBM wants the sequential stream.)
    const int Dq = D >> 2;                     // 16-byte columns per row
    const int kpz = 2 + L + (A - 1);           // first all-zero pair
    int i = tid / Dq, q4 = tid - i * Dq;       // one division per thread, then incremental
    const int di = nthr / Dq, dq = nthr - di * Dq;
    for (; i < A;) {
      const float2 me = s.pos[i];
      float2 o[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kp = 2 * q4 + h;
        // source entity of pair kp: landmark kp-2, or other agent jj = kp-2-L shifted past i
        const int jj = kp - 2 - L;
        int src = kp < 2 + L ? A + (kp - 2) : jj + (jj >= i ? 1 : 0);
        src = min(max(src, 0), E - 1);
        const float2 p = s.pos[src];
        float2 v = make_float2(p.x - me.x, p.y - me.y);
        if (kp == 0) v = s.vel[i];
        if (kp == 1) v = me;
        if (kp >= kpz) v = make_float2(0.f, 0.f);
        o[h] = v;
      }
      *reinterpret_cast<float4 *>(obs_w + i * rowlen + 4 * q4) = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
      i += di;
      q4 += dq;
      if (q4 >= Dq) { q4 -= Dq; ++i; }
    }
  } else {
    const int Dp = D >> 1;  // D is even when dim_c is (checked on the host)
    const float inv = 1.0f / (float)Dp;
    const int total = A * Dp;
    for (int idx = tid; idx < total; idx += nthr) {
      const int i = (int)(((float)idx + 0.5f) * inv);  // exact: |err| << 0.5/Dp for idx < 2^20
      const int kp = idx - i * Dp;
      const float2 v = spread_pair(s, A, L, i, kp, s.pos[i]);
      float *g = obs_w + i * rowlen + 2 * kp;
      g[0] = v.x;
      g[1] = v.y;
    }
  }
  } else if (w + gridDim.x < B) {
    prefetch(w + gridDim.x);
  }
  __syncthreads();  // every LDS read of this world is done before the next one is committed
  }
}

int launch_wide(bool phys, bool out, const WideDesc &d, const MpeBuffers &b, size_t B, hipStream_t stream) {
  if (out && d.kind != MPE_SCN_SPREAD) return MPE_EUNSUPPORTED;
  if (out && (d.dim_c != 2 || (d.D & 1))) return MPE_EUNSUPPORTED;
  const int A = d.A;
  int nthr = A >= 48 ? 256 : A >= 24 ? 128 : 64;
  while (nthr * kPF < d.A + d.L) nthr *= 2;   // register prefetch covers kPF entities per thread
  if (const char *e = std::getenv("MPE_WIDE_NTHR")) nthr = std::atoi(e);  // tuning experiment switch
  if (nthr > 1024 || nthr * kPF < d.A + d.L) return MPE_EUNSUPPORTED;
  const size_t lds = wide_lds_bytes(d.A, d.L, nthr);
  if (lds > 64 * 1024) return MPE_EUNSUPPORTED;
  // persistent grid: enough workgroups to fill every CU (256 CUs x 2048 threads on MI355X), each
  // walking over worlds blockIdx.x, + gridDim.x, ...
  int bpc = 2048 / nthr;
  if (const char *e = std::getenv("MPE_WIDE_BPC")) bpc = std::atoi(e);
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
    else n_cu = 256;
  }
  const size_t want = (size_t)n_cu * (size_t)(bpc > 0 ? bpc : 1);
  const dim3 grid((unsigned)(B < want ? B : want)), block(nthr);
  if (phys && out) hipLaunchKernelGGL((k_wide<true, true>), grid, block, lds, stream, d, b, B);
  else if (phys) hipLaunchKernelGGL((k_wide<true, false>), grid, block, lds, stream, d, b, B);
  else hipLaunchKernelGGL((k_wide<false, true>), grid, block, lds, stream, d, b, B);
  return (int)hipGetLastError();
}

}  // namespace mpe
