// Write-pattern probe for the C4 observation buffer (tools/write_pattern.py): the same 402.7 MB written once per launch
//  mode 0  linear: workgroup g writes one contiguous 1/grid of the buffer (what a fill does)
//  mode 1  rows:   workgroup g owns `wpw` consecutive worlds and visits the 64 agent blocks in turn, writing the
//                  wpw rows of 1536 B of each (pieces of wpw * 1536 B at 6 MiB strides: what k_duo's row stream does)
// 16-byte stores (plain, nontemporal or sc1); 256 threads per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void st(v4 v, v4 *p) {
  if (NT == 1) __builtin_nontemporal_store(v, p);
  else if (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v));
  else *p = v;
}

template <int NT>
__global__ void __launch_bounds__(256) k_linear(v4 *out, size_t n4) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  const v4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (size_t k = lo + threadIdx.x; k < hi; k += 256) st<NT>(v, out + k);
}

template <int NT>
__global__ void __launch_bounds__(256) k_rows(v4 *out, int A, int B, int D4, int wpw, int rotate) {
  const int w0 = blockIdx.x * wpw;
  const v4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  const int piece = wpw * D4;   // float4 per (agent, world group)
  for (int a0 = 0; a0 < A; ++a0) {
    const int a = rotate ? (a0 + blockIdx.x) % A : a0;
    v4 *p = out + ((size_t)a * B + w0) * D4;
    for (int k = threadIdx.x; k < piece; k += 256) st<NT>(v, p + k);
  }
}

// mode 2  affinity: pages of `wpw` KiB; page p is written by a workgroup on XCD (p + rotate) % 8 (workgroup ids go round-robin
//         over the 8 XCDs), `grid` workgroups, each looping over its XCD's pages: is there an XCD <-> memory-channel affinity?
template <int NT>
__global__ void __launch_bounds__(256) k_aff(v4 *out, size_t n4, int page4, int shift) {
  const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const size_t pages = n4 / page4;
  const v4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  // pages p with (p + shift) % 8 == xcd, i.e. p = 8 q + ((xcd - shift) & 7); this workgroup takes q = j, j + per_xcd, ...
  for (size_t q = j; 8 * q + ((xcd - shift) & 7u) < pages; q += per_xcd) {
    v4 *p = out + (8 * q + ((xcd - shift) & 7u)) * page4;
    for (int k = threadIdx.x; k < page4; k += 256) st<NT>(v, p + k);
  }
}

// mode 3  transient: one short-lived workgroup per `wpw` KiB piece, pieces in address order (what torch's fill_ launches)
template <int NT>
__global__ void __launch_bounds__(256) k_transient(v4 *out, size_t n4, int piece4) {
  const v4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  v4 *p = out + (size_t)blockIdx.x * piece4;
  for (int k = threadIdx.x; k < piece4; k += 256) st<NT>(v, p + k);
}

// mode 4  rows in lock step: as mode 1, but no workgroup starts agent block a + K before every workgroup has finished
//         block a (K = `rotate`; a counter per agent block, all workgroups co-resident): the chip-wide write window is
//         K + 1 agent blocks instead of all 64
template <int NT>
__global__ void __launch_bounds__(256) k_rows_lock(v4 *out, int A, int B, int D4, int wpw, int K, unsigned *done) {
  const int w0 = blockIdx.x * wpw;
  const v4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  const int piece = wpw * D4;
  for (int a = 0; a < A; ++a) {
    if (a >= K) {
      if (threadIdx.x == 0)
        while (__hip_atomic_load(done + (a - K), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(2);
      __syncthreads();
    }
    v4 *p = out + ((size_t)a * B + w0) * D4;
    for (int k = threadIdx.x; k < piece; k += 256) st<NT>(v, p + k);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// mode: 0 linear, 1 rows, 2 affinity, 3 transient, 4 rows in lock step; + 16 * store kind (0 plain, 1 nontemporal, 2 sc1)
extern "C" int wp_launch(int mode, void *out, int A, int B, int D, int wpw, int rotate, int grid, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const int kind = mode >> 4;
  const size_t n4 = (size_t)A * B * D / 4;
  if ((mode & 15) == 0) {
    if (kind == 0) hipLaunchKernelGGL(k_linear<0>, dim3(grid), dim3(256), 0, s, (v4 *)out, n4);
    else if (kind == 1) hipLaunchKernelGGL(k_linear<1>, dim3(grid), dim3(256), 0, s, (v4 *)out, n4);
    else hipLaunchKernelGGL(k_linear<2>, dim3(grid), dim3(256), 0, s, (v4 *)out, n4);
  } else if ((mode & 15) == 4) {
    static unsigned *done = nullptr;
    if (!done && hipMalloc(&done, 256 * sizeof(unsigned)) != hipSuccess) return -1;
    (void)hipMemsetAsync(done, 0, 256 * sizeof(unsigned), s);
    if (kind == 1) hipLaunchKernelGGL(k_rows_lock<1>, dim3(B / wpw), dim3(256), 0, s, (v4 *)out, A, B, D / 4, wpw, rotate, done);
    else hipLaunchKernelGGL(k_rows_lock<2>, dim3(B / wpw), dim3(256), 0, s, (v4 *)out, A, B, D / 4, wpw, rotate, done);
  } else if ((mode & 15) == 3) {
    const int piece4 = wpw * 1024 / 16;
    const unsigned g = (unsigned)(n4 / piece4);
    if (kind == 0) hipLaunchKernelGGL(k_transient<0>, dim3(g), dim3(256), 0, s, (v4 *)out, n4, piece4);
    else if (kind == 1) hipLaunchKernelGGL(k_transient<1>, dim3(g), dim3(256), 0, s, (v4 *)out, n4, piece4);
    else hipLaunchKernelGGL(k_transient<2>, dim3(g), dim3(256), 0, s, (v4 *)out, n4, piece4);
  } else if ((mode & 15) == 2) {
    const int page4 = wpw * 1024 / 16;
    if (kind == 0) hipLaunchKernelGGL(k_aff<0>, dim3(grid), dim3(256), 0, s, (v4 *)out, n4, page4, rotate);
    else hipLaunchKernelGGL(k_aff<1>, dim3(grid), dim3(256), 0, s, (v4 *)out, n4, page4, rotate);
  } else {
    if (kind == 0) hipLaunchKernelGGL(k_rows<0>, dim3(B / wpw), dim3(256), 0, s, (v4 *)out, A, B, D / 4, wpw, rotate);
    else if (kind == 1) hipLaunchKernelGGL(k_rows<1>, dim3(B / wpw), dim3(256), 0, s, (v4 *)out, A, B, D / 4, wpw, rotate);
    else hipLaunchKernelGGL(k_rows<2>, dim3(B / wpw), dim3(256), 0, s, (v4 *)out, A, B, D / 4, wpw, rotate);
  }
  return (int)hipGetLastError();
}
