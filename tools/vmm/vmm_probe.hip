// vmm_probe.hip -- round-4 experiment: can a FAST observation buffer be CONSTRUCTED?
//
// Round 3 found that the rate at which the N=64 step streams its 384 MiB of rows is a property of the ALLOCATION (67-68 us into
// some buffers, 81-84 us into others) and could only SELECT among whole allocations.  "A 768 MiB block is fast in
// its upper half only" says speed is a property of physical regions of ~100s of MB.  The HIP virtual-memory API lets user code
// own physical chunks and decide which of them back a virtual range: this helper creates N physical chunks, times a write
// kernel on each (mapped alone), and maps any chosen sequence of them under one contiguous address range.
//
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/vmm/libvmm.so tools/vmm/vmm_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

namespace {
struct State {
  int dev = 0;
  size_t chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> h;
  void *scratch = nullptr;                 // VA for probing one chunk at a time
  std::vector<std::pair<void *, size_t>> ranges;
} g;
char g_err[256] = "";
int fail(hipError_t e, const char *what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e ? (int)e : -1;
}
#define CK(x, what) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(e_, what); } while (0)

hipMemAccessDesc access_desc() {
  hipMemAccessDesc d = {};
  d.location.type = hipMemLocationTypeDevice;
  d.location.id = g.dev;
  d.flags = hipMemAccessFlagsProtReadWrite;
  return d;
}

// persistent workgroups, each writing ONE contiguous slice (the pattern that tells fast from slow buffers: DESIGN 2.7, second pass)
__global__ void __launch_bounds__(256) k_slices(float4 *p, size_t n4, float v) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  typedef float vf4 __attribute__((ext_vector_type(4)));
  const vf4 x = {v, v, v, v};
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) __builtin_nontemporal_store(x, reinterpret_cast<vf4 *>(p + i));
}
// fill-shaped: one short-lived workgroup per 4 KB in address order (placement-insensitive in round 3: the control)
__global__ void __launch_bounds__(256) k_fill(float4 *p, size_t n4, float v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) p[i] = make_float4(v, v, v, v);
}
}  // namespace

extern "C" {
const char *vmm_last_error() { return g_err; }

// n physical chunks of `chunk_bytes` (rounded up to the granularity) on the current device; returns the chunk size or < 0
long long vmm_create(int n, long long chunk_bytes) {
  CK(hipGetDevice(&g.dev), "hipGetDevice");
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = g.dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended), "hipMemGetAllocationGranularity");
  g.chunk = ((size_t)chunk_bytes + gran - 1) / gran * gran;
  for (int k = 0; k < n; ++k) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, g.chunk, &prop, 0), "hipMemCreate");
    g.h.push_back(h);
  }
  CK(hipMemAddressReserve(&g.scratch, g.chunk, 0, nullptr, 0), "hipMemAddressReserve(scratch)");
  return (long long)g.chunk;
}

// time `reps` launches of write pattern `mode` (0 slices, 1 fill) on chunk k, mapped alone: us per launch, or < 0
double vmm_probe(int k, int mode, int reps, int grid) {
  if (k < 0 || k >= (int)g.h.size()) return -1.0;
  if (hipMemMap(g.scratch, g.chunk, 0, g.h[k], 0) != hipSuccess) return -2.0;
  hipMemAccessDesc d = access_desc();
  if (hipMemSetAccess(g.scratch, g.chunk, &d, 1) != hipSuccess) return -3.0;
  const size_t n4 = g.chunk / 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&](float v) {
    if (mode == 0) hipLaunchKernelGGL(k_slices, dim3(grid), dim3(256), 0, 0, (float4 *)g.scratch, n4, v);
    else hipLaunchKernelGGL(k_fill, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (float4 *)g.scratch, n4, v);
  };
  for (int r = 0; r < 3; ++r) launch(1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch((float)r);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipDeviceSynchronize();
  if (hipMemUnmap(g.scratch, g.chunk) != hipSuccess) return -4.0;
  return (double)ms * 1e3 / reps;
}

// one contiguous virtual range backed by chunks order[0..n) in that order; returns the device pointer (0 on failure)
void *vmm_compose(const int *order, int n) {
  void *base = nullptr;
  const size_t total = g.chunk * (size_t)n;
  if (hipMemAddressReserve(&base, total, 0, nullptr, 0) != hipSuccess) { snprintf(g_err, sizeof(g_err), "reserve failed"); return nullptr; }
  for (int i = 0; i < n; ++i) {
    if (order[i] < 0 || order[i] >= (int)g.h.size()) return nullptr;
    hipError_t e = hipMemMap((char *)base + (size_t)i * g.chunk, g.chunk, 0, g.h[order[i]], 0);
    if (e != hipSuccess) { fail(e, "hipMemMap"); return nullptr; }
  }
  hipMemAccessDesc d = access_desc();
  hipError_t e = hipMemSetAccess(base, total, &d, 1);
  if (e != hipSuccess) { fail(e, "hipMemSetAccess"); return nullptr; }
  g.ranges.push_back({base, total});
  return base;
}

// release every range and every physical chunk (a new vmm_create may follow, e.g. with another chunk size)
int vmm_destroy() {
  for (auto &r : g.ranges)
    if (r.first) {
      hipMemUnmap(r.first, r.second);
      hipMemAddressFree(r.first, r.second);
    }
  g.ranges.clear();
  for (auto h : g.h) hipMemRelease(h);
  g.h.clear();
  if (g.scratch) hipMemAddressFree(g.scratch, g.chunk);
  g.scratch = nullptr;
  return 0;
}

int vmm_release_range(void *base) {
  for (auto &r : g.ranges)
    if (r.first == base) {
      CK(hipMemUnmap(base, r.second), "hipMemUnmap");
      CK(hipMemAddressFree(base, r.second), "hipMemAddressFree");
      r.first = nullptr;
      return 0;
    }
  return -1;
}
}
