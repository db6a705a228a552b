// launch_floor.hip -- what does ONE dependent launch of the simple_spread N=3 step cost on this GPU
// before any arithmetic?  Replays, from a HIP graph, chains of
//   null      an empty kernel with the step kernel's grid (1024 x 192 threads at B = 65536)
//   stream    a kernel that only moves the step's algorithmic bytes with the step's access pattern
//             (33 coalesced dword loads per world, 54+ floats of 16-byte stores, ...)
// and prints microseconds per launch.  The gap between `stream` and the real kernel is what the
// arithmetic / latency chain costs; the gap between `null` and 0 is the fixed launch cost.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_null(float *p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 0; }

// per world: read 12 pos + 6 vel + 15 act floats, write 6 pos/vel... as the step does (agents only: 12),
// 54 obs floats, 3 rew floats, 3 done bytes.  One lane per (world, agent) like k_split: WG = 3 waves.
__global__ void __launch_bounds__(192) k_stream(const float *__restrict__ pos, const float *__restrict__ vel,
                                                 const float *__restrict__ act, float *__restrict__ pos_o,
                                                 float *__restrict__ vel_o, float *__restrict__ obs,
                                                 float *__restrict__ rew, unsigned char *__restrict__ done, size_t B) {
  const int lane = threadIdx.x & 63, i = threadIdx.x >> 6;
  const size_t w0 = (size_t)blockIdx.x * 64, w = w0 + lane;
  if (w >= B) return;
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 12; ++e) acc += pos[(size_t)e * B + w];
  acc += vel[(size_t)(2 * i) * B + w] + vel[(size_t)(2 * i + 1) * B + w];
  const float *a = act + ((size_t)i * B + w) * 5;
  acc += (a[1] - a[2]) + (a[3] - a[4]);
  pos_o[(size_t)(2 * i) * B + w] = acc;
  pos_o[(size_t)(2 * i + 1) * B + w] = acc;
  vel_o[(size_t)(2 * i) * B + w] = acc;
  vel_o[(size_t)(2 * i + 1) * B + w] = acc;
  float4 *o = reinterpret_cast<float4 *>(obs + (size_t)i * B * 18 + w0 * 18);
  for (int q = lane; q < 288; q += 64) o[q] = make_float4(acc, acc, acc, acc);  // 64 rows x 18 floats
  rew[(size_t)i * B + w] = acc;
  done[(size_t)i * B + w] = 0;
}

template <class F>
static float chain_us(F launch, int n, hipStream_t s) {
  hipGraph_t g; hipGraphExec_t ge;
  launch(); CK(hipStreamSynchronize(s));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < n; ++k) launch();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3f / n;
}

int main(int argc, char **argv) {
  const size_t B = argc > 1 ? strtoull(argv[1], 0, 10) : 65536;
  hipStream_t s; CK(hipStreamCreate(&s));
  float *pos, *vel, *pos_o, *vel_o, *obs, *rew; unsigned char *done;
  std::vector<float *> acts(16);
  CK(hipMalloc(&pos, 12 * B * 4)); CK(hipMalloc(&vel, 6 * B * 4)); CK(hipMalloc(&pos_o, 12 * B * 4)); CK(hipMalloc(&vel_o, 6 * B * 4));
  CK(hipMalloc(&obs, 2 * 54 * B * 4)); CK(hipMalloc(&rew, 2 * 3 * B * 4)); CK(hipMalloc(&done, 2 * 3 * B));
  for (auto &a : acts) { CK(hipMalloc(&a, 15 * B * 4)); CK(hipMemsetAsync(a, 0, 15 * B * 4, s)); }
  CK(hipMemsetAsync(pos, 0, 12 * B * 4, s)); CK(hipMemsetAsync(vel, 0, 6 * B * 4, s));
  const unsigned grid = (unsigned)((B + 63) / 64);
  int k = 0;
  const double bytes = (33.0 + 69.0) * 4 * B + 3 * B;
  float t_null = chain_us([&] { hipLaunchKernelGGL(k_null, dim3(grid), dim3(192), 0, s, pos); }, 400, s);
  float t_null1 = chain_us([&] { hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s, pos); }, 400, s);
  float t_str = chain_us([&] {
    hipLaunchKernelGGL(k_stream, dim3(grid), dim3(192), 0, s, pos, vel, acts[k & 15], pos, vel, obs + (k & 1) * 54 * B,
                       rew + (k & 1) * 3 * B, done + (k & 1) * 3 * B, B); ++k; }, 400, s);
  printf("B=%zu grid=%u x 192\n", B, grid);
  printf("null kernel, 1 WG            : %7.2f us/launch\n", t_null1);
  printf("null kernel, step grid       : %7.2f us/launch\n", t_null);
  printf("stream kernel (step's bytes) : %7.2f us/launch  -> %.0f GB/s of the %.1f MB algorithmic bytes\n", t_str,
         bytes / t_str / 1e3, bytes / 1e6);
  return 0;
}
