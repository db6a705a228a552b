// store_pattern.hip -- how fast can one MI355X absorb the simple_spread N=64 observation block
// (B x A rows of D floats = 403 MB at B=4096, A=64, D=384) under different store organisations?
// Every variant writes the same bytes (a constant, no LDS, no arithmetic) so only the store stream
// differs.  Prints TB/s per variant (HIP-graph replay of 20 launches, best of 5).
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>

typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(float4 v, float4 *p) {
  vf4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<vf4 *>(p));
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int A = 64, D = 384, Dq = D / 4;

// V0: one float4 per thread, linear (what a fill does)
__global__ void k_linear(float4 *o, size_t n4) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n4) o[t] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// V9: linear order like V0, but every float4 is computed the way a two-kernel observation emitter would: thread t owns
// piece q of row (agent i, world w) in MEMORY order, reads the world's positions from a world-major scratch copy
// Qs [B][128] float2 (coalesced: consecutive pieces read consecutive entities) and its own position (broadcast).
__global__ void __launch_bounds__(256) k_linear_rows(float4 *o, const float2 *__restrict__ Qs, unsigned Bw) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [A][B][Dq]
  const unsigned per_agent = Bw * (unsigned)Dq;
  const unsigned i = (unsigned)(t / per_agent);
  const unsigned rem = (unsigned)(t - (size_t)i * per_agent);
  const unsigned w = rem / (unsigned)Dq, q = rem - w * (unsigned)Dq;
  if (i >= (unsigned)A) return;
  const float2 *Qw = Qs + (size_t)w * 128;
  const float2 me = Qw[64 + i];
  const int idx0 = 2 * (int)q - 2, thr = 64 + (int)i;
  int s0 = idx0 + (idx0 >= thr), s1 = idx0 + 1 + (idx0 + 1 >= thr);
  s0 = s0 < 0 ? 0 : (s0 > 127 ? 127 : s0);
  s1 = s1 < 0 ? 0 : (s1 > 127 ? 127 : s1);
  const float2 p0 = Qw[s0], p1 = Qw[s1];
  float4 v = make_float4(p0.x - me.x, p0.y - me.y, p1.x - me.x, p1.y - me.y);
  if (2 * q >= 129) { v.x = 0.f; v.y = 0.f; }
  if (2 * q + 1 >= 129) { v.z = 0.f; v.w = 0.f; }
  if (q == 0) v = make_float4(me.x, me.y, me.x, me.y);
  o[t] = v;
}

// V6: grid-stride fill with a small persistent grid
__global__ void k_gridstride(float4 *o, size_t n4) {
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (size_t)gridDim.x * blockDim.x)
    o[t] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// wave per world; rows of the world are `rowlen` floats apart; MODE 0: 96 full flat stores, 1: per row 64+32 lanes
template <int MODE, bool NT, int WPW /* waves per world */>
__global__ void __launch_bounds__(256) k_world(float *obs, size_t B, size_t rowlen, size_t worldlen) {
  const int lane = threadIdx.x & 63;
  const size_t gw = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // global wave
  const size_t w = gw / WPW;
  const int part = (int)(gw % WPW);
  if (w >= B) return;
  float *base = obs + w * worldlen;
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
  const int r0 = part * (A / WPW), r1 = r0 + A / WPW;
  if (MODE == 0) {
    for (int f = r0 * Dq + lane; f < r1 * Dq; f += 64) {
      const int i = f / Dq, q = f - i * Dq;
      float4 *p = reinterpret_cast<float4 *>(base + (size_t)i * rowlen) + q;
      if (NT) nt_store(v, p); else *p = v;
    }
  } else {
    for (int i = r0; i < r1; ++i) {
      float4 *p = reinterpret_cast<float4 *>(base + (size_t)i * rowlen);
      if (NT) { nt_store(v, p + lane); if (lane < Dq - 64) nt_store(v, p + 64 + lane); }
      else { p[lane] = v; if (lane < Dq - 64) p[64 + lane] = v; }
    }
  }
}
// wave per world, per row 64+32 lanes, but the row ORDER is rotated: world w starts at row (w / GROUP) % A, so that at
// any instant the waves are spread evenly over the A row windows (each window still sees contiguous GROUP-world runs)
template <int GROUP>
__global__ void __launch_bounds__(256) k_world_rot(float *obs, size_t B, size_t rowlen, size_t worldlen) {
  const int lane = threadIdx.x & 63;
  const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w >= B) return;
  float *base = obs + w * worldlen;
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
  const int start = (int)((w / GROUP) % A);
  for (int k = 0; k < A; ++k) {
    int i = start + k;
    if (i >= A) i -= A;
    float4 *p = reinterpret_cast<float4 *>(base + (size_t)i * rowlen);
    p[lane] = v;
    if (lane < Dq - 64) p[64 + lane] = v;
  }
}

// workgroup (256 threads) per world, flat: thread writes pieces tid, tid+256, ...
__global__ void __launch_bounds__(256) k_wgworld(float *obs, size_t B, size_t rowlen, size_t worldlen) {
  const size_t w = blockIdx.x;
  float *base = obs + w * worldlen;
  for (int f = threadIdx.x; f < A * Dq; f += 256) {
    const int i = f / Dq, q = f - i * Dq;
    reinterpret_cast<float4 *>(base + (size_t)i * rowlen)[q] = make_float4(1.f, 2.f, 3.f, 4.f);
  }
}

static float run(std::function<void()> launch, hipStream_t s) {
  const int n = 20;
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
  const size_t B = argc > 1 ? strtoull(argv[1], 0, 10) : 4096;
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t nfl = B * A * D;
  float *obs; CK(hipMalloc(&obs, nfl * 4));
  const double mb = nfl * 4 / 1e6;
  auto rep = [&](const char *name, float us) { printf("%-58s %8.2f us  %6.2f TB/s\n", name, us, mb / us / 1e6 * 1e6 / 1e6 * 1e0 * 1.0); };
  const size_t n4 = nfl / 4;
  const size_t am_row = B * D, am_world = D;        // agent-major: rows B*D apart, worlds D apart
  const size_t wm_row = D, wm_world = (size_t)A * D;  // world-major
  const unsigned gw1 = (unsigned)((B * 64 + 255) / 256), gw4 = (unsigned)((B * 4 * 64 + 255) / 256);
  printf("B=%zu  block = %.1f MB\n", B, mb);
  rep("V0 linear, one float4 per thread", run([&] { hipLaunchKernelGGL(k_linear, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (float4 *)obs, n4); }, s));
  float2 *Qs; CK(hipMalloc(&Qs, B * 128 * sizeof(float2))); CK(hipMemsetAsync(Qs, 0, B * 128 * sizeof(float2), s));
  rep("V9 linear order, rows computed from a world-major scratch", run([&] { hipLaunchKernelGGL(k_linear_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (float4 *)obs, Qs, (unsigned)B); }, s));
  rep("V6 grid-stride fill, 2048 x 256 threads", run([&] { hipLaunchKernelGGL(k_gridstride, dim3(2048), dim3(256), 0, s, (float4 *)obs, n4); }, s));
  rep("V6b grid-stride fill, 1024 x 256 threads", run([&] { hipLaunchKernelGGL(k_gridstride, dim3(1024), dim3(256), 0, s, (float4 *)obs, n4); }, s));
  rep("V2 wave/world, agent-major, 96 flat full stores", run([&] { hipLaunchKernelGGL((k_world<0, false, 1>), dim3(gw1), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  rep("V1 wave/world, agent-major, per row 64+32 lanes", run([&] { hipLaunchKernelGGL((k_world<1, false, 1>), dim3(gw1), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  // the same wave-per-world row pattern with the agent blocks PADDED apart: rows of one world are then no longer an
  // exact multiple of 2 MiB apart (B*D*4 = 6 MiB at B=4096) -- does the per-process bimodality (60 vs 73 us) follow
  // the power-of-two stride between the concurrently written streams?
  {
    float *big; CK(hipMalloc(&big, (nfl + (size_t)A * (1 << 20)) * 4));
    for (size_t pad : {(size_t)64, (size_t)1024, (size_t)(8192 + 64), (size_t)(1 << 18) + 64}) {
      char nm[96]; snprintf(nm, sizeof nm, "V1p wave/world, agent-major, block stride + %zu floats", pad);
      rep(nm, run([&] { hipLaunchKernelGGL((k_world<1, false, 1>), dim3(gw1), dim3(256), 0, s, big, B, am_row + pad, am_world); }, s));
    }
    rep("V1  (same buffer, unpadded)", run([&] { hipLaunchKernelGGL((k_world<1, false, 1>), dim3(gw1), dim3(256), 0, s, big, B, am_row, am_world); }, s));
    rep("V0  linear fill of the same buffer", run([&] { hipLaunchKernelGGL(k_linear, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (float4 *)big, n4); }, s));
    CK(hipFree(big));
  }
  rep("V8 wave/world, agent-major, rows rotated per 64 worlds", run([&] { hipLaunchKernelGGL((k_world_rot<64>), dim3(gw1), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  rep("V8b wave/world, agent-major, rows rotated per 4 worlds", run([&] { hipLaunchKernelGGL((k_world_rot<4>), dim3(gw1), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  rep("V8c wave/world, agent-major, rows rotated per 512 worlds", run([&] { hipLaunchKernelGGL((k_world_rot<512>), dim3(gw1), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  rep("V3 wave/world, world-major, 96 flat full stores", run([&] { hipLaunchKernelGGL((k_world<0, false, 1>), dim3(gw1), dim3(256), 0, s, obs, B, wm_row, wm_world); }, s));
  rep("V4 wave/world, agent-major, flat, nontemporal", run([&] { hipLaunchKernelGGL((k_world<0, true, 1>), dim3(gw1), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  rep("V4b wave/world, world-major, flat, nontemporal", run([&] { hipLaunchKernelGGL((k_world<0, true, 1>), dim3(gw1), dim3(256), 0, s, obs, B, wm_row, wm_world); }, s));
  rep("V5 4 waves/world (16 rows each), agent-major, flat", run([&] { hipLaunchKernelGGL((k_world<0, false, 4>), dim3(gw4), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  rep("V5b 4 waves/world, agent-major, flat, nontemporal", run([&] { hipLaunchKernelGGL((k_world<0, true, 4>), dim3(gw4), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  rep("V7 workgroup/world (256 thr), agent-major, flat", run([&] { hipLaunchKernelGGL(k_wgworld, dim3((unsigned)B), dim3(256), 0, s, obs, B, am_row, am_world); }, s));
  rep("V7b workgroup/world (256 thr), world-major, flat", run([&] { hipLaunchKernelGGL(k_wgworld, dim3((unsigned)B), dim3(256), 0, s, obs, B, wm_row, wm_world); }, s));
  return 0;
}
