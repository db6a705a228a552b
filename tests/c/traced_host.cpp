// Host harness of tests/test_symtrace.py::test_generated_code_on_the_host_against_the_numpy_evaluation: the device functions
// symtrace.hip_source generates (traced_obs / traced_shared / traced_rew / traced_done) compiled for the HOST with plain-C++
// stand-ins for the few device intrinsics they use, run world by world on states read from a file.  Test infrastructure: checks the
// code generator without a GPU (the kernels themselves are tested on the device, tests/test_gpu_traced.py).
//   g++ -O1 -std=c++17 -ffp-contract=off -include generated.h traced_host.cpp -o traced_host
//   ./traced_host in.bin out.bin     in: int32 B, E, A, DC, NK, n_shared, widths[A]; float P[B][E][2], V[B][E][2], W[B][A][DC]; int32 K[B][NK]
//                                    out: per world: rows of every agent, rewards of every agent, dones of every agent (as floats)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

static inline float fast_sqrt(float x) { return sqrtf(x); }
static inline bool sqrt_lt(float s2, float m) { return sqrtf(s2) < m; }
static inline float host_exp2f(float x) { return exp2f(x); }
static inline float host_logf2(float x) { return log2f(x); }
#define __builtin_amdgcn_exp2f host_exp2f
#define __builtin_amdgcn_logf host_logf2
#define __device__
#define __forceinline__ inline

#ifndef MPE_HOST_TRACED_SOURCE
#error "compile with -DMPE_HOST_TRACED_SOURCE=\"<file with the generated functions>\""
#endif
#include MPE_HOST_TRACED_SOURCE

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[6];
  if (fread(hdr, 4, 6, f) != 6) return 4;
  const int B = hdr[0], E = hdr[1], A = hdr[2], DC = hdr[3], NK = hdr[4], NS = hdr[5];
  std::vector<int32_t> widths(A);
  if ((int)fread(widths.data(), 4, A, f) != A) return 4;
  std::vector<float> P((size_t)B * E * 2), V((size_t)B * E * 2), W((size_t)B * A * (DC ? DC : 1));
  std::vector<int32_t> K((size_t)B * (NK ? NK : 1));
  if (fread(P.data(), 4, P.size(), f) != P.size() || fread(V.data(), 4, V.size(), f) != V.size()) return 4;
  if (DC && fread(W.data(), 4, (size_t)B * A * DC, f) != (size_t)B * A * DC) return 4;
  if (NK && fread(K.data(), 4, (size_t)B * NK, f) != (size_t)B * NK) return 4;
  fclose(f);
  FILE *o = fopen(argv[2], "wb");
  if (!o) return 5;
  int dmax = 1;
  for (int i = 0; i < A; ++i) dmax = widths[i] > dmax ? widths[i] : dmax;
  std::vector<float> row(dmax), shared(NS ? NS : 1);
  for (int b = 0; b < B; ++b) {
    auto p = [&](int e, int c) { return P[((size_t)b * E + e) * 2 + c]; };
    auto v = [&](int e, int c) { return V[((size_t)b * E + e) * 2 + c]; };
    auto w = [&](int j, int c) { return W[((size_t)b * A + j) * DC + c]; };
    auto k = [&](int q) { return (int)K[(size_t)b * NK + q]; };
    auto s = [&](int q) { return shared[q]; };
    for (int q = 0; q < NS; ++q) mpe::traced_shared(q, &shared[q], p, v, w, k);
    for (int i = 0; i < A; ++i) {
      mpe::traced_obs(i, row.data(), p, v, w, k);
      fwrite(row.data(), 4, widths[i], o);
    }
    for (int i = 0; i < A; ++i) { const float r = mpe::traced_rew(i, p, v, w, k, s); fwrite(&r, 4, 1, o); }
    for (int i = 0; i < A; ++i) { const float d = mpe::traced_done(i, p, v, w, k) ? 1.f : 0.f; fwrite(&d, 4, 1, o); }
  }
  fclose(o);
  return 0;
}
