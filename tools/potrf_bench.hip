// Time of the 128x128 diagonal-block factorisation (chol.hpp chol_potrf_block) and of its phases (skip masks):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igraphite_amd/csrc tools/potrf_bench.hip -o build/potrf_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "common.hpp"
#include "chol.hpp"
using namespace gr;
template <typename T, int NT> void run(const char *name) {
  const int n = CH_NB;
  std::vector<T> h((size_t)n * n);
  std::mt19937 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> g((size_t)n * n);
  for (auto &v : g) v = nd(rng);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += g[i * n + k] * g[j * n + k]; h[(size_t)i * n + j] = (T)(s / n + (i == j ? 1.0 : 0.0)); }
  T *A, *A0, *Linv; int *fail;
  hipMalloc(&A, sizeof(T) * n * n); hipMalloc(&A0, sizeof(T) * n * n); hipMalloc(&Linv, sizeof(T) * n * n); hipMalloc(&fail, 4);
  hipMemcpy(A0, h.data(), sizeof(T) * n * n, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chol_potrf<T, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_potrf_lds(sizeof(T)));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int skip : {0, 1, 14, 13, 11, 15}) {
    float tot = 0;
    const int reps = 50;
    for (int r = 0; r < reps + 3; ++r) {
      hipMemcpyAsync(A, A0, sizeof(T) * n * n, hipMemcpyDeviceToDevice, 0);
      hipEventRecord(a, 0);
      k_chol_potrf<T, NT><<<1, NT, chol_potrf_lds(sizeof(T)), 0>>>(A, n, 0, Linv, fail, skip);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (r >= 3) tot += ms;
    }
    std::printf("%s NT=%d skip=%2d (1 diag16, 2 solve+update, 4 inverse, 8 store): %.1f us\n", name, NT, skip, tot * 1e3 / reps);
  }
}
int main() { run<double, 256>("f64"); run<double, 512>("f64"); run<double, 1024>("f64"); run<float, 256>("f32"); run<float, 512>("f32"); return 0; }
