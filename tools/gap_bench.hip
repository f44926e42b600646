#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
int main() {
  float *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
  hipStream_t s; hipStreamCreate(&s);
  for (int N : {4, 7, 16}) {
    const int REP = 200;
    // stream: REP groups of N kernels, sync only at the end
    for (int w = 0; w < 2; ++w) {
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < REP; ++r) for (int i = 0; i < N; ++i) k_tiny<<<64, 256, 0, s>>>(d, 1 << 14);
      hipStreamSynchronize(s);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (w) printf("N=%2d stream: %.2f us per group (%.2f per kernel)\n", N, us / REP, us / REP / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) k_tiny<<<64, 256, 0, s>>>(d, 1 << 14);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 2; ++w) {
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < REP; ++r) hipGraphLaunch(ge, s);
      hipStreamSynchronize(s);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (w) printf("N=%2d graph : %.2f us per group (%.2f per kernel)\n", N, us / REP, us / REP / N);
    }
    // latency of ONE group from a drained stream (host round trip in between, like the LM loop)
    for (int mode = 0; mode < 2; ++mode) {
      double tot = 0;
      for (int r = 0; r < 50; ++r) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        if (mode == 0) for (int i = 0; i < N; ++i) k_tiny<<<64, 256, 0, s>>>(d, 1 << 14); else hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      printf("N=%2d %s from idle: %.2f us per group\n", N, mode ? "graph " : "stream", tot / 50);
    }
  }
  return 0;
}
