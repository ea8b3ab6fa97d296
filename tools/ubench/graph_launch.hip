// Host and device cost of a chain of tiny dependent kernels: launched one by one vs as one captured graph.
//   hipcc --offload-arch=gfx950 -O2 -o graph_launch graph_launch.hip && ./graph_launch [kernels per chain]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void bump(float* x) { x[threadIdx.x] += 1.0f; }

#define CHECK(e)                                                              \
  do {                                                                        \
    hipError_t err__ = (e);                                                   \
    if (err__ != hipSuccess) {                                                \
      printf("%s: %s\n", #e, hipGetErrorString(err__));                       \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main(int argc, char** argv) {
  const int chain = argc > 1 ? atoi(argv[1]) : 16, reps = 2000;
  float* x;
  CHECK(hipMalloc(&x, 256 * sizeof(float)));
  CHECK(hipMemset(x, 0, 256 * sizeof(float)));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int mode = 0; mode < 2; ++mode) {
    hipGraphExec_t exec = nullptr;
    if (mode == 1) {
      hipGraph_t graph;
      CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < chain; ++k) bump<<<1, 256, 0, s>>>(x);
      CHECK(hipStreamEndCapture(s, &graph));
      CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    }
    for (int warm = 0; warm < 2; ++warm) {
      CHECK(hipStreamSynchronize(s));
      const auto t0 = now();
      for (int r = 0; r < reps; ++r) {
        if (mode == 1) {
          CHECK(hipGraphLaunch(exec, s));
        } else {
          for (int k = 0; k < chain; ++k) bump<<<1, 256, 0, s>>>(x);
        }
      }
      const auto t1 = now();
      CHECK(hipStreamSynchronize(s));
      const auto t2 = now();
      if (warm)
        printf("%s: chain of %d kernels: host issue %.2f us per chain (%.2f per kernel), until done %.2f us per chain (%.2f per kernel)\n",
               mode ? "graph " : "eager ", chain, us(t0, t1) / reps, us(t0, t1) / reps / chain, us(t0, t2) / reps,
               us(t0, t2) / reps / chain);
    }
  }
  return 0;
}
