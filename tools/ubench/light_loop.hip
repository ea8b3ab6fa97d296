// Timing harness for the loop launch that carries no per-state product (leaf_loop_p1 with
// STATE = kLoopChunk) on the 784-leaf model's geometry: 784 families (10,2), 60 000 observations.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../bayesiannetwork_amd/csrc -I../../include light_loop.hip -o light_loop
//   ./light_loop [N] [chunks] [split 0|1]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kernels_leaf_loop.h"

#define CK(x)                                              \
  do {                                                     \
    hipError_t e__ = (x);                                  \
    if (e__ != hipSuccess) {                               \
      std::printf("%s: %s\n", #x, hipGetErrorString(e__)); \
      std::exit(1);                                        \
    }                                                      \
  } while (0)

template <typename U>
U* dev(const std::vector<U>& h) {
  U* d = nullptr;
  CK(hipMalloc(&d, h.size() * sizeof(U)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice));
  return d;
}

template <bool SPLIT, int OUT>
float run(float* arena, const float* cpts, bn::LeafFamilyArgs<float> a, int reps) {
  using namespace bn;
  auto fn = leaf_loop_p1<float, 10, 2, OUT, kLoopChunk, SPLIT>;
  const size_t lds = (size_t)kWavesPerBlock * LeafRing<float, 2, false, true>::kWaveBytes;
  const int64_t items = (int64_t)a.chunks * a.ntiles * 2;
  const uint32_t blocks = SPLIT ? (uint32_t)items : (uint32_t)((items + kWavesPerBlock - 1) / kWavesPerBlock);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn<<<blocks, 256, lds>>>(arena, cpts, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) fn<<<blocks, 256, lds>>>(arena, cpts, a);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1000.f;
}

int main(int argc, char** argv) {
  const int K = 784, D0 = 10, S = 2, TILE = 256;
  const int64_t N = argc > 1 ? atoll(argv[1]) : 60000;
  const int chunks = argc > 2 ? atoi(argv[2]) : 8;
  const int64_t ntiles = (N + TILE - 1) / TILE, npad = ntiles * TILE;
  const int chunk_len = (K + chunks - 1) / chunks;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> uni(0.f, 1.f);
  std::vector<float> arena((size_t)K * npad * S);
  for (size_t i = 0; i < arena.size(); ++i) arena[i] = 0.05f + 0.9f * uni(rng);
  std::vector<float> cpts((size_t)K * D0 * S);
  for (size_t r = 0; r < (size_t)K * D0; ++r) {
    const float x = 0.05f + 0.9f * uni(rng);
    cpts[r * 2] = x;
    cpts[r * 2 + 1] = 1.f - x;
  }
  std::vector<double> scaled((size_t)D0 * npad);
  for (auto& x : scaled) x = 0.01 + uni(rng);
  std::vector<int64_t> cpt_off(K), obs_off(K);
  for (int k = 0; k < K; ++k) {
    cpt_off[k] = (int64_t)k * D0 * S;
    obs_off[k] = (int64_t)k * npad * S;
  }
  std::vector<double> zm((size_t)chunks * npad, 0.75);
  std::vector<int32_t> ze((size_t)chunks * npad, 0);
  bn::LeafFamilyArgs<float> a{};
  a.cpt_off = dev(cpt_off);
  a.obs_off = dev(obs_off);
  a.K = K;
  a.chunks = chunks;
  a.chunk_len = chunk_len;
  a.chunk_base = 0;
  a.scaled = dev(scaled);
  a.npad = npad;
  a.ntiles = ntiles;
  a.N = N;
  a.zdiv_mant = dev(zm);
  a.zdiv_exp = dev(ze);
  a.zdiv_next_mant = dev(zm);
  a.zdiv_next_exp = dev(ze);
  std::vector<float> z((size_t)K * npad, 0.f);
  a.z_state = dev(z);
  a.split = 1;
  float* d_arena = dev(arena);
  float* d_cpts = dev(cpts);
  std::printf("N %lld chunks %d  light: one wave per chunk %.1f us, block per chunk %.1f us;  last: %.1f / %.1f us\n",
              (long long)N, chunks, run<false, bn::kLoopLight>(d_arena, d_cpts, a, 20),
              run<true, bn::kLoopLight>(d_arena, d_cpts, a, 20), run<false, bn::kLoopLast>(d_arena, d_cpts, a, 20),
              run<true, bn::kLoopLast>(d_arena, d_cpts, a, 20));
  return 0;
}
