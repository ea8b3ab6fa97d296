// Stand-alone timing and equivalence harness for the steady-state loop launch of the leaf-family
// kernel on the 784-leaf model's geometry (784 families (10,2), 60 000 observations): runs the
// library's kernel and an experimental variant on the same synthetic state and compares what they
// write (the leaf normalisers and the chunk partial products).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../bayesiannetwork_amd/csrc -I../../include leaf_loop.hip -o leaf_loop
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kernels_leaf.h"
#include "leaf_loop_variant.h"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e__ = (x);                                                              \
    if (e__ != hipSuccess) {                                                           \
      std::printf("%s: %s\n", #x, hipGetErrorString(e__));                             \
      std::exit(1);                                                                    \
    }                                                                                  \
  } while (0)

template <typename U>
U* dev(const std::vector<U>& h) {
  U* d = nullptr;
  CK(hipMalloc(&d, h.size() * sizeof(U)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int K = 784, D0 = 10, S = 2, TILE = 256;
  const int64_t N = argc > 1 ? atoll(argv[1]) : 60000;
  const int64_t ntiles = (N + TILE - 1) / TILE, npad = ntiles * TILE;
  const int chunks = argc > 2 ? atoi(argv[2]) : 8;
  const int chunk_len = (K + chunks - 1) / chunks;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> uni(0.f, 1.f);

  // arena: observation arrays [K][ntiles][S][TILE]
  std::vector<float> arena((size_t)K * npad * S);
  for (size_t k = 0; k < (size_t)K; ++k)
    for (int64_t t = 0; t < ntiles; ++t)
      for (int c = 0; c < TILE; ++c) {
        const float x = uni(rng) < 0.2f ? 0.9999995f : 5e-7f;
        arena[(k * ntiles + t) * S * TILE + c] = 1.f - x;
        arena[(k * ntiles + t) * S * TILE + TILE + c] = x;
      }
  std::vector<float> cpts((size_t)K * D0 * S);
  for (size_t r = 0; r < (size_t)K * D0; ++r) {
    const float a = 0.05f + uni(rng);
    const float b = 0.05f + uni(rng);
    cpts[r * 2] = a / (a + b);
    cpts[r * 2 + 1] = b / (a + b);
  }
  std::vector<double> scaled((size_t)D0 * npad);
  for (int64_t n = 0; n < npad; ++n) {
    double mx = 0;
    for (int i = 0; i < D0; ++i) {
      scaled[i * npad + n] = std::pow((double)uni(rng), 6.0) + 1e-9;
      mx = std::max(mx, scaled[i * npad + n]);
    }
    for (int i = 0; i < D0; ++i) scaled[i * npad + n] /= mx;
  }
  std::vector<float> z((size_t)K * npad);
  for (auto& x : z) x = 0.2f + 0.8f * uni(rng);
  std::vector<int64_t> f_in((size_t)K * 2, 0), f_out((size_t)K * 2, 0), cpt_off(K), obs_off(K), c_off(K, 0);
  for (int k = 0; k < K; ++k) {
    cpt_off[k] = (int64_t)k * D0 * S;
    obs_off[k] = (int64_t)k * npad * S;
  }

  float* d_arena = dev(arena);
  float* d_cpts = dev(cpts);
  double* d_scaled = dev(scaled);
  float* d_z0 = dev(z);
  float *d_z1 = nullptr, *d_z2 = nullptr;
  CK(hipMalloc(&d_z1, z.size() * 4));
  CK(hipMalloc(&d_z2, z.size() * 4));
  const size_t rows = (size_t)chunks * D0 * npad;
  double *d_pm1, *d_pm2;
  int32_t *d_pe1, *d_pe2;
  CK(hipMalloc(&d_pm1, rows * 8));
  CK(hipMalloc(&d_pm2, rows * 8));
  CK(hipMalloc(&d_pe1, rows * 4));
  CK(hipMalloc(&d_pe2, rows * 4));

  bn::LeafFamilyArgs<float> a{};
  a.f_in = dev(f_in);
  a.f_out = dev(f_out);
  a.cpt_off = dev(cpt_off);
  a.obs_off = dev(obs_off);
  a.c_off = dev(c_off);
  a.K = K;
  a.chunks = chunks;
  a.chunk_len = chunk_len;
  a.chunk_base = 0;
  a.scaled = d_scaled;
  a.npad = npad;
  a.ntiles = ntiles;
  a.N = N;
  a.slots = (int32_t)(ntiles * 2);

  {
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bnx::leaf_loop_v3<10, 2, 2, false>, 256, 4 * bnx::kRing * bnx::kSlotBytes));
    int nb0 = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb0, bn::leaf_family_p1<float, 10, 2, 2, 0>, 256, 0));
    std::printf("occupancy API: library kernel %d blocks/CU, v3 %d blocks/CU; chunks %d\n", nb0, nb, chunks);
  }
  const int64_t waves = (int64_t)chunks * ntiles * 2;
  const dim3 grid((uint32_t)((waves + 3) / 4)), block(256);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_it = [&](auto&& launch, float* zbuf, double* pm, int32_t* pe, const char* name) {
    float best = 1e9f, total = 0;
    const int reps = 12;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemcpy(zbuf, d_z0, z.size() * 4, hipMemcpyDeviceToDevice));
      a.z_state = zbuf;
      a.part_mant = pm;
      a.part_exp = pe;
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) {
        best = std::min(best, ms);
        total += ms;
      }
    }
    std::printf("%-28s best %.1f us  mean %.1f us\n", name, best * 1e3, total / (reps - 2) * 1e3);
  };
  time_it([&] { bn::leaf_family_p1<float, 10, 2, 2, 0><<<grid, block>>>(d_arena, d_cpts, a); }, d_z1, d_pm1, d_pe1,
          "library steady loop");
  time_it([&] { bnx::launch_variant(grid, block, d_arena, d_cpts, a); }, d_z2, d_pm2, d_pe2, "variant");
  // chunk-level normaliser products of the input state, computed on the host
  const size_t zc_n = (size_t)chunks * npad;
  std::vector<double> zc_m(zc_n);
  std::vector<int32_t> zc_e(zc_n);
  for (int c = 0; c < chunks; ++c)
    for (int64_t nn = 0; nn < npad; ++nn) {
      double m = 1.0;
      int e = 0;
      for (int k = c * chunk_len; k < std::min(K, (c + 1) * chunk_len); ++k) {
        int de;
        m = std::frexp(m * (double)z[(size_t)k * npad + nn], &de);
        e += de;
      }
      zc_m[c * npad + nn] = m;
      zc_e[c * npad + nn] = e;
    }
  double* d_zc_m0 = dev(zc_m);
  int32_t* d_zc_e0 = dev(zc_e);
  double* d_zc_m;
  int32_t* d_zc_e;
  CK(hipMalloc(&d_zc_m, zc_n * 8));
  CK(hipMalloc(&d_zc_e, zc_n * 4));
  double *d_pm3, *d_pm4;
  int32_t *d_pe3, *d_pe4;
  float* d_z4;
  CK(hipMalloc(&d_pm3, rows * 8));
  CK(hipMalloc(&d_pm4, rows * 8));
  CK(hipMalloc(&d_pe3, rows * 4));
  CK(hipMalloc(&d_pe4, rows * 4));
  CK(hipMalloc(&d_z4, z.size() * 4));
  auto reset_zc = [&] {
    CK(hipMemcpy(d_zc_m, d_zc_m0, zc_n * 8, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(d_zc_e, d_zc_e0, zc_n * 4, hipMemcpyDeviceToDevice));
  };
  reset_zc();
  time_it([&] { bnx::launch_v3<true>(grid, block, d_arena, d_cpts, a, d_zc_m, d_zc_e); }, d_z4, d_pm4, d_pe4,
          "v3 writing every z");
  time_it([&] { reset_zc(); bnx::launch_v3<false>(grid, block, d_arena, d_cpts, a, d_zc_m, d_zc_e); }, d_z2, d_pm3, d_pe3,
          "v3 chunk products (+2 copies)");
  double* d_pm5; int32_t* d_pe5;
  CK(hipMalloc(&d_pm5, rows * 8));
  CK(hipMalloc(&d_pe5, rows * 4));
  time_it([&] { reset_zc(); bnx::launch_v3_pair(grid, block, d_arena, d_cpts, a, d_zc_m, d_zc_e); }, d_z2, d_pm5, d_pe5,
          "v3 chunk, pairs (+2 copies)");
  {
    std::vector<double> pm5(rows); std::vector<int32_t> pe5(rows), pe1h(rows); std::vector<double> pm1h(rows);
    CK(hipMemcpy(pm5.data(), d_pm5, rows * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pe5.data(), d_pe5, rows * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pm1h.data(), d_pm1, rows * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pe1h.data(), d_pe1, rows * 4, hipMemcpyDeviceToHost));
    double w5 = 0;
    for (size_t i = 0; i < rows; ++i) w5 = std::max(w5, std::fabs(std::ldexp(pm5[i], pe5[i] - pe1h[i]) - pm1h[i]) / std::fabs(pm1h[i]));
    std::printf("pairs: partial products worst relative %.3g\n", w5);
  }
  {
    reset_zc();
    CK(hipEventRecord(e0));
    bnx::launch_v3<false>(grid, block, d_arena, d_cpts, a, d_zc_m, d_zc_e);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("v3 chunk products, one launch alone: %.1f us\n", ms * 1e3);
  }

  // compare
  std::vector<float> z1(z.size()), z2(z.size());
  std::vector<double> pm1(rows), pm2(rows);
  std::vector<int32_t> pe1(rows), pe2(rows);
  CK(hipMemcpy(z1.data(), d_z1, z.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(z2.data(), d_z2, z.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(pm1.data(), d_pm1, rows * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(pm2.data(), d_pm2, rows * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(pe1.data(), d_pe1, rows * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(pe2.data(), d_pe2, rows * 4, hipMemcpyDeviceToHost));
  size_t zbad = 0, pbad = 0;
  double zworst = 0, pworst = 0;
  for (size_t i = 0; i < z.size(); ++i) {
    const double d = std::fabs((double)z1[i] - z2[i]) / std::fabs((double)z1[i]);
    zworst = std::max(zworst, d);
    zbad += z1[i] != z2[i];
  }
  for (size_t i = 0; i < rows; ++i) {
    const double x1 = std::ldexp(pm1[i], 0), x2 = std::ldexp(pm2[i], pe2[i] - pe1[i]);
    const double d = x1 == x2 ? 0.0 : std::fabs(x1 - x2) / std::fabs(x1);
    pworst = std::max(pworst, d);
    pbad += d > 1e-12;
  }
  {
    // v3: partial products against the library's; z of the writing mode against the library's; the
    // chunk products of the other mode against the product of the library's z
    std::vector<double> pm3(rows), pm4(rows), zo_m(zc_n);
    std::vector<int32_t> pe3(rows), pe4(rows), zo_e(zc_n);
    std::vector<float> z4(z.size());
    CK(hipMemcpy(pm3.data(), d_pm3, rows * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pm4.data(), d_pm4, rows * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pe3.data(), d_pe3, rows * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pe4.data(), d_pe4, rows * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(z4.data(), d_z4, z.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(zo_m.data(), d_zc_m, zc_n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(zo_e.data(), d_zc_e, zc_n * 4, hipMemcpyDeviceToHost));
    double w3 = 0, w4 = 0, wz = 0, wc = 0;
    for (size_t i = 0; i < rows; ++i) {
      const double r3 = std::ldexp(pm3[i], pe3[i] - pe1[i]), r4 = std::ldexp(pm4[i], pe4[i] - pe1[i]);
      w3 = std::max(w3, std::fabs(r3 - pm1[i]) / std::fabs(pm1[i]));
      w4 = std::max(w4, std::fabs(r4 - pm1[i]) / std::fabs(pm1[i]));
    }
    for (size_t i = 0; i < z.size(); ++i) wz = std::max(wz, std::fabs((double)z4[i] - z1[i]) / std::fabs((double)z1[i]));
    for (int c = 0; c < chunks; ++c)
      for (int64_t nn = 0; nn < N; ++nn) {
        double m = 1.0;
        int e = 0;
        for (int k = c * chunk_len; k < std::min(K, (c + 1) * chunk_len); ++k) {
          int de;
          m = std::frexp(m * (double)z1[(size_t)k * npad + nn], &de);
          e += de;
        }
        const double got = std::ldexp(zo_m[c * npad + nn], zo_e[c * npad + nn] - e);
        wc = std::max(wc, std::fabs(got - m) / m);
      }
    std::printf("v3: partial products worst relative %.3g (chunk mode) %.3g (z mode); z worst relative %.3g; chunk products worst relative %.3g\n",
                w3, w4, wz, wc);
  }
  std::printf("z: %zu of %zu differ in bits, worst relative %.3g;  partial products: %zu of %zu differ by more than 1e-12, worst %.3g\n",
              zbad, z.size(), zworst, pbad, rows, pworst);
  return 0;
}
