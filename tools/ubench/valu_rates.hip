// Issue cost of the VALU operations the leaf-family kernels lean on, relative to v_fma_f32:
// each test runs 8 independent chains of one operation per lane, on every SIMD, with 1, 4 or 8 waves
// per SIMD, long enough that only the issue rate matters.
//   hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
  float a[8];
  f2 p[8];
  double d[8];
  int e[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 1e-3f + i;
    p[i] = f2{a[i], a[i] + 0.5f};
    d[i] = a[i];
    e[i] = i;
  }
  const f2 c0 = {0.999f, 0.998f}, c1 = {1e-3f, 2e-3f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = fmaf(a[i], 0.999f, 1e-3f);
      if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == 2) d[i] = fma(d[i], 0.999, 1e-3);
      if (OP == 3) { asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d[i]) : "v"(a[i])); }
      if (OP == 4) { asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(a[i]) : "v"(d[i])); }
      if (OP == 5) { asm volatile("v_frexp_mant_f64_e32 %0, %1" : "=v"(d[i]) : "v"(d[i])); }
      if (OP == 6) { asm volatile("v_frexp_exp_i32_f64_e32 %0, %1" : "=v"(e[i]) : "v"(d[i])); }
      if (OP == 7) d[i] = d[i] * 0.999;
      if (OP == 8) a[i] = __builtin_amdgcn_logf(a[i] + 2.f);
      if (OP == 9) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c0), "v"(c1)); }
      if (OP == 10) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c0)); }
      if (OP == 11) { asm volatile("v_mov_b64_e32 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7])); }
      if (OP == 12) { asm volatile("v_frexp_mant_f32_e32 %0, %0" : "+v"(a[i])); }
      if (OP == 13) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(e[i]) : "v"(e[(i + 1) & 7])); }
      if (OP == 14) { asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); }
      if (OP == 15) { asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i])); }
      if (OP == 16) { asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a[i])); }
      if (OP == 17) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c0)); }
      if (OP == 18) { asm volatile("v_rcp_f32_e32 %0, %0\n\tv_mul_f64 %1, %1, %2" : "+v"(a[i]), "+v"(d[i]) : "v"(d[(i + 1) & 7])); }
      if (OP == 19) { asm volatile("v_rcp_f32_e32 %0, %0\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(a[i]), "+v"(p[i]) : "v"(c0), "v"(c1)); }
      if (OP == 20) { asm volatile("v_rcp_f32_e32 %0, %0\n\tv_cvt_f64_f32_e32 %1, %3\n\tv_mul_f64 %2, %2, %1" : "+v"(a[i]), "=&v"(d[(i + 1) & 7]), "+v"(d[i]) : "v"(a[(i + 2) & 7])); }
    }
  }
  float acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i] + (float)d[i] + e[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
float run(float* out, int waves_per_simd) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd;
  k<OP><<<blocks, 256>>>(out, 1.5f);
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(out, 1.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  const char* names[] = {"v_fma_f32", "v_rcp_f32", "v_fma_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_frexp_mant_f64",
                         "v_frexp_exp_i32_f64", "v_mul_f64", "v_log_f32 (+add)", "v_pk_fma_f32", "v_pk_mul_f32",
                         "v_mov_b64", "v_frexp_mant_f32", "v_mul_hi_u32", "v_mul_f32", "v_ldexp_f64", "v_exp_f32",
                         "v_pk_add_f32", "rcp_f32 + mul_f64", "rcp_f32 + pk_fma_f32", "rcp + cvt + mul_f64"};
  for (int w : {1, 4, 8}) {
    float ms[21];
    ms[0] = run<0>(out, w); ms[1] = run<1>(out, w); ms[2] = run<2>(out, w); ms[3] = run<3>(out, w);
    ms[4] = run<4>(out, w); ms[5] = run<5>(out, w); ms[6] = run<6>(out, w); ms[7] = run<7>(out, w);
    ms[8] = run<8>(out, w); ms[9] = run<9>(out, w); ms[10] = run<10>(out, w); ms[11] = run<11>(out, w);
    ms[12] = run<12>(out, w); ms[13] = run<13>(out, w); ms[14] = run<14>(out, w); ms[15] = run<15>(out, w);
    ms[16] = run<16>(out, w); ms[17] = run<17>(out, w); ms[18] = run<18>(out, w); ms[19] = run<19>(out, w); ms[20] = run<20>(out, w);
    // cycles per wave-instruction on one SIMD at ~2.4 GHz: ms * 2.4e6 / (ITERS * 8 instr * waves per SIMD)
    for (int i = 0; i < 21; ++i)
      printf("waves/SIMD %d  %-22s %8.3f ms  x%.2f of fma_f32  ~%.2f cyc per wave-instruction\n", w, names[i], ms[i],
             ms[i] / ms[0], ms[i] * 2.4e6 / (ITERS * 8.0 * w));
  }
  return 0;
}
