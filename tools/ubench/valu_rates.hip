// Issue cost of the VALU operations the leaf-family kernel leans on, relative to v_fma_f32:
// each test runs 8 independent chains of one operation per lane, on every SIMD, long enough
// that only the issue rate matters.   hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
  float a[8];
  double d[8];
  int e[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 1e-3f + i;
    d[i] = a[i];
    e[i] = i;
  }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = fmaf(a[i], 0.999f, 1e-3f);
      if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == 2) d[i] = fma(d[i], 0.999, 1e-3);
      if (OP == 3) { d[i] = (double)a[i]; a[i] = a[i] + (float)(it & 1); asm volatile("" : "+v"(d[i])); }
      if (OP == 4) { a[i] = (float)d[i]; asm volatile("" : "+v"(a[i])); d[i] += 1.0; }
      if (OP == 5) { int x; d[i] = frexp(d[i] + 3.0, &x); e[i] += x; }
      if (OP == 6) a[i] = ldexpf(a[i], e[i] & 1);
      if (OP == 7) d[i] = d[i] * 0.999;
      if (OP == 8) a[i] = __builtin_amdgcn_logf(a[i] + 2.f);
      if (OP == 9) d[i] = __builtin_amdgcn_rcp(d[i]);
    }
  }
  float acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i] + (float)d[i] + e[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
float run(float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * 8;  // 8 waves per SIMD
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
  const char* names[] = {"v_fma_f32", "v_rcp_f32", "v_fma_f64", "cvt f32->f64 (+add)", "cvt f64->f32 (+add f64)",
                         "frexp f64 (mant+exp+2 add)", "v_ldexp_f32 (+and)", "v_mul_f64", "v_log_f32 (+add)", "v_rcp_f64"};
  float ms[10];
  ms[0] = run<0>(out); ms[1] = run<1>(out); ms[2] = run<2>(out); ms[3] = run<3>(out); ms[4] = run<4>(out);
  ms[5] = run<5>(out); ms[6] = run<6>(out); ms[7] = run<7>(out); ms[8] = run<8>(out); ms[9] = run<9>(out);
  for (int i = 0; i < 10; ++i) printf("%-28s %8.3f ms  x%.2f of fma_f32\n", names[i], ms[i], ms[i] / ms[0]);
  return 0;
}
