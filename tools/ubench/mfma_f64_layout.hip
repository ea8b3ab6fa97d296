// Which element of D = A * B each accumulator register of v_mfma_f64_16x16x4_f64 holds.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f64_layout mfma_f64_layout.hip && ./mfma_f64_layout
#include <hip/hip_runtime.h>

#include <cstdio>

using f64x4 = __attribute__((ext_vector_type(4))) double;

__global__ void probe(double* out) {
  const int lane = threadIdx.x;
  // assumed operand layouts: A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16]
  const int i = lane % 16, k = lane / 16, j = lane % 16;
  const double a = k == 0 ? 100.0 * (i + 1) : 0.0;  // D[i][j] = 100 (i + 1) * (j + 1) ... encodes both
  const double b = k == 0 ? (double)(j + 1) : 0.0;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

int main() {
  double* d;
  hipMalloc(&d, 256 * sizeof(double));
  probe<<<1, 64>>>(d);
  double h[256];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int lane = 0; lane < 64; lane += 5)
    for (int r = 0; r < 4; ++r) {
      const int v = (int)(h[lane * 4 + r] + 0.5);
      // v = 100 (i + 1) (j + 1): with j + 1 <= 16 and i + 1 <= 16 the pair is recovered by trying j = lane % 16
      const int jj = lane % 16 + 1;
      printf("lane %2d reg %d: value %6d -> i = %d (if j = lane %% 16 = %d)\n", lane, r, v, v / (100 * jj) - 1, jj - 1);
    }
  return 0;
}
