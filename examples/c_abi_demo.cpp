// The C ABI of libbn_hip.so used from plain C++ with HIP device buffers: no Python, no torch.
// Network and numbers are the reference's first docs example (docs/1. Simple Bayesian
// network.md:56-57 for the CPTs, :70-75 for the posteriors): Q(2) -> Y(3), three observations,
// one per state of Y.  Prints P(Q | Y = y) for each and the summed log-likelihood; exits non-zero
// if a posterior is off by more than 1e-12 from the documented values.
//
//   make demo && ./examples/c_abi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bn_hip.h"

#define HIP_OK(expr)                                                          \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess) {                                                  \
      std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e__));        \
      return 2;                                                               \
    }                                                                         \
  } while (0)
#define BN_OK_OR_DIE(expr)                                                    \
  do {                                                                        \
    if ((expr) != BN_OK) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #expr, bn_last_error());               \
      return 2;                                                               \
    }                                                                         \
  } while (0)

int main() {
  // topology in network order: node 0 = Q, node 1 = Y with parent Q; Y observed
  const int32_t num_states[2] = {2, 3};
  const int32_t parent_offsets[3] = {0, 0, 1}, parent_indices[1] = {0};
  const int32_t child_offsets[3] = {0, 1, 1}, child_indices[1] = {1};
  const int32_t observed[1] = {1};
  const int64_t N = 3;

  bn_plan* plan = nullptr;
  BN_OK_OR_DIE(bn_plan_create(BN_F64, N, 2, num_states, parent_offsets, parent_indices, child_offsets, child_indices,
                              observed, 1, &plan));
  int32_t is_tree = 0, diameter = 0;
  BN_OK_OR_DIE(bn_plan_topology(plan, &is_tree, &diameter));

  // flat CPT buffer: Q's prior, then Y's [2, 3] table
  const std::vector<double> cpts = {0.4, 0.6, 0.1, 0.6, 0.3, 0.5, 0.1, 0.4};
  if ((int64_t)cpts.size() != bn_plan_cpt_elements(plan)) return 2;
  const std::vector<double> evidence = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // [N, 3] one-hot rows

  double *d_cpts = nullptr, *d_ev = nullptr, *d_post = nullptr, *d_ll = nullptr;
  HIP_OK(hipMalloc(&d_cpts, cpts.size() * sizeof(double)));
  HIP_OK(hipMalloc(&d_ev, evidence.size() * sizeof(double)));
  HIP_OK(hipMalloc(&d_post, N * 2 * sizeof(double)));
  HIP_OK(hipMalloc(&d_ll, sizeof(double)));
  HIP_OK(hipMemcpy(d_cpts, cpts.data(), cpts.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_ev, evidence.data(), evidence.size() * sizeof(double), hipMemcpyHostToDevice));

  BN_OK_OR_DIE(bn_set_cpts(plan, d_cpts, nullptr));
  const void* sources[1] = {d_ev};
  BN_OK_OR_DIE(bn_enter_evidence(plan, sources, nullptr));
  BN_OK_OR_DIE(bn_iterate(plan, diameter, nullptr));  // the reference's default: as many iterations as the diameter
  BN_OK_OR_DIE(bn_marginal_single(plan, 0, d_post, nullptr));
  BN_OK_OR_DIE(bn_log_likelihood(plan, 0, d_ll, nullptr));

  double post[6], ll = 0;
  HIP_OK(hipMemcpy(post, d_post, sizeof(post), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&ll, d_ll, sizeof(ll), hipMemcpyDeviceToHost));

  const double want[6] = {2.0 / 17, 15.0 / 17, 0.8, 0.2, 1.0 / 3, 2.0 / 3};
  double worst = 0;
  std::printf("tree=%d iterations=%d\n", is_tree, diameter);
  for (int n = 0; n < 3; ++n) {
    std::printf("P(Q | Y=%d) = [%.12f, %.12f]\n", n, post[2 * n], post[2 * n + 1]);
    for (int s = 0; s < 2; ++s) worst = std::fmax(worst, std::fabs(post[2 * n + s] - want[2 * n + s]));
  }
  std::printf("log-likelihood (sum) = %.12f\n", ll);

  bn_plan_destroy(plan);
  (void)hipFree(d_cpts);
  (void)hipFree(d_ev);
  (void)hipFree(d_post);
  (void)hipFree(d_ll);
  if (!(worst <= 1e-12)) {
    std::fprintf(stderr, "posterior off by %.3e\n", worst);
    return 1;
  }
  return 0;
}
