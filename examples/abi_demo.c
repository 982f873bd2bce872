/* abi_demo.c -- drive the EM core through include/mmsbm_hip.h from plain C99 (no Python, no C++).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/abi_demo.c -o abi_demo -Lmmsbm_amd -lmmsbm_hip \
 *       -Wl,-rpath,$PWD/mmsbm_amd -lm
 *   ./abi_demo [n_obs users items ratings K L iterations]
 *
 * Builds a small synthetic problem (LCG ids, parameters from the same LCG), runs `iterations`
 * EM iterations on GPU 0 and prints the likelihood, the row sums of theta (must be 1) and a
 * checksum that tests/test_gpu_parity.py compares with the Python path on the same inputs. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mmsbm_hip.h"

static uint64_t lcg_state = 88172645463325252ULL;
static uint32_t lcg(void) {
  lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (uint32_t)(lcg_state >> 33);
}
static double unit(void) { return (lcg() + 1.0) / 2147483649.0; }

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != MMSBM_OK) {                                                       \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mmsbm_hip_last_error());     \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 5000;
  const int U = argc > 2 ? atoi(argv[2]) : 300, I = argc > 3 ? atoi(argv[3]) : 120;
  const int R = argc > 4 ? atoi(argv[4]) : 5, K = argc > 5 ? atoi(argv[5]) : 6, L = argc > 6 ? atoi(argv[6]) : 9;
  const int iters = argc > 7 ? atoi(argv[7]) : 25;
  int32_t *u = malloc(sizeof(int32_t) * n), *it = malloc(sizeof(int32_t) * n), *r = malloc(sizeof(int32_t) * n);
  for (int64_t j = 0; j < n; ++j) { u[j] = lcg() % U; it[j] = lcg() % I; r[j] = lcg() % R; }
  double *theta = malloc(sizeof(double) * U * K), *eta = malloc(sizeof(double) * I * L);
  double *pr = malloc(sizeof(double) * K * L * R);
  for (int j = 0; j < U * K; ++j) theta[j] = unit();
  for (int j = 0; j < I * L; ++j) eta[j] = unit();
  for (int j = 0; j < K * L * R; ++j) pr[j] = unit();

  int ndev = 0;
  CHECK(mmsbm_hip_device_count(&ndev));
  mmsbm_hip_ctx *ctx = NULL;
  CHECK(mmsbm_hip_create(0, n, U, I, R, K, L, u, it, r, -1, &ctx));
  CHECK(mmsbm_hip_set_params(ctx, theta, eta, pr));
  CHECK(mmsbm_hip_em_iterate(ctx, iters));
  CHECK(mmsbm_hip_synchronize(ctx));
  double lik = 0.0;
  CHECK(mmsbm_hip_likelihood(ctx, &lik));
  CHECK(mmsbm_hip_get_params(ctx, theta, eta, pr));
  double worst = 0.0, checksum = 0.0;
  int64_t deg_dims[8];
  CHECK(mmsbm_hip_dims(ctx, deg_dims));
  int64_t *du = malloc(sizeof(int64_t) * U), *di = malloc(sizeof(int64_t) * I);
  CHECK(mmsbm_hip_degrees(ctx, du, di));
  for (int a = 0; a < U; ++a) {
    double s = 0.0;
    for (int k = 0; k < K; ++k) { s += theta[a * K + k]; checksum += (a % 7 + 1) * theta[a * K + k]; }
    int64_t present = 0;  /* users without rows keep theta = 0 */
    for (int64_t j = 0; j < n && !present; ++j) present = (u[j] == a);
    if (present && (s - 1.0 > worst || 1.0 - s > worst)) worst = s > 1.0 ? s - 1.0 : 1.0 - s;
  }
  printf("devices %d pairs %lld likelihood %.12e max|rowsum-1| %.3e checksum %.12e\n", ndev,
         (long long)deg_dims[6], lik, worst, checksum);
  CHECK(mmsbm_hip_destroy(ctx));
  free(u); free(it); free(r); free(theta); free(eta); free(pr); free(du); free(di);
  return worst < 1e-12 ? 0 : 2;
}
