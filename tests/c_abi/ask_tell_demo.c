/* ask_tell_demo.c — the C ABI of include/ahmc_hip.h driven from plain C99 (no C++, no Python, no torch):
 * 64 chains of a user log-density (a 3-D Gaussian with unequal scales, evaluated HERE, in the caller) sampled
 * with NUTS through the ask / tell calls, step sizes found by ahmc_ext_find_good_stepsize_begin and adapted with
 * a StepSizeAdaptor.  Linked against whichever implementation of the ABI is given on the link line — the test
 * (tests/test_capi_and_host.py) uses the CPU checker; on a GPU box the same object links against libahmc_hip.so.
 * Prints "ok <mean acceptance> <pooled variances>" and exits 0 when the pooled variances match the target's.      */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ahmc_hip.h"

#define D 3
#define N 64

static const double SCALE[D] = {0.5, 1.0, 2.0};

#define CHECK(call)                                                                       \
  do {                                                                                    \
    int32_t rc_ = (call);                                                                 \
    if (rc_ != AHMC_OK) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, ahmc_last_error(ctx));           \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

/* the caller's h.∂ℓπ∂θ(θ): ℓπ and -∇ℓπ for every chain (columns of the (D,N) column-major arrays) */
static void density(const double* theta, double* lp, double* grad_neg) {
  for (int c = 0; c < N; ++c) {
    double v = 0;
    for (int d = 0; d < D; ++d) {
      const double x = theta[c * D + d], s2 = SCALE[d] * SCALE[d];
      v += -0.5 * x * x / s2;
      grad_neg[c * D + d] = x / s2;
    }
    lp[c] = v;
  }
}

/* serve the engine's requests until the run is complete */
static int drive(ahmc_ctx* ctx, double* theta, double* lp, double* grad_neg) {
  for (;;) {
    int64_t n = 0;
    CHECK(ahmc_ext_pending(ctx, &n, NULL, theta));
    if (n == 0) return 0;
    density(theta, lp, grad_neg);
    CHECK(ahmc_ext_advance(ctx, lp, grad_neg));
  }
}

int main(void) {
  ahmc_ctx* ctx = NULL;
  static double theta[D * N], r[D * N], lp[N], grad_neg[D * N], alpha[N], sum2[D];
  if (ahmc_create(0, AHMC_F64, D, N, NULL, &ctx) != AHMC_OK) {
    fprintf(stderr, "ahmc_create: %s\n", ahmc_last_error(NULL));
    return 1;
  }
  CHECK(ahmc_set_target(ctx, AHMC_TARGET_EXTERNAL, NULL, 0));
  CHECK(ahmc_set_metric(ctx, AHMC_METRIC_UNIT, NULL, 0));
  const double eps0 = 0.1;
  CHECK(ahmc_set_stepsize(ctx, &eps0, 1));
  CHECK(ahmc_seed(ctx, 2026, 0, 1, 0));
  for (int i = 0; i < D * N; ++i) theta[i] = 0.1 * ((i * 37) % 19 - 9);
  memset(r, 0, sizeof r);
  density(theta, lp, grad_neg);
  CHECK(ahmc_set_phasepoint(ctx, theta, r, lp, grad_neg));
  CHECK(ahmc_ext_find_good_stepsize_begin(ctx, 0.1, 100));
  if (drive(ctx, theta, lp, grad_neg)) return 1;
  CHECK(ahmc_adaptor_init(ctx, AHMC_ADAPT_STEPSIZE, 0.8, 75, 50, 25));
  ahmc_kernel_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.nuts = 1;
  cfg.sampler = AHMC_TS_MULTINOMIAL;
  cfg.criterion = AHMC_TC_GENERALISED;
  cfg.max_depth = 8;
  cfg.delta_max = 1000.0;
  const int n_adapts = 150, n_samples = 450;
  double acc = 0;
  long kept = 0;
  for (int i = 1; i <= n_samples; ++i) {
    CHECK(ahmc_ext_begin(ctx, &cfg, 1));
    if (drive(ctx, theta, lp, grad_neg)) return 1;
    CHECK(ahmc_adapt(ctx, i, n_adapts, NULL, NULL));
    if (i > n_adapts) {
      CHECK(ahmc_get_phasepoint(ctx, theta, NULL, NULL, NULL, NULL));
      CHECK(ahmc_get_stat(ctx, AHMC_STAT_ACCEPTANCE_RATE, alpha));
      for (int c = 0; c < N; ++c) {
        acc += alpha[c];
        for (int d = 0; d < D; ++d) sum2[d] += theta[c * D + d] * theta[c * D + d];
      }
      kept += N;
    }
  }
  int64_t it = 0;
  CHECK(ahmc_get_info(ctx, AHMC_INFO_ITERATION, &it));
  int bad = it != n_samples;
  printf("ok %.3f", acc / kept);
  for (int d = 0; d < D; ++d) {
    const double var = sum2[d] / kept, want = SCALE[d] * SCALE[d];
    printf(" %.3f", var);
    if (fabs(var / want - 1) > 0.12) bad = 1;
  }
  printf("\n");
  ahmc_destroy(ctx);
  return bad;
}
