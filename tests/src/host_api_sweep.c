/* Host-side API sweep in plain C (tests/test_examples.py; also run under AddressSanitizer + UBSan by
 * scripts/sanitize_host.sh): every batched entry point of include/hector_mpc.h at least once,
 * including the argument-error paths, an empty batch, the host-pointer convenience calls that use the handle's scratch,
 * the safe pass, the binary64 copy-out enabled before and after a solve, tick-to-tick warm start and a device group.
 * Exit code 0 = every call behaved as documented. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hector_mpc.h"

#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) {                                                             \
      fprintf(stderr, "FAILED line %d: %s (%s)\n", __LINE__, #cond, hmpc_last_hip_error()); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

enum { H = 10, N = 96 };

static void make_records(unsigned char *recs, size_t stride, double tilt) {
  double Q[12] = {100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1};
  double A[12] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
  for (int k = 0; k < N; ++k) {
    double vx = -0.5 + k * (1.0 / (N - 1)), roll = tilt * sin(0.7 * k), pitch = tilt * cos(1.3 * k);
    double p[3] = {0, 0, 0.55}, v[3] = {vx, 0.1 * tilt, 0}, w[3] = {tilt, -tilt, 0.5 * tilt};
    double q[4] = {cos(roll / 2) * cos(pitch / 2), sin(roll / 2) * cos(pitch / 2), cos(roll / 2) * sin(pitch / 2),
                   -sin(roll / 2) * sin(pitch / 2)};
    double r[6] = {0.02, -0.01, 0.06, -0.06, -0.55, -0.55}, ja[10] = {0}, traj[12 * H] = {0};
    int gait[2 * H];
    for (int i = 0; i < H; ++i) {
      traj[12 * i + 3] = i * 0.04 * vx, traj[12 * i + 5] = 0.55, traj[12 * i + 9] = vx;
      gait[2 * i] = (k % 3 == 0) ? 1 : ((i + k) % H < H / 2), gait[2 * i + 1] = (k % 3 == 0) ? 1 : !gait[2 * i];
    }
    hmpc_pack_record(recs + k * stride, H, p, v, q, w, r, ja, 0.0, Q, traj, A, gait);
  }
}

int main(void) {
  struct problem_setup ps = {0.04f, 0.25f, 500.f, H};
  hmpc_handle *h = NULL;
  /* argument errors never crash and never throw */
  CHECK(hmpc_create(NULL, &ps, N, 0) == HMPC_E_ARG);
  CHECK(hmpc_create(&h, NULL, N, 0) == HMPC_E_ARG);
  struct problem_setup bad = ps;
  bad.horizon = 21;
  CHECK(hmpc_create(&h, &bad, N, 0) == HMPC_E_HORIZON);
  CHECK(hmpc_solve(NULL, NULL) == HMPC_E_ARG && hmpc_destroy(NULL) == HMPC_E_ARG);
  int rc = hmpc_create(&h, &ps, N, 0);
  if (rc == HMPC_E_NO_DEVICE) {
    fprintf(stderr, "no HIP device visible: %s\n", hmpc_last_hip_error());
    return 3;
  }
  CHECK(rc == HMPC_OK);
  const size_t stride = hmpc_record_stride(H);
  unsigned char *recs = (unsigned char *)calloc(N, stride);
  float *forces = (float *)calloc((size_t)N * 12 * H, sizeof(float)), *forces2 = (float *)calloc((size_t)N * 12 * H, sizeof(float));
  uint32_t *st = (uint32_t *)calloc(N, sizeof(uint32_t));
  double *x64 = (double *)calloc((size_t)N * 12 * H, sizeof(double)), *obj = (double *)calloc(N, sizeof(double));
  make_records(recs, stride, 0.05);

  /* empty batch: everything is a no-op */
  CHECK(hmpc_upload_records(h, recs, 0) == HMPC_OK && hmpc_solve(h, NULL) == HMPC_OK);
  CHECK(hmpc_download(h, forces, st) == HMPC_OK && hmpc_download_f64(h, x64, obj) == HMPC_OK);
  float ms = -1.f;
  CHECK(hmpc_time_solve(h, NULL, 3, &ms) == HMPC_OK && ms == 0.f);
  CHECK(hmpc_upload_records(h, recs, N + 1) == HMPC_E_BATCH);

  /* plain solve, then the binary64 copy-out requested AFTER the solve (re-runs the batch) */
  CHECK(hmpc_upload_records(h, recs, N) == HMPC_OK && hmpc_batch(h) == N && hmpc_horizon(h) == H);
  CHECK(hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
  int bad_n = 0;
  for (int k = 0; k < N; ++k) bad_n += HMPC_STATUS_CODE(st[k]) != HMPC_S_OK;
  CHECK(bad_n == 0);
  CHECK(hmpc_download_f64(h, x64, obj) == HMPC_OK);
  for (int k = 0; k < N * 12 * H; ++k) CHECK(fabs(x64[k] - (double)forces[k]) <= 1e-4 * (1.0 + fabs(x64[k])));
  /* enabled beforehand: no second launch needed, same numbers */
  CHECK(hmpc_enable_f64_output(h) == HMPC_OK && hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces2, st) == HMPC_OK);
  CHECK(memcmp(forces, forces2, sizeof(float) * N * 12 * H) == 0);
  CHECK(hmpc_time_solve(h, NULL, 2, &ms) == HMPC_OK && ms > 0.f);

  /* cold start = same optimum; safe pass on inputs far outside the nominal range */
  CHECK(hmpc_set_warm_start(h, 0) == HMPC_OK && hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces2, st) == HMPC_OK);
  for (int k = 0; k < N * 12 * H; ++k) CHECK(fabsf(forces[k] - forces2[k]) <= 1e-4f * (1.f + fabsf(forces[k])));
  CHECK(hmpc_set_warm_start(h, 1) == HMPC_OK);
  make_records(recs, stride, 0.6);
  CHECK(hmpc_set_auto_resolve(h, 0) == HMPC_OK);
  CHECK(hmpc_upload_records(h, recs, N) == HMPC_OK && hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
  int flagged = 0, resolved = -1;
  for (int k = 0; k < N; ++k) {
    const uint32_t c = HMPC_STATUS_CODE(st[k]);
    flagged += (c == HMPC_S_WORKSET || c == HMPC_S_MAXITER || c == HMPC_S_KKT || c == HMPC_S_INFEASIBLE);
  }
  CHECK(hmpc_resolve_failed(h, &resolved) == HMPC_OK && resolved == flagged);
  CHECK(hmpc_set_auto_resolve(h, 1) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
  printf("hard batch: %d flagged by the fast pass, re-solved by the safe pass\n", flagged);

  /* tick-to-tick warm start: second solve of the same data needs (almost) no iterations and gives the same forces */
  make_records(recs, stride, 0.05);
  CHECK(hmpc_set_tick_warm_start(h, 1, 0) == HMPC_OK);
  CHECK(hmpc_upload_records(h, recs, N) == HMPC_OK && hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
  CHECK(hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces2, st) == HMPC_OK);
  long it2 = 0;
  for (int k = 0; k < N; ++k) it2 += HMPC_STATUS_ITERS(st[k]);
  for (int k = 0; k < N * 12 * H; ++k) CHECK(fabsf(forces[k] - forces2[k]) <= 1e-4f * (1.f + fabsf(forces[k])));
  CHECK(it2 <= N && hmpc_reset_tick_warm_start(h) == HMPC_OK && hmpc_set_tick_warm_start(h, 0, 0) == HMPC_OK);

  /* rows f1-f3 through the host-pointer forms (handle scratch, twice to exercise its reuse and growth) */
  struct hmpc_tick_inputs *ticks = (struct hmpc_tick_inputs *)calloc(N, sizeof(*ticks));
  double *wpd = (double *)calloc(2 * N, sizeof(double)), *rb = (double *)calloc(9 * N, sizeof(double));
  double *lq = (double *)calloc(10 * N, sizeof(double)), *fff = (double *)calloc(12 * N, sizeof(double)), *tau = (double *)calloc(10 * N, sizeof(double));
  for (int k = 0; k < N; ++k) {
    struct hmpc_tick_inputs *t = &ticks[k];
    t->position[2] = 0.55, t->orientation[0] = 1.0;
    t->rBody[0] = t->rBody[4] = t->rBody[8] = 1.0;
    t->pFoot[1] = 0.06, t->pFoot[4] = -0.06;
    t->v_des_robot[0] = 0.2;
    t->gait_offsets[1] = 5, t->gait_durations[0] = t->gait_durations[1] = 5, t->gait_iteration = k % H;
    t->flags = (k & 1) ? HMPC_TICK_LEG_Q_MOTOR : 0;
    rb[9 * k] = rb[9 * k + 4] = rb[9 * k + 8] = 1.0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hmpc_build_records(h, ticks, N, 0.04, wpd) == HMPC_OK);
    CHECK(hmpc_download_records(h, recs) == HMPC_OK);
    CHECK(hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
    CHECK(hmpc_body_wrench(h, rb, fff) == HMPC_OK && hmpc_leg_torques(h, rb, lq, fff, tau) == HMPC_OK);
  }
  for (int k = 0; k < N; ++k) CHECK(HMPC_STATUS_CODE(st[k]) == HMPC_S_OK && fabs(fff[12 * k + 2] + (double)forces[(size_t)k * 12 * H + 2]) < 1e-12);
  CHECK(hmpc_build_records(h, NULL, N, 0.04, wpd) == HMPC_E_ARG && hmpc_body_wrench(h, NULL, fff) == HMPC_E_ARG);

  /* device-side safe pass (round 3): hard batch again, repaired on the device without hmpc_resolve_failed */
  make_records(recs, stride, 0.6);
  CHECK(hmpc_set_device_repair(NULL, 1) == HMPC_E_ARG && hmpc_set_device_repair(h, 1) == HMPC_OK);
  CHECK(hmpc_set_auto_resolve(h, 0) == HMPC_OK);
  CHECK(hmpc_upload_records(h, recs, N) == HMPC_OK && hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
  for (int k = 0; k < N; ++k) CHECK(HMPC_STATUS_CODE(st[k]) != HMPC_S_WORKSET);
  CHECK(hmpc_set_device_repair(h, 0) == HMPC_OK && hmpc_set_auto_resolve(h, 1) == HMPC_OK);

  /* external-QP parity hook (round 3): the kernel's own QP handed back in must reproduce the kernel's own forces */
  {
    make_records(recs, stride, 0.05);
    CHECK(hmpc_upload_records(h, recs, N) == HMPC_OK && hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
    const int ld = HMPC_MAX_VARS;
    float *Hx = (float *)calloc((size_t)N * ld * ld, sizeof(float)), *gx = (float *)calloc((size_t)N * ld, sizeof(float));
    float *Fx = (float *)calloc((size_t)N * 16 * 12, sizeof(float)), *Hk = (float *)calloc((size_t)ld * ld, sizeof(float));
    for (int k = 0; k < N; ++k) {
      int nk = 0, mk = 0;
      CHECK(hmpc_debug_assemble(h, k, &nk, &mk, NULL, Hk, gx + (size_t)k * ld, Fx + (size_t)k * 192, NULL, NULL, NULL, NULL, NULL) == HMPC_OK);
      CHECK(nk > 0 && nk <= ld);
      for (int i = 0; i < nk; ++i) memcpy(Hx + ((size_t)k * ld + i) * ld, Hk + (size_t)i * nk, sizeof(float) * nk);
    }
    CHECK(hmpc_debug_solve_external_qp(h, Hx, gx, NULL, ld) == HMPC_E_ARG);
    CHECK(hmpc_debug_solve_external_qp(h, Hx, gx, Fx, ld) == HMPC_OK && hmpc_download(h, forces2, st) == HMPC_OK);
    CHECK(memcmp(forces, forces2, sizeof(float) * N * 12 * H) == 0);
    free(Hx), free(gx), free(Fx), free(Hk);
  }

  /* parity hook */
  int n = 0, m = 0;
  CHECK(hmpc_debug_assemble(h, 0, &n, &m, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == HMPC_OK && n > 0 && m > 0);
  CHECK(hmpc_debug_assemble(h, N, &n, &m, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == HMPC_E_ARG);
  CHECK(hmpc_destroy(h) == HMPC_OK);

  /* device group: three members on device 0 (P2P transport), ragged slices */
  hmpc_group *g = NULL;
  int devs[3] = {0, 0, 0};
  CHECK(hmpc_group_create(&g, &ps, devs, 3, N, HMPC_GROUP_RCCL) == HMPC_E_ARG); /* RCCL refuses repeated devices */
  CHECK(hmpc_group_create(&g, &ps, devs, 3, N, HMPC_GROUP_AUTO) == HMPC_OK && hmpc_group_transport(g) == HMPC_GROUP_P2P);
  make_records(recs, stride, 0.05);
  float *wrench = (float *)calloc(12 * N, sizeof(float));
  CHECK(hmpc_group_upload_records(g, recs, N - 1) == HMPC_OK && hmpc_group_solve(g) == HMPC_OK);
  CHECK(hmpc_group_gather_wrench(g, wrench, st) == HMPC_OK && hmpc_group_download(g, forces, NULL) == HMPC_OK);
  for (int k = 0; k < N - 1; ++k) CHECK(memcmp(wrench + 12 * k, forces + (size_t)12 * H * k, 48) == 0);
  /* round 3: a collected exchange is not handed out again -- solve, post, wait, solve, gather returns the SECOND solve */
  make_records(recs, stride, 0.08);
  CHECK(hmpc_group_post_gather(g) == HMPC_OK && hmpc_group_wait_gather(g) == HMPC_OK);
  CHECK(hmpc_group_upload_records(g, recs, N) == HMPC_OK && hmpc_group_solve(g) == HMPC_OK);
  CHECK(hmpc_group_gather_wrench(g, wrench, st) == HMPC_OK && hmpc_group_download(g, forces, NULL) == HMPC_OK);
  for (int k = 0; k < N; ++k) CHECK(memcmp(wrench + 12 * k, forces + (size_t)12 * H * k, 48) == 0);
  CHECK(hmpc_group_set_exchange_repair(g, 0) == HMPC_OK && hmpc_group_solve(g) == HMPC_OK && hmpc_group_gather_wrench(g, wrench, st) == HMPC_OK);
  CHECK(hmpc_group_set_exchange_repair(g, 1) == HMPC_OK && hmpc_group_set_exchange_repair(NULL, 1) == HMPC_E_ARG);
  CHECK(hmpc_group_upload_records(g, recs, 0) == HMPC_OK && hmpc_group_solve(g) == HMPC_OK && hmpc_group_gather_wrench(g, wrench, st) == HMPC_OK);
  CHECK(hmpc_group_synchronize(g) == HMPC_OK && hmpc_group_destroy(g) == HMPC_OK);

  free(recs), free(forces), free(forces2), free(st), free(x64), free(obj), free(ticks), free(wpd), free(rb), free(lq), free(fff), free(tau), free(wrench);
  printf("host API sweep ok\n");
  return 0;
}
