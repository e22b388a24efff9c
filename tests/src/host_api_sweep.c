/* Host-side API sweep in plain C (tests/test_examples.py; also run under AddressSanitizer + UBSan by
 * scripts/sanitize_host.sh): every batched entry point of include/hector_mpc.h at least once,
 * including the argument-error paths, an empty batch, the host-pointer convenience calls that use the handle's scratch,
 * the safe pass, the binary64 copy-out enabled before and after a solve, tick-to-tick warm start, a device group, and (round 4)
 * the iteration cap, the wide variant's global-memory safe pass on hard inputs and a group of three-contact handles.
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

  /* round 4: the iteration cap (update_solver_settings' max_iter for the batched interface) and the one-call tick entry */
  CHECK(hmpc_set_max_iterations(NULL, 1) == HMPC_E_ARG && hmpc_set_max_iterations(h, -1) == HMPC_E_ARG);
  make_records(recs, stride, 0.05);
  CHECK(hmpc_upload_records(h, recs, N) == HMPC_OK && hmpc_set_max_iterations(h, 1) == HMPC_OK);
  CHECK(hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
  {
    int capped = 0;
    for (int k = 0; k < N; ++k) {
      CHECK(HMPC_STATUS_CODE(st[k]) == HMPC_S_OK || HMPC_STATUS_CODE(st[k]) == HMPC_S_MAXITER);
      capped += HMPC_STATUS_CODE(st[k]) == HMPC_S_MAXITER;
    }
    CHECK(capped > 0); /* ... and the safe pass of hmpc_download left the caller's cap alone */
  }
  CHECK(hmpc_set_max_iterations(h, 0) == HMPC_OK && hmpc_solve(h, NULL) == HMPC_OK && hmpc_download(h, forces, st) == HMPC_OK);
  for (int k = 0; k < N; ++k) CHECK(HMPC_STATUS_CODE(st[k]) == HMPC_S_OK);
  CHECK(hmpc_tick_solve_device(h, NULL, N, 0.04, NULL, NULL, NULL, NULL) == HMPC_E_ARG);
  CHECK(hmpc_tick_solve_device(NULL, ticks, N, 0.04, NULL, NULL, tau, NULL) == HMPC_E_ARG);

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
  /* round 5: the striped deal (member i holds instances i, i + 3, ...): same host-facing results, in instance order */
  {
    float *wr2 = (float *)calloc(12 * N, sizeof(float));
    float *fo2 = (float *)calloc((size_t)N * 12 * H, sizeof(float));
    uint32_t *st3 = (uint32_t *)calloc(N, sizeof(uint32_t));
    make_records(recs, stride, 0.08);
    CHECK(hmpc_group_deal(g) == HMPC_DEAL_CONTIGUOUS && hmpc_group_set_deal(g, 7) == HMPC_E_ARG && hmpc_group_set_deal(NULL, 1) == HMPC_E_ARG);
    CHECK(hmpc_group_upload_records(g, recs, N - 1) == HMPC_OK && hmpc_group_solve(g) == HMPC_OK);
    CHECK(hmpc_group_gather_wrench(g, wrench, st) == HMPC_OK && hmpc_group_download(g, forces, NULL) == HMPC_OK);
    CHECK(hmpc_group_member_step(g, 1) == 1);
    CHECK(hmpc_group_set_deal(g, HMPC_DEAL_STRIPED) == HMPC_OK && hmpc_group_deal(g) == HMPC_DEAL_STRIPED);
    CHECK(hmpc_group_upload_records(g, recs, N - 1) == HMPC_OK && hmpc_group_solve(g) == HMPC_OK);
    CHECK(hmpc_group_gather_wrench(g, wr2, st3) == HMPC_OK && hmpc_group_download(g, fo2, NULL) == HMPC_OK);
    {
      int lo = -1, nn = -1;
      CHECK(hmpc_group_member(g, 2, NULL, NULL, &lo, &nn, NULL) == HMPC_OK && lo == 2 && hmpc_group_member_step(g, 2) == 3);
      CHECK(nn == (N - 1 - 2 + 2) / 3 && hmpc_group_member_step(g, 3) == HMPC_E_ARG);
    }
    /* the same optimum for every instance: to solver precision, not bit for bit -- this batch alternates double-support and
     * walking instances with period 3, so the striped deal hands members 1 and 2 walking instances only and their handles
     * pick the 60-variable kernel variant where the contiguous slices (both kinds in every slice) ran the 120-variable one */
    for (int k = 0; k < N - 1; ++k) {
      CHECK(HMPC_STATUS_CODE(st[k]) == HMPC_S_OK && HMPC_STATUS_CODE(st3[k]) == HMPC_S_OK);
      for (int c = 0; c < 12 * H; ++c) {
        const double a = forces[(size_t)k * 12 * H + c], b = fo2[(size_t)k * 12 * H + c];
        CHECK(fabs(a - b) <= 1e-5 * fmax(1.0, fabs(a)));
      }
      CHECK(memcmp(wr2 + 12 * k, fo2 + (size_t)12 * H * k, 48) == 0); /* the gathered wrench IS step 0 of the member's forces */
    }
    CHECK(hmpc_group_set_deal(g, HMPC_DEAL_CONTIGUOUS) == HMPC_OK);
    free(wr2), free(fo2), free(st3);
  }
  CHECK(hmpc_group_synchronize(g) == HMPC_OK && hmpc_group_destroy(g) == HMPC_OK);

  /* round 5: strided upload (every second record of a host array), the dispatch-order modes, the legacy iteration cap */
  {
    hmpc_handle *h5 = NULL;
    CHECK(hmpc_create(&h5, &ps, N, 0) == HMPC_OK);
    unsigned char *wide = (unsigned char *)calloc((size_t)2 * N, stride);
    make_records(recs, stride, 0.05);
    for (int k = 0; k < N; ++k) memcpy(wide + (size_t)2 * k * stride, recs + (size_t)k * stride, stride);
    float *fa = (float *)calloc((size_t)N * 12 * H, sizeof(float)), *fb = (float *)calloc((size_t)N * 12 * H, sizeof(float));
    uint32_t *sa = (uint32_t *)calloc(N, sizeof(uint32_t)), *sb = (uint32_t *)calloc(N, sizeof(uint32_t));
    CHECK(hmpc_upload_records(h5, recs, N) == HMPC_OK && hmpc_solve(h5, NULL) == HMPC_OK && hmpc_download(h5, fa, sa) == HMPC_OK);
    CHECK(hmpc_upload_records_strided_async(h5, wide, N, stride / 2, NULL) == HMPC_E_ARG); /* pitch below one record */
    CHECK(hmpc_upload_records_strided_async(h5, wide, N, 2 * stride, NULL) == HMPC_OK);
    for (int mode = 0; mode <= 2; ++mode) {
      CHECK(hmpc_set_dispatch_order(h5, mode) == HMPC_OK && hmpc_solve(h5, NULL) == HMPC_OK && hmpc_download(h5, fb, sb) == HMPC_OK);
      CHECK(memcmp(fa, fb, sizeof(float) * (size_t)N * 12 * H) == 0 && memcmp(sa, sb, sizeof(uint32_t) * N) == 0);
    }
    CHECK(hmpc_set_dispatch_order(h5, 3) == HMPC_E_ARG && hmpc_set_dispatch_order(h5, 1) == HMPC_OK);
    CHECK(hmpc_destroy(h5) == HMPC_OK);
    free(wide), free(fa), free(fb), free(sa), free(sb);
    CHECK(hmpc_legacy_set_max_iterations(-1) == HMPC_E_ARG && hmpc_legacy_set_max_iterations(0) == HMPC_OK);
  }

  /* round 4: double support over h = 20 far outside the nominal ranges -- instances that outgrow the wide variant's working set
   * go through the safe pass whose packed Schur inverse lives in global memory (scratch grown on demand inside hmpc_download) */
  {
    enum { H2 = 20, N2 = 24 };
    struct problem_setup ps2 = {0.04f, 0.25f, 500.f, H2};
    hmpc_handle *hw = NULL;
    CHECK(hmpc_create(&hw, &ps2, N2, 0) == HMPC_OK);
    const size_t st2 = hmpc_record_stride(H2);
    unsigned char *r2 = (unsigned char *)calloc(N2, st2);
    double Q[12] = {100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1};
    double A[12] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
    for (int k = 0; k < N2; ++k) {
      double tl = 0.9 * sin(1.1 * k + 0.3), p[3] = {0, 0, 0.55}, v[3] = {1.5 * cos(0.4 * k), 1.2 * sin(0.9 * k), 0.4}, w[3] = {3.0 * tl, -2.5 * tl, 2.0 * cos(k)};
      double q[4] = {cos(tl / 2), sin(tl / 2), 0, 0}, r[6] = {0.05, -0.04, 0.06, -0.06, -0.55, -0.55}, ja[10] = {0}, traj[12 * H2] = {0};
      int gait[2 * H2];
      for (int i = 0; i < H2; ++i) traj[12 * i + 5] = 0.55, traj[12 * i + 9] = 2.0 * sin(k), gait[2 * i] = gait[2 * i + 1] = 1;
      CHECK(hmpc_pack_record(r2 + k * st2, H2, p, v, q, w, r, ja, 0.0, Q, traj, A, gait) == HMPC_OK);
    }
    float *f2 = (float *)calloc((size_t)N2 * 12 * H2, sizeof(float));
    uint32_t *s2 = (uint32_t *)calloc(N2, sizeof(uint32_t));
    CHECK(hmpc_upload_records(hw, r2, N2) == HMPC_OK && hmpc_set_auto_resolve(hw, 0) == HMPC_OK);
    CHECK(hmpc_solve(hw, NULL) == HMPC_OK && hmpc_download(hw, f2, s2) == HMPC_OK);
    int flagged = 0, nres = -1, left = 0;
    for (int k = 0; k < N2; ++k) flagged += HMPC_STATUS_CODE(s2[k]) != HMPC_S_OK;
    CHECK(hmpc_resolve_failed(hw, &nres) == HMPC_OK && nres == flagged && hmpc_download(hw, f2, s2) == HMPC_OK);
    for (int k = 0; k < N2; ++k) {
      CHECK(HMPC_STATUS_CODE(s2[k]) != HMPC_S_WORKSET && HMPC_STATUS_CODE(s2[k]) != HMPC_S_TOO_LARGE);
      left += HMPC_STATUS_CODE(s2[k]) != HMPC_S_OK && HMPC_STATUS_CODE(s2[k]) != HMPC_S_OK_RELAXED;
    }
    printf("wide variant, hard inputs: %d of %d flagged by the fast pass, %d left after the safe pass\n", flagged, N2, left);
    CHECK(hmpc_destroy(hw) == HMPC_OK);
    free(r2), free(f2), free(s2);
  }

  /* round 4: a group of three-contact handles (BASELINE config 5's split): 18 step-0 values + status per instance */
  {
    hmpc_group *g3 = NULL;
    int d2[2] = {0, 0};
    CHECK(hmpc_group_create_ex(&g3, &ps, d2, 2, N, HMPC_GROUP_P2P, 4) == HMPC_E_ARG);
    CHECK(hmpc_group_create_ex(&g3, &ps, d2, 2, N, HMPC_GROUP_P2P, 3) == HMPC_OK && hmpc_group_contacts(g3) == 3);
    const size_t s3 = hmpc_record_stride_ex(H, 3);
    unsigned char *r3 = (unsigned char *)calloc(N, s3);
    double Q[12] = {100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1}, Rh[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double A3[18] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
    for (int k = 0; k < N; ++k) {
      double vx = -0.5 + k * (1.0 / (N - 1)), p[3] = {0, 0, 0.55}, v[3] = {vx, 0, 0}, q[4] = {1, 0, 0, 0}, w[3] = {0, 0, 0};
      double r9[9] = {0, 0, 0.25, 0.06, -0.06, -0.15, -0.55, -0.55, 0.10}, ja[10] = {0}, traj[12 * H] = {0};
      int gait3[3 * H];
      for (int i = 0; i < H; ++i) traj[12 * i + 5] = 0.55, traj[12 * i + 9] = vx, gait3[3 * i] = gait3[3 * i + 1] = 1, gait3[3 * i + 2] = (i + k) % H < 6;
      CHECK(hmpc_pack_record_ex(r3 + k * s3, H, 3, p, v, q, w, r9, ja, 0.0, Q, traj, A3, gait3, Rh, 100.0) == HMPC_OK);
    }
    float *w3 = (float *)calloc((size_t)18 * N, sizeof(float)), *f3 = (float *)calloc((size_t)N * 18 * H, sizeof(float));
    CHECK(hmpc_group_upload_records(g3, r3, N) == HMPC_OK && hmpc_group_solve(g3) == HMPC_OK);
    CHECK(hmpc_group_gather_wrench(g3, w3, st) == HMPC_OK && hmpc_group_download(g3, f3, NULL) == HMPC_OK);
    for (int k = 0; k < N; ++k) CHECK(HMPC_STATUS_CODE(st[k]) == HMPC_S_OK && memcmp(w3 + 18 * k, f3 + (size_t)18 * H * k, 72) == 0);
    CHECK(hmpc_group_destroy(g3) == HMPC_OK);
    free(r3), free(w3), free(f3);
  }

  /* round 6: robot constants as data, the hand-over of full working sets (host- and device-driven, continuation only), command sweeps */
  {
    hmpc_handle *h6 = NULL;
    CHECK(hmpc_create(&h6, &ps, N, 0) == HMPC_OK);
    struct hmpc_params prm, got;
    hmpc_default_params(&prm);
    CHECK(prm.mass == 9.0f && prm.mu == 2.0f && prm.lt == 0.09f && prm.lh == 0.06f && prm.inertia[2] == 0.0691f);
    CHECK(hmpc_set_params(NULL, &prm) == HMPC_E_ARG && hmpc_get_params(h6, NULL) == HMPC_E_ARG);
    struct hmpc_params neg = prm;
    neg.mass = -1.f;
    CHECK(hmpc_set_params(h6, &neg) == HMPC_E_ARG);
    make_records(recs, stride, 0.05);
    CHECK(hmpc_upload_records(h6, recs, N) == HMPC_OK && hmpc_solve(h6, NULL) == HMPC_OK && hmpc_download(h6, forces, st) == HMPC_OK);
    prm.mass = 11.5f, prm.mu = 0.8f;
    CHECK(hmpc_set_params(h6, &prm) == HMPC_OK && hmpc_get_params(h6, &got) == HMPC_OK && got.mass == 11.5f && got.mu == 0.8f);
    CHECK(hmpc_solve(h6, NULL) == HMPC_OK && hmpc_download(h6, forces2, st) == HMPC_OK);
    double dmax = 0;
    for (int k = 0; k < N * 12 * H; ++k) dmax = fmax(dmax, fabs((double)forces2[k] - (double)forces[k]));
    CHECK(dmax > 1e-3); /* a heavier robot on a slipperier floor gets other forces */
    CHECK(hmpc_set_params(h6, NULL) == HMPC_OK && hmpc_get_params(h6, &got) == HMPC_OK && got.mass == 9.0f);
    CHECK(hmpc_legacy_set_params(&neg) == HMPC_E_ARG && hmpc_legacy_set_params(NULL) == HMPC_OK);
    /* hard inputs: working sets beyond the fast variant's 64 rows, continued (default) and re-solved (hand-over off); device repair 1, 2 */
    make_records(recs, stride, 1.1);
    int flagged[4] = {0, 0, 0, 0}, left[4] = {0, 0, 0, 0};
    for (int mode = 0; mode < 4; ++mode) {
      CHECK(hmpc_set_handover(h6, mode != 1) == HMPC_OK && hmpc_set_device_repair(h6, mode >= 2 ? mode - 1 : 0) == HMPC_OK);
      CHECK(hmpc_set_auto_resolve(h6, 0) == HMPC_OK);
      CHECK(hmpc_upload_records(h6, recs, N) == HMPC_OK && hmpc_solve(h6, NULL) == HMPC_OK && hmpc_download(h6, forces2, st) == HMPC_OK);
      for (int k = 0; k < N; ++k) flagged[mode] += HMPC_STATUS_CODE(st[k]) != HMPC_S_OK;
      CHECK(hmpc_set_auto_resolve(h6, 1) == HMPC_OK && hmpc_download(h6, forces2, st) == HMPC_OK);
      for (int k = 0; k < N; ++k) left[mode] += HMPC_STATUS_CODE(st[k]) != HMPC_S_OK && HMPC_STATUS_CODE(st[k]) != HMPC_S_OK_RELAXED;
      if (mode == 0) memcpy(forces, forces2, sizeof(float) * (size_t)N * 12 * H);
      else
        for (int k = 0; k < N * 12 * H; ++k) CHECK(fabs((double)forces2[k] - (double)forces[k]) <= 2e-5 * (1.0 + fabs((double)forces[k])));
    }
    printf("hard inputs, flagged before the host's safe pass: hand-over %d, cold %d, device repair %d, continuation only %d; left after it: %d %d %d %d\n",
           flagged[0], flagged[1], flagged[2], flagged[3], left[0], left[1], left[2], left[3]);
    CHECK(left[0] == 0 && left[1] == 0 && left[2] == 0 && left[3] == 0 && flagged[2] <= flagged[3] && flagged[3] <= flagged[0]);
    CHECK(hmpc_set_device_repair(h6, 0) == HMPC_OK);
    /* command sweeps: 12 groups of 8 records that differ in the trajectory only == the independent solves, bit for bit */
    make_records(recs, stride, 0.05);
    for (int k = 0; k < N; ++k) {
      if (k % 8 == 0) continue;
      unsigned char *dst = recs + k * stride, *src = recs + (k - k % 8) * stride;
      float trj[12 * H];
      memcpy(trj, dst + 4 * 54, sizeof trj);          /* keep this record's trajectory ... */
      memcpy(dst, src, stride);                       /* ... on the group's first record */
      for (int i = 0; i < H; ++i) trj[12 * i + 10] = 0.01f * (float)(k % 8);
      memcpy(dst + 4 * 54, trj, sizeof trj);
    }
    CHECK(hmpc_upload_records(h6, recs, N) == HMPC_OK && hmpc_solve(h6, NULL) == HMPC_OK && hmpc_download(h6, forces, st) == HMPC_OK);
    uint32_t *st2 = (uint32_t *)calloc(N, sizeof(uint32_t));
    CHECK(hmpc_solve_command_sweep(h6, 7, NULL) == HMPC_E_ARG && hmpc_solve_command_sweep(h6, 0, NULL) == HMPC_E_ARG);
    CHECK(hmpc_solve_command_sweep(h6, 8, NULL) == HMPC_OK && hmpc_download(h6, forces2, st2) == HMPC_OK);
    CHECK(memcmp(forces, forces2, sizeof(float) * (size_t)N * 12 * H) == 0 && memcmp(st, st2, sizeof(uint32_t) * N) == 0);
    recs[5 * stride + 4 * 3] ^= 1; /* one bit of v of record 5: no longer its group's state */
    CHECK(hmpc_upload_records(h6, recs, N) == HMPC_OK && hmpc_solve_command_sweep(h6, 8, NULL) == HMPC_OK && hmpc_download(h6, forces2, st2) == HMPC_OK);
    CHECK(HMPC_STATUS_CODE(st2[5]) == HMPC_S_SWEEP_MISMATCH && HMPC_STATUS_CODE(st2[4]) == HMPC_S_OK && forces2[5 * 12 * H] == 0.f);
    free(st2);
    CHECK(hmpc_destroy(h6) == HMPC_OK);
  }

  /* round 6: a Hessian that is not positive definite -> HMPC_S_INDEFINITE -> the reference's two regularised QPs inside hmpc_download
   * (tests/golden/indefinite_*.bin: three h = 20 records, the last two indefinite, and what the reference's qpOASES returns for them;
   * tests/golden/make_indefinite_golden.py).  Run from the repository root; skipped with a note when the files are not there. */
  {
    enum { H20 = 20, NI = 3 };
    const size_t stride20 = hmpc_record_stride(H20);
    FILE *fr = fopen("tests/golden/indefinite_records.bin", "rb"), *ff = fopen("tests/golden/indefinite_forces.bin", "rb");
    if (fr && ff) {
      unsigned char *r20 = (unsigned char *)malloc(NI * stride20);
      double *want = (double *)malloc(sizeof(double) * NI * 12 * H20);
      float *got = (float *)malloc(sizeof(float) * NI * 12 * H20);
      uint32_t st20[NI];
      CHECK(fread(r20, stride20, NI, fr) == NI && fread(want, sizeof(double) * 12 * H20, NI, ff) == NI);
      struct problem_setup s20 = {0.04f, 0.25f, 500.0f, H20};
      hmpc_handle *h20 = NULL;
      CHECK(hmpc_create(&h20, &s20, NI, 0) == HMPC_OK);
      CHECK(hmpc_set_auto_resolve(h20, 0) == HMPC_OK);
      CHECK(hmpc_upload_records(h20, r20, NI) == HMPC_OK && hmpc_solve(h20, NULL) == HMPC_OK && hmpc_download(h20, got, st20) == HMPC_OK);
      CHECK(HMPC_STATUS_CODE(st20[1]) != HMPC_S_OK && HMPC_STATUS_CODE(st20[2]) != HMPC_S_OK); /* the fast pass alone never calls them solved */
      int nres = -1;
      CHECK(hmpc_resolve_failed(h20, &nres) == HMPC_OK && nres >= 2 && hmpc_download(h20, got, st20) == HMPC_OK);
      for (int k = 0; k < NI; ++k) {
        double fmax = 1.0, err = 0.0;
        CHECK(HMPC_STATUS_CODE(st20[k]) == HMPC_S_OK);
        for (int i = 0; i < 12 * H20; ++i) fmax = fmax > fabs(want[k * 12 * H20 + i]) ? fmax : fabs(want[k * 12 * H20 + i]);
        for (int i = 0; i < 12 * H20; ++i) err = err > fabs((double)got[k * 12 * H20 + i] - want[k * 12 * H20 + i]) ? err : fabs((double)got[k * 12 * H20 + i] - want[k * 12 * H20 + i]);
        CHECK(err < 1e-6 * fmax);
      }
      /* a second batch on the same handle (the rho buffer is reused) with the repair on the device: no host pass needed ... */
      CHECK(hmpc_set_device_repair(h20, 1) == HMPC_OK);
      CHECK(hmpc_upload_records(h20, r20, NI) == HMPC_OK && hmpc_solve(h20, NULL) == HMPC_OK && hmpc_download(h20, got, st20) == HMPC_OK);
      for (int k = 0; k < NI; ++k) CHECK(HMPC_STATUS_CODE(st20[k]) == HMPC_S_OK);
      /* ... and a third with the repair left to hmpc_download */
      CHECK(hmpc_set_auto_resolve(h20, 1) == HMPC_OK && hmpc_set_device_repair(h20, 0) == HMPC_OK);
      CHECK(hmpc_upload_records(h20, r20, NI) == HMPC_OK && hmpc_solve(h20, NULL) == HMPC_OK && hmpc_download(h20, got, st20) == HMPC_OK);
      for (int k = 0; k < NI; ++k) CHECK(HMPC_STATUS_CODE(st20[k]) == HMPC_S_OK);
      CHECK(hmpc_destroy(h20) == HMPC_OK);
      printf("indefinite Hessians: 2 of 3 regularised as the reference does, forces within 1e-6\n");
      free(r20), free(want), free(got);
    } else {
      printf("indefinite Hessians: fixture files not found (run from the repository root) -- skipped\n");
    }
    if (fr) fclose(fr);
    if (ff) fclose(ff);
  }

  free(recs), free(forces), free(forces2), free(st), free(x64), free(obj), free(ticks), free(wpd), free(rb), free(lq), free(fff), free(tau), free(wrench);
  printf("host API sweep ok\n");
  fflush(stdout); /* (so that the line survives a tool that aborts the process during runtime teardown, e.g. a sanitizer) */
  return 0;
}
