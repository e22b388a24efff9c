/* Batched C ABI from plain C: pack records from update_problem_data-style arguments, solve, download.
 *   gcc -std=c11 -Iinclude examples/batched.c -Lhector_simulation_amd -lhector_mpc_hip -Wl,-rpath,$PWD/hector_simulation_amd -o batched */
#include <stdio.h>
#include <stdlib.h>

#include "hector_mpc.h"

int main(void) {
  enum { H = 10, N = 256 };
  struct problem_setup ps = {0.04f, 0.25f, 500.f, H};
  hmpc_handle *h = NULL;
  int rc = hmpc_create(&h, &ps, N, 0);
  if (rc != HMPC_OK) {
    fprintf(stderr, "hmpc_create failed (%d): %s\n", rc, hmpc_last_hip_error());
    return 2;
  }
  const size_t stride = hmpc_record_stride(H);
  unsigned char *recs = (unsigned char *)calloc(N, stride);
  double Q[12] = {100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1};
  double A[12] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
  for (int k = 0; k < N; ++k) { /* a sweep over the commanded forward velocity, walking gait */
    double vx = -0.5 + k * (1.0 / (N - 1));
    double p[3] = {0, 0, 0.55}, v[3] = {vx, 0, 0}, q[4] = {1, 0, 0, 0}, w[3] = {0, 0, 0};
    double r[6] = {0, 0, 0.06, -0.06, -0.55, -0.55}, ja[10] = {0}, traj[12 * H] = {0};
    int gait[2 * H];
    for (int i = 0; i < H; ++i) {
      traj[12 * i + 3] = i * 0.04 * vx, traj[12 * i + 5] = 0.55, traj[12 * i + 9] = vx;
      gait[2 * i] = (i + k) % H < H / 2, gait[2 * i + 1] = !gait[2 * i];
    }
    hmpc_pack_record(recs + k * stride, H, p, v, q, w, r, ja, 0.0, Q, traj, A, gait);
  }
  float *forces = (float *)malloc(sizeof(float) * N * 12 * H);
  uint32_t *st = (uint32_t *)malloc(sizeof(uint32_t) * N);
  rc = hmpc_upload_records(h, recs, N);
  if (rc == HMPC_OK) rc = hmpc_solve(h, NULL);
  if (rc == HMPC_OK) rc = hmpc_download(h, forces, st);
  int bad = 0;
  for (int k = 0; k < N; ++k) bad += HMPC_STATUS_CODE(st[k]) != HMPC_S_OK;
  printf("rc %d, %d of %d instances not ok; instance 0: Fz_L %.3f Fz_R %.3f\n", rc, bad, N, forces[2], forces[5]);
  hmpc_destroy(h);
  free(recs), free(forces), free(st);
  return (rc == HMPC_OK && bad == 0) ? 0 : 1;
}
