/* Device group from plain C: one process, every GPU of the node (hipGetDeviceCount through the library: a group over
 * devices 0..G-1), contiguous slices per device, the solved step-0 wrenches gathered on every device and on the host.
 *   gcc -std=c11 -Iinclude examples/batched_multi.c -Lhector_simulation_amd -lhector_mpc_hip \
 *       -Wl,-rpath,$PWD/hector_simulation_amd -o batched_multi
 *   ./batched_multi [n_devices [p2p|auto [contacts [striped]]]]   (default: as many as hmpc_group_create accepts, probing 8,4,2,1)
 * "striped": the batch -- a sweep ordered by commanded velocity -- is dealt round-robin (hmpc_group_set_deal: member i holds
 * instances i, i + G, ...) instead of cut into contiguous slices, so that every GPU gets the same mix of easy and hard instances.
 * "p2p" with n_devices > visible devices lists device 0 repeatedly (how a one-GPU box exercises the multi-member path).
 * contacts = 3: the loco-manipulation extension (BASELINE config 5 runs it on 4 GPUs) -- hmpc_group_create_ex, records of
 * hmpc_pack_record_ex, and an exchange of 18 step-0 values [F_L F_R F_H M_L M_R M_H] + status per instance. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hector_mpc.h"

int main(int argc, char **argv) {
  enum { H = 10, N = 1000 }; /* 1000 over 3 members = 334 + 333 + 333: the ragged case */
  struct problem_setup ps = {0.04f, 0.25f, 500.f, H};
  const int want = argc > 1 ? atoi(argv[1]) : 0;
  const int p2p = argc > 2 && !strcmp(argv[2], "p2p");
  const int nc = (argc > 3 && atoi(argv[3]) == 3) ? 3 : 2, W = 6 * nc; /* contacts; step-0 wrench width */
  const int striped = argc > 4 && !strcmp(argv[4], "striped");
  hmpc_group *g = NULL;
  int rc = HMPC_E_ARG, G = 0;
  if (p2p) {
    int devs[16] = {0};
    G = want > 0 && want <= 16 ? want : 3;
    rc = hmpc_group_create_ex(&g, &ps, devs, G, N, HMPC_GROUP_P2P, nc);
  } else {
    const int probe[4] = {8, 4, 2, 1};
    for (int i = 0; i < 4 && rc != HMPC_OK; ++i) {
      G = want > 0 ? want : probe[i];
      rc = hmpc_group_create_ex(&g, &ps, NULL, G, N, HMPC_GROUP_AUTO, nc);
      if (want > 0) break;
    }
  }
  if (rc != HMPC_OK) {
    fprintf(stderr, "hmpc_group_create failed (%d): %s\n", rc, hmpc_group_last_error());
    return 2;
  }
  const size_t stride = hmpc_record_stride_ex(H, nc);
  unsigned char *recs = (unsigned char *)calloc(N, stride);
  double Q[12] = {100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1};
  double A[12] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
  double A3[18] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
  double Rhand[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < N; ++k) { /* velocity-command x gait-phase sweep */
    double vx = -0.5 + k * (1.0 / (N - 1));
    double p[3] = {0, 0, 0.55}, v[3] = {vx, 0, 0}, q[4] = {1, 0, 0, 0}, w[3] = {0, 0, 0};
    double r[6] = {0, 0, 0.06, -0.06, -0.55, -0.55}, ja[10] = {0}, traj[12 * H] = {0};
    int gait[2 * H];
    for (int i = 0; i < H; ++i) {
      traj[12 * i + 3] = i * 0.04 * vx, traj[12 * i + 5] = 0.55, traj[12 * i + 9] = vx;
      gait[2 * i] = (i + k) % H < H / 2, gait[2 * i + 1] = !gait[2 * i];
    }
    if (nc == 2) {
      hmpc_pack_record(recs + k * stride, H, p, v, q, w, r, ja, 0.0, Q, traj, A, gait);
    } else { /* both feet down, the hand on a surface in front of the body for part of the horizon */
      double r3[9] = {0, 0, 0.25, 0.06, -0.06, -0.15, -0.55, -0.55, 0.10}; /* r[3*axis + contact] */
      int gait3[3 * H];
      for (int i = 0; i < H; ++i) gait3[3 * i] = gait3[3 * i + 1] = 1, gait3[3 * i + 2] = (i + k) % H < 7;
      hmpc_pack_record_ex(recs + k * stride, H, 3, p, v, q, w, r3, ja, 0.0, Q, traj, A3, gait3, Rhand, 100.0);
    }
  }
  float *wrench = (float *)malloc(sizeof(float) * N * W);
  float *forces = (float *)malloc(sizeof(float) * N * W * H);
  uint32_t *st = (uint32_t *)malloc(sizeof(uint32_t) * N), *st2 = (uint32_t *)malloc(sizeof(uint32_t) * N);
  rc = hmpc_group_set_deal(g, striped ? HMPC_DEAL_STRIPED : HMPC_DEAL_CONTIGUOUS);
  if (rc == HMPC_OK) rc = hmpc_group_upload_records(g, recs, N);
  if (rc == HMPC_OK) rc = hmpc_group_solve(g);
  if (rc == HMPC_OK) rc = hmpc_group_gather_wrench(g, wrench, st); /* the exchange step */
  if (rc == HMPC_OK) rc = hmpc_group_download(g, forces, st2);     /* everything, for the cross-check below */
  int bad = 0, mismatch = 0;
  for (int k = 0; rc == HMPC_OK && k < N; ++k) {
    bad += HMPC_STATUS_CODE(st[k]) != HMPC_S_OK;
    mismatch += st[k] != st2[k] || memcmp(wrench + W * k, forces + (size_t)W * H * k, W * sizeof(float)) != 0;
  }
  for (int i = 0; i < G; ++i) {
    int dev, lo, n;
    hmpc_group_member(g, i, NULL, &dev, &lo, &n, NULL);
    const int step = hmpc_group_member_step(g, i);
    if (step == 1) printf("member %d: device %d, instances [%d, %d)\n", i, dev, lo, lo + n);
    else printf("member %d: device %d, %d instances %d, %d, %d, ...\n", i, dev, n, lo, lo + step, lo + 2 * step);
  }
  printf("rc %d, group of %d (%s), %d contacts, %d of %d not ok, %d gathered rows differ from the full download; instance 0: Fz_L %.3f Fz_R %.3f\n",
         rc, G, hmpc_group_transport(g) == HMPC_GROUP_RCCL ? "rccl" : "p2p", hmpc_group_contacts(g), bad, N, mismatch, wrench[2], wrench[5]);
  fflush(stdout); /* (so that the line survives a tool that aborts the process during runtime teardown, e.g. a sanitizer) */
  if (rc != HMPC_OK) fprintf(stderr, "error: %s\n", hmpc_group_last_error());
  hmpc_group_destroy(g);
  free(recs), free(wrench), free(forces), free(st), free(st2);
  return (rc == HMPC_OK && bad == 0 && mismatch == 0) ? 0 : 1;
}
