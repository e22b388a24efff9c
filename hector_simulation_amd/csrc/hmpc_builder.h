// hmpc_builder.h -- the caller-side rows either side of the solve, batched on the device:
//   f1  record builder      = ConvexMPCLocomotion::updateMPCIfNeeded's input construction
//                             (ConvexMPC/ConvexMPCLocomotion.cpp:283-406) + the double->float narrowing of
//                             update_problem_data (ConvexMPC/convexMPC_interface.cpp:83-103)
//   f2  gait table          = Gait::mpc_gait (ConvexMPC/GaitGenerator.cpp:85-103), fused into f1
//   f3  body-frame wrenches = f_ff[leg] = -rBody [GRF; GRM] (ConvexMPCLocomotion.cpp:419-440)
// Both are plain streaming kernels (HBM-bound): binary64 arithmetic in the order the reference writes it, no contraction
// (this translation unit is built with -ffp-contract=off), so the packed records are bit-identical to the CPU restatement,
// which is pinned bit for bit against the reference's own GaitGenerator.cpp / ConvexMPCLocomotion.cpp / LegController.cpp
// executed (oracle/_ref/libcaller_ref.so; tests/test_caller_reference.py, also the GPU leg against reference-generated goldens).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hector_mpc.h"
#include "hmpc_math.h"

namespace hmpc {

// One workgroup per instance; thread t produces 32-bit word t of the packed record (coalesced 720-B burst out).
// cls (optional): the instance's stance leg-step count -- the size class hmpc_solve routes it by (KernelArgs::cls), counted
// with the solver's own criterion (|f_max * gait| >= 1e-4, SolverMPC.cpp:589-637).
__global__ __launch_bounds__(256) void build_records_kernel(const hmpc_tick_inputs *ticks, int batch, int h, double dtMPC,
                                                            unsigned char *records, int stride, double *wpd_out, float f_max,
                                                            unsigned char *cls) {
  const int inst = blockIdx.x;
  if (inst >= batch) return;
  const hmpc_tick_inputs &tk = ticks[inst];
  const int nwords = stride >> 2, ntraj = 12 * h;
  uint32_t *out = reinterpret_cast<uint32_t *>(records + (size_t)inst * stride);
  const double PI = 3.14159265359, PI2 = 2 * PI;
  const double *p = tk.position;
  // shared scalars (recomputed per thread: a handful of flops on broadcast loads)
  double vdw[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    vdw[i] = (tk.rBody[0 * 3 + i] * tk.v_des_robot[0] + tk.rBody[1 * 3 + i] * tk.v_des_robot[1]) + tk.rBody[2 * 3 + i] * 0.0;
  const double max_pos_error = .05;
  double xStart = tk.world_position_desired[0], yStart = tk.world_position_desired[1];
  if (xStart - p[0] > max_pos_error) xStart = p[0] + max_pos_error;
  if (p[0] - xStart > max_pos_error) xStart = p[0] - max_pos_error;
  if (yStart - p[1] > max_pos_error) yStart = p[1] + max_pos_error;
  if (p[1] - yStart > max_pos_error) yStart = p[1] - max_pos_error;
  if (threadIdx.x == 0 && wpd_out) {
    wpd_out[2 * inst + 0] = xStart;
    wpd_out[2 * inst + 1] = yStart;
  }
  if (cls && threadIdx.x == 64) {  // (a lane of the second wave: thread 0 has the clamp to write)
    int cnt = 0;
    for (int gi = 0; gi < 2 * h; ++gi) {
      const int i = gi >> 1, j = gi & 1;
      int progress = (i + tk.gait_iteration) % h - tk.gait_offsets[j];
      if (progress < 0) progress += h;
      const float ubc = f_max * (float)((progress < tk.gait_durations[j]) ? 1 : 0);
      cnt += !(ubc < 0.0001 && ubc > -.0001);
    }
    cls[inst] = (unsigned char)cnt;
  }
  for (int t = threadIdx.x; t < nwords; t += blockDim.x) {
    uint32_t word = 0;
    if (t < 54 + ntraj) {
      double v;
      if (t < 3) v = p[t];
      else if (t < 6) v = tk.vWorld[t - 3];
      else if (t < 10) v = tk.orientation[t - 6];
      else if (t < 13) v = tk.omegaWorld[t - 10];
      else if (t < 19) {
        const int i = t - 13;
        v = tk.pFoot[3 * (i % 2) + i / 2] - p[i / 2];
      } else if (t < 29) {
        const int i = t - 19, k = i % 5;
        double a = tk.leg_q[i];
        if (tk.flags & HMPC_TICK_LEG_Q_MOTOR) {  // LegController.cpp:111-113 mutates data[leg].q before the MPC reads it
          if (k == 2 || k == 4) a = a + 0.3 * 3.14159;
          if (k == 3) a = a - 0.6 * 3.14159;
        }
        if (k == 2 || k == 4) a += 0.3 * PI;
        if (k == 3) a -= 0.6 * PI;
        v = (__builtin_fabs(a) < PI2) ? a : fmod(a, PI2);  // fmod(x,y) == x exactly when |x| < y
      } else if (t == 29) v = tk.rpy[2];
      else if (t < 42) {
        const int i = t - 30;
        v = (i < 2) ? 100.0 : (i == 2 ? 250.0 : (i < 5 ? 200.0 : (i == 5 ? 300.0 : 1.0)));
      } else if (t < 54) {
        const int i = t - 42;
        v = (i >= 6) ? 1e-2 : ((i == 2 || i == 5) ? 5e-4 : 1e-4);
      } else {
        const int i = (t - 54) / 12, j = (t - 54) % 12;
        // trajInitial (ConvexMPCLocomotion.cpp:351-362)
        double ti;
        switch (j) {
          case 0: ti = tk.roll_des; break;
          case 1: ti = tk.pitch_des; break;
          case 3: ti = xStart; break;
          case 4: ti = yStart; break;
          case 5: ti = 0.55; break;
          case 8: ti = tk.yaw_rate_des; break;
          case 9: ti = vdw[0]; break;
          case 10: ti = vdw[1]; break;
          default: ti = 0.0; break;
        }
        v = ti;
        if (i == 0) {
          if (j < 3) v = tk.rpy[j];
          else if (j < 6) v = p[j - 3];
        } else {
          if (j == 3) v = (vdw[0] == 0) ? xStart + i * dtMPC * vdw[0] : p[0] + i * dtMPC * vdw[0];
          if (j == 4) v = (vdw[1] == 0) ? yStart + i * dtMPC * vdw[1] : p[1] + i * dtMPC * vdw[1];
          if (j == 2) v = (tk.yaw_rate_des == 0) ? 0.0 : tk.rpy[2] + i * dtMPC * tk.yaw_rate_des;
        }
      }
      word = __float_as_uint((float)v);
    } else {
      // gait bytes (GaitGenerator.cpp:85-103), four per word
      const int b0 = 4 * (t - 54 - ntraj);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int gi = b0 + bb;
        uint32_t byte = 0;
        if (gi < 2 * h) {
          const int i = gi >> 1, j = gi & 1;
          const int iter = (i + tk.gait_iteration) % h;
          int progress = iter - tk.gait_offsets[j];
          if (progress < 0) progress += h;
          byte = (progress < tk.gait_durations[j]) ? 1u : 0u;
        }
        word |= byte << (8 * bb);
      }
    }
    out[t] = word;
  }
}

// size classes of records that are already in HBM (hmpc_set_device_records without a hint): one thread per instance counts
// the stance leg-steps from the record's gait bytes, with the solve kernel's criterion (two-contact records)
__global__ __launch_bounds__(256) void classify_records_kernel(const unsigned char *records, int stride, int batch, int h,
                                                               float f_max, unsigned char *cls) {
  const int inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= batch) return;
  const unsigned char *g = records + (size_t)inst * stride + 4 * (54 + 12 * h);
  int cnt = 0;
  for (int i = 0; i < 2 * h; ++i) {
    const float ubc = f_max * (float)g[i];
    cnt += !(ubc < 0.0001 && ubc > -.0001);
  }
  cls[inst] = (unsigned char)(cnt > 255 ? 255 : cnt);
}

// Cost predictor for a COLD handle (round 5): which instances will take many active-set iterations, read off the record alone --
// no previous solve needed.  What the iteration count of this QP follows (fitted offline on the iteration counts of the bench's
// and the stress sets' instances, scripts/dev/fit_predictor.py; correlation 0.8 with the iterations of the 2-contact and
// 3-contact sets, 0.6-0.7 on single support / 3x ranges): the forward acceleration the tick asks for, u = (vx_cmd - vx) +
// 2 * mean foot x -- a body slower than commanded with its feet ahead of it has to push through the heel edges of the line
// contacts and the friction rows, many rows become active (15 iterations on average at u = 0.75 against 2 at u = 0); the other
// direction costs a twentieth of that; tilt adds a little.  Returned as one of 64 buckets (higher = longer), the key the
// counting sort below uses in place of the previous solve's iteration count.
__device__ __forceinline__ int predicted_cost_bucket(const unsigned char *rec, int h, int nc) {
  const float *f = reinterpret_cast<const float *>(rec);
  const int nf = (nc == 3) ? 73 : 54;
  const float vx = f[3], qw = f[6], qx = f[7], qy = f[8], qz = f[9];
  const float mrx = 0.5f * (f[13] + f[14]);            // r_feet(axis, contact) = r[nc * axis + contact]: x of the two feet
  const float vcmd = f[nf + 9];                        // reference trajectory, step 0, v_x (ConvexMPCLocomotion.cpp:330-406)
  const float sr = 2.0f * (qw * qx + qy * qz), sp = 2.0f * (qw * qy - qx * qz);  // ~ roll, pitch (their sines)
  const float u = (vcmd - vx) + 2.0f * mrx;
  const float score = (u > 0.0f ? u : -0.05f * u) + 0.5f * (fabsf(sr) + fabsf(sp));
  const int b = (int)(score * 48.0f);
  return b < 0 ? 0 : (b > 63 ? 63 : b);
}

// Longest-first dispatch (hmpc_set_dispatch_order): the instances of the batch ordered by the active-set iterations their
// PREVIOUS solve took (status word bits 8-19), most first -- or, when there is no previous solve of this batch (records !=
// nullptr: a cold handle, a new batch size, the first tick of a device-built pipeline), by predicted_cost_bucket of their
// records (keys != nullptr, written by predicted_cost_kernel) -- a counting sort by one workgroup (64 buckets, iterations >= 63
// share the first).  The solve kernels then take instance order[blockIdx.x]: the hardware starts workgroups in index
// order, so the long solves start first and the short ones fill the last, partly occupied round of workgroup slots.  The
// position inside a bucket is decided by atomics and may differ from run to run -- that only changes which slot an
// instance runs in, never its result.
// keys of the predictor, one thread per instance over the whole chip (a single sorting workgroup reading 8 192 scattered records
// twice took ~30 us -- 2 % of a b8192 solve; this launch takes ~3 us and the sort then reads one byte per instance)
__global__ __launch_bounds__(256) void predicted_cost_kernel(const unsigned char *records, int stride, int batch, int h, int nc,
                                                             unsigned char *keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) keys[i] = (unsigned char)predicted_cost_bucket(records + (size_t)i * stride, h, nc);
}

__global__ __launch_bounds__(1024) void dispatch_order_kernel(const uint32_t *status, int batch, int *order, const unsigned char *keys) {
  // per-wave counters: a batch whose iteration counts all fall into two or three buckets (walking) would otherwise send every
  // one of its atomics to the same few LDS words (measured: 15 us for 8 192 instances; ~2 us this way)
  __shared__ int cnt[16][64];
  __shared__ int base[64];
  const int tid = threadIdx.x, wv = tid >> 6;
  auto key = [&](const int i) -> int {
    if (keys) return (int)keys[i];  // (uniform branch) predicted cost buckets, 0..63
    const int it = (int)((status[i] >> 8) & 0xFFFu);
    return it > 63 ? 63 : it;
  };
  cnt[wv][tid & 63] = 0;
  __syncthreads();
  for (int i = tid; i < batch; i += 1024) atomicAdd(&cnt[wv][key(i)], 1);
  __syncthreads();
  if (tid < 64) {
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += cnt[w][tid];
    base[tid] = tot;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 63; b >= 0; --b) {
      const int t = base[b];
      base[b] = run, run += t;
    }
  }
  __syncthreads();
  if (tid < 64) {
    int run = base[tid];
    for (int w = 0; w < 16; ++w) {
      const int t = cnt[w][tid];
      cnt[w][tid] = run, run += t;
    }
  }
  __syncthreads();
  for (int i = tid; i < batch; i += 1024) order[atomicAdd(&cnt[wv][key(i)], 1)] = i;
}

// thread g -> (instance g/12, leg (g%12)/6, row (g%6)): f_ff = -rBody * [GRF; GRM]
__global__ __launch_bounds__(256) void body_wrench_kernel(const float *forces, int batch, int h, const double *rBody,
                                                          double *f_ff) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= 12 * batch) return;
  const int inst = g / 12, c = g % 12, leg = c / 6, r6 = c % 6, i = r6 % 3;
  const float *sol = forces + (size_t)inst * 12 * h;
  const int base = (r6 < 3) ? leg * 3 : leg * 3 + 6;  // GRF = sol[3leg..], GRM = sol[3leg+6..]
  const double *R = rBody + (size_t)inst * 9 + 3 * i;
  const double v0 = (double)sol[base], v1 = (double)sol[base + 1], v2 = (double)sol[base + 2];
  f_ff[g] = ((-R[0]) * v0 + (-R[1]) * v1) + (-R[2]) * v2;
}

// thread g -> (instance g/2, leg g%2): body-frame wrench of that leg, force-moment Jacobian of the leg
// (common/LegController.cpp:108-167, repeated factors named) and tau = J' f (LegController.cpp:57-61)
// ticks != nullptr (the tick pipeline, hmpc_tick_solve_device): rBody and the joint angles are read from the tick structs
// instead; a tick's leg_q is the motor angle when HMPC_TICK_LEG_Q_MOTOR is set and otherwise data[leg].q AFTER the
// LegController's in-place offset -- which is exactly the value the Jacobian's formulas use (LegController.cpp:111-113).
__global__ __launch_bounds__(256) void leg_torque_kernel(const float *forces, int batch, int h, const double *rBody,
                                                         const double *leg_q, double *f_ff, double *tau,
                                                         const hmpc_tick_inputs *ticks) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= 2 * batch) return;
  const int inst = g >> 1, leg = g & 1;
  const float *sol = forces + (size_t)inst * 12 * h;
  const double *R = ticks ? ticks[inst].rBody : rBody + (size_t)inst * 9;
  double f[6];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int base = leg * 3 + 6 * half;
    const double v0 = (double)sol[base], v1 = (double)sol[base + 1], v2 = (double)sol[base + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i) f[3 * half + i] = ((-R[3 * i]) * v0 + (-R[3 * i + 1]) * v1) + (-R[3 * i + 2]) * v2;
  }
  if (f_ff) {
#pragma unroll
    for (int i = 0; i < 6; ++i) f_ff[(size_t)inst * 12 + 6 * leg + i] = f[i];
  }
  const double *q = (ticks ? ticks[inst].leg_q : leg_q + (size_t)inst * 10) + 5 * leg;
  const bool offset_applied = ticks && !(ticks[inst].flags & HMPC_TICK_LEG_Q_MOTOR);
  const double q0 = q[0], q1 = q[1];
  const double q2 = offset_applied ? q[2] : q[2] + 0.3 * 3.14159, q3 = offset_applied ? q[3] : q[3] - 0.6 * 3.14159,
               q4 = offset_applied ? q[4] : q[4] + 0.3 * 3.14159;
  const double side = (leg == 0) ? 1.0 : -1.0;
  double s0, c0, s1, c1, s2, c2, s23, c23, s234, c234;
  det_sincos(q0, s0, c0);
  det_sincos(q1, s1, c1);
  det_sincos(q2, s2, c2);
  det_sincos(q2 + q3, s23, c23);
  det_sincos(q2 + q3 + q4, s234, c234);
  const double Ls = 0.04 * s234 + 0.22 * s23 + 0.22 * s2, Lc = 0.04 * c234 + 0.22 * c23 + 0.22 * c2;
  const double Ls3 = 0.04 * s234 + 0.22 * s23, Lc3 = 0.04 * c234 + 0.22 * c23;
  const double k1 = 0.018 * side + 0.0025, k0 = 0.015 * side;
  const double hip = k0 + c1 * k1 - 1.0 * s1 * Lc;
  const double lat = s1 * k1 + c1 * Lc;
  double J[6][5];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) J[r][c] = 0.0;
  J[0][0] = s0 * (Ls + 0.0135) + c0 * hip;
  J[1][0] = s0 * hip - 1.0 * c0 * (Ls + 0.0135);
  J[5][0] = 1.0;
  J[0][1] = -1.0 * s0 * lat;
  J[1][1] = c0 * lat;
  J[2][1] = s1 * Lc - 1.0 * c1 * k1;
  J[3][1] = c0;
  J[4][1] = s0;
  J[0][2] = s0 * s1 * Ls - 1.0 * c0 * Lc;
  J[1][2] = -1.0 * s0 * Lc - 1.0 * c0 * s1 * Ls;
  J[2][2] = c1 * Ls;
  J[0][3] = s0 * s1 * Ls3 - 1.0 * c0 * Lc3;
  J[1][3] = -1.0 * s0 * Lc3 - 1.0 * c0 * s1 * Ls3;
  J[2][3] = c1 * Ls3;
  J[0][4] = 0.04 * s234 * s0 * s1 - 0.04 * c234 * c0;
  J[1][4] = -0.04 * c234 * s0 - 0.04 * s234 * c0 * s1;
  J[2][4] = 0.04 * s234 * c1;
#pragma unroll
  for (int c = 2; c < 5; ++c) {
    J[3][c] = -c1 * s0;
    J[4][c] = c0 * c1;
    J[5][c] = s1;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc = acc + J[i][j] * f[i];
    tau[(size_t)inst * 10 + 5 * leg + j] = acc;
  }
}

}  // namespace hmpc
