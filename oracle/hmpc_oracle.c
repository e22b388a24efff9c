/*
 * hmpc_oracle.c -- CPU ORACLE (plain C restatement of the reference's convex-MPC QP path).
 * TEST INFRASTRUCTURE ONLY: imported/linked only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  See oracle/hmpc_oracle.h for scope, provenance and the parity status (assembly half pinned
 * against the reference's own source compiled with oracle/mini_eigen; solver half = the reference's own qpOASES).
 *
 * Reference files restated (paths under Hector_ROS_Simulation/hector_control/ConvexMPC/):
 *   SolverMPC.cpp:65-89    euler_to_rotation      -> orc_euler_rate_inverse
 *   SolverMPC.cpp:133-193  c2qp                   -> step 9 of orc_assemble
 *   SolverMPC.cpp:302-331  cross_mat, ct_ss_mats  -> step 7
 *   SolverMPC.cpp:333-342  quat_to_rpy            -> step 3
 *   SolverMPC.cpp:371-577  solve_mpc (assembly)   -> orc_assemble
 *   SolverMPC.cpp:589-697  swing elimination      -> orc_reduce
 *   SolverMPC.cpp:699-732  qpOASES call + scatter -> orc_solve_mpc
 *   RobotState.cpp:9-53    RobotState::set        -> steps 2, 5
 *   convexMPC_interface.cpp:42-110                -> orc_setup_problem / orc_update_problem_data / orc_get_solution
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  No implicit FMA contraction is allowed;
 * every fused multiply-add below is an explicit fmaf()/fma() call.
 */
#define _GNU_SOURCE
#include "hmpc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#if defined(__GNUC__) && defined(__x86_64__)
#define ORC_CLONES __attribute__((target_clones("fma", "default")))
#else
#define ORC_CLONES
#endif

#define ORC_BIG 5e10 /* BIG_NUMBER, SolverMPC.cpp:16 */

/* ------------------------------------------------------------------------------------------------------
 * Deterministic binary64 trigonometry.  Only +,-,*,/,sqrt,rint and explicit fma: every operation is a
 * correctly rounded IEEE-754 operation, so any conforming implementation of the same sequence produces the
 * same bits.  Accuracy is a few 1e-16 relative; callers round the result once to binary32.
 * ------------------------------------------------------------------------------------------------------ */
static const double PIO2_1 = 0x1.921fb54400000p+0;  /* first 33 bits of pi/2 */
static const double PIO2_2 = 0x1.0b4611a600000p-34; /* next 33 bits */
static const double PIO2_3 = 0x1.3198a2e037073p-69; /* remainder */
static const double TWO_OVER_PI = 0x1.45f306dc9c883p-1;
static const double PIO2_D = 0x1.921fb54442d18p+0;
static const double PI_D = 0x1.921fb54442d18p+1;

static const double SC[9] = {/* (-1)^k/(2k+1)!, k=1..9 */
                             -0x1.5555555555555p-3,  0x1.1111111111111p-7,  -0x1.a01a01a01a01ap-13,
                             0x1.71de3a556c734p-19,  -0x1.ae64567f544e4p-26, 0x1.6124613a86d09p-33,
                             -0x1.ae7f3e733b81fp-41, 0x1.952c77030ad4ap-49,  -0x1.2f49b46814157p-57};
static const double CC[9] = {/* (-1)^k/(2k)!, k=1..9 */
                             -0x1.0000000000000p-1,  0x1.5555555555555p-5,  -0x1.6c16c16c16c17p-10,
                             0x1.a01a01a01a01ap-16,  -0x1.27e4fb7789f5cp-22, 0x1.1eed8eff8d898p-29,
                             -0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-45,  -0x1.6827863b97d97p-53};
static const double AC[8] = {/* (-1)^k/(2k+1), k=1..8 */
                             -0x1.5555555555555p-2, 0x1.999999999999ap-3, -0x1.2492492492492p-3, 0x1.c71c71c71c71cp-4,
                             -0x1.745d1745d1746p-4, 0x1.3b13b13b13b14p-4, -0x1.1111111111111p-4, 0x1.e1e1e1e1e1e1ep-5};
static const double ATAN_TAB[9] = {/* atan(i/8) */
                                   0x0.0p+0,
                                   0x1.fd5ba9aac2f6ep-4,
                                   0x1.f5b75f92c80ddp-3,
                                   0x1.6f61941e4def1p-2,
                                   0x1.dac670561bb4fp-2,
                                   0x1.1e00babdefeb4p-1,
                                   0x1.4978fa3269ee1p-1,
                                   0x1.700a7c5784634p-1,
                                   0x1.921fb54442d18p-1};

ORC_CLONES void orc_sincos(double x, double *s, double *c) {
  double k = rint(x * TWO_OVER_PI);
  double r = fma(-k, PIO2_1, x);
  r = fma(-k, PIO2_2, r);
  r = fma(-k, PIO2_3, r);
  double z = r * r;
  double ps = SC[8], pc = CC[8];
  for (int i = 7; i >= 0; --i) {
    ps = fma(ps, z, SC[i]);
    pc = fma(pc, z, CC[i]);
  }
  double sn = fma(r * z, ps, r);
  double cs = fma(z, pc, 1.0);
  long q = (long)k & 3; /* two's complement: correct for negative k too */
  double so, co;
  switch (q) {
    case 0: so = sn, co = cs; break;
    case 1: so = cs, co = -sn; break;
    case 2: so = -sn, co = -cs; break;
    default: so = -cs, co = sn; break;
  }
  *s = so;
  *c = co;
}

/* atan on [0,1] */
static inline double orc_atan01(double t) {
  double fi = rint(t * 8.0);
  int i = (int)fi;
  double cpt = fi * 0.125;
  double u = (t - cpt) / fma(t, cpt, 1.0);
  double z = u * u;
  double p = AC[7];
  for (int k = 6; k >= 0; --k) p = fma(p, z, AC[k]);
  double a = fma(u * z, p, u);
  return ATAN_TAB[i] + a;
}

ORC_CLONES double orc_atan2(double y, double x) {
  double ay = fabs(y), ax = fabs(x);
  double a;
  if (ax == 0.0 && ay == 0.0) {
    a = 0.0;
  } else if (ay <= ax) {
    a = orc_atan01(ay / ax);
  } else {
    a = PIO2_D - orc_atan01(ax / ay);
  }
  if (x < 0.0) a = PI_D - a;
  return (y < 0.0) ? -a : a;
}

ORC_CLONES double orc_asin(double v) { return orc_atan2(v, sqrt((1.0 - v) * (1.0 + v))); }

static inline float sinf_orc(float a) {
  double s, c;
  orc_sincos((double)a, &s, &c);
  return (float)s;
}
static inline float cosf_orc(float a) {
  double s, c;
  orc_sincos((double)a, &s, &c);
  return (float)c;
}

/* ------------------------------------------------------------------------------------------------------ */
orc_qp_t *orc_qp_alloc(int h) {
  orc_qp_t *q = (orc_qp_t *)calloc(1, sizeof(orc_qp_t));
  q->horizon = h;
  q->Phi = (float *)calloc((size_t)h * 13 * 18, sizeof(float));
  q->Apow = (float *)calloc((size_t)(h + 1) * 169, sizeof(float));
  q->H = (float *)calloc((size_t)324 * h * h, sizeof(float));
  q->g = (float *)calloc((size_t)18 * h, sizeof(float));
  q->lb = (float *)calloc((size_t)24 * h, sizeof(float));
  q->ub = (float *)calloc((size_t)24 * h, sizeof(float));
  return q;
}
void orc_qp_free(orc_qp_t *q) {
  if (!q) return;
  free(q->Phi), free(q->Apow), free(q->H), free(q->g), free(q->lb), free(q->ub), free(q);
}
orc_red_t *orc_red_alloc(int h) {
  orc_red_t *r = (orc_red_t *)calloc(1, sizeof(orc_red_t));
  int N = 18 * h, M = 24 * h;
  r->var_ind = (int *)calloc(N, sizeof(int));
  r->con_ind = (int *)calloc(M, sizeof(int));
  r->H = (double *)calloc((size_t)N * N, sizeof(double));
  r->g = (double *)calloc(N, sizeof(double));
  r->A = (double *)calloc((size_t)M * N, sizeof(double));
  r->lb = (double *)calloc(M, sizeof(double));
  r->ub = (double *)calloc(M, sizeof(double));
  return r;
}
void orc_red_free(orc_red_t *r) {
  if (!r) return;
  free(r->var_ind), free(r->con_ind), free(r->H), free(r->g), free(r->A), free(r->lb), free(r->ub), free(r);
}

/* 3x3 inverse by the adjugate, the closed form Eigen uses for fixed 3x3 (SolverMPC.cpp:87, :320):
 * cof(i,j) = m[i+1][j+1]*m[i+2][j+2] - m[i+1][j+2]*m[i+2][j+1] (indices mod 3), det = sum_i cof(i,0)*m[i][0],
 * inv[r][c] = cof(c,r) * (1/det). */
static void inverse3(const float m[9], float inv[9]) {
  float cof[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      float a = m[i1 * 3 + j1] * m[i2 * 3 + j2];
      float b = m[i1 * 3 + j2] * m[i2 * 3 + j1];
      cof[i * 3 + j] = a - b;
    }
  float d0 = cof[0] * m[0], d1 = cof[3] * m[3], d2 = cof[6] * m[6];
  float det = (d0 + d1) + d2;
  float invdet = 1.0f / det;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) inv[r * 3 + c] = cof[c * 3 + r] * invdet;
}

/* Study switch (tests only): 1 = every chain step rounds the product and the sum separately, acc = fl(acc + fl(a*b)),
 * as an SSE2 build without FMA -- the reference's own CMake flags -- does, instead of the contract's fused step.  Used
 * to measure how far the optimal forces move between the two arithmetics (tests/test_oracle.py); never the default. */
static int g_unfused = 0;
void orc_set_unfused_chain(int on) { g_unfused = on; }
static inline float chain_step(float a, float b, float acc) {
  if (g_unfused) {
    volatile float p = a * b; /* volatile: keep the product a separately rounded binary32 value */
    return acc + p;
  }
  return fmaf(a, b, acc);
}

/* k-ascending fmaf chain: out(MxN) = A(MxK) * B(KxN), row-major, acc starts at +0 */
ORC_CLONES static void chain_matmul(const float *A, const float *B, float *out, int M, int K, int N) {
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k) acc = chain_step(A[i * K + k], B[k * N + j], acc);
      out[i * N + j] = acc;
    }
}

/* Study switch (tests only): 1 = take sin/cos/asin/atan2 from the host's libm with the overloads the reference's
 * C++ selects (sinf/cosf/asinf for float arguments, SolverMPC.cpp:74-85, 340, 428-433; double atan2 for the mixed
 * float/double arguments of :339,:341) instead of the contract's deterministic binary64 routines rounded once.  With
 * this and orc_set_unfused_chain(1) the oracle reproduces, bit for bit, the reference's own sources compiled against
 * oracle/mini_eigen (tests/test_reference_source.py); never the default, because libm's float routines are not
 * reproducible on the GPU (glibc's sinf is within 1 ulp, not correctly rounded: the test states how often they differ). */
static int g_libm_trig = 0;
void orc_set_libm_trig(int on) { g_libm_trig = on; }
static inline float sin_f(float a) { return g_libm_trig ? sinf(a) : sinf_orc(a); }
static inline float cos_f(float a) { return g_libm_trig ? cosf(a) : cosf_orc(a); }

/* foot rotation = Rz(q0) Rx(q1) Ry(q2) Ry(q3) Ry(q4) in the expanded sin/cos form of SolverMPC.cpp:428-433, with the
 * common factors named and EVERY operation in the type C++ gives it there: sin()/cos() of a float are float; the
 * literals "1.0" / "-1.0" are double and promote the product they start and everything that product is combined with
 * ("cos(q0)*cos(q2) - 1.0*sin(q0)*sin(q1)*sin(q2)" is a double difference of a float product and a double triple
 * product); each matrix entry is narrowed to float once, by the comma initialiser. */
static void foot_rotation(const float q[5], float Rf[9]) {
  const float s0 = sin_f(q[0]), c0 = cos_f(q[0]);
  const float s1 = sin_f(q[1]), c1 = cos_f(q[1]);
  const float s2 = sin_f(q[2]), c2 = cos_f(q[2]);
  const float s3 = sin_f(q[3]), c3 = cos_f(q[3]);
  const float s4 = sin_f(q[4]), c4 = cos_f(q[4]);
  const float a = c0 * s2 + (c2 * s0) * s1;                                      /* float */
  const double b = (double)(c0 * c2) - (((double)s0 * (double)s1) * (double)s2); /* ... - 1.0*s0*s1*s2 */
  const float d = c2 * s0 + (c0 * s1) * s2;                                      /* float */
  const double e = (double)(s0 * s2) - (((double)c0 * (double)c2) * (double)s1); /* ... - 1.0*c0*c2*s1 */
  const double X = (double)(c3 * a) + (double)s3 * b;            /* cos(q3)*(a) + sin(q3)*(b) */
  const double Y = ((double)s3 * (double)a) - (double)c3 * b;    /* 1.0*sin(q3)*(a) - cos(q3)*(b) */
  const double P = (double)(c3 * d) - ((double)s3) * e;          /* cos(q3)*(d) - 1.0*sin(q3)*(e) */
  const double Q = (double)(s3 * d) + (double)c3 * e;            /* sin(q3)*(d) + cos(q3)*(e) */
  const float q234 = (q[2] + q[3]) + q[4];
  const float s234 = sin_f(q234), c234 = cos_f(q234);
  Rf[0] = (float)((-(double)s4) * X - (double)c4 * Y); /* - 1.0*sin(q4)*(X) - cos(q4)*(Y) */
  Rf[1] = -(c1 * s0);                                  /* -1.0*cos(q1)*sin(q0): an exact double product, rounded once */
  Rf[2] = (float)((double)c4 * X - (double)s4 * Y);    /* cos(q4)*(X) - sin(q4)*(Y) */
  Rf[3] = (float)((double)c4 * P - (double)s4 * Q);    /* cos(q4)*(P) - 1.0*sin(q4)*(Q) */
  Rf[4] = c0 * c1;
  Rf[5] = (float)((double)c4 * Q + (double)s4 * P);    /* cos(q4)*(Q) + sin(q4)*(P) */
  Rf[6] = -(s234 * c1);                                /* -1.0*sin(q2+q3+q4)*cos(q1): exact double product, rounded once */
  Rf[7] = s1;
  Rf[8] = c234 * c1;
}

static int g_dense_chain = 0;
void orc_set_dense_chain(int on) { g_dense_chain = on; }

/* Robot / contact constants: the reference's literals (SolverMPC.cpp:420, 423, 488-490; RobotState.cpp:45) unless a test sets
 * others -- the same struct the product takes (include/hector_mpc.h struct hmpc_params), so that non-default values are checked
 * bitwise as well.  Process-global like the study switches above; NULL restores the defaults. */
static const orc_params_t ORC_DEFAULT_PARAMS = {9.0f, {0.5413f, 0.5200f, 0.0691f}, 2.0f, 0.09f, 0.06f, 9.81f};
static orc_params_t g_params = {9.0f, {0.5413f, 0.5200f, 0.0691f}, 2.0f, 0.09f, 0.06f, 9.81f};
void orc_set_params(const orc_params_t *p) { g_params = p ? *p : ORC_DEFAULT_PARAMS; }
void orc_get_params(orc_params_t *p) { *p = g_params; }

ORC_CLONES void orc_assemble(const orc_update_t *u, const orc_setup_t *st, orc_qp_t *o) {
  const int h = st->horizon;
  o->horizon = h;
  const int nc = (u->nc == 3) ? 3 : 2, U = 6 * nc, C8 = 8 * nc;
  o->nc = nc;

  /* 1. joint angles (SolverMPC.cpp:374-393): float += double offsets, then fmod by 2*PI in double */
  const double PI = 3.14159265359;
  const double PI2 = 2 * PI;
  float qj[10];
  for (int i = 0; i < 10; ++i) qj[i] = u->joint_angles[i];
  qj[2] = (float)((double)qj[2] + 0.3 * PI);
  qj[3] = (float)((double)qj[3] - 0.6 * PI);
  qj[4] = (float)((double)qj[4] + 0.3 * PI);
  qj[7] = (float)((double)qj[7] + 0.3 * PI);
  qj[8] = (float)((double)qj[8] - 0.6 * PI);
  qj[9] = (float)((double)qj[9] + 0.3 * PI);
  for (int i = 0; i < 10; ++i) qj[i] = (float)fmod((double)qj[i], PI2);
  memcpy(o->qj, qj, sizeof qj);

  /* 2. body rotation from the quaternion (RobotState.cpp:17-30; Eigen toRotationMatrix closed form) */
  const float qw = u->q[0], qx = u->q[1], qy = u->q[2], qz = u->q[3];
  float *R = o->R;
  {
    float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
    float twx = tx * qw, twy = ty * qw, twz = tz * qw;
    float txx = tx * qx, txy = ty * qx, txz = tz * qx;
    float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1.0f - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1.0f - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1.0f - (txx + tyy);
  }

  /* 3. roll/pitch/yaw (SolverMPC.cpp:333-342): float products, double literals where the reference has them */
  float rpy[3];
  {
    float t = qw * qy - qx * qz;
    double asd = 2.0 * (double)t;
    if (!(asd < 0.99999)) asd = 0.99999; /* t_min(a, .99999) */
    float as = (float)asd;
    float n0 = 2.0f * (qw * qx + qy * qz);
    double d0 = 1.0 - (double)(2.0f * (qx * qx + qy * qy));
    float n2 = 2.0f * (qw * qz + qx * qy);
    double d2 = 1.0 - (double)(2.0f * (qy * qy + qz * qz));
    if (g_libm_trig) { /* study switch: atan2(float, double) is the double overload, asin(float) the float one */
      rpy[0] = (float)atan2((double)n0, d0);
      rpy[1] = asinf(as);
      rpy[2] = (float)atan2((double)n2, d2);
    } else {
      rpy[0] = (float)orc_atan2((double)n0, d0);
      rpy[1] = (float)orc_asin((double)as);
      rpy[2] = (float)orc_atan2((double)n2, d2);
    }
  }
  memcpy(o->rpy, rpy, sizeof rpy);

  /* 4. inverse Euler-rate map (SolverMPC.cpp:65-89, used at :417) */
  float Rbi[9];
  {
    float cy = cos_f(rpy[2]), sy = sin_f(rpy[2]);
    float cp = cos_f(rpy[1]), sp = sin_f(rpy[1]);
    float Rb[9] = {cy * cp, -sy, 0.0f, sy * cp, cy, 0.0f, -sp, 0.0f, 1.0f};
    inverse3(Rb, Rbi);
  }

  /* 5. x0 and world inertia (SolverMPC.cpp:420-421, RobotState.cpp:45) */
  float *x0 = o->x0;
  for (int i = 0; i < 3; ++i) {
    x0[i] = rpy[i];
    x0[3 + i] = u->p[i];
    x0[6 + i] = u->w[i];
    x0[9 + i] = u->v[i];
  }
  x0[12] = g_params.gravity;                                                       /* 9.81f (SolverMPC.cpp:420) */
  const float Ib[3] = {g_params.inertia[0], g_params.inertia[1], g_params.inertia[2]}; /* 0.5413, 0.5200, 0.0691 (RobotState.cpp:45) */
  float RI[9], Rt[9], Iw[9], Iinv[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      RI[i * 3 + k] = R[i * 3 + k] * Ib[k];
      Rt[k * 3 + i] = R[i * 3 + k];
    }
  chain_matmul(RI, Rt, Iw, 3, 3, 3);
  inverse3(Iw, Iinv);

  /* 7. continuous model (SolverMPC.cpp:312-331), mass 9.0 (:423); r_feet(axis,leg) = r[2*axis+leg] */
  float Act[169], Bct[13 * 18];
  memset(Act, 0, sizeof Act);
  memset(Bct, 0, sizeof Bct);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Act[i * 13 + 6 + j] = Rbi[i * 3 + j];
  for (int i = 0; i < 3; ++i) Act[(3 + i) * 13 + 9 + i] = 1.0f;
  Act[11 * 13 + 12] = -1.0f;
  const float inv_m = 1.0f / g_params.mass; /* mass 9.0 (SolverMPC.cpp:423) */
  for (int leg = 0; leg < nc; ++leg) {
    float r0 = u->r[0 * nc + leg], r1 = u->r[1 * nc + leg], r2 = u->r[2 * nc + leg];
    float cm[9] = {0.0f, -r2, r1, r2, 0.0f, -r0, -r1, r0, 0.0f};
    float blk[9];
    chain_matmul(Iinv, cm, blk, 3, 3, 3);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        Bct[(6 + i) * U + 3 * leg + j] = blk[i * 3 + j];
        Bct[(6 + i) * U + 3 * nc + 3 * leg + j] = Iinv[i * 3 + j];
      }
    for (int i = 0; i < 3; ++i) Bct[(9 + i) * U + 3 * leg + i] = inv_m;
  }

  /* 8. forward-Euler discretisation (SolverMPC.cpp:145-146) */
  const float dt = st->dt;
  float *Acd = o->Acd, *Bcd = o->Bcd;
  for (int i = 0; i < 13; ++i)
    for (int j = 0; j < 13; ++j) Acd[i * 13 + j] = ((i == j) ? 1.0f : 0.0f) + dt * Act[i * 13 + j];
  for (int i = 0; i < 13 * U; ++i) Bcd[i] = dt * Bct[i];

  /* 9. powers by repeated right-multiplication from the identity and Phi_k = Acd^k Bcd (SolverMPC.cpp:148-178);
   *    the reference hard-codes 10 here; the oracle uses the horizon. */
  for (int i = 0; i < 169; ++i) o->Apow[i] = (i % 14 == 0) ? 1.0f : 0.0f;
  for (int k = 0; k < h; ++k) chain_matmul(o->Apow + k * 169, Acd, o->Apow + (k + 1) * 169, 13, 13, 13);
  const int PS = 13 * U; /* floats per Phi_k */
  for (int k = 0; k < h; ++k) chain_matmul(o->Apow + k * 169, Bcd, o->Phi + k * PS, 13, 13, U);

  /* 10. tracking error e = A_qp x0 - X_d (SolverMPC.cpp:457-461, :570) */
  float *e = (float *)malloc(sizeof(float) * 13 * h);
  for (int i = 0; i < h; ++i) {
    float ax[13];
    chain_matmul(o->Apow + (i + 1) * 169, x0, ax, 13, 13, 1);
    for (int s = 0; s < 13; ++s) {
      float xd = (s < 12) ? u->traj[12 * i + s] : 0.0f;
      e[13 * i + s] = ax[s] - xd;
    }
  }

  /* 11. cost (SolverMPC.cpp:450-454, :557-570).  (B' S) is formed first, as Eigen evaluates B'*S*B left to right:
   *     SPhi = fl(w_s * Phi).  H_IJ (I<=J) = 2*fl( chain_k fmaf(SB[k][I], B[k][J]) + alpha*delta_IJ ), mirrored;
   *     g_J = 2 * chain_k fmaf(SB[k][J], e[k]).  Rows of B_qp above the block diagonal are exact zeros, which are
   *     bitwise neutral in a chain started at +0, so the chain may start at k = 13*max(step_I, step_J). */
  float W[13];
  for (int s = 0; s < 12; ++s) W[s] = u->weights[s];
  W[12] = 0.0f;
  float *SPhi = (float *)malloc(sizeof(float) * h * PS);
  for (int k = 0; k < h; ++k)
    for (int s = 0; s < 13; ++s)
      for (int c = 0; c < U; ++c) SPhi[k * PS + s * U + c] = W[s] * o->Phi[k * PS + s * U + c];
  const int N = U * h;
  for (int I = 0; I < N; ++I) {
    int a = I / U, r = I % U;
    for (int J = I; J < N; ++J) {
      int b = J / U, c = J % U;
      float acc = 0.0f;
      int i0 = g_dense_chain ? 0 : b; /* b >= a because J >= I */
      for (int i = i0; i < h; ++i) {
        for (int s = 0; s < 13; ++s) {
          float sb = (i >= a) ? SPhi[(i - a) * PS + s * U + r] : 0.0f;
          float bb = (i >= b) ? o->Phi[(i - b) * PS + s * U + c] : 0.0f;
          acc = chain_step(sb, bb, acc);
        }
      }
      float hv = 2.0f * (acc + ((I == J) ? u->Alpha_K[r] : 0.0f));
      o->H[(size_t)I * N + J] = hv;
      o->H[(size_t)J * N + I] = hv;
    }
  }
  for (int J = 0; J < N; ++J) {
    int b = J / U, c = J % U;
    float acc = 0.0f;
    int i0 = g_dense_chain ? 0 : b;
    for (int i = i0; i < h; ++i)
      for (int s = 0; s < 13; ++s) {
        float sb = (i >= b) ? SPhi[(i - b) * PS + s * U + c] : 0.0f;
        acc = chain_step(sb, e[13 * i + s], acc);
      }
    o->g[J] = 2.0f * acc;
  }
  free(SPhi);
  free(e);

  /* 12. foot rotations and constraint block (SolverMPC.cpp:426-433, :488-548); mu, lt, lh hard-coded there */
  foot_rotation(qj, o->Rfoot[0]);
  foot_rotation(qj + 5, o->Rfoot[1]);
  if (nc == 3) memcpy(o->Rfoot[2], u->Rhand, sizeof(float) * 9); /* extension: the hand contact frame is an input */
  const float mu = g_params.mu, lt = g_params.lt, lh = g_params.lh; /* 2.0, 0.09, 0.06 (SolverMPC.cpp:488-490) */
  float *Fc = o->Fc;
  memset(Fc, 0, sizeof(float) * C8 * U);
  for (int leg = 0; leg < nc; ++leg) {
    const float *Rf = o->Rfoot[leg];
    float col0[3], col1[3], vlt[3], vlh[3], t0[3], t1[3], flt[3], flh[3];
    for (int k = 0; k < 3; ++k) {
      col0[k] = Rf[k * 3 + 0];
      col1[k] = Rf[k * 3 + 1];
      vlt[k] = -lt * Rf[k * 3 + 2];
      vlh[k] = -lh * Rf[k * 3 + 2];
    }
    chain_matmul(col0, Rt, t0, 1, 3, 3); /* (e_x' Rfoot') R'  */
    chain_matmul(col1, Rt, t1, 1, 3, 3); /* (e_y' Rfoot') R'  */
    chain_matmul(vlt, Rt, flt, 1, 3, 3); /* (-lt e_z' Rfoot') R' */
    chain_matmul(vlh, Rt, flh, 1, 3, 3);
    float *row = Fc + (8 * leg) * U;
    const int cf = 3 * leg, cmo = 3 * nc + 3 * leg;
    row[0 * U + cf + 0] = -mu, row[0 * U + cf + 2] = 1.0f;
    row[1 * U + cf + 0] = mu, row[1 * U + cf + 2] = 1.0f;
    row[2 * U + cf + 1] = -mu, row[2 * U + cf + 2] = 1.0f;
    row[3 * U + cf + 1] = mu, row[3 * U + cf + 2] = 1.0f;
    for (int j = 0; j < 3; ++j) {
      row[4 * U + cmo + j] = t0[j];
      row[5 * U + cf + j] = flt[j];
      row[5 * U + cmo + j] = t1[j];
      row[6 * U + cf + j] = flh[j];
      row[6 * U + cmo + j] = (leg != 1) ? -t1[j] : t1[j]; /* SolverMPC.cpp:526 vs :546; the hand follows the left foot */
    }
    row[7 * U + cf + 2] = 2.0f;
  }

  /* 13. bounds (SolverMPC.cpp:466-482) */
  const float big = (float)ORC_BIG;
  for (int i = 0; i < h; ++i)
    for (int leg = 0; leg < nc; ++leg) {
      float *lb = o->lb + C8 * i + 8 * leg, *ub = o->ub + C8 * i + 8 * leg;
      for (int j = 0; j < 4; ++j) lb[j] = 0.0f, ub[j] = big;
      lb[4] = 0.0f, ub[4] = 0.01f;
      lb[5] = -big, ub[5] = 0.0f;
      lb[6] = -big, ub[6] = 0.0f;
      lb[7] = 0.0f, ub[7] = ((leg == 2) ? u->f_max_hand : st->f_max) * (float)u->gait[nc * i + leg];
    }
}

static int near_zero(float a) { return (a < 0.0001 && a > -.0001); }

/* SolverMPC.cpp:589-697.  A row whose two bounds are both ~0 (only the Fz row of a swing leg-step can be) and whose
 * coefficient ~2 sits in column j removes that leg-step's force (j-2..j) and moment (j+4..j+6) variables and its 8
 * constraint rows; the survivors are gathered in ascending original order. */
void orc_reduce(const orc_qp_t *qp, orc_red_t *red) {
  const int nc = qp->nc, U = 6 * nc, C8 = 8 * nc;
  const int h = qp->horizon, N = U * h, M = C8 * h;
  char *ve = (char *)calloc(N, 1), *ce = (char *)calloc(M, 1);
  for (int i = 0; i < M; ++i) {
    if (!(near_zero((float)(double)qp->lb[i]) && near_zero((float)(double)qp->ub[i]))) continue;
    int step = i / C8;
    const float *crow = qp->Fc + (i % C8) * U;
    for (int jj = 0; jj < U; ++jj) {
      if (!near_zero(crow[jj] - 2)) continue;
      /* column jj is Fz of contact c = jj/3: its force (jj-2..jj), its moment (3 nc further) and its 8 rows go.
       * For nc = 2 this is the reference's index arithmetic (j+4..j+6, rows (j+4)/6*8-1 resp. (j+1)/6*8+7 and the 7 before). */
      int j = U * step + jj, c = jj / 3;
      int last = C8 * step + 8 * c + 7;
      for (int k = 0; k < 3; ++k) ve[j - k] = 1, ve[j - 2 + 3 * nc + k] = 1;
      for (int k = 0; k < 8; ++k) ce[last - k] = 1;
    }
  }
  int n = 0, m = 0;
  for (int i = 0; i < N; ++i)
    if (!ve[i]) red->var_ind[n++] = i;
  for (int i = 0; i < M; ++i)
    if (!ce[i]) red->con_ind[m++] = i;
  red->n = n;
  red->m = m;
  for (int i = 0; i < n; ++i) {
    int oa = red->var_ind[i];
    red->g[i] = (double)qp->g[oa];
    for (int j = 0; j < n; ++j) red->H[(size_t)i * n + j] = (double)qp->H[(size_t)oa * N + red->var_ind[j]];
  }
  for (int c = 0; c < m; ++c) {
    int oc = red->con_ind[c];
    int step = oc / C8;
    const float *crow = qp->Fc + (oc % C8) * U;
    for (int j = 0; j < n; ++j) {
      int ov = red->var_ind[j];
      red->A[(size_t)c * n + j] = (ov / U == step) ? (double)crow[ov % U] : 0.0; /* fmat is block diagonal (:552-555) */
    }
    red->lb[c] = (double)qp->lb[oc];
    red->ub[c] = (double)qp->ub[oc];
  }
  free(ve);
  free(ce);
}

int orc_solve_reduced(const orc_red_t *red, double *x_red, double *y_red, int *nwsr, double *obj) {
  return ref_qpoases_solve(red->n, red->m, red->H, red->g, red->A, red->lb, red->ub, 500, x_red, y_red, obj, nwsr);
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int solve_with(const orc_update_t *u, const orc_setup_t *s, orc_qp_t *qp, orc_red_t *red, double *q_soln,
                      int *nwsr, double *obj, double *t_asm, double *t_sol) {
  const int N = 6 * ((u->nc == 3) ? 3 : 2) * s->horizon;
  double t0 = now_s();
  orc_assemble(u, s, qp);
  orc_reduce(qp, red);
  double t1 = now_s();
  double *x = (double *)calloc(N, sizeof(double));
  int rv = orc_solve_reduced(red, x, NULL, nwsr, obj);
  /* scatter (SolverMPC.cpp:720-732): eliminated variables are reported as exactly 0 */
  for (int i = 0; i < N; ++i) q_soln[i] = 0.0;
  for (int i = 0; i < red->n; ++i) q_soln[red->var_ind[i]] = x[i];
  free(x);
  double t2 = now_s();
  if (t_asm) *t_asm += t1 - t0;
  if (t_sol) *t_sol += t2 - t1;
  return rv;
}

int orc_solve_mpc(const orc_update_t *u, const orc_setup_t *s, double *q_soln, int *nwsr, double *obj, int *n_red,
                  int *m_red) {
  orc_qp_t *qp = orc_qp_alloc(s->horizon);
  orc_red_t *red = orc_red_alloc(s->horizon);
  int rv = solve_with(u, s, qp, red, q_soln, nwsr, obj, NULL, NULL);
  if (n_red) *n_red = red->n;
  if (m_red) *m_red = red->m;
  orc_qp_free(qp);
  orc_red_free(red);
  return rv;
}

/* ---- legacy interface (convexMPC_interface.cpp:42-110): globals, blocking solve, get_solution 0 before first solve */
static orc_setup_t g_setup;
static orc_update_t g_update;
static double *g_q_soln = NULL;
static int g_has_solved = 0;

void orc_setup_problem(double dt, int horizon, double mu, double f_max) {
  g_setup.horizon = horizon;
  g_setup.f_max = (float)f_max;
  g_setup.mu = (float)mu;
  g_setup.dt = (float)dt;
  free(g_q_soln); /* resize_qp_mats re-allocates every tick (SolverMPC.cpp:196-299) */
  g_q_soln = (double *)calloc((size_t)12 * horizon, sizeof(double));
}

void orc_update_problem_data(double *p, double *v, double *q, double *w, double *r, double *joint_angles, double yaw,
                             double *weights, double *state_trajectory, double *Alpha_K, int *gait) {
  const int h = g_setup.horizon;
  g_update.nc = 2;
  for (int i = 0; i < 3; ++i) g_update.p[i] = (float)p[i], g_update.v[i] = (float)v[i], g_update.w[i] = (float)w[i];
  for (int i = 0; i < 4; ++i) g_update.q[i] = (float)q[i];
  for (int i = 0; i < 6; ++i) g_update.r[i] = (float)r[i];
  for (int i = 0; i < 10; ++i) g_update.joint_angles[i] = (float)joint_angles[i];
  g_update.yaw = (float)yaw;
  for (int i = 0; i < 12; ++i) g_update.weights[i] = (float)weights[i], g_update.Alpha_K[i] = (float)Alpha_K[i];
  for (int i = 0; i < 12 * h; ++i) g_update.traj[i] = (float)state_trajectory[i];
  for (int i = 0; i < 2 * h; ++i) g_update.gait[i] = (unsigned char)gait[i];
  int nwsr;
  double obj;
  orc_solve_mpc(&g_update, &g_setup, g_q_soln, &nwsr, &obj, NULL, NULL);
  g_has_solved = 1;
}

double orc_get_solution(int index) {
  if (!g_has_solved) return 0.0;
  return g_q_soln[index];
}

/* ---- batched CPU baseline over packed records ---- */
void orc_unpack_record(const unsigned char *rec, int horizon, orc_update_t *u) {
  const float *f = (const float *)rec;
  memcpy(u->p, f + 0, 12), memcpy(u->v, f + 3, 12), memcpy(u->q, f + 6, 16), memcpy(u->w, f + 10, 12);
  memcpy(u->r, f + 13, 24), memcpy(u->joint_angles, f + 19, 40);
  u->yaw = f[29];
  memcpy(u->weights, f + 30, 48), memcpy(u->Alpha_K, f + 42, 48);
  memcpy(u->traj, f + 54, sizeof(float) * 12 * horizon);
  memcpy(u->gait, rec + 4 * (54 + 12 * horizon), 2 * horizon);
  u->nc = 2;
}

/* extension record (nc = 3): p3 v3 q4 w3 r9 joint10 yaw weights12 alpha18 Rhand9 f_max_hand traj12h | gait 3h bytes */
void orc_unpack_record3(const unsigned char *rec, int horizon, orc_update_t *u) {
  const float *f = (const float *)rec;
  memcpy(u->p, f + 0, 12), memcpy(u->v, f + 3, 12), memcpy(u->q, f + 6, 16), memcpy(u->w, f + 10, 12);
  memcpy(u->r, f + 13, 36), memcpy(u->joint_angles, f + 22, 40);
  u->yaw = f[32];
  memcpy(u->weights, f + 33, 48), memcpy(u->Alpha_K, f + 45, 72), memcpy(u->Rhand, f + 63, 36);
  u->f_max_hand = f[72];
  memcpy(u->traj, f + 73, sizeof(float) * 12 * horizon);
  memcpy(u->gait, rec + 4 * (73 + 12 * horizon), 3 * horizon);
  u->nc = 3;
}

static int g_records_nc = 2;
void orc_set_records_nc(int nc) { g_records_nc = (nc == 3) ? 3 : 2; }

int orc_solve_records(const unsigned char *records, int stride, int first, int count, int horizon, float dt,
                      float f_max, double *q_soln, int *nwsr_out, double *obj_out, double *t_assemble,
                      double *t_solve) {
  return orc_solve_records_ex(records, stride, first, count, horizon, dt, f_max, q_soln, nwsr_out, obj_out, t_assemble,
                              t_solve, 0);
}

/* ... with qpOASES' return value per instance (0 = solved; SolverMPC.cpp:712-715 only prints "failed to solve!"), so that a
 * comparison can leave out the instances the reference itself did not solve. */
int orc_solve_records_ex(const unsigned char *records, int stride, int first, int count, int horizon, float dt,
                         float f_max, double *q_soln, int *nwsr_out, double *obj_out, double *t_assemble,
                         double *t_solve, int *rv_out) {
  orc_setup_t s = {dt, 0.25f, f_max, horizon};
  orc_qp_t *qp = orc_qp_alloc(horizon);
  orc_red_t *red = orc_red_alloc(horizon);
  orc_update_t u;
  memset(&u, 0, sizeof u);
  const int nc = g_records_nc, NU = 6 * nc;
  int bad = 0;
  for (int k = 0; k < count; ++k) {
    if (nc == 3) orc_unpack_record3(records + (size_t)(first + k) * stride, horizon, &u);
    else orc_unpack_record(records + (size_t)(first + k) * stride, horizon, &u);
    int nwsr = 0;
    double obj = 0;
    int rv = solve_with(&u, &s, qp, red, q_soln + (size_t)k * NU * horizon, &nwsr, &obj, t_assemble, t_solve);
    if (nwsr_out) nwsr_out[k] = nwsr;
    if (obj_out) obj_out[k] = obj;
    if (rv_out) rv_out[k] = rv;
    if (rv != 0) ++bad;
  }
  orc_qp_free(qp);
  orc_red_free(red);
  return bad;
}


/* ================= SURVEY.md section 8(f): f1 input builder, f2 gait table, f3 wrench consumer ================= */

/* GaitGenerator.cpp:85-103 */
void orc_mpc_gait(int n, const int offsets[2], const int durations[2], int iteration, int *table) {
  for (int i = 0; i < n; ++i) {
    int iter = (i + iteration) % n;
    for (int j = 0; j < 2; ++j) {
      int progress = iter - offsets[j];
      if (progress < 0) progress += n;
      table[2 * i + j] = (progress < durations[j]) ? 1 : 0;
    }
  }
}

/* ConvexMPCLocomotion.cpp:283-406, then the double->float / int->u8 narrowing of convexMPC_interface.cpp:83-103 */
void orc_build_record(const orc_tick_t *t, int h, double dtMPC, unsigned char *record, double wpd_out[2]) {
  double q[10];
  for (int i = 0; i < 10; ++i) q[i] = t->leg_q[i];
  if (t->flags & 1) { /* raw motor angles: common/LegController.cpp:111-113 mutates data[leg].q before the MPC reads it */
    for (int l = 0; l < 2; ++l) {
      q[5 * l + 2] = q[5 * l + 2] + 0.3 * 3.14159;
      q[5 * l + 3] = q[5 * l + 3] - 0.6 * 3.14159;
      q[5 * l + 4] = q[5 * l + 4] + 0.3 * 3.14159;
    }
  }
  const double PI = 3.14159265359;
  q[2] += 0.3 * PI, q[3] -= 0.6 * PI, q[4] += 0.3 * PI;
  q[7] += 0.3 * PI, q[8] -= 0.6 * PI, q[9] += 0.3 * PI;
  const double PI2 = 2 * PI;
  for (int i = 0; i < 10; ++i) q[i] = fmod(q[i], PI2);
  double r[6];
  for (int i = 0; i < 6; ++i) r[i] = t->pFoot[3 * (i % 2) + i / 2] - t->position[i / 2];
  const double Q[12] = {100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1};
  const double Alpha[12] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
  const double yaw = t->rpy[2];
  /* v_des_world = rBody' * (vx, vy, 0), three-term dot products in index order */
  double vdw[3];
  for (int i = 0; i < 3; ++i)
    vdw[i] = (t->rBody[0 * 3 + i] * t->v_des_robot[0] + t->rBody[1 * 3 + i] * t->v_des_robot[1]) + t->rBody[2 * 3 + i] * 0.0;
  const double max_pos_error = .05;
  double xStart = t->world_position_desired[0], yStart = t->world_position_desired[1];
  const double *p = t->position;
  if (xStart - p[0] > max_pos_error) xStart = p[0] + max_pos_error;
  if (p[0] - xStart > max_pos_error) xStart = p[0] - max_pos_error;
  if (yStart - p[1] > max_pos_error) yStart = p[1] + max_pos_error;
  if (p[1] - yStart > max_pos_error) yStart = p[1] - max_pos_error;
  wpd_out[0] = xStart, wpd_out[1] = yStart;
  double trajInitial[12] = {t->roll_des, t->pitch_des, 0.0, xStart, yStart, 0.55, 0, 0, t->yaw_rate_des, vdw[0], vdw[1], 0};
  double *trajAll = (double *)malloc(sizeof(double) * 12 * h);
  for (int i = 0; i < h; ++i) {
    for (int j = 0; j < 12; ++j) trajAll[12 * i + j] = trajInitial[j];
    if (i == 0) {
      /* the reference writes indices 0..5 of the WHOLE array here (trajAll[0..5]), i.e. step 0 */
      trajAll[0] = t->rpy[0], trajAll[1] = t->rpy[1], trajAll[2] = t->rpy[2];
      trajAll[3] = p[0], trajAll[4] = p[1], trajAll[5] = p[2];
    } else {
      if (vdw[0] == 0) trajAll[12 * i + 3] = trajInitial[3] + i * dtMPC * vdw[0];
      else trajAll[12 * i + 3] = p[0] + i * dtMPC * vdw[0];
      if (vdw[1] == 0) trajAll[12 * i + 4] = trajInitial[4] + i * dtMPC * vdw[1];
      else trajAll[12 * i + 4] = p[1] + i * dtMPC * vdw[1];
      if (t->yaw_rate_des == 0) trajAll[12 * i + 2] = trajInitial[2];
      else trajAll[12 * i + 2] = yaw + i * dtMPC * t->yaw_rate_des;
    }
  }
  int *table = (int *)malloc(sizeof(int) * 2 * h);
  orc_mpc_gait(h, t->gait_offsets, t->gait_durations, t->gait_iteration, table);
  /* narrowing + packing */
  const int stride = ((54 + 12 * h) * 4 + 2 * h + 15) / 16 * 16;
  memset(record, 0, stride);
  float *f = (float *)record;
  for (int i = 0; i < 3; ++i) f[i] = (float)p[i], f[3 + i] = (float)t->vWorld[i], f[10 + i] = (float)t->omegaWorld[i];
  for (int i = 0; i < 4; ++i) f[6 + i] = (float)t->orientation[i];
  for (int i = 0; i < 6; ++i) f[13 + i] = (float)r[i];
  for (int i = 0; i < 10; ++i) f[19 + i] = (float)q[i];
  f[29] = (float)yaw;
  for (int i = 0; i < 12; ++i) f[30 + i] = (float)Q[i], f[42 + i] = (float)Alpha[i];
  for (int i = 0; i < 12 * h; ++i) f[54 + i] = (float)trajAll[i];
  unsigned char *g = record + 4 * (54 + 12 * h);
  for (int i = 0; i < 2 * h; ++i) g[i] = (unsigned char)table[i];
  free(trajAll);
  free(table);
}

/* ConvexMPCLocomotion.cpp:419-440: GRF_R = -rBody*GRF, GRM_R = -rBody*GRM, f = [GRF_R; GRM_R] per leg */
void orc_body_wrench(const double *sol, const double *rBody, double *f_ff) {
  for (int leg = 0; leg < 2; ++leg) {
    double GRF[3], GRM[3];
    for (int axis = 0; axis < 3; ++axis) GRF[axis] = sol[leg * 3 + axis], GRM[axis] = sol[leg * 3 + axis + 6];
    for (int i = 0; i < 3; ++i) {
      f_ff[6 * leg + i] = ((-rBody[3 * i + 0]) * GRF[0] + (-rBody[3 * i + 1]) * GRF[1]) + (-rBody[3 * i + 2]) * GRF[2];
      f_ff[6 * leg + 3 + i] = ((-rBody[3 * i + 0]) * GRM[0] + (-rBody[3 * i + 1]) * GRM[1]) + (-rBody[3 * i + 2]) * GRM[2];
    }
  }
}


/* LegController.cpp:108-167 computeLegJacobianAndPosition (force-moment Jacobian only) and :57-61 legtau = J' f.
 * The reference's expression is written here with its repeated factors named (Ls = 0.04 s234 + 0.22 s23 + 0.22 s2, ...);
 * sin/cos through the deterministic routines above (binary64, not narrowed).  Note the 3.14159 (not 3.14159265359)
 * in the offsets, as in the reference. */
void orc_leg_jacobian(const double q_in[5], int leg, double J[30] /* row-major 6x5 */) {
  const double q0 = q_in[0], q1 = q_in[1];
  const double q2 = q_in[2] + 0.3 * 3.14159, q3 = q_in[3] - 0.6 * 3.14159, q4 = q_in[4] + 0.3 * 3.14159;
  const double side = (leg == 0) ? 1.0 : -1.0;
  double s0, c0, s1, c1, s2, c2, s23, c23, s234, c234;
  if (g_libm_trig) { /* study switch: libm's double sin/cos, what the reference's build calls (LegController.cpp:130-165) */
    /* g++ -O3 turns SOME of the reference's sin(x)/cos(x) pairs into sincos(x) calls (its binary here mixes sin, cos and
     * sincos), and glibc's sincos differs from its sin/cos in the last bit on ~0.2 % of arguments: which entry gets which is
     * a compiler decision the reference does not pin.  Plain sin/cos is the closest single choice (measured: 7 of 60 000
     * Jacobian entries differ from the reference build by one ulp; tests/test_caller_reference.py). */
    s0 = sin(q0), c0 = cos(q0), s1 = sin(q1), c1 = cos(q1), s2 = sin(q2), c2 = cos(q2);
    s23 = sin(q2 + q3), c23 = cos(q2 + q3), s234 = sin(q2 + q3 + q4), c234 = cos(q2 + q3 + q4);
  } else {
    orc_sincos(q0, &s0, &c0);
    orc_sincos(q1, &s1, &c1);
    orc_sincos(q2, &s2, &c2);
    orc_sincos(q2 + q3, &s23, &c23);
    orc_sincos(q2 + q3 + q4, &s234, &c234);
  }
  const double Ls = 0.04 * s234 + 0.22 * s23 + 0.22 * s2, Lc = 0.04 * c234 + 0.22 * c23 + 0.22 * c2;
  const double Ls3 = 0.04 * s234 + 0.22 * s23, Lc3 = 0.04 * c234 + 0.22 * c23;
  const double k1 = 0.018 * side + 0.0025, k0 = 0.015 * side;
  const double hip = k0 + c1 * k1 - 1.0 * s1 * Lc;
  const double lat = s1 * k1 + c1 * Lc;
  memset(J, 0, sizeof(double) * 30);
#define JJ(r, c) J[(r) * 5 + (c)]
  JJ(0, 0) = s0 * (Ls + 0.0135) + c0 * hip;
  JJ(1, 0) = s0 * hip - 1.0 * c0 * (Ls + 0.0135);
  JJ(5, 0) = 1.0;
  JJ(0, 1) = -1.0 * s0 * lat;
  JJ(1, 1) = c0 * lat;
  JJ(2, 1) = s1 * Lc - 1.0 * c1 * k1;
  JJ(3, 1) = c0;
  JJ(4, 1) = s0;
  JJ(0, 2) = s0 * s1 * Ls - 1.0 * c0 * Lc;
  JJ(1, 2) = -1.0 * s0 * Lc - 1.0 * c0 * s1 * Ls;
  JJ(2, 2) = c1 * Ls;
  JJ(0, 3) = s0 * s1 * Ls3 - 1.0 * c0 * Lc3;
  JJ(1, 3) = -1.0 * s0 * Lc3 - 1.0 * c0 * s1 * Ls3;
  JJ(2, 3) = c1 * Ls3;
  JJ(0, 4) = 0.04 * s234 * s0 * s1 - 0.04 * c234 * c0;
  JJ(1, 4) = -0.04 * c234 * s0 - 0.04 * s234 * c0 * s1;
  JJ(2, 4) = 0.04 * s234 * c1;
  for (int c = 2; c < 5; ++c) {
    JJ(3, c) = -c1 * s0;
    JJ(4, c) = c0 * c1;
    JJ(5, c) = s1;
  }
#undef JJ
}

/* tau[leg][j] = sum_i J(i,j) f_ff[leg][i], i ascending (LegController.cpp:60) */
void orc_leg_torques(const double *f_ff /*[2][6]*/, const double *leg_q /*[10]*/, double *tau /*[2][5]*/) {
  for (int leg = 0; leg < 2; ++leg) {
    double J[30];
    orc_leg_jacobian(leg_q + 5 * leg, leg, J);
    for (int j = 0; j < 5; ++j) {
      double acc = 0.0;
      for (int i = 0; i < 6; ++i) acc = acc + J[i * 5 + j] * f_ff[6 * leg + i];
      tau[5 * leg + j] = acc;
    }
  }
}
