// hmpc_sweep_mfma.h -- experimental alternative to the scalar symmetric sweeps of hmpc_kernel.h (phase S): the same
// inversion M = H^-1 as a panel-blocked sweep whose trailing update runs on the fp64 matrix cores.
// Built only with -DHMPC_MFMA_SWEEP=1.  Numerically equivalent (every GPU parity test passes with it); measured on MI355X
// at the same speed as the scalar sweeps for 120 variables and 4-5 % slower for 60 / 180: the
// per-panel chain  publish -> 4x4 inverse -> MFMA -> patch  is as long as the four scalar pivots it replaces.
#pragma once

namespace hmpc {

// Blocked form of the same sweeps on the fp64 matrix cores.  During the inversion the symmetric matrix lives in the
// accumulator layout of v_mfma_f64_16x16x4_f64: 16x16 tiles (I <= J) of the matrix padded with the identity to a
// multiple of 16 (diagonal tiles kept full).  Wave w owns the tile columns w and NTL-1-w (NTL+1 tiles each way round):
// slot s <= NTL-1-w holds tile (s, NTL-1-w), slot s >= NTL-w holds tile (NTL-s, w) -- so for a given tile row the slots
// involved are compile-time constants and only the roles depend (uniformly) on the wave.
// Panel p = pivots 4p..4p+3 (tile row Ik = p/4, register / lane group g = p%4), D = A(K,K):
//     A(K,K) <- -D^-1,   A(K,J) <- D^-1 A(K,J),   A(I,J) <- A(I,J) - A(I,K) D^-1 A(K,J)      (I, J outside K)
//  (a) the panel columns P = A(:,K) go to LDS: from the tiles of tile column Ik (one wave) and, by symmetry, from
//      register g of the tiles of tile row Ik (at most two per wave);
//  (b) threads i < 16 ntl invert the 4x4 D in registers (2x2 block formulas) and form row i of  Tn = -(P D^-1)
//      (zero for the pivot rows) and of  Fx = the values the pivot rows / columns take (P D^-1; -D^-1 inside K);
//  (c) one MFMA per tile: A(I,J) += Tn(I) P(J)'; then pivot rows and pivot columns are overwritten from Fx.
// Two barriers per panel of four pivots.  Tiles of pure padding see Tn = 0 and stay the identity.
template <int NMAX, int NT, class SM>
__device__ __forceinline__ void sweep_mfma(SM &S, double (&a)[GS][GS], const int n, const int ng, const int tid, const int wv,
                                           const int ln, const bool owner, const int e1, const int i0, const int j0,
                                           const bool diag) {
  constexpr int NW = SM::NW;
  auto &A = S.u.a;
  auto &Q = S.u.s;
  {
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int NTL = (NMAX + 15) / 16, NPAD = 16 * NTL, NWS = NTL / 2, TPW = NTL + 1;
    static_assert(NTL % 2 == 0 && NWS <= NW && NPAD <= NT, "tile columns are dealt to the waves in pairs");
    static_assert(16 * NPAD <= SM::QMAX * (SM::QMAX + 1) / 2, "panel scratch / row strip live in the (idle) E storage");
    double *Pb = Q.Ep;              // [2][NPAD][4] panel columns, double buffered by panel parity
    double *Tn = Q.Ep + 8 * NPAD;   // [NPAD][4]
    double *Fx = Q.Ep + 12 * NPAD;  // [NPAD][4]
    const int l15 = ln & 15, kq = ln >> 4;
    const int ntl = (n + 15) >> 4;  // tile rows that hold matrix entries
    const bool sw = wv < NWS;       // this wave holds tiles
    const int c1 = wv, c2 = NTL - 1 - wv;
    const bool cl0 = (l15 >> 2) == 0, cl1 = (l15 >> 2) == 1, cl2 = (l15 >> 2) == 2;
    int offa[TPW], offb[TPW];
    d4 acc[TPW];
#pragma unroll
    for (int s2 = 0; s2 < TPW; ++s2) {
      const bool second = s2 <= c2;
      const int I = second ? s2 : NTL - s2, J = second ? c2 : c1;
      offa[s2] = (16 * I + l15) * 4 + kq;  // this lane's operand in Tn (rows of tile row I) ...
      offb[s2] = (16 * J + l15) * 4 + kq;  // ... and in P / Fx (rows of tile column J)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = 16 * I + kq + 4 * v, j = 16 * J + l15;
        double val = (i == j) ? 1.0 : 0.0;
        if (sw && i < n && j < n) {
          const int oi = S.s2o[i], oj = S.s2o[j];
          val = (double)A.Hs[hs_index<NMAX>(oi < oj ? oi : oj, oi < oj ? oj : oi)];
        }
        acc[s2][v] = val;
      }
    }
    __syncthreads();  // every tile is loaded before the solver state (which aliases the staging area) is written
    for (int t = tid; t < 16 * NPAD; t += NT) Q.Ep[t] = 0.0;
    __syncthreads();
    auto pick = [&](const d4 &x, int g) -> double { return g == 0 ? x[0] : (g == 1 ? x[1] : (g == 2 ? x[2] : x[3])); };
    auto put = [&](d4 &x, int g, double val) {
      x[0] = g == 0 ? val : x[0], x[1] = g == 1 ? val : x[1], x[2] = g == 2 ? val : x[2], x[3] = g == 3 ? val : x[3];
    };
#pragma unroll
    for (int Ik = 0; Ik < NTL; ++Ik) {
      if (16 * Ik < n) {  // uniform
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
          const int p = 4 * Ik + g;
          if (4 * p >= n) break;  // uniform
          double *Pc = Pb + (p & 1) * 4 * NPAD;
          const bool mycols = g == 0 ? cl0 : (g == 1 ? cl1 : (g == 2 ? cl2 : (l15 >> 2) == 3));
          const int pcol = 4 * kq + (l15 & 3);  // + 64 I + 16 v: entry (16 I + kq + 4 v, l15 & 3) of P / Fx
          // (a)
          if (sw) {
            if (c2 == Ik) {
              if (mycols) {
#pragma unroll
                for (int I = 0; I <= Ik; ++I)
#pragma unroll
                  for (int v = 0; v < 4; ++v) Pc[64 * I + 16 * v + pcol] = acc[I][v];
              }
            } else if (Ik < c2) {
              Pc[offb[Ik]] = pick(acc[Ik], g);
            }
            if (c1 == Ik) {
              if (mycols) {
#pragma unroll
                for (int I = 0; I <= Ik; ++I)
#pragma unroll
                  for (int v = 0; v < 4; ++v) Pc[64 * I + 16 * v + pcol] = acc[NTL - I][v];
              }
            } else if (Ik < c1) {
              Pc[offb[NTL - Ik]] = pick(acc[NTL - Ik], g);
            }
          }
          __syncthreads();
          // (b)
          if (tid < 16 * ntl) {
            double D[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double2 lo = *reinterpret_cast<const double2 *>(Pc + (4 * p + r) * 4);
              const double2 hi = *reinterpret_cast<const double2 *>(Pc + (4 * p + r) * 4 + 2);
              D[r][0] = lo.x, D[r][1] = lo.y, D[r][2] = hi.x, D[r][3] = hi.y;
            }
            // D^-1 through its 2x2 blocks [[A, B], [B', C]]: two reciprocals on the critical path instead of four.
            // X = A^-1, Y = X B, S = C - B'Y, Z = S^-1:   D^-1 = [[X + Y Z Y', -Y Z], [-Z Y', Z]].  D is overwritten
            // with -D^-1 (the sign the sweeps leave).
            {
              auto rcp = [](double x) {
                double r = __builtin_amdgcn_rcp(x);
                r = dfma(dfma(-x, r, 1.0), r, r);
                return dfma(dfma(-x, r, 1.0), r, r);
              };
              const double a00 = D[0][0], a01 = D[0][1], a11 = D[1][1];
              const double ia = rcp(dfma(a00, a11, -(a01 * a01)));
              const double x00 = a11 * ia, x01 = -a01 * ia, x11 = a00 * ia;
              const double b00 = D[0][2], b01 = D[0][3], b10 = D[1][2], b11 = D[1][3];
              const double y00 = dfma(x00, b00, x01 * b10), y01 = dfma(x00, b01, x01 * b11);
              const double y10 = dfma(x01, b00, x11 * b10), y11 = dfma(x01, b01, x11 * b11);
              const double s00 = D[2][2] - dfma(b00, y00, b10 * y10), s01 = D[2][3] - dfma(b00, y01, b10 * y11);
              const double s11 = D[3][3] - dfma(b01, y01, b11 * y11);
              const double is = rcp(dfma(s00, s11, -(s01 * s01)));
              const double z00 = s11 * is, z01 = -s01 * is, z11 = s00 * is;
              const double w00 = dfma(y00, z00, y01 * z01), w01 = dfma(y00, z01, y01 * z11);  // W = Y Z
              const double w10 = dfma(y10, z00, y11 * z01), w11 = dfma(y10, z01, y11 * z11);
              D[0][0] = -(x00 + dfma(w00, y00, w01 * y01));
              D[0][1] = D[1][0] = -(x01 + dfma(w00, y10, w01 * y11));
              D[1][1] = -(x11 + dfma(w10, y10, w11 * y11));
              D[0][2] = D[2][0] = w00, D[0][3] = D[3][0] = w01;
              D[1][2] = D[2][1] = w10, D[1][3] = D[3][1] = w11;
              D[2][2] = -z00, D[2][3] = D[3][2] = -z01, D[3][3] = -z11;
            }
            const double2 lo = *reinterpret_cast<const double2 *>(Pc + tid * 4);
            const double2 hi = *reinterpret_cast<const double2 *>(Pc + tid * 4 + 2);
            const bool pivr = (tid >> 2) == p;
            const int ar = tid & 3;
            // pivot row a: -e_a in place of its panel row gives -(e_a' D^-1) = row a of -D^-1
            const double pr[4] = {pivr ? (ar == 0 ? -1.0 : 0.0) : lo.x, pivr ? (ar == 1 ? -1.0 : 0.0) : lo.y,
                                  pivr ? (ar == 2 ? -1.0 : 0.0) : hi.x, pivr ? (ar == 3 ? -1.0 : 0.0) : hi.y};
            double f4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              double acc2 = 0.0;
#pragma unroll
              for (int b = 0; b < 4; ++b) acc2 = dfma(pr[b], -D[b][c], acc2);  // D holds -D^-1
              f4[c] = acc2;
            }
            *reinterpret_cast<double2 *>(Fx + tid * 4) = make_double2(f4[0], f4[1]);
            *reinterpret_cast<double2 *>(Fx + tid * 4 + 2) = make_double2(f4[2], f4[3]);
            *reinterpret_cast<double2 *>(Tn + tid * 4) = make_double2(pivr ? 0.0 : -f4[0], pivr ? 0.0 : -f4[1]);
            *reinterpret_cast<double2 *>(Tn + tid * 4 + 2) = make_double2(pivr ? 0.0 : -f4[2], pivr ? 0.0 : -f4[3]);
          }
          __syncthreads();
          // (c)
          if (sw) {
            double av[TPW], bv[TPW];
#pragma unroll
            for (int s2 = 0; s2 < TPW; ++s2) av[s2] = Tn[offa[s2]], bv[s2] = Pc[offb[s2]];
#pragma unroll
            for (int s2 = 0; s2 < TPW; ++s2)
              acc[s2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[s2], acc[s2], 0, 0, 0);
            if (c2 == Ik) {
              put(acc[Ik], g, Fx[offb[Ik]]);  // diagonal tile: its pivot rows ...
              if (mycols) {                   // ... and the pivot columns of the whole tile column
#pragma unroll
                for (int I = 0; I <= Ik; ++I)
#pragma unroll
                  for (int v = 0; v < 4; ++v) acc[I][v] = Fx[64 * I + 16 * v + pcol];
              }
            } else if (Ik < c2) {
              put(acc[Ik], g, Fx[offb[Ik]]);
            }
            if (c1 == Ik) {
              put(acc[NTL - Ik], g, Fx[offb[NTL - Ik]]);
              if (mycols) {
#pragma unroll
                for (int I = 0; I <= Ik; ++I)
#pragma unroll
                  for (int v = 0; v < 4; ++v) acc[NTL - I][v] = Fx[64 * I + 16 * v + pcol];
              }
            } else if (Ik < c1) {
              put(acc[NTL - Ik], g, Fx[offb[NTL - Ik]]);
            }
          }
        }
      }
    }
    // the tiles now hold -M: hand the upper triangle over to the 6x6 register blocks, one 16-row strip at a time
    __syncthreads();
    double *strip = Q.Ep;  // [16][NPAD]
#pragma unroll
    for (int ii = 0; ii < GS; ++ii)
#pragma unroll
      for (int jj = 0; jj < GS; ++jj) a[ii][jj] = 0.0;
#pragma unroll
    for (int I = 0; I < NTL; ++I) {
      if (I < ntl) {  // uniform
        if (sw) {
          if (I <= c2) {
#pragma unroll
            for (int v = 0; v < 4; ++v) strip[(kq + 4 * v) * NPAD + 16 * c2 + l15] = -acc[I][v];
          }
          if (I <= c1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) strip[(kq + 4 * v) * NPAD + 16 * c1 + l15] = -acc[NTL - I][v];
          }
        }
        __syncthreads();
        if (owner && e1 < ng) {
#pragma unroll
          for (int ii = 0; ii < GS; ++ii) {
            const int row = i0 + ii;
            if ((row >> 4) == I) {
              const double *sp = strip + (row & 15) * NPAD + j0;
#pragma unroll
              for (int jj = 0; jj < GS; jj += 2) {
                const double2 t2 = *reinterpret_cast<const double2 *>(sp + jj);
                a[ii][jj] = t2.x, a[ii][jj + 1] = t2.y;
              }
            }
          }
        }
        __syncthreads();
      }
    }
    // diagonal blocks keep the full symmetric 6x6: the lower triangle mirrors the upper one bit for bit
#pragma unroll
    for (int ii = 1; ii < GS; ++ii)
#pragma unroll
      for (int jj = 0; jj < ii; ++jj) a[ii][jj] = diag ? a[jj][ii] : a[ii][jj];
  }
}

}  // namespace hmpc
