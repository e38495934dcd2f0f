// Gauss <-> GLL tensor contractions of the hexahedral pressure operators on the fp64 matrix cores (lx1 = 8, lx2 = 6).
//
// opgradt3 (D^T p) and opdiv3 (weak divergence) are, per velocity component, three passes OUT[m][n] = sum_k A[m][k] X[k][n] with
// the 6 x 8 matrices J12 / D12 (or their transposes) as A and the other two tensor indices flattened into n (36, 48 or 64
// columns).  The thread-per-node loops issue two LDS reads per multiply-add and are bound by the LDS pipe (k_gradt, which is
// opgradt3 alone, ran at 2.4x its streaming floor); here A lives in registers (v_mfma_f64_16x16x4_f64 fragments), X is read
// from LDS once per 16-column tile.  Where an operator applies J12 and D12 to the SAME data the two 6-row matrices are stacked
// into one 12-row fragment (75 % of the tile); contracted lengths of 6 are padded to 8 with zero operands (no out-of-range
// read: the padded lanes supply 0.0), column counts of 36 to 48 with masked lanes.
#pragma once
#include "nsk_kernels.hpp"

namespace nsk {
namespace k3 {

typedef double mo_d4 __attribute__((ext_vector_type(4)));

// One pass with one or two accumulated chains:  OUT(m, n) = sum_k A1(m,k) X1(k,n) [+ sum_k A2(m,k) X2(k,n)],
// k < K (<= 8), n < NCOL, m < 16 as far as ST::store keeps it.  X(k, n) at X[XC::col(n) + k * XC::ks].
template <int K, int NCOL, class XC, class ST, bool TWO>
__device__ inline void mo_pass(const double (&a1)[2], const double* X1, const double (&a2)[2], const double* X2, double* OUT,
                               int wave, int nwaves, int lane) {
  const int n16 = lane & 15, kq = lane >> 4;
  constexpr int NT16 = (NCOL + 15) / 16;
  for (int tile = wave; tile < NT16; tile += nwaves) {
    const int n = tile * 16 + n16;
    const bool nok = n < NCOL;
    const int cb = nok ? XC::col(n) : 0;
    mo_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = 4 * q + kq;
      const bool ok = nok && k < K;
      const double b1 = ok ? X1[cb + k * XC::ks] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1, acc, 0, 0, 0);
      if (TWO) {
        const double b2 = ok ? X2[cb + k * XC::ks] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[q], b2, acc, 0, 0, 0);
      }
    }
    if (nok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ST::store(OUT, n, kq + 4 * r, acc[r]);
    }
  }
}

// column maps of the three tensor layouts (M = 6, N = 8)
struct ColT { static constexpr int ks = 36; static __device__ inline int col(int n) { return n; } };                      // [k|cc][ba]: contract the slowest index
struct ColS6 { static constexpr int ks = 6; static __device__ inline int col(int n) { return (n / 6) * 36 + n % 6; } };   // [kk][b][a] (6 x 6 planes): contract b
struct ColS8 { static constexpr int ks = 6; static __device__ inline int col(int n) { return (n / 6) * 48 + n % 6; } };   // [k][j][a] (8 x 6 planes): contract j
struct ColR6 { static constexpr int ks = 1; static __device__ inline int col(int n) { return n * 6; } };                  // [(k,j)][a]: contract a
struct ColR8 { static constexpr int ks = 1; static __device__ inline int col(int n) { return n * 8; } };                  // [(k,j)][i]: contract i
// stores: rows m < 8 (or < 6) of the produced index with stride MS at column offset col(n)
template <class C, int MS, int MR> struct StLin { static __device__ inline void store(double* o, int n, int m, double v) { if (m < MR) o[C::col(n) + m * MS] = v; } };
// stacked operator [D12; J12]: rows 0..5 -> first array, rows 6..11 -> second array (offset SPLIT doubles further)
template <class C, int MS, int SPLIT> struct StSplit {
  static __device__ inline void store(double* o, int n, int m, double v) {
    if (m < 6) o[C::col(n) + m * MS] = v;
    else if (m < 12) o[SPLIT + C::col(n) + (m - 6) * MS] = v;
  }
};

// D^T p for lx1 = 8: same contract as opgradt3<8> (sP [3][216], sC [3][288], sE [2][384] scratch; result g[3] for GLL node tid)
__device__ inline void opgradt3_mfma8(const double* sJ12, const double* sD12, double pval, const double (&w2)[9], double* sP,
                                      double* sC, double* sE, int tid, double (&g)[3]) {
  constexpr int N = 8, M = 6, MM = 216, NNM = 384, NMM = 288;
  const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
  // fragments of J12^T and D12^T (8 x 6, zero padded): A[m][k] = J12[k][m]
  double aJ[2], aD[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = 4 * q + kq;
    const bool ok = m16 < N && k < M;
    aJ[q] = ok ? sJ12[k * N + m16] : 0.0;
    aD[q] = ok ? sD12[k * N + m16] : 0.0;
  }
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    if (tid < MM) {
      sP[tid] = pval * w2[0 * 3 + c];
      sP[MM + tid] = pval * w2[1 * 3 + c];
      sP[2 * MM + tid] = pval * w2[2 * 3 + c];
    }
    lds_barrier();
    // axis t: sC_a[kk][ba] = sum_cc A_a[kk][cc] sP_a[cc][ba],  A_0 = A_1 = J12^T, A_2 = D12^T
    mo_pass<M, 36, ColT, StLin<ColT, 36, N>, false>(aJ, sP, aJ, sP, sC, wave, 8, lane);
    mo_pass<M, 36, ColT, StLin<ColT, 36, N>, false>(aJ, sP + MM, aJ, sP, sC + NMM, wave, 8, lane);
    mo_pass<M, 36, ColT, StLin<ColT, 36, N>, false>(aD, sP + 2 * MM, aD, sP, sC + 2 * NMM, wave, 8, lane);
    lds_barrier();
    // axis s: sE_0[kk][jj][a] = J12^T sC_0,   sE_1 = D12^T sC_1 + J12^T sC_2
    mo_pass<M, 48, ColS6, StLin<ColS8, 6, N>, false>(aJ, sC, aJ, sC, sE, wave, 8, lane);
    mo_pass<M, 48, ColS6, StLin<ColS8, 6, N>, true>(aD, sC + NMM, aJ, sC + 2 * NMM, sE + NNM, wave, 8, lane);
    lds_barrier();
    // axis r: g[(k,j)][i] = D12^T sE_0 + J12^T sE_1   -> staged in sP, one value per GLL node
    mo_pass<M, 64, ColR6, StLin<ColR8, 1, N>, true>(aD, sE, aJ, sE + NNM, sP, wave, 8, lane);
    lds_barrier();
    g[c] = sP[tid];
    lds_barrier();
  }
}

// weak divergence for lx1 = 8: same contract as opdiv3<8> (su [3][512]; sA [2][384], sB [3][288] scratch); value for Gauss node tid
__device__ inline double opdiv3_mfma8(const double* sJ12, const double* sD12, const double* su, double* sA, double* sB,
                                      int tid, const double (&w2)[9]) {
  constexpr int N = 8, M = 6, MM = 216, NN = 512, NNM = 384, NMM = 288;
  const int lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
  // A[m][k]: rows 0..5 = D12[m][k], rows 6..11 = J12[m-6][k] (stacked), and the two 6-row operators alone
  double aDJ[2], aJ[2], aD[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = 4 * q + kq;
    aJ[q] = (m16 < M) ? sJ12[m16 * N + k] : 0.0;
    aD[q] = (m16 < M) ? sD12[m16 * N + k] : 0.0;
    aDJ[q] = (m16 < M) ? sD12[m16 * N + k] : ((m16 < 2 * M) ? sJ12[(m16 - M) * N + k] : 0.0);
  }
  double div = 0.0;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const double* u = su + c * NN;
    // axis r: sA_0[(k,j)][a] = D12 u, sA_1 = J12 u   (stacked)
    mo_pass<N, 64, ColR8, StSplit<ColR6, 1, NNM>, false>(aDJ, u, aDJ, u, sA, wave, 8, lane);
    lds_barrier();
    // axis s: sB_0[k][b][a] = J12 sA_0;  sB_1 = D12 sA_1, sB_2 = J12 sA_1 (stacked)
    mo_pass<N, 48, ColS8, StLin<ColS6, 6, M>, false>(aJ, sA, aJ, sA, sB, wave, 8, lane);
    mo_pass<N, 48, ColS8, StSplit<ColS6, 6, NMM>, false>(aDJ, sA + NNM, aDJ, sA, sB + NMM, wave, 8, lane);
    lds_barrier();
    // axis t: ur = J12 sB_0, us = J12 sB_1, ut = D12 sB_2  -> staged in sA ([3][216]), one triple per Gauss node
    mo_pass<N, 36, ColT, StLin<ColT, 36, M>, false>(aJ, sB, aJ, sB, sA, wave, 8, lane);
    mo_pass<N, 36, ColT, StLin<ColT, 36, M>, false>(aJ, sB + NMM, aJ, sB, sA + MM, wave, 8, lane);
    mo_pass<N, 36, ColT, StLin<ColT, 36, M>, false>(aD, sB + 2 * NMM, aD, sB, sA + 2 * MM, wave, 8, lane);
    lds_barrier();
    if (tid < MM) div += w2[0 * 3 + c] * sA[tid] + w2[1 * 3 + c] * sA[MM + tid] + w2[2 * 3 + c] * sA[2 * MM + tid];
    lds_barrier();
  }
  return div;
}

}  // namespace k3
}  // namespace nsk
