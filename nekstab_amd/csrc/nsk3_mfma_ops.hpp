// Gauss <-> GLL tensor contractions of the hexahedral pressure operators on the fp64 matrix cores (lx1 = 6, 8, 10).
//
// opgradt3 (D^T p) and opdiv3 (weak divergence) are, per velocity component, three passes OUT[m][n] = sum_k A[m][k] X[k][n] with
// the (lx1-2) x lx1 matrices J12 / D12 (or their transposes) as A and the other two tensor indices flattened into n (at lx1 = 8:
// 36, 48 or 64 columns).  The thread-per-node loops issue two LDS reads per multiply-add and are bound by the LDS pipe (k_gradt, which is
// opgradt3 alone, ran at 2.4x its streaming floor); here A lives in registers (v_mfma_f64_16x16x4_f64 fragments), X is read
// from LDS once per 16-column tile.  Where an operator applies J12 and D12 to the SAME data the two 6-row matrices are stacked
// into one fragment (12 of 16 rows at lx1 = 8, all 16 at lx1 = 10); contracted lengths are padded to a multiple of 4 with zero
// operands (no out-of-range read: the padded lanes supply 0.0), column counts to a multiple of 16 with masked lanes.
#pragma once
#include "nsk_kernels.hpp"

namespace nsk {
namespace k3 {

typedef double mo_d4 __attribute__((ext_vector_type(4)));

// One pass with one or two accumulated chains:  OUT(m, n) = sum_k A1(m,k) X1(k,n) [+ sum_k A2(m,k) X2(k,n)],
// k < K (<= 4 KQ), n < NCOL, m < 16 as far as ST::store keeps it.  X(k, n) at X[XC::col(n) + k * XC::ks].
template <int K, int KQ, int NCOL, class XC, class ST, bool TWO>
__device__ inline void mo_pass(const double (&a1)[KQ], const double* X1, const double (&a2)[KQ], const double* X2, double* OUT,
                               int wave, int nwaves, int lane) {
  const int n16 = lane & 15, kq = lane >> 4;
  constexpr int NT16 = (NCOL + 15) / 16;
  for (int tile = wave; tile < NT16; tile += nwaves) {
    const int n = tile * 16 + n16;
    const bool nok = n < NCOL;
    const int cb = nok ? XC::col(n) : 0;
    mo_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
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

// column maps of the tensor layouts: n -> LDS offset, `ks` = stride of the contracted index
template <int KS> struct ColLinear { static constexpr int ks = KS; static __device__ inline int col(int n) { return n; } };                 // contract the slowest index
template <int DIV, int HI, int KS> struct ColPlane { static constexpr int ks = KS; static __device__ inline int col(int n) { return (n / DIV) * HI + n % DIV; } };   // [plane][contracted][fast]
template <int CS> struct ColRow { static constexpr int ks = 1; static __device__ inline int col(int n) { return n * CS; } };                 // contract the fastest index
// stores: rows m < MR of the produced index with stride MS at column offset col(n)
template <class C, int MS, int MR> struct StLin { static __device__ inline void store(double* o, int n, int m, double v) { if (m < MR) o[C::col(n) + m * MS] = v; } };
// stacked operator [D12; J12]: rows 0..H-1 -> first array, rows H..2H-1 -> second array (SPLIT doubles further)
template <class C, int MS, int H, int SPLIT> struct StSplit {
  static __device__ inline void store(double* o, int n, int m, double v) {
    if (m < H) o[C::col(n) + m * MS] = v;
    else if (m < 2 * H) o[SPLIT + C::col(n) + (m - H) * MS] = v;
  }
};

// D^T p: same contract as the thread-per-node opgradt3<N> (sP [3][M^3], sC [3][N M^2], sE [2][N^2 M] scratch; result g[3] for GLL
// node tid; the last pass is staged through sC, free by then, which holds N^3 <= 3 N M^2 doubles)
// PRE: sP holds ALL nine products p * w2[a][c] ([c][a][M^3], filled by the caller before a barrier): the metrics are then dead
// before the passes start (18 fewer live registers, two barriers fewer) at the price of 6 M^3 more doubles of LDS
template <int N, bool PRE = false>
__device__ inline void opgradt3_mfma(const double* sJ12, const double* sD12, double pval, const double (&w2)[9], double* sP,
                                     double* sC, double* sE, int tid, int nt, double (&g)[3]) {
  constexpr int M = N - 2, MM = M * M * M, NN = N * N * N, NNM = N * N * M, NMM = N * M * M, KQ = (M + 3) / 4;
  static_assert(NN <= 3 * NMM && N <= 16, "staging through sC");
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6, m16 = lane & 15, kq = lane >> 4;
  // fragments of J12^T and D12^T (N x M, zero padded): A[m][k] = J12[k][m]
  double aJ[KQ], aD[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = 4 * q + kq;
    const bool ok = m16 < N && k < M;
    aJ[q] = ok ? sJ12[k * N + m16] : 0.0;
    aD[q] = ok ? sD12[k * N + m16] : 0.0;
  }
  typedef ColLinear<M * M> CT;                 // [cc | kk][(b,a)]
  typedef ColPlane<M, M * M, M> CS_in;         // [kk][b][a]   -> contract b
  typedef ColPlane<M, N * M, M> CS_out;        // [kk][jj][a]
  typedef ColRow<M> CR_in;                     // [(k,j)][a]   -> contract a
  typedef ColRow<N> CR_out;                    // [(k,j)][i]
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    if constexpr (!PRE) {
      if (tid < MM) {
        sP[tid] = pval * w2[0 * 3 + c];
        sP[MM + tid] = pval * w2[1 * 3 + c];
        sP[2 * MM + tid] = pval * w2[2 * 3 + c];
      }
      lds_barrier();
    }
    const double* sPc = PRE ? sP + c * 3 * MM : sP;
    // axis t: sC_a[kk][ba] = sum_cc A_a[kk][cc] sP_a[cc][ba],  A_0 = A_1 = J12^T, A_2 = D12^T
    mo_pass<M, KQ, M * M, CT, StLin<CT, M * M, N>, false>(aJ, sPc, aJ, sPc, sC, wave, nw, lane);
    mo_pass<M, KQ, M * M, CT, StLin<CT, M * M, N>, false>(aJ, sPc + MM, aJ, sPc, sC + NMM, wave, nw, lane);
    mo_pass<M, KQ, M * M, CT, StLin<CT, M * M, N>, false>(aD, sPc + 2 * MM, aD, sPc, sC + 2 * NMM, wave, nw, lane);
    lds_barrier();
    // axis s: sE_0[kk][jj][a] = J12^T sC_0,   sE_1 = D12^T sC_1 + J12^T sC_2
    mo_pass<M, KQ, N * M, CS_in, StLin<CS_out, M, N>, false>(aJ, sC, aJ, sC, sE, wave, nw, lane);
    mo_pass<M, KQ, N * M, CS_in, StLin<CS_out, M, N>, true>(aD, sC + NMM, aJ, sC + 2 * NMM, sE + NNM, wave, nw, lane);
    lds_barrier();
    // axis r: g[(k,j)][i] = D12^T sE_0 + J12^T sE_1   -> staged in sC, one value per GLL node
    mo_pass<M, KQ, N * N, CR_in, StLin<CR_out, 1, N>, true>(aD, sE, aJ, sE + NNM, sC, wave, nw, lane);
    lds_barrier();
    g[c] = (tid < NN) ? sC[tid] : 0.0;
    lds_barrier();
  }
}

// weak divergence: same contract as the thread-per-node opdiv3<N> (su [3][N^3]; sA [2][N^2 M], sB [3][N M^2] scratch); value for
// Gauss node tid; the last pass is staged through sA, which holds 3 M^3 <= 2 N^2 M doubles
template <int N>
__device__ inline double opdiv3_mfma(const double* sJ12, const double* sD12, const double* su, double* sA, double* sB,
                                     int tid, int nt, const double (&w2)[9]) {
  constexpr int M = N - 2, MM = M * M * M, NN = N * N * N, NNM = N * N * M, NMM = N * M * M, KQ = (N + 3) / 4;
  static_assert(3 * MM <= 2 * NNM && 2 * M <= 16, "staging through sA, stacked operator in one tile");
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6, m16 = lane & 15, kq = lane >> 4;
  // A[m][k]: rows 0..M-1 = D12[m][k], rows M..2M-1 = J12[m-M][k] (stacked), and the two M-row operators alone
  double aDJ[KQ], aJ[KQ], aD[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = 4 * q + kq;
    const bool kok = k < N;
    aJ[q] = (m16 < M && kok) ? sJ12[m16 * N + k] : 0.0;
    aD[q] = (m16 < M && kok) ? sD12[m16 * N + k] : 0.0;
    aDJ[q] = !kok ? 0.0 : ((m16 < M) ? sD12[m16 * N + k] : ((m16 < 2 * M) ? sJ12[(m16 - M) * N + k] : 0.0));
  }
  typedef ColRow<N> CR_in;                     // [(k,j)][i]   -> contract i
  typedef ColRow<M> CR_out;                    // [(k,j)][a]
  typedef ColPlane<M, N * M, M> CS_in;         // [k][j][a]    -> contract j
  typedef ColPlane<M, M * M, M> CS_out;        // [k][b][a]
  typedef ColLinear<M * M> CT;                 // [k | cc][(b,a)]
  double div = 0.0;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const double* u = su + c * NN;
    // axis r: sA_0[(k,j)][a] = D12 u, sA_1 = J12 u   (stacked)
    mo_pass<N, KQ, N * N, CR_in, StSplit<CR_out, 1, M, NNM>, false>(aDJ, u, aDJ, u, sA, wave, nw, lane);
    lds_barrier();
    // axis s: sB_0[k][b][a] = J12 sA_0;  sB_1 = D12 sA_1, sB_2 = J12 sA_1 (stacked)
    mo_pass<N, KQ, N * M, CS_in, StLin<CS_out, M, M>, false>(aJ, sA, aJ, sA, sB, wave, nw, lane);
    mo_pass<N, KQ, N * M, CS_in, StSplit<CS_out, M, M, NMM>, false>(aDJ, sA + NNM, aDJ, sA, sB + NMM, wave, nw, lane);
    lds_barrier();
    // axis t: ur = J12 sB_0, us = J12 sB_1, ut = D12 sB_2  -> staged in sA ([3][M^3]), one triple per Gauss node
    mo_pass<N, KQ, M * M, CT, StLin<CT, M * M, M>, false>(aJ, sB, aJ, sB, sA, wave, nw, lane);
    mo_pass<N, KQ, M * M, CT, StLin<CT, M * M, M>, false>(aJ, sB + NMM, aJ, sB, sA + MM, wave, nw, lane);
    mo_pass<N, KQ, M * M, CT, StLin<CT, M * M, M>, false>(aD, sB + 2 * NMM, aD, sB, sA + 2 * MM, wave, nw, lane);
    lds_barrier();
    if (tid < MM) div += w2[0 * 3 + c] * sA[tid] + w2[1 * 3 + c] * sA[MM + tid] + w2[2 * 3 + c] * sA[2 * MM + tid];
    lds_barrier();
  }
  return div;
}

}  // namespace k3
}  // namespace nsk
