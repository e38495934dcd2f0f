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

// general column map: columns n = p * Q + q over two tensor indices with strides SP, SQ; the contracted / produced index has stride KS
template <int Q, int SP, int SQ, int KS> struct Col2 { static constexpr int ks = KS; static __device__ inline int col(int n) { return (n / Q) * SP + (n % Q) * SQ; } };

// LDS layouts of the tiles the passes write and read: strides of the three tensor indices, slowest first.  Compact by default.
// At lx1 = 8 the compact strides (64, 8, 1) put whole 16-lane groups of a matrix-core store on one or two banks (ds_write_b64:
// 16 contiguous lanes per LDS cycle, bank = double index mod 16): counters at config 4's size gave 52 % of k_schwarz's LDS cycles as
// bank conflicts, and the LDS array busy for 60 % of the kernel.  Every tile stage has ONE writer pattern and ONE reader pattern, so
// each gets its own strides (searched with the bank rules of the CDNA4 guide: all passes below conflict-free or within one cycle):
//   F[s]: the N^3 tile of the fast-diagonalisation solve after stage s (0: filled, 1..3: forward r, s, t; 4..6: back t, s, r);
//   C: D^T intermediates [N][M][M] (written along t, read along s), E: [N][N][M] (written along s, read along r), G: the N^3 result.
template <int N> struct CompactLay {
  static constexpr int M = N - 2;
  static constexpr int F[7][3] = {{N * N, N, 1}, {N * N, N, 1}, {N * N, N, 1}, {N * N, N, 1}, {N * N, N, 1}, {N * N, N, 1}, {N * N, N, 1}};
  static constexpr int FEXT = N * N * N;
  static constexpr int C[3] = {M * M, M, 1}, CEXT = N * M * M;
  static constexpr int E[3] = {N * M, M, 1}, EEXT = N * N * M;
  static constexpr int G[3] = {N * N, N, 1};
};
template <int N> struct PadLay : CompactLay<N> {};
template <> struct PadLay<8> {
  static constexpr int F[7][3] = {{66, 8, 1}, {8, 65, 1}, {65, 16, 2}, {16, 65, 2}, {16, 65, 2}, {9, 76, 2}, {8, 71, 1}};
  static constexpr int FEXT = 616;
  static constexpr int C[3] = {38, 6, 1}, CEXT = 304;
  static constexpr int E[3] = {6, 52, 1}, EEXT = 416;
  static constexpr int G[3] = {8, 69, 1};
};

// D^T p: same contract as the thread-per-node opgradt3<N> (sP [3][M^3], sC [3][N M^2], sE [2][N^2 M] scratch; result g[3] for GLL
// node tid; the last pass is staged through sC, free by then, which holds N^3 <= 3 N M^2 doubles)
// PRE: sP holds ALL nine products p * w2[a][c] ([c][a][M^3], filled by the caller before a barrier): the metrics are then dead
// before the passes start (18 fewer live registers, two barriers fewer) at the price of 6 M^3 more doubles of LDS
// L: layout of sC (three arrays of L::CEXT) and sE (two of L::EEXT); the result is staged through sC with strides L::G
template <int N, bool PRE = false, class L = CompactLay<N>>
__device__ inline void opgradt3_mfma(const double* sJ12, const double* sD12, double pval, const double (&w2)[9], double* sP,
                                     double* sC, double* sE, int tid, int nt, double (&g)[3]) {
  constexpr int M = N - 2, MM = M * M * M, NN = N * N * N, NNM = L::EEXT, NMM = L::CEXT, KQ = (M + 3) / 4;
  static_assert((N - 1) * (L::G[0] + L::G[1] + L::G[2]) < 3 * NMM && N <= 16, "staging through sC");
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6, m16 = lane & 15, kq = lane >> 4;
  // fragments of J12^T and D12^T (N x M, zero padded): A[m][k] = J12[k][m]
  double aJ[KQ], aD[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = 4 * q + kq;
    const bool ok = m16 < N && k < M;
    aJ[q] = ok ? sJ12[k * N + m16] : 0.0;
    aD[q] = ok ? sD12[k * N + m16] : 0.0;
  }
  typedef ColLinear<M * M> CT;                                   // sP [cc][(b,a)]            -> contract cc
  typedef Col2<M, L::C[1], L::C[2], L::C[0]> CT_out;             // sC [kk][b][a], columns (b,a): produce kk
  typedef Col2<M, L::C[0], L::C[2], L::C[1]> CS_in;              // sC, columns (kk,a)        -> contract b
  typedef Col2<M, L::E[0], L::E[2], L::E[1]> CS_out;             // sE [kk][jj][a], columns (kk,a): produce jj
  typedef Col2<N, L::E[0], L::E[1], L::E[2]> CR_in;              // sE, columns (kk,jj)       -> contract a
  typedef Col2<N, L::G[0], L::G[1], L::G[2]> CR_out;             // result [k][j][i], columns (k,j): produce i
  const int gk = tid / (N * N), gj = (tid / N) % N, gi = tid % N;
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
    mo_pass<M, KQ, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aJ, sPc, aJ, sPc, sC, wave, nw, lane);
    mo_pass<M, KQ, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aJ, sPc + MM, aJ, sPc, sC + NMM, wave, nw, lane);
    mo_pass<M, KQ, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aD, sPc + 2 * MM, aD, sPc, sC + 2 * NMM, wave, nw, lane);
    lds_barrier();
    // axis s: sE_0[kk][jj][a] = J12^T sC_0,   sE_1 = D12^T sC_1 + J12^T sC_2
    mo_pass<M, KQ, N * M, CS_in, StLin<CS_out, CS_out::ks, N>, false>(aJ, sC, aJ, sC, sE, wave, nw, lane);
    mo_pass<M, KQ, N * M, CS_in, StLin<CS_out, CS_out::ks, N>, true>(aD, sC + NMM, aJ, sC + 2 * NMM, sE + NNM, wave, nw, lane);
    lds_barrier();
    // axis r: g[(k,j)][i] = D12^T sE_0 + J12^T sE_1   -> staged in sC, one value per GLL node
    mo_pass<M, KQ, N * N, CR_in, StLin<CR_out, CR_out::ks, N>, true>(aD, sE, aJ, sE + NNM, sC, wave, nw, lane);
    lds_barrier();
    g[c] = (tid < NN) ? sC[gk * L::G[0] + gj * L::G[1] + gi * L::G[2]] : 0.0;
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
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6, m16 = lane & 15, kq = lane >> 4;
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

// Fast-diagonalisation solve of one Schwarz patch (k_schwarz): six N x N passes over an N^3 tile,
//   z = (S_t x S_s x S_r) diag(1/(lr+ls+lt)) (S_t x S_s x S_r)^T w,
// each OUT(m, n) = sum_k A(m,k) X(k,n) with N^2 columns.  sS = [3][N*N] (S_d[pos][mode]), sL = [3][N]; in: sa, out: sa, scratch sb.
// The thread-per-node form read two LDS operands per multiply-add (96 reads per thread and solve); here the operator is a register
// fragment and the tile is read once per 16-column block (2-3 reads per lane and pass).  Ends with a barrier.
template <int N, class XC, class OC, bool FWD, class F>
__device__ inline void fd_pass(const double* sSd, const double* X, int wave, int nwaves, int lane, F&& store) {
  constexpr int KQ = (N + 3) / 4, NCOL = N * N, NT16 = (NCOL + 15) / 16;
  const int n16 = lane & 15, kq = lane >> 4;
  double a[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = 4 * q + kq;
    // forward: A(m, k) = S[k][m] (S^T);  back: A(m, k) = S[m][k]
    a[q] = (n16 < N && k < N) ? (FWD ? sSd[k * N + n16] : sSd[n16 * N + k]) : 0.0;
  }
  for (int tile = wave; tile < NT16; tile += nwaves) {
    const int n = tile * 16 + n16;
    const bool nok = n < NCOL;
    const int cb = nok ? XC::col(n) : 0, ob = nok ? OC::col(n) : 0;
    mo_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int k = 4 * q + kq;
      const double b = (nok && k < N) ? X[cb + k * XC::ks] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b, acc, 0, 0, 0);
    }
    if (nok) {
#pragma unroll
      for (int r = 0; r < (N + 3) / 4; ++r)
        if (kq + 4 * r < N) store(ob + (kq + 4 * r) * OC::ks, n, kq + 4 * r, acc[r]);
    }
  }
}
// column maps of stage ST of the tile (strides L::F[ST] of (k, j, i)): contract / produce i, j or k
template <int N, class L, int ST> using FdR = Col2<N, L::F[ST][0], L::F[ST][1], L::F[ST][2]>;      // columns (k, j)
template <int N, class L, int ST> using FdS = Col2<N, L::F[ST][0], L::F[ST][2], L::F[ST][1]>;      // columns (k, i)
template <int N, class L, int ST> using FdT = Col2<N, L::F[ST][1], L::F[ST][2], L::F[ST][0]>;      // columns (j, i)
// pass boundary: workgroup barrier, or -- one wavefront per element -- only the wait for the wave's own LDS traffic (the LDS
// serves a wavefront's instructions in order; the "memory" clobber keeps the compiler from moving accesses across)
__device__ inline void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <bool WAVE> __device__ inline void pass_sync() { if constexpr (WAVE) wave_sync(); else lds_barrier(); }

template <int N, class L, bool WAVE = false>
__device__ inline void fd_forward_mfma(const double* sS, const double* sL, double* sa, double* sb, double eps, int tid, int nt) {
  const int lane = tid & 63, wave = WAVE ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6), nw = WAVE ? 1 : nt >> 6;
  fd_pass<N, FdR<N, L, 0>, FdR<N, L, 1>, true>(sS, sa, wave, nw, lane, [&](int o, int, int, double v) { sb[o] = v; });
  pass_sync<WAVE>();
  fd_pass<N, FdS<N, L, 1>, FdS<N, L, 2>, true>(sS + N * N, sb, wave, nw, lane, [&](int o, int, int, double v) { sa[o] = v; });
  pass_sync<WAVE>();
  fd_pass<N, FdT<N, L, 2>, FdT<N, L, 3>, true>(sS + 2 * N * N, sa, wave, nw, lane, [&](int o, int n, int m, double v) {
    const double lam = sL[n % N] + sL[N + n / N] + sL[2 * N + m];
    sb[o] = (lam > eps) ? v / lam : 0.0;
  });
  pass_sync<WAVE>();
}
template <int N, class L, bool WAVE = false>
__device__ inline void fd_back_mfma(const double* sS, double* sa, double* sb, int tid, int nt) {
  const int lane = tid & 63, wave = WAVE ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6), nw = WAVE ? 1 : nt >> 6;
  fd_pass<N, FdT<N, L, 3>, FdT<N, L, 4>, false>(sS + 2 * N * N, sb, wave, nw, lane, [&](int o, int, int, double v) { sa[o] = v; });
  pass_sync<WAVE>();
  fd_pass<N, FdS<N, L, 4>, FdS<N, L, 5>, false>(sS + N * N, sa, wave, nw, lane, [&](int o, int, int, double v) { sb[o] = v; });
  pass_sync<WAVE>();
  fd_pass<N, FdR<N, L, 5>, FdR<N, L, 6>, false>(sS, sb, wave, nw, lane, [&](int o, int, int, double v) { sa[o] = v; });
  pass_sync<WAVE>();
}

// D^T p by ONE wavefront (k_schwarz_w): z[r], w2[m][r] = value / metrics of Gauss node r * 64 + lane; the three components go
// straight from the accumulators of the last pass to yl (element base `y`, component stride cs; a tile's two stores cover whole
// 64-byte lines).  One buffer of GT_BUF doubles, regions reused in the order the passes allow (what is live together: the two
// intermediates of the D-along-s / D-along-t terms + their sum's tile, then the three tiles of the last two passes):
//   sC1 [0, C) sC2 [C, 2C) | products P1, P2 and later sE1 [C+E, ...) | then P0 [C, ..) -> sC0 [0, C) -> sE0 [C, C+E)
template <int N, class L> struct GtWave {
  static constexpr int M = N - 2, MM = M * M * M, C = L::CEXT, E = L::EEXT;
  static constexpr int oC1 = 0, oC2 = C, oE1 = C + E, oP12 = C + E, oP0 = C, oC0 = 0, oE0 = C;
  static constexpr int BUF = (oE1 + E > oP12 + 2 * MM) ? oE1 + E : oP12 + 2 * MM;
  static_assert(2 * C <= oP12 && oP0 + MM <= oE1 && oE0 + E <= oE1, "regions");
};
template <int N, class L, int RM>
__device__ inline void opgradt3_wave(const double* sJ12, const double* sD12, const double (&z)[RM], const double (&w2)[9][RM],
                                     double* buf, int lane, double* __restrict__ y, long long cs) {
  using G = GtWave<N, L>;
  constexpr int M = N - 2, MM = M * M * M, KQ = (M + 3) / 4;
  const int m16 = lane & 15, kq = lane >> 4;
  double aJ[KQ], aD[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = 4 * q + kq;
    const bool ok = m16 < N && k < M;
    aJ[q] = ok ? sJ12[k * N + m16] : 0.0;
    aD[q] = ok ? sD12[k * N + m16] : 0.0;
  }
  typedef ColLinear<M * M> CT;
  typedef Col2<M, L::C[1], L::C[2], L::C[0]> CT_out;
  typedef Col2<M, L::C[0], L::C[2], L::C[1]> CS_in;
  typedef Col2<M, L::E[0], L::E[2], L::E[1]> CS_out;
  typedef Col2<N, L::E[0], L::E[1], L::E[2]> CR_in;
  typedef ColRow<N> CR_out;                                  // the element's own [(k,j)][i] order in global memory
#pragma unroll
  for (int c = 0; c < 3; ++c) {                          // (unrolled: w2 stays in registers)
    // terms with the derivative along s (axis 1) and along t (axis 2): products, t-axis passes, s-axis pass
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int idx = r * 64 + lane;
      if (idx < MM) { buf[G::oP12 + idx] = z[r] * w2[1 * 3 + c][r]; buf[G::oP12 + MM + idx] = z[r] * w2[2 * 3 + c][r]; }
    }
    wave_sync();
    mo_pass<M, KQ, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aJ, buf + G::oP12, aJ, buf, buf + G::oC1, 0, 1, lane);
    mo_pass<M, KQ, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aD, buf + G::oP12 + MM, aD, buf, buf + G::oC2, 0, 1, lane);
    wave_sync();
    mo_pass<M, KQ, N * M, CS_in, StLin<CS_out, CS_out::ks, N>, true>(aD, buf + G::oC1, aJ, buf + G::oC2, buf + G::oE1, 0, 1, lane);
    wave_sync();
    // term with the derivative along r (axis 0)
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int idx = r * 64 + lane;
      if (idx < MM) buf[G::oP0 + idx] = z[r] * w2[0 * 3 + c][r];
    }
    wave_sync();
    mo_pass<M, KQ, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aJ, buf + G::oP0, aJ, buf, buf + G::oC0, 0, 1, lane);
    wave_sync();
    mo_pass<M, KQ, N * M, CS_in, StLin<CS_out, CS_out::ks, N>, false>(aJ, buf + G::oC0, aJ, buf, buf + G::oE0, 0, 1, lane);
    wave_sync();
    // axis r: g[(k,j)][i] = D12^T sE_0 + J12^T sE_1  -> global
    mo_pass<M, KQ, N * N, CR_in, StLin<CR_out, 1, N>, true>(aD, buf + G::oE0, aJ, buf + G::oE1, y + c * cs, 0, 1, lane);
    wave_sync();
  }
}

// Weak divergence by ONE wavefront (k_divgs_w), component by component: `su` (the gathered, mass-scaled component c, [k][j][i])
// -> r-axis (stacked [D12; J12]) -> s-axis -> t-axis, whose three products stay in the accumulators and are combined with the
// metrics there (w2 is loaded in the accumulator's own node order: rows cc = kq + 4 r, columns (b, a) = 16 tile + n16).
// One buffer, regions reused as the passes allow:  sA0 [0, A) sA1 [A, 2A) | su [2A, 2A + N^3) -> sB0 [2A, ..) sB2 [2A + B, ..) | sB1 [0, B)
template <int N> struct DvWave {
  static constexpr int M = N - 2, A = N * N * M, B = N * M * M, NN = N * N * N;
  static constexpr int oA0 = 0, oA1 = A, oU = 2 * A, oB0 = 2 * A, oB2 = 2 * A + B, oB1 = 0;
  static constexpr int BUF = (2 * A + NN > 2 * A + 2 * B) ? 2 * A + NN : 2 * A + 2 * B;
  static constexpr int NT3 = (M * M + 15) / 16, RQ = (M + 3) / 4;
  static_assert(B <= A && 2 * M <= 16, "sB1 in sA0's place; stacked operator in one tile");
};
template <int K, int KQ, class XC>
__device__ inline mo_d4 mo_tile(const double (&a)[KQ], const double* X, int tile, int ncol, int lane) {
  const int n = tile * 16 + (lane & 15), kq = lane >> 4;
  const bool nok = n < ncol;
  const int cb = nok ? XC::col(n) : 0;
  mo_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = 4 * q + kq;
    const double b = (nok && k < K) ? X[cb + k * XC::ks] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b, acc, 0, 0, 0);
  }
  return acc;
}
// one component: buf + oU holds it; div[t][r] += the component's three terms.  wa / wb / wc -> the element's w2[0*3+c], w2[1*3+c],
// w2[2*3+c]; they are read in accumulator order, issued after the first pass (the gather's registers are free by then) and
// consumed after the last.
template <int N>
__device__ inline void opdiv3_wave_comp(const double (&aDJ)[(N + 3) / 4], const double (&aJ)[(N + 3) / 4], const double (&aD)[(N + 3) / 4],
                                        double* buf, int lane, const double* __restrict__ wa, const double* __restrict__ wb,
                                        const double* __restrict__ wc, double (&div)[DvWave<N>::NT3][DvWave<N>::RQ]) {
  using W = DvWave<N>;
  constexpr int M = N - 2, KQ = (N + 3) / 4;
  typedef ColRow<N> CR_in;                     // [(k,j)][i]   -> contract i
  typedef ColRow<M> CR_out;                    // [(k,j)][a]
  typedef ColPlane<M, N * M, M> CS_in;         // [k][j][a]    -> contract j
  typedef ColPlane<M, M * M, M> CS_out;        // [k][b][a]
  typedef ColLinear<M * M> CT;                 // [k][(b,a)]   -> contract k
  // axis r: sA_0[(k,j)][a] = D12 u, sA_1 = J12 u   (stacked)
  mo_pass<N, KQ, N * N, CR_in, StSplit<CR_out, 1, M, W::oA1 - W::oA0>, false>(aDJ, buf + W::oU, aDJ, buf, buf + W::oA0, 0, 1, lane);
  wave_sync();
  double ma[W::NT3][W::RQ], mb[W::NT3][W::RQ], mc[W::NT3][W::RQ];
  {
    const int m16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < W::NT3; ++t)
#pragma unroll
      for (int r = 0; r < W::RQ; ++r) {
        const int n = t * 16 + m16, cc = kq + 4 * r;
        const bool ok = n < M * M && cc < M;
        const unsigned qo = (unsigned)(ok ? cc * M * M + n : 0) * 8u;
        ma[t][r] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wa) + qo);
        mb[t][r] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wb) + qo);
        mc[t][r] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wc) + qo);
      }
  }
  // axis s: sB_0[k][b][a] = J12 sA_0;  sB_1 = D12 sA_1, sB_2 = J12 sA_1 (stacked)
  mo_pass<N, KQ, N * M, CS_in, StLin<CS_out, M, M>, false>(aJ, buf + W::oA0, aJ, buf, buf + W::oB0, 0, 1, lane);
  mo_pass<N, KQ, N * M, CS_in, StSplit<CS_out, M, M, W::oB2 - W::oB1>, false>(aDJ, buf + W::oA1, aDJ, buf, buf + W::oB1, 0, 1, lane);
  wave_sync();
  // axis t: ur = J12 sB_0, us = J12 sB_1, ut = D12 sB_2, combined with the metrics in the accumulators (rows / columns outside
  // the tile carry zeros in the accumulators: whatever metric value was read there does not count)
#pragma unroll
  for (int t = 0; t < W::NT3; ++t) {
    const mo_d4 ur = mo_tile<N, KQ, CT>(aJ, buf + W::oB0, t, M * M, lane);
    const mo_d4 us = mo_tile<N, KQ, CT>(aJ, buf + W::oB1, t, M * M, lane);
    const mo_d4 ut = mo_tile<N, KQ, CT>(aD, buf + W::oB2, t, M * M, lane);
#pragma unroll
    for (int r = 0; r < W::RQ; ++r) div[t][r] += ma[t][r] * ur[r] + mb[t][r] * us[r] + mc[t][r] * ut[r];
  }
  wave_sync();
}

// ---- passes on v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction) ----------------------------------------
// Timed on the MI355X (scripts/mfma_f64_rate.hip): 69 TFLOP/s against 49 for v_mfma_f64_16x16x4_f64 (and 54 for v_fma_f64) -- and a
// 4-row block wastes nothing on an 8-row operator where the 16-row tile is half empty: 2.1x fewer flops issued at 1.4x the rate.
// Operand layout, found by experiment (scripts/mfma_f64_4x4_layout.hip): A[b][i][k] in lane i + 4 b + 16 k, B[b][k][j] in lane
// j + 4 b + 16 k, D[b][i][j] in lane j + 4 b + 16 i.  With the four blocks b = four groups of four columns and the SAME A in each,
// B and D sit exactly where the 16x16x4 form has them (column lane & 15, k resp. row lane >> 4): one instruction per 4-row block
// of the operator replaces one 16x16x4 instruction, the callers' column maps and stores do not change.
// Used by the wavefront-per-element Schwarz kernel (k_schwarz_w16: 470 -> 422-434 us at config 4's size), which had reached 90 % of
// the 16x16x4 rate.  NOT by the workgroup-per-element kernels (opgradt3_mfma, opdiv3_mfma, k_convect_mfma8): measured there, the
// extra fragment registers cost a workgroup per CU (k_divgs<8> 61 -> 74 VGPRs, 675 -> 762 us; k_convect_mfma8 2.8 -> 4.1 ms).
template <int KQ, int RB> struct Frag4 { double a[RB][KQ]; };
template <int KQ, int RB, class F>
__device__ inline Frag4<KQ, RB> make_frag4(int lane, F&& elem) {      // elem(row, k): the operator, zero outside its range
  Frag4<KQ, RB> f;
  const int i = lane & 3, kq = lane >> 4;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int q = 0; q < KQ; ++q) f.a[rb][q] = elem(rb * 4 + i, 4 * q + kq);
  return f;
}
template <int K, int KQ, int RB, int NCOL, class XC, class ST, bool TWO>
__device__ inline void mo_pass4(const Frag4<KQ, RB>& a1, const double* X1, const Frag4<KQ, RB>& a2, const double* X2, double* OUT,
                                int wave, int nwaves, int lane) {
  const int n16 = lane & 15, kq = lane >> 4;
  constexpr int NT16 = (NCOL + 15) / 16;
  for (int tile = wave; tile < NT16; tile += nwaves) {
    const int n = tile * 16 + n16;
    const bool nok = n < NCOL;
    const int cb = nok ? XC::col(n) : 0;
    double acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int k = 4 * q + kq;
      const bool ok = nok && k < K;
      const double b1 = ok ? X1[cb + k * XC::ks] : 0.0;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1.a[rb][q], b1, acc[rb], 0, 0, 0);
      if (TWO) {
        const double b2 = ok ? X2[cb + k * XC::ks] : 0.0;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2.a[rb][q], b2, acc[rb], 0, 0, 0);
      }
    }
    if (nok) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) ST::store(OUT, n, kq + 4 * rb, acc[rb]);
    }
  }
}

// ---- slim forms for k_schwarz_w<N, true>: sixteen elements in flight per CU instead of twelve --------------------------------
// One tile buffer for the fast-diagonalisation solve: a wavefront that owns the tile can run a pass IN PLACE -- every operand read of
// all tiles is issued before the first store, and the LDS serves a wavefront in order.  One layout for all six passes then
// (strides (k, j, i) = (PS, RS, 1); at lx1 = 8 (72, 9, 1): 224 LDS cycles per three passes in the bank model against 144 with the
// per-stage strides of PadLay and 480 compact).
template <int N> struct SlimLay { static constexpr int PS = N * N, RS = N, EXT = N * N * N; };
template <> struct SlimLay<8> { static constexpr int PS = 72, RS = 9, EXT = 576; };
template <int N, class XC, bool FWD, class F>
__device__ inline void fd_pass_inplace(const double* sSd, double* X, int lane, F&& xform) {
  constexpr int KQ = (N + 3) / 4, NCOL = N * N, NT16 = (NCOL + 15) / 16, RB = (N + 3) / 4;
  const int n16 = lane & 15, kq = lane >> 4;
  // forward: A(m, k) = S[k][m] (S^T);  back: A(m, k) = S[m][k]
  const Frag4<KQ, RB> a = make_frag4<KQ, RB>(lane, [&](int mrow, int k) { return (mrow < N && k < N) ? (FWD ? sSd[k * N + mrow] : sSd[mrow * N + k]) : 0.0; });
  double b[NT16][KQ];
#pragma unroll
  for (int t = 0; t < NT16; ++t) {
    const int n = t * 16 + n16;
    const int cb = (n < NCOL) ? XC::col(n) : 0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int k = 4 * q + kq;
      b[t][q] = (n < NCOL && k < N) ? X[cb + k * XC::ks] : 0.0;
    }
  }
  wave_sync();                                   // every read of the pass has returned before the first store is issued
#pragma unroll
  for (int t = 0; t < NT16; ++t) {
    const int n = t * 16 + n16;
    double acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.a[rb][q], b[t][q], acc[rb], 0, 0, 0);
    if (n < NCOL) {
      const int cb = XC::col(n);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        if (kq + 4 * rb < N) X[cb + (kq + 4 * rb) * XC::ks] = xform(acc[rb], n, kq + 4 * rb);
    }
  }
  wave_sync();
}
template <int N>
__device__ inline void fd_solve_inplace_wave(const double* sS, const double* sL, double* X, double eps, int lane) {
  using S = SlimLay<N>;
  typedef Col2<N, S::PS, S::RS, 1> CR;           // columns (k, j): contract i
  typedef Col2<N, S::PS, 1, S::RS> CS;           // columns (k, i): contract j
  typedef Col2<N, S::RS, 1, S::PS> CT;           // columns (j, i): contract k
  auto same = [](double v, int, int) { return v; };
  fd_pass_inplace<N, CR, true>(sS, X, lane, same);
  fd_pass_inplace<N, CS, true>(sS + N * N, X, lane, same);
  fd_pass_inplace<N, CT, true>(sS + 2 * N * N, X, lane, [&](double v, int n, int m) {
    const double lam = sL[n % N] + sL[N + n / N] + sL[2 * N + m];
    return (lam > eps) ? v / lam : 0.0;
  });
  fd_pass_inplace<N, CT, false>(sS + 2 * N * N, X, lane, same);
  fd_pass_inplace<N, CS, false>(sS + N * N, X, lane, same);
  fd_pass_inplace<N, CR, false>(sS, X, lane, same);
}
// ---- D^T G D of one component on the matrix cores, one workgroup per element (k_helm, k_rhs) ----------------------------------------
// Six passes of the 1-D derivative operator (three directions forward, the metric products per node, three directions back) as
// tile jobs of sixteen columns spread over the workgroup's wavefronts; the operator's fragments (D and D^T, 2 x RB x KQ doubles a
// lane) come straight from global memory with the kernel's first loads.  Tiles in the (PS, RS, 1) strides of SlimLay.
// LDS traffic per component: 84 KB against the 393 KB of the per-node sums of axhelm3 (lx1 = 8), which kept the LDS of a CU busy
// for 41 % of k_helm<8>'s time.
template <int N> struct AxFrag { Frag4<(N + 3) / 4, (N + 3) / 4> d, dt; };
template <int N>
__device__ inline AxFrag<N> ax_frags(const double* __restrict__ D, int lane) {        // D[a][b] row-major, as Dev::D
  constexpr int KQ = (N + 3) / 4, RB = (N + 3) / 4;
  AxFrag<N> f;
  f.d = make_frag4<KQ, RB>(lane, [&](int m, int k) { return (m < N && k < N) ? D[m * N + k] : 0.0; });
  f.dt = make_frag4<KQ, RB>(lane, [&](int m, int k) { return (m < N && k < N) ? D[k * N + m] : 0.0; });
  return f;
}
// Z: the component's tile (overwritten), W: three tiles, O12: two tiles.  Returns (A z)(node tn) and z itself in zout.
// Ends without a barrier: the caller's next barrier separates the reads of Z / O12 from the next component's writes.
template <int N>
__device__ inline double axhelm3_mfma(const AxFrag<N>& F, double* Z, double* W, double* O12, const double (&g)[6], bool act, int tn,
                                      int wave, int nwaves, int lane, double& zout) {
  using S = SlimLay<N>;
  typedef Col2<N, S::PS, S::RS, 1> CR;           // columns (k, j): contract i
  typedef Col2<N, S::PS, 1, S::RS> CS;           // columns (k, i): contract j
  typedef Col2<N, S::RS, 1, S::PS> CT;           // columns (j, i): contract k
  constexpr int KQ = (N + 3) / 4, RB = (N + 3) / 4, NCOL = N * N, NT16 = (NCOL + 15) / 16, EXT = S::EXT;
  // job (direction d, tile t) goes to wavefront (d NT16 + t) mod nwaves
  const int w1 = (wave + nwaves - NT16 % nwaves) % nwaves, w2 = (wave + nwaves - (2 * NT16) % nwaves) % nwaves;
  mo_pass4<N, KQ, RB, NCOL, CR, StLin<CR, 1, N>, false>(F.d, Z, F.d, Z, W, wave, nwaves, lane);
  mo_pass4<N, KQ, RB, NCOL, CS, StLin<CS, S::RS, N>, false>(F.d, Z, F.d, Z, W + EXT, w1, nwaves, lane);
  mo_pass4<N, KQ, RB, NCOL, CT, StLin<CT, S::PS, N>, false>(F.d, Z, F.d, Z, W + 2 * EXT, w2, nwaves, lane);
  lds_barrier();
  zout = 0.0;
  if (act) {
    const double ur = W[tn], us = W[EXT + tn], ut = W[2 * EXT + tn];
    zout = Z[tn];
    W[tn] = g[0] * ur + g[3] * us + g[4] * ut;
    W[EXT + tn] = g[3] * ur + g[1] * us + g[5] * ut;
    W[2 * EXT + tn] = g[4] * ur + g[5] * us + g[2] * ut;
  }
  lds_barrier();
  mo_pass4<N, KQ, RB, NCOL, CR, StLin<CR, 1, N>, false>(F.dt, W, F.dt, W, Z, wave, nwaves, lane);
  mo_pass4<N, KQ, RB, NCOL, CS, StLin<CS, S::RS, N>, false>(F.dt, W + EXT, F.dt, W + EXT, O12, w1, nwaves, lane);
  mo_pass4<N, KQ, RB, NCOL, CT, StLin<CT, S::PS, N>, false>(F.dt, W + 2 * EXT, F.dt, W + 2 * EXT, O12 + EXT, w2, nwaves, lane);
  lds_barrier();
  return act ? ((Z[tn] + O12[tn]) + O12[EXT + tn]) : 0.0;
}

// opgradt3_wave with the metrics of a component loaded when its turn comes (the next component's are in flight under the current
// one's passes): 24 registers of metrics instead of 72.  wm = the element's first metric, npr = stride between the nine.
template <int N, class L, int RM>
__device__ inline void opgradt3_wave_ld(const double* sJ12, const double* sD12, const double (&z)[RM], const double* __restrict__ wm,
                                        long long npr, unsigned zmask, double* buf, int lane, double* __restrict__ y, long long cs) {
  using G = GtWave<N, L>;
  constexpr int M = N - 2, MM = M * M * M, KQ = (M + 3) / 4;
  constexpr int RB = (N + 3) / 4;
  // fragments of J12^T and D12^T (N x M, zero padded): A[m][k] = J12[k][m]
  const Frag4<KQ, RB> aJ = make_frag4<KQ, RB>(lane, [&](int mrow, int k) { return (mrow < N && k < M) ? sJ12[k * N + mrow] : 0.0; });
  const Frag4<KQ, RB> aD = make_frag4<KQ, RB>(lane, [&](int mrow, int k) { return (mrow < N && k < M) ? sD12[k * N + mrow] : 0.0; });
  typedef ColLinear<M * M> CT;
  typedef Col2<M, L::C[1], L::C[2], L::C[0]> CT_out;
  typedef Col2<M, L::C[0], L::C[2], L::C[1]> CS_in;
  typedef Col2<M, L::E[0], L::E[2], L::E[1]> CS_out;
  typedef Col2<N, L::E[0], L::E[1], L::E[2]> CR_in;
  typedef ColRow<N> CR_out;
  double wn[3][RM];                                      // metrics (axis a, this component) of the component in turn
  auto load_w = [&](int c) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const int idx = r * 64 + lane;
        wn[a][r] = ((zmask >> (a * 3 + c)) & 1u) ? 0.0 : *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wm + (size_t)(a * 3 + c) * npr) + (unsigned)(idx < MM ? idx : 0) * 8u);      // (zmask: Dev::zmask)
      }
  };
  load_w(0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double p0[RM], p1[RM], p2[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) { p0[r] = z[r] * wn[0][r]; p1[r] = z[r] * wn[1][r]; p2[r] = z[r] * wn[2][r]; }
    if (c < 2) load_w(c + 1);                              // in flight under this component's passes
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int idx = r * 64 + lane;
      if (idx < MM) { buf[G::oP12 + idx] = p1[r]; buf[G::oP12 + MM + idx] = p2[r]; }
    }
    wave_sync();
    mo_pass4<M, KQ, RB, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aJ, buf + G::oP12, aJ, buf, buf + G::oC1, 0, 1, lane);
    mo_pass4<M, KQ, RB, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aD, buf + G::oP12 + MM, aD, buf, buf + G::oC2, 0, 1, lane);
    wave_sync();
    mo_pass4<M, KQ, RB, N * M, CS_in, StLin<CS_out, CS_out::ks, N>, true>(aD, buf + G::oC1, aJ, buf + G::oC2, buf + G::oE1, 0, 1, lane);
    wave_sync();
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int idx = r * 64 + lane;
      if (idx < MM) buf[G::oP0 + idx] = p0[r];
    }
    wave_sync();
    mo_pass4<M, KQ, RB, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aJ, buf + G::oP0, aJ, buf, buf + G::oC0, 0, 1, lane);
    wave_sync();
    mo_pass4<M, KQ, RB, N * M, CS_in, StLin<CS_out, CS_out::ks, N>, false>(aJ, buf + G::oC0, aJ, buf, buf + G::oE0, 0, 1, lane);
    wave_sync();
    mo_pass4<M, KQ, RB, N * N, CR_in, StLin<CR_out, 1, N>, true>(aD, buf + G::oE0, aJ, buf + G::oE1, y + c * cs, 0, 1, lane);
    wave_sync();
  }
}

// opgradt3_wave_ld for a SMALL WORKGROUP of NW wavefronts that owns the element (k_schwarz_q: lx1 = 10, where sixteen nodes per lane
// cost a single wavefront 224 registers): Gauss node r * 64 * NW + tid per thread, the tiles of a pass spread over the NW waves,
// a workgroup barrier (of NW waves) where the wavefront form waits for its own LDS traffic.
template <int N, class L, int RM, int NW>
__device__ inline void opgradt3_wg_ld(const double* sJ12, const double* sD12, const double (&z)[RM], const double* __restrict__ wm,
                                      long long npr, unsigned zmask, double* buf, int tid, double* __restrict__ y, long long cs) {
  using G = GtWave<N, L>;
  constexpr int M = N - 2, MM = M * M * M, KQ = (M + 3) / 4, NT = 64 * NW;
  constexpr int RB = (N + 3) / 4;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Frag4<KQ, RB> aJ = make_frag4<KQ, RB>(lane, [&](int mrow, int k) { return (mrow < N && k < M) ? sJ12[k * N + mrow] : 0.0; });
  const Frag4<KQ, RB> aD = make_frag4<KQ, RB>(lane, [&](int mrow, int k) { return (mrow < N && k < M) ? sD12[k * N + mrow] : 0.0; });
  typedef ColLinear<M * M> CT;
  typedef Col2<M, L::C[1], L::C[2], L::C[0]> CT_out;
  typedef Col2<M, L::C[0], L::C[2], L::C[1]> CS_in;
  typedef Col2<M, L::E[0], L::E[2], L::E[1]> CS_out;
  typedef Col2<N, L::E[0], L::E[1], L::E[2]> CR_in;
  typedef ColRow<N> CR_out;
  double wn[3][RM];                                      // metrics (axis a, this component) of the component in turn
  auto load_w = [&](int c) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const int idx = r * NT + tid;
        wn[a][r] = ((zmask >> (a * 3 + c)) & 1u) ? 0.0 : *reinterpret_cast<const double*>(reinterpret_cast<const char*>(wm + (size_t)(a * 3 + c) * npr) + (unsigned)(idx < MM ? idx : 0) * 8u);
      }
  };
  load_w(0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double p0[RM], p1[RM], p2[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) { p0[r] = z[r] * wn[0][r]; p1[r] = z[r] * wn[1][r]; p2[r] = z[r] * wn[2][r]; }
    if (c < 2) load_w(c + 1);                              // in flight under this component's passes
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int idx = r * NT + tid;
      if (idx < MM) { buf[G::oP12 + idx] = p1[r]; buf[G::oP12 + MM + idx] = p2[r]; }
    }
    lds_barrier();
    mo_pass4<M, KQ, RB, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aJ, buf + G::oP12, aJ, buf, buf + G::oC1, wave, NW, lane);
    mo_pass4<M, KQ, RB, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aD, buf + G::oP12 + MM, aD, buf, buf + G::oC2, wave, NW, lane);
    lds_barrier();
    mo_pass4<M, KQ, RB, N * M, CS_in, StLin<CS_out, CS_out::ks, N>, true>(aD, buf + G::oC1, aJ, buf + G::oC2, buf + G::oE1, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int idx = r * NT + tid;
      if (idx < MM) buf[G::oP0 + idx] = p0[r];
    }
    lds_barrier();
    mo_pass4<M, KQ, RB, M * M, CT, StLin<CT_out, CT_out::ks, N>, false>(aJ, buf + G::oP0, aJ, buf, buf + G::oC0, wave, NW, lane);
    lds_barrier();
    mo_pass4<M, KQ, RB, N * M, CS_in, StLin<CS_out, CS_out::ks, N>, false>(aJ, buf + G::oC0, aJ, buf, buf + G::oE0, wave, NW, lane);
    lds_barrier();
    mo_pass4<M, KQ, RB, N * N, CR_in, StLin<CR_out, 1, N>, true>(aD, buf + G::oE0, aJ, buf + G::oE1, y + c * cs, wave, NW, lane);
    lds_barrier();
  }
}

// Weak divergence with the THREE COMPONENTS' pass chains running side by side (k_divgs_c3): waves w, w + 3, ... of the workgroup own
// component w % 3 and its buffer region (DvWave's regions: 3 x 10.5 KB at lx1 = 8), so the workgroup meets at 5 barriers instead
// of 12 and every pass has three times the tiles to spread over the waves.  The last pass stays in the accumulators and is
// combined with the metrics there (read in accumulator order by the wave that owns the tile); the three components' partial sums
// meet in LDS.  buf: 3 regions of DvWave<N>::BUF doubles, component c gathered at region c + oU.  Returns the value of Gauss node tid.
template <int N> constexpr int nt_min_sub() { return ((((N * N * N + 63) / 64)) / 3) < 1 ? 1 : (((N * N * N + 63) / 64)) / 3; }   // fewest waves a component gets
// The metrics of a wavefront's tiles of the last pass, in accumulator order: addressable from (wavefront, lane) alone, so the
// kernel issues them with its first loads (dv3_metrics) and hands them to opdiv3_mfma_c3.
template <int N> struct Dv3Met {
  static constexpr int TPW = (DvWave<N>::NT3 + (nt_min_sub<N>()) - 1) / nt_min_sub<N>();
  double wa[TPW][DvWave<N>::RQ], wb[TPW][DvWave<N>::RQ], wc[TPW][DvWave<N>::RQ];
};
template <int N>
__device__ inline void dv3_metrics(const Dev& d, long long e, int tid, int nt, Dv3Met<N>& mt) {
  using W = DvWave<N>;
  constexpr int M = N - 2, MM = M * M * M;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6, m16 = lane & 15, kq = lane >> 4;
  const int comp = wave % 3, sub = wave / 3, nsub = (nw - comp + 2) / 3;
  const bool za = (d.zmask >> (0 * 3 + comp)) & 1u, zb = (d.zmask >> (1 * 3 + comp)) & 1u, zc = (d.zmask >> (2 * 3 + comp)) & 1u;   // (Dev::zmask: zero on every node)
#pragma unroll
  for (int u = 0; u < Dv3Met<N>::TPW; ++u)
#pragma unroll
    for (int r = 0; r < W::RQ; ++r) {
      const int t = sub + u * nsub, n = t * 16 + m16, cc = kq + 4 * r;
      const bool ok = t < W::NT3 && n < M * M && cc < M;
      const size_t q = (size_t)e * MM + (ok ? cc * M * M + n : 0);
      mt.wa[u][r] = za ? 0.0 : d.w2m[(size_t)(0 * 3 + comp) * d.npr + q];
      mt.wb[u][r] = zb ? 0.0 : d.w2m[(size_t)(1 * 3 + comp) * d.npr + q];
      mt.wc[u][r] = zc ? 0.0 : d.w2m[(size_t)(2 * 3 + comp) * d.npr + q];
    }
}
template <int N>
__device__ inline double opdiv3_mfma_c3(const Dv3Met<N>& mt, const double* sJ12, const double* sD12, double* buf, int tid, int nt) {
  using W = DvWave<N>;
  constexpr int M = N - 2, MM = M * M * M, KQ = (N + 3) / 4, oPart = W::B;      // partial sums: [B, B + M^3) of the region (free after the s-axis)
  static_assert(W::B + MM <= 2 * W::A, "partial sums inside the dead sA tiles");
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6, m16 = lane & 15, kq = lane >> 4;
  const int comp = wave % 3, sub = wave / 3, nsub = (nw - comp + 2) / 3;
  double* rb = buf + comp * W::BUF;
  double aDJ[KQ], aJ[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = 4 * q + kq;
    const bool kok = k < N;
    aJ[q] = (m16 < M && kok) ? sJ12[m16 * N + k] : 0.0;
    aDJ[q] = !kok ? 0.0 : ((m16 < M) ? sD12[m16 * N + k] : ((m16 < 2 * M) ? sJ12[(m16 - M) * N + k] : 0.0));
  }
  typedef ColRow<N> CR_in;
  typedef ColRow<M> CR_out;
  typedef ColPlane<M, N * M, M> CS_in;
  typedef ColPlane<M, M * M, M> CS_out;
  typedef ColLinear<M * M> CT;
  constexpr int TPW = Dv3Met<N>::TPW;
  mo_pass<N, KQ, N * N, CR_in, StSplit<CR_out, 1, M, W::oA1 - W::oA0>, false>(aDJ, rb + W::oU, aDJ, rb, rb + W::oA0, sub, nsub, lane);
  lds_barrier();
  mo_pass<N, KQ, N * M, CS_in, StLin<CS_out, M, M>, false>(aJ, rb + W::oA0, aJ, rb, rb + W::oB0, sub, nsub, lane);
  lds_barrier();                                 // (sB_1 lands where sA_0 was: every wave is done reading it)
  mo_pass<N, KQ, N * M, CS_in, StSplit<CS_out, M, M, W::oB2 - W::oB1>, false>(aDJ, rb + W::oA1, aDJ, rb, rb + W::oB1, sub, nsub, lane);
  lds_barrier();
  double aD[KQ];                                 // D12 alone = the first M rows of the stacked fragment
#pragma unroll
  for (int q = 0; q < KQ; ++q) aD[q] = (m16 < M) ? aDJ[q] : 0.0;
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int t = sub + u * nsub;
    if (t < W::NT3) {
      const mo_d4 ur = mo_tile<N, KQ, CT>(aJ, rb + W::oB0, t, M * M, lane);
      const mo_d4 us = mo_tile<N, KQ, CT>(aJ, rb + W::oB1, t, M * M, lane);
      const mo_d4 ut = mo_tile<N, KQ, CT>(aD, rb + W::oB2, t, M * M, lane);
#pragma unroll
      for (int r = 0; r < W::RQ; ++r) {
        const int n = t * 16 + m16, cc = kq + 4 * r;
        if (n < M * M && cc < M) rb[oPart + cc * M * M + n] = mt.wa[u][r] * ur[r] + mt.wb[u][r] * us[r] + mt.wc[u][r] * ut[r];
      }
    }
  }
  lds_barrier();
  return (tid < MM) ? (buf[oPart + tid] + buf[W::BUF + oPart + tid]) + buf[2 * W::BUF + oPart + tid] : 0.0;
}

}  // namespace k3
}  // namespace nsk
