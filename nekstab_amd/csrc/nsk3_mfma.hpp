// Hexahedral convection kernel with the tensor contractions on the fp64 matrix cores (lx1 = 8, lxd = 12).
//
// north_star: "MFMA used only for the small dense N x N x N tensor contractions".  Every pass of the dealiasing
// interpolation (8 -> 12 points per direction), of the fine-mesh gradient (12 x 12 derivative matrix) and of the
// projection back (12 -> 8) is OUT[m][n] = sum_k A[m][k] X[k][n] with a small constant A and the other two tensor
// indices flattened into n (64, 96 or 144 columns: multiples of 16 at this order, which is why this kernel exists for
// lx1 = 8 only).  v_mfma_f64_16x16x4_f64 computes a 16 x 16 tile of OUT per K = 4: A rows padded 12 -> 16 (75 % tile use
// going up and for the gradient) or 8 -> 16 (50 % coming down).  The thread-per-node form (k3::k_convect) issues two LDS
// reads per multiply-add and is bound by the LDS pipe; here every X element is read from LDS once per tile (one
// ds_read_b64 per lane per MFMA) and reused 16 times from registers, A never leaves registers.
//   fragment layouts (as k_basis_gemm_mfma): A[lane & 15][lane >> 4], B[lane >> 4][lane & 15], D: col = lane & 15,
//   row = (lane >> 4) + 4 * reg.
// Modes 0 (direct) and 1 (adjoint); the full-equation mode keeps the thread-per-node kernel.
#pragma once
#include "nsk3_kernels.hpp"

namespace nsk {
namespace k3 {

typedef double mf_d4 __attribute__((ext_vector_type(4)));

// column n -> LDS offset, plus the stride of the contracted (input) / produced (output) index
template <int CS, int KS> struct ColLin { static constexpr int ks = KS; static __device__ inline int col(int n) { return n * CS; } };           // n * CS
template <int DIV, int HI, int KS> struct ColSplit { static constexpr int ks = KS; static __device__ inline int col(int n) { return (n / DIV) * HI + (n % DIV); } };   // (n / DIV) * HI + n % DIV

// OUT[m][n] = sum_k A[m][k] X[k][n]:  m < MR <= 16, k < 4 * KQ, n < NCOL (multiple of 16); tiles of 16 columns over the waves
template <int MR, int KQ, int NCOL, class IN, class OT>
__device__ inline void mfma_pass(const double (&afrag)[KQ], const double* X, double* OUT, int wave, int nwaves, int lane) {
  const int n16 = lane & 15, kq = lane >> 4;
  for (int tile = wave; tile < NCOL / 16; tile += nwaves) {
    const int n = tile * 16 + n16;
    const double* xb = X + IN::col(n) + kq * IN::ks;
    mf_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KQ; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[q], xb[4 * q * IN::ks], acc, 0, 0, 0);
    double* ob = OUT + OT::col(n);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = kq + 4 * r;
      if (m < MR) ob[m * OT::ms_()] = acc[r];
    }
  }
}
// output descriptors reuse the column maps; `ks` of an output descriptor is the stride of the produced index m
template <int CS, int MS> struct OutLin : ColLin<CS, MS> { static __device__ inline constexpr int ms_() { return MS; } };
template <int DIV, int HI, int MS> struct OutSplit : ColSplit<DIV, HI, MS> { static __device__ inline constexpr int ms_() { return MS; } };

__global__ __launch_bounds__(512) void k_convect_mfma8(Dev d, const double* __restrict__ uin, double* __restrict__ bf, int adjoint) {
  const double* __restrict__ bfc = d.bfc + (d.bf_stride ? (size_t)(*d.bstep) * (size_t)d.bf_stride : (size_t)0);      // steady set or orbit slot
  constexpr int N = 8, ND = 12, NN = 512, NDD = 1728, NT = 512, PPT = 4, NW = 8;
  __shared__ double su[NN], t1[N * N * ND], t2[N * ND * ND], sf[NDD], gr[NDD], gs[NDD], gt[NDD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long e = blockIdx.x;
  const int m16 = lane & 15, kq = lane >> 4;
  // A fragments, constant for the whole kernel:  up: J (12 x 8);  gradient: Dd (12 x 12);  down: J^T (8 x 12)
  double aJ[2], aD[3], aJt[3];
#pragma unroll
  for (int q = 0; q < 2; ++q) aJ[q] = (m16 < ND) ? d.Jd[m16 * N + 4 * q + kq] : 0.0;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    aD[q] = (m16 < ND) ? d.Dd[m16 * ND + 4 * q + kq] : 0.0;
    aJt[q] = (m16 < N) ? d.Jd[(4 * q + kq) * N + m16] : 0.0;
  }
  const size_t nf = (size_t)d.nfine;
  double o[PPT][3];
#pragma unroll
  for (int r = 0; r < PPT; ++r) o[r][0] = o[r][1] = o[r][2] = 0.0;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    su[tid] = uin[c * d.cs + e * NN + tid];
    lds_barrier();
    // u[(k,j)][i] -> t1[(k,j)][a] -> t2[k][b][a] -> sf[c'][b][a]
    mfma_pass<ND, 2, 64, ColLin<N, 1>, OutLin<ND, 1>>(aJ, su, t1, wave, NW, lane);
    lds_barrier();
    mfma_pass<ND, 2, 96, ColSplit<ND, N * ND, ND>, OutSplit<ND, ND * ND, ND>>(aJ, t1, t2, wave, NW, lane);
    lds_barrier();
    mfma_pass<ND, 2, 144, ColLin<1, ND * ND>, OutLin<1, ND * ND>>(aJ, t2, sf, wave, NW, lane);
    lds_barrier();
    // fine-mesh gradient: d/dr, d/ds, d/dt
    mfma_pass<ND, 3, 144, ColLin<ND, 1>, OutLin<ND, 1>>(aD, sf, gr, wave, NW, lane);
    mfma_pass<ND, 3, 144, ColSplit<ND, ND * ND, ND>, OutSplit<ND, ND * ND, ND>>(aD, sf, gs, wave, NW, lane);
    mfma_pass<ND, 3, 144, ColLin<1, ND * ND>, OutLin<1, ND * ND>>(aD, sf, gt, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int p = tid + r * NT;
      if (p < NDD) {
        const double g0 = gr[p], g1 = gs[p], g2 = gt[p], uf = sf[p];
        const size_t q = (size_t)e * NDD + p;
        const double conv = ld_bfc(d, bfc, 0, nf, q) * g0 + ld_bfc(d, bfc, 1, nf, q) * g1 + ld_bfc(d, bfc, 2, nf, q) * g2;   // (U.grad) u'_c
        const double sg = adjoint ? -conv : conv;
        if (c == 0) o[r][0] += sg; else if (c == 1) o[r][1] += sg; else o[r][2] += sg;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          const double G = ld_bfc(d, bfc, adjoint ? 3 + 3 * c + x : 3 + 3 * x + c, nf, q);
          o[r][x] += uf * G;
        }
      }
    }
    lds_barrier();
  }
  const long long l = e * NN + tid;
  const double sb = d.spng[l] * d.bm1[l];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int p = tid + r * NT;
      if (p < NDD) sf[p] = o[r][c];
    }
    lds_barrier();
    // sf[c'][b][a] -> t2[k][b][a] -> t1[(k,j)][a] -> su[(k,j)][i]
    mfma_pass<N, 3, 144, ColLin<1, ND * ND>, OutLin<1, ND * ND>>(aJt, sf, t2, wave, NW, lane);
    lds_barrier();
    mfma_pass<N, 3, 96, ColSplit<ND, ND * ND, ND>, OutSplit<ND, N * ND, ND>>(aJt, t2, t1, wave, NW, lane);
    lds_barrier();
    mfma_pass<N, 3, 64, ColLin<ND, 1>, OutLin<N, 1>>(aJt, t1, su, wave, NW, lane);
    lds_barrier();
    const double s = su[tid];
    const double un = uin[c * d.cs + l];
    bf[c * d.cs + l] = -(sb * un + s);
    lds_barrier();
  }
}

}  // namespace k3
}  // namespace nsk
