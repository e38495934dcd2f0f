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

// ---- the same kernel for orders whose column counts are no multiples of 16 (lx1 = 10, lxd = 15: 100, 150 and 225 columns) ----
// k_convect<10> (convect_lds) walks a lane's fine-mesh points one at a time with the base-flow constants loaded inside the loop, one
// workgroup per CU: 116 ms per launch at config 5's size (99 452 elements), 11 % of a time step, seventeen times its streaming floor.
// Here: every pass on the matrix cores through mo_pass (nsk3_mfma_ops.hpp: masked columns and contracted lengths), 512 threads with
// seven fine-mesh points each, and THREE LDS regions of lxd^3 doubles (81 008 B: two workgroups per CU fill its 160 KB) --
//   R1: the fine-mesh field sf / the accumulated output of a component; t1 (the [N][N][ND] intermediate) while sf is dead
//   R2: one fine-mesh derivative at a time (d/dr, d/ds, d/dt in turn: the products with the convecting field are summed in a
//       register per point); t2 ([N][ND][ND]) and the element's N^3 tile while the derivative is dead
//   R3: the accumulator of the third output component (the other two live in registers: all three there spill 280 B per lane)
// The constants of a phase are requested before the pass that precedes it (a barrier's memory clobber keeps them there).
template <int N>
__global__ __launch_bounds__(512, 4) void k_convect_mfma(Dev d, const double* __restrict__ uin, double* __restrict__ bf, int adjoint) {
  const double* __restrict__ bfc = d.bfc + (d.bf_stride ? (size_t)(*d.bstep) * (size_t)d.bf_stride : (size_t)0);      // steady set or orbit slot
  constexpr int ND = 3 * N / 2, NN = N * N * N, NDD = ND * ND * ND, NT = 512, NW = NT / 64;
  constexpr int PPT = (NDD + NT - 1) / NT, PPN = (NN + NT - 1) / NT, KQU = (N + 3) / 4, KQD = (ND + 3) / 4;
  static_assert(ND <= 16 && N * N * ND <= NDD && N * ND * ND + NN <= NDD, "one 16-row tile per operator; t1 inside R1, t2 and the element tile inside R2");
  __shared__ double R1[NDD], R2[NDD], R3[NDD];
  double* sf = R1; double* t1 = R1; double* sg = R2; double* t2 = R2; double* su = R2 + N * ND * ND;
  double* so2 = R3;                                             // the third output component's accumulator (each lane its own points)
  const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long e = blockIdx.x;
  const int m16 = lane0 & 15, kq = lane0 >> 4;
  // A fragments, constant for the whole kernel:  up: J (ND x N);  gradient: Dd (ND x ND);  down: J^T (N x ND)
  double aJ[KQU], aD[KQD];
#pragma unroll
  for (int q = 0; q < KQU; ++q) { const int k = 4 * q + kq; aJ[q] = (m16 < ND && k < N) ? d.Jd[m16 * N + k] : 0.0; }
#pragma unroll
  for (int q = 0; q < KQD; ++q) { const int k = 4 * q + kq; aD[q] = (m16 < ND && k < ND) ? d.Dd[m16 * ND + k] : 0.0; }
  typedef ColRow<N> UR_in;                       typedef ColRow<ND> UR_out;                     // [(k,j)][i] -> [(k,j)][a]
  typedef ColPlane<ND, N * ND, ND> US_in;        typedef ColPlane<ND, ND * ND, ND> US_out;      // [k][j][a]  -> [k][b][a]
  typedef ColLinear<ND * ND> UT;                                                                // [k][(b,a)] -> [c'][(b,a)]
  const size_t nf = (size_t)d.nfine;
  const double* __restrict__ bfe = bfc + (size_t)e * NDD;       // uniform base + 32-bit lane offsets (global_load saddr form)
  // byte offsets of this lane's points: tid * 8 + r * NT * 8, the last one clamped (its value is never used)
  unsigned po0 = (unsigned)tid * 8u, pol = (unsigned)((tid + (PPT - 1) * NT < NDD) ? tid + (PPT - 1) * NT : NDD - 1) * 8u;
  double o[PPT][2];
#pragma unroll
  for (int r = 0; r < PPT; ++r) { o[r][0] = o[r][1] = 0.0; const int p = tid + r * NT; if (p < NDD) so2[p] = 0.0; }
  // constant `qi` at this lane's points (not loaded where the mask says it vanishes on the whole mesh: a wave-uniform branch)
  auto ldc = [&](int qi, double (&b)[PPT]) {
    if ((d.bfmask >> qi) & 1u) {
#pragma unroll
      for (int r = 0; r < PPT; ++r) b[r] = 0.0;
    } else {
      const double* __restrict__ bq = bfe + (size_t)qi * nf;
#pragma unroll
      for (int r = 0; r < PPT; ++r) b[r] = (r < PPT - 1) ? ld_boff(bq + r * NT, po0) : ld_boff(bq, pol);
    }
  };
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    // (the lane index is made opaque per iteration: otherwise every lane-dependent LDS offset of the nine passes below is hoisted out
    //  of the loop and the kernel spills)
    int lane = lane0;
    asm volatile("" : "+v"(lane), "+v"(po0), "+v"(pol));          // (the offsets too: hoisted, the constants' loads become 21 64-bit addresses)
#pragma unroll
    for (int r = 0; r < PPN; ++r) { const int p = tid + r * NT; if (p < NN) su[p] = uin[c * d.cs + e * NN + p]; }
    lds_barrier();
    mo_pass<N, KQU, N * N, UR_in, StLin<UR_out, 1, ND>, false>(aJ, su, aJ, su, t1, wave, NW, lane);
    lds_barrier();
    mo_pass<N, KQU, N * ND, US_in, StLin<US_out, ND, ND>, false>(aJ, t1, aJ, t1, t2, wave, NW, lane);
    lds_barrier();
    mo_pass<N, KQU, ND * ND, UT, StLin<UT, ND * ND, ND>, false>(aJ, t2, aJ, t2, sf, wave, NW, lane);
    double b[PPT], conv[PPT];
    ldc(0, b);
    lds_barrier();
    // fine-mesh gradient, one direction at a time: d/dr, d/ds, d/dt
    mo_pass<ND, KQD, ND * ND, ColRow<ND>, StLin<ColRow<ND>, 1, ND>, false>(aD, sf, aD, sf, sg, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPT; ++r) { const int p = tid + r * NT; conv[r] = (p < NDD) ? b[r] * sg[p] : 0.0; }
    ldc(1, b);
    lds_barrier();
    mo_pass<ND, KQD, ND * ND, US_out, StLin<US_out, ND, ND>, false>(aD, sf, aD, sf, sg, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPT; ++r) { const int p = tid + r * NT; if (p < NDD) conv[r] += b[r] * sg[p]; }
    ldc(2, b);
    lds_barrier();
    mo_pass<ND, KQD, ND * ND, UT, StLin<UT, ND * ND, ND>, false>(aD, sf, aD, sf, sg, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int p = tid + r * NT;
      if (p < NDD) conv[r] += b[r] * sg[p];                              // (U.grad) u'_c
      const double sgn = adjoint ? -conv[r] : conv[r];
      if (c == 0) o[r][0] += sgn; else if (c == 1) o[r][1] += sgn; else if (p < NDD) so2[p] += sgn;
    }
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      // direct:  + u'_c dU_x/dx_c   (u'.grad) U ;   adjoint:  + u'_c dU_c/dx_x   (grad U)^T u'
      ldc(adjoint ? 3 + 3 * c + x : 3 + 3 * x + c, b);
#pragma unroll
      for (int r = 0; r < PPT; ++r) {
        const int p = tid + r * NT;
        if (p < NDD) { if (x < 2) o[r][x] += sf[p] * b[r]; else so2[p] += sf[p] * b[r]; }
      }
      __builtin_amdgcn_sched_barrier(0);                                // (one constant's loads in flight at a time: three would spill)
    }
    lds_barrier();
  }
  double aJt[KQD];
  {
    int ln = lane0;
    asm volatile("" : "+v"(ln));                                 // (m16 / kq recomputed: kept across the loop they are spilled)
    const int m16b = ln & 15, kqb = ln >> 4;
#pragma unroll
    for (int q = 0; q < KQD; ++q) { const int k = 4 * q + kqb; aJt[q] = (m16b < N && k < ND) ? d.Jd[k * N + m16b] : 0.0; }
  }
  double sb[PPN], un[PPN];
#pragma unroll
  for (int r = 0; r < PPN; ++r) {
    const int p = tid + r * NT;
    const long long l = e * NN + (p < NN ? p : 0);
    sb[r] = d.spng[l] * d.bm1[l];
  }
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int r = 0; r < PPT; ++r) { const int p = tid + r * NT; if (p < NDD && c < 2) sf[p] = (c == 0) ? o[r][0] : o[r][1]; }
    const double* src = (c < 2) ? sf : so2;
#pragma unroll
    for (int r = 0; r < PPN; ++r) { const int p = tid + r * NT; un[r] = uin[c * d.cs + e * NN + (p < NN ? p : 0)]; }
    lds_barrier();
    // sf[c'][b][a] -> t2[k][b][a] -> t1[(k,j)][a] -> su[(k,j)][i]
    mo_pass<ND, KQD, ND * ND, UT, StLin<UT, ND * ND, N>, false>(aJt, src, aJt, src, t2, wave, NW, lane);
    lds_barrier();
    mo_pass<ND, KQD, N * ND, US_out, StLin<US_in, ND, N>, false>(aJt, t2, aJt, t2, t1, wave, NW, lane);
    lds_barrier();
    mo_pass<ND, KQD, N * N, UR_out, StLin<UR_in, 1, N>, false>(aJt, t1, aJt, t1, su, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPN; ++r) {
      const int p = tid + r * NT;
      if (p < NN) bf[c * d.cs + e * NN + p] = -(sb[r] * un[r] + su[p]);
    }
    lds_barrier();
  }
}

// ---- the FULL equations' convection term (u . grad) u (mode 2: Newton-Krylov, core/matvec.f:64-108) on the matrix cores, lx1 = 10 ----
// [UPSTREAM advab]  v_c = sum_a c_a d(u_c)/d(xi_a),  c_a = sum_x w_d J d(xi_a)/d(x_x) u_x  on the dealiasing mesh.  The convecting field needs
// all three fine-mesh components at every point, so they all live in LDS (U: 81 KB) next to one derivative tile G (27 KB) and t2
// (18 KB): 126 KB, one 1024-thread workgroup per CU (as k_convect<10>, whose lanes walk their points one at a time with the nine
// metric terms loaded inside the loop: 116 ms per launch at config 5's size).  c_a of a lane's four points is formed ONCE (nine metric
// loads per point, less the terms that vanish on the whole mesh) and kept in registers across the three components.
//   G also holds t1 on the way up (dead before the first derivative) and v_c, then t1, on the way down; the element tile rides in t2's place.
template <int N>
__global__ __launch_bounds__(1024, 4) void k_convect_mfma_nl(Dev d, const double* __restrict__ uin, double* __restrict__ bf) {
  constexpr int ND = 3 * N / 2, NN = N * N * N, NDD = ND * ND * ND, NT = 1024, NW = NT / 64;
  constexpr int PPT = (NDD + NT - 1) / NT, KQU = (N + 3) / 4, KQD = (ND + 3) / 4;
  static_assert(ND <= 16 && N * N * ND <= NDD && NN <= N * ND * ND && NN <= NT, "one 16-row tile per operator; t1 inside G, the element tile inside t2, one GLL node per thread");
  __shared__ double U[3 * NDD], G[NDD], T2[N * ND * ND];
  double* t1 = G; double* su = T2;
  const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long e = blockIdx.x;
  double aJ[KQU], aD[KQD];
  {
    const int m16 = lane0 & 15, kq = lane0 >> 4;
#pragma unroll
    for (int q = 0; q < KQU; ++q) { const int k = 4 * q + kq; aJ[q] = (m16 < ND && k < N) ? d.Jd[m16 * N + k] : 0.0; }
#pragma unroll
    for (int q = 0; q < KQD; ++q) { const int k = 4 * q + kq; aD[q] = (m16 < ND && k < ND) ? d.Dd[m16 * ND + k] : 0.0; }
  }
  typedef ColRow<N> UR_in;                       typedef ColRow<ND> UR_out;                     // [(k,j)][i] -> [(k,j)][a]
  typedef ColPlane<ND, N * ND, ND> US_in;        typedef ColPlane<ND, ND * ND, ND> US_out;      // [k][j][a]  -> [k][b][a]
  typedef ColLinear<ND * ND> UT;                                                                // [k][(b,a)] -> [c'][(b,a)]
  const size_t nf = (size_t)d.nfine;
  // ---- the three components on the dealiasing mesh
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    if (tid < NN) su[tid] = uin[c * d.cs + e * NN + tid];
    lds_barrier();
    mo_pass<N, KQU, N * N, UR_in, StLin<UR_out, 1, ND>, false>(aJ, su, aJ, su, t1, wave, NW, lane);
    lds_barrier();
    mo_pass<N, KQU, N * ND, US_in, StLin<US_out, ND, ND>, false>(aJ, t1, aJ, t1, T2, wave, NW, lane);
    lds_barrier();
    mo_pass<N, KQU, ND * ND, UT, StLin<UT, ND * ND, ND>, false>(aJ, T2, aJ, T2, U + c * NDD, wave, NW, lane);
    lds_barrier();
  }
  // ---- the convecting field at this lane's points: c_a = sum_x mtd[a][x] u_x
  double ca[PPT][3];
  {
    const double* __restrict__ me = d.mtd + (size_t)e * NDD;
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int p = tid + r * NT;
      const unsigned po = (unsigned)(p < NDD ? p : NDD - 1) * 8u;
      const double u0 = U[p < NDD ? p : 0], u1 = U[NDD + (p < NDD ? p : 0)], u2 = U[2 * NDD + (p < NDD ? p : 0)];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double m0 = ((d.zmask >> (a * 3 + 0)) & 1u) ? 0.0 : ld_boff(me + (size_t)(a * 3 + 0) * nf, po);      // (zmask: wave-uniform branches)
        const double m1 = ((d.zmask >> (a * 3 + 1)) & 1u) ? 0.0 : ld_boff(me + (size_t)(a * 3 + 1) * nf, po);
        const double m2 = ((d.zmask >> (a * 3 + 2)) & 1u) ? 0.0 : ld_boff(me + (size_t)(a * 3 + 2) * nf, po);
        ca[r][a] = (m0 * u0 + m1 * u1) + m2 * u2;
      }
    }
  }
  double aJt[KQD];
  {
    const int m16 = lane0 & 15, kq = lane0 >> 4;
#pragma unroll
    for (int q = 0; q < KQD; ++q) { const int k = 4 * q + kq; aJt[q] = (m16 < N && k < ND) ? d.Jd[k * N + m16] : 0.0; }
  }
  const long long l = e * NN + (tid < NN ? tid : 0);
  const double kk = d.spng[l] * d.bm1[l] * d.nl_spng_str;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const double* uc = U + c * NDD;
    double v[PPT];
    // d/dr, d/ds, d/dt of component c, one at a time through G
    mo_pass<ND, KQD, ND * ND, ColRow<ND>, StLin<ColRow<ND>, 1, ND>, false>(aD, uc, aD, uc, G, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPT; ++r) { const int p = tid + r * NT; v[r] = (p < NDD) ? ca[r][0] * G[p] : 0.0; }
    lds_barrier();
    mo_pass<ND, KQD, ND * ND, US_out, StLin<US_out, ND, ND>, false>(aD, uc, aD, uc, G, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPT; ++r) { const int p = tid + r * NT; if (p < NDD) v[r] += ca[r][1] * G[p]; }
    lds_barrier();
    mo_pass<ND, KQD, ND * ND, UT, StLin<UT, ND * ND, ND>, false>(aD, uc, aD, uc, G, wave, NW, lane);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPT; ++r) { const int p = tid + r * NT; if (p < NDD) v[r] += ca[r][2] * G[p]; }
    lds_barrier();                                               // (every lane has read the derivative: G takes v_c)
#pragma unroll
    for (int r = 0; r < PPT; ++r) { const int p = tid + r * NT; if (p < NDD) G[p] = v[r]; }
    lds_barrier();
    // v[c'][b][a] -> t2[k][b][a] -> t1[(k,j)][a] (in G: v is dead) -> su[(k,j)][i] (in t2's place)
    mo_pass<ND, KQD, ND * ND, UT, StLin<UT, ND * ND, N>, false>(aJt, G, aJt, G, T2, wave, NW, lane);
    lds_barrier();
    mo_pass<ND, KQD, N * ND, US_out, StLin<US_in, ND, N>, false>(aJt, T2, aJt, T2, t1, wave, NW, lane);
    lds_barrier();
    mo_pass<ND, KQD, N * N, UR_out, StLin<UR_in, 1, N>, false>(aJt, t1, aJt, t1, su, wave, NW, lane);
    lds_barrier();
    if (tid < NN) {
      const double s = su[tid];
      bf[c * d.cs + l] = ((kk != 0.0) ? kk * (d.spng_vr[c * d.cs + l] - uin[c * d.cs + l]) : 0.0) - s;
    }
    lds_barrier();
  }
}

}  // namespace k3
}  // namespace nsk
