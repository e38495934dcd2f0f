// HIP kernels (gfx950) of the linearised PnPn-2 time step on hexahedral elements: the 3-D
// counterparts of nsk_kernels.hpp, same kernel names and launch signatures (namespace nsk::k3),
// so the host driver (step graphs, adaptive budgets, Krylov algebra) is shared.
// One workgroup per element, one thread per GLL node (lx1 = 8: 512 threads = 8 wavefronts),
// tensor contractions in three LDS passes, dssum as the same fused gather.
#pragma once
#include "nsk_kernels.hpp"
#include "nsk3_mfma_ops.hpp"

namespace nsk {
namespace k3 {

// Nodes of valence 5..8 (hexahedral vertices) do not fit the 4-wide gather table.  Their index lists
// are the same for every component and iteration, so a launch stages them once in LDS (8 ints per
// thread, private to the thread: no barrier) and every later gather issues its value loads at once
// instead of walking offsets -> indices -> values.  Same left-to-right sum as gs_csr.
__device__ inline void gs_wide_stage(const Dev& d, const int4 tab, long long l, int* sw, const int4* corner = nullptr) {
  if (corner) {                                  // element corner: its list is addressed directly (issued before tab is known)
    const int4 a = corner[0], b = corner[1];
    if (a.x != -2) {
      if (tab.x >= 0) return;
      sw[0] = a.x; sw[1] = a.y; sw[2] = a.z; sw[3] = a.w; sw[4] = b.x; sw[5] = b.y; sw[6] = b.z; sw[7] = b.w;
      return;
    }
  }
  if (tab.x >= 0) return;
  const int o0 = d.gs_off[l], o1 = d.gs_off[l + 1];
  if (o1 - o0 > 8) { sw[0] = -2; return; }
  int id[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) id[q] = (o0 + q < o1) ? d.gs_idx[o0 + q] : -1;
#pragma unroll
  for (int q = 0; q < 8; ++q) sw[q] = id[q];
}
__device__ inline double gs_wide_sum(const double* __restrict__ f, const Dev& d, long long l, const int* sw) {
  int id[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) id[q] = sw[q];
  if (id[0] == -2) return gs_csr(f, d, l);
  double v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = (id[q] >= 0) ? f[id[q]] : 0.0;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (id[q] >= 0) s += v[q];
  return s;
}

// gs_csr as a plain loop, one entry per trip: for the paths a regular mesh never takes (its eight-wide form costs the kernel
// around it twenty registers)
__device__ inline double gs_csr_rolled(const double* __restrict__ f, const Dev& d, long long l) {
  const int o0 = d.gs_off[l], o1 = d.gs_off[l + 1];
  double s = 0.0;
#pragma unroll 1
  for (int k = o0; k < o1; ++k) s += f[d.gs_idx[k]];
  return s;
}

// Element corners are the nodes of valence 5..8 of a regular hexahedral mesh.  Their lists live in a table addressed by
// (element, corner) -- known from the thread index -- so the index loads go out with the kernel's first loads instead of
// behind gs_tab -> gs_off -> gs_idx (measured at config 4's size, k_divgs: the loading phase of a workgroup took 10.4 us = five
// dependent round trips, all 512 threads waiting at the barrier for the 8 corner threads).  Same left-to-right sum as gs_csr.
template <int N>
__device__ inline int corner_id(int k, int j, int i) {
  const bool c = (i == 0 || i == N - 1) && (j == 0 || j == N - 1) && (k == 0 || k == N - 1);
  return c ? ((k ? 4 : 0) | (j ? 2 : 0) | (i ? 1 : 0)) : -1;
}
struct CornerList { int id[8]; };
__device__ inline void corner_issue(const Dev& d, long long e, int c, CornerList& L) {
  if (c < 0 || !d.gs_corner) { L.id[0] = -2; return; }
  const int4* p = reinterpret_cast<const int4*>(d.gs_corner + ((size_t)e * 8 + c) * 8);
  const int4 a = p[0], b = p[1];
  L.id[0] = a.x; L.id[1] = a.y; L.id[2] = a.z; L.id[3] = a.w; L.id[4] = b.x; L.id[5] = b.y; L.id[6] = b.z; L.id[7] = b.w;
}
__device__ inline double corner_sum(const double* __restrict__ f, const Dev& d, long long l, const CornerList& L) {
  if (L.id[0] == -2) return gs_csr(f, d, l);
  double v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = (L.id[q] >= 0) ? f[L.id[q]] : 0.0;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (L.id[q] >= 0) s += v[q];
  return s;
}
// gs_sum with the corner lists for the wide nodes
__device__ inline double gs_sum3(const GsVals& v, const double* __restrict__ f, const Dev& d, const int4 t, long long l, const CornerList& L) {
  if (t.x < 0) return corner_sum(f, d, l, L);
  return ((v.a + v.b) + v.c) + v.d;
}

// uniform base (scalar registers) + 32-bit byte offset (one vector register for every basis vector of a node): the
// global_load saddr form; a 64-bit address per load costs two vector registers per load in flight
__device__ inline double ld_boff(const double* __restrict__ base, unsigned boff) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + boff);
}
__device__ inline void st_boff(double* __restrict__ base, unsigned boff, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + boff) = v;
}
// gs_load with the uniform field base in scalar registers and 32-bit byte offsets (node indices * 8 fit 32 bits up to 5e8 nodes):
// half the address registers of the 64-bit form when eight nodes per lane are in flight
__device__ inline GsVals gs_load_o(const double* __restrict__ f, const int4 t, unsigned l) {
  GsVals v;
  v.a = ld_boff(f, (unsigned)(t.x >= 0 ? t.x : (int)l) * 8u);
  v.b = (t.y >= 0) ? ld_boff(f, (unsigned)t.y * 8u) : 0.0;
  v.c = (t.z >= 0) ? ld_boff(f, (unsigned)t.z * 8u) : 0.0;
  v.d = (t.w >= 0) ? ld_boff(f, (unsigned)t.w * 8u) : 0.0;
  return v;
}
template <int N>
struct Cfg {
  static constexpr int NN = N * N * N, M = N - 2, MM = M * M * M, ND = 3 * N / 2, NDD = ND * ND * ND;
  static constexpr int EPB = 1;
  static constexpr int NT = ((NN + 63) / 64) * 64;
  static constexpr int NTD = NT;
  static constexpr int NNM = N * N * M, NMM = N * M * M;            // intermediate tile sizes GLL <-> Gauss
};

// ---------------------------------------------------------------------------
// element-local building blocks (one element per workgroup, LDS tiles [k][j][i])
// ---------------------------------------------------------------------------
// D^T G D on NC tiles su[c][NN]; st scratch [NC][3][NN]   [UPSTREAM hmholtz.f axhelm, 3-D branch]
// g = (G1..G6) = (rr, ss, tt, rs, rt, st)
template <int N, int NC>
__device__ inline void axhelm3(const double* sD, const double* sDt, const double* su, double* st, bool act,
                               int k, int j, int i, const double (&g)[6], double (&out)[NC]) {
  constexpr int NN = N * N * N;
  const int t = (k * N + j) * N + i;
  if (act) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double* z = su + c * NN;
      double ur = 0.0, us = 0.0, ut = 0.0;
#pragma unroll
      for (int m = 0; m < N; ++m) {
        ur += sDt[m * N + i] * z[(k * N + j) * N + m];
        us += sDt[m * N + j] * z[(k * N + m) * N + i];
        ut += sDt[m * N + k] * z[(m * N + j) * N + i];
      }
      st[(c * 3 + 0) * NN + t] = g[0] * ur + g[3] * us + g[4] * ut;
      st[(c * 3 + 1) * NN + t] = g[3] * ur + g[1] * us + g[5] * ut;
      st[(c * 3 + 2) * NN + t] = g[4] * ur + g[5] * us + g[2] * ut;
    }
  }
  lds_barrier();
  if (act) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double* t1 = st + (c * 3 + 0) * NN;
      const double* t2 = st + (c * 3 + 1) * NN;
      const double* t3 = st + (c * 3 + 2) * NN;
      double au = 0.0;
#pragma unroll
      for (int m = 0; m < N; ++m)
        au += sD[m * N + i] * t1[(k * N + j) * N + m] + sD[m * N + j] * t2[(k * N + m) * N + i] + sD[m * N + k] * t3[(m * N + j) * N + i];
      out[c] = au;
    }
  }
}

// weak divergence GLL -> Gauss  [UPSTREAM navier1.f opdiv/multd, 3-D]; su = [3][NN]
// scratch sA [2][NNM], sB [3][NMM]; w2m = [9][npr] metrics (a*3+c) premultiplied by the Gauss weights.
// Returns the value for Gauss node tid (< MM).
template <int N>
__device__ inline double opdiv3(const double* sJ12, const double* sD12, const double* su, double* sA, double* sB,
                                int tid, int nt, const double (&w2)[9]) {
  constexpr int NN = N * N * N, M = N - 2, MM = M * M * M, NNM = N * N * M, NMM = N * M * M;
#ifndef NSK_NO_MFMA_OPS
  return opdiv3_mfma<N>(sJ12, sD12, su, sA, sB, tid, nt, w2);      // matrix cores (nsk3_mfma_ops.hpp)
#endif
  double div = 0.0;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const double* u = su + c * NN;
    for (int p = tid; p < NNM; p += nt) {            // axis r: [k][j][a]
      const int a = p % M, kj = p / M;
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) { const double x = u[kj * N + i]; s1 += sD12[a * N + i] * x; s2 += sJ12[a * N + i] * x; }
      sA[p] = s1; sA[NNM + p] = s2;
    }
    lds_barrier();
    for (int p = tid; p < NMM; p += nt) {            // axis s: [k][b][a]
      const int a = p % M, b = (p / M) % M, k = p / (M * M);
      double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const double x1 = sA[(k * N + j) * M + a], x2 = sA[NNM + (k * N + j) * M + a];
        s1 += sJ12[b * N + j] * x1; s2 += sD12[b * N + j] * x2; s3 += sJ12[b * N + j] * x2;
      }
      sB[p] = s1; sB[NMM + p] = s2; sB[2 * NMM + p] = s3;
    }
    lds_barrier();
    if (tid < MM) {                                  // axis t
      const int ba = tid % (M * M), cc = tid / (M * M);
      double ur = 0.0, us = 0.0, ut = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        ur += sJ12[cc * N + k] * sB[k * M * M + ba];
        us += sJ12[cc * N + k] * sB[NMM + k * M * M + ba];
        ut += sD12[cc * N + k] * sB[2 * NMM + k * M * M + ba];
      }
      div += w2[0 * 3 + c] * ur + w2[1 * 3 + c] * us + w2[2 * 3 + c] * ut;
    }
  }
  return div;
}

// D^T p  [UPSTREAM navier1.f opgradt/cdtp, 3-D]; pw = p (x Gauss weights are inside w2) at Gauss node tid.
// scratch sP [3][MM], sC [3][NMM], sE [2][NNM]
template <int N>
__device__ inline void opgradt3(const double* sJ12, const double* sD12, double pval, const double (&w2)[9], double* sP,
                                double* sC, double* sE, int tid, int nt, bool act, int k, int j, int i, double (&g)[3]) {
  constexpr int M = N - 2, MM = M * M * M, NNM = N * N * M, NMM = N * M * M;
#ifndef NSK_NO_MFMA_OPS
  opgradt3_mfma<N>(sJ12, sD12, pval, w2, sP, sC, sE, tid, nt, g); return;      // matrix cores (nsk3_mfma_ops.hpp)
#endif
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    if (tid < MM) {
      sP[tid] = pval * w2[0 * 3 + c];
      sP[MM + tid] = pval * w2[1 * 3 + c];
      sP[2 * MM + tid] = pval * w2[2 * 3 + c];
    }
    lds_barrier();
    for (int p = tid; p < NMM; p += nt) {            // axis t: [kk][b][a]
      const int ba = p % (M * M), kk = p / (M * M);
      double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int cc = 0; cc < M; ++cc) {
        const double jj = sJ12[cc * N + kk], dd = sD12[cc * N + kk];
        s1 += jj * sP[cc * M * M + ba]; s2 += jj * sP[MM + cc * M * M + ba]; s3 += dd * sP[2 * MM + cc * M * M + ba];
      }
      sC[p] = s1; sC[NMM + p] = s2; sC[2 * NMM + p] = s3;
    }
    lds_barrier();
    for (int p = tid; p < NNM; p += nt) {            // axis s: [kk][jj][a]
      const int a = p % M, jj = (p / M) % N, kk = p / (M * N);
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int b = 0; b < M; ++b) {
        const double jw = sJ12[b * N + jj], dw = sD12[b * N + jj];
        s1 += jw * sC[(kk * M + b) * M + a];
        s2 += dw * sC[NMM + (kk * M + b) * M + a] + jw * sC[2 * NMM + (kk * M + b) * M + a];
      }
      sE[p] = s1; sE[NNM + p] = s2;
    }
    lds_barrier();
    double s = 0.0;
    if (act) {
#pragma unroll
      for (int a = 0; a < M; ++a) s += sD12[a * N + i] * sE[(k * N + j) * M + a] + sJ12[a * N + i] * sE[NNM + (k * N + j) * M + a];
    }
    g[c] = s;
  }
}

template <int N>
__device__ inline void load_basis3(const Dev& d, double* sD, double* sDt, double* sJ12, double* sD12, int tid, int nt) {
  constexpr int M = N - 2;
  for (int k = tid; k < N * N; k += nt) {
    if (sD) sD[k] = d.D[k];
    if (sDt) sDt[(k % N) * N + k / N] = d.D[k];
  }
  if (sJ12)
    for (int k = tid; k < M * N; k += nt) { sJ12[k] = d.J12[k]; sD12[k] = d.D12[k]; }
}

// Metric and G arrays that are zero on every node (Dev::zmask: a spanwise-extruded or a box mesh) are not loaded: 4 of the 9
// metric terms and 2 of the 6 G factors at config 4, 6 and 3 at config 5.  The branch is on a kernel argument: wave-uniform.
__device__ inline void load_w2(const Dev& d, long long q, double (&w2)[9]) {
#pragma unroll
  for (int a = 0; a < 9; ++a) w2[a] = ((d.zmask >> a) & 1u) ? 0.0 : d.w2m[(size_t)a * d.npr + q];
}
__device__ inline void load_g6(const Dev& d, long long l, double (&g)[6]) {
  g[0] = d.g1[l]; g[1] = d.g2[l]; g[2] = d.g3[l];
  g[3] = (d.zmask & (1u << 9)) ? 0.0 : d.g4[l];
  g[4] = (d.zmask & (1u << 10)) ? 0.0 : d.g5[l];
  g[5] = (d.zmask & (1u << 11)) ? 0.0 : d.g6[l];
}
// constant q of the twelve base-flow constants on the dealiasing mesh (Dev::bfmask: a two-dimensional base flow on an extruded mesh
// leaves 6 of the 12 zero)
__device__ inline double ld_bfc(const Dev& d, const double* __restrict__ bfc, int q, size_t nf, size_t node) {
  return ((d.bfmask >> q) & 1u) ? 0.0 : bfc[(size_t)q * nf + node];
}

// interpolation GLL -> dealiasing mesh of one component: in (global, [NN]) -> sf [NDD]; scratch t1 [N*N*ND], t2 [N*ND*ND]
template <int N>
__device__ inline void to_fine(const double* sJ, const double* __restrict__ uin, double* sf, double* t1, double* t2,
                               int tid, int nt) {
  constexpr int NN = N * N * N, ND = 3 * N / 2;
  for (int p = tid; p < NN; p += nt) t2[p] = uin[p];
  lds_barrier();
  for (int p = tid; p < N * N * ND; p += nt) {       // r: [k][j][a]
    const int a = p % ND, kj = p / ND;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += sJ[a * N + i] * t2[kj * N + i];
    t1[p] = s;
  }
  lds_barrier();
  for (int p = tid; p < N * ND * ND; p += nt) {      // s: [k][b][a]
    const int a = p % ND, b = (p / ND) % ND, k = p / (ND * ND);
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) s += sJ[b * N + j] * t1[(k * N + j) * ND + a];
    t2[p] = s;
  }
  lds_barrier();
  for (int p = tid; p < ND * ND * ND; p += nt) {     // t: [c][b][a]
    const int ba = p % (ND * ND), c = p / (ND * ND);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) s += sJ[c * N + k] * t2[k * ND * ND + ba];
    sf[p] = s;
  }
  lds_barrier();
}

// transpose of to_fine: sf [NDD] -> value at GLL node (k,j,i) of this thread
template <int N>
__device__ inline double from_fine(const double* sJ, const double* sf, double* t1, double* t2, int tid, int nt,
                                   bool act, int k, int j, int i) {
  constexpr int ND = 3 * N / 2;
  for (int p = tid; p < N * ND * ND; p += nt) {      // t: [k][b][a]
    const int ba = p % (ND * ND), kk = p / (ND * ND);
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < ND; ++c) s += sJ[c * N + kk] * sf[c * ND * ND + ba];
    t2[p] = s;
  }
  lds_barrier();
  for (int p = tid; p < N * N * ND; p += nt) {       // s: [k][j][a]
    const int a = p % ND, jj = (p / ND) % N, kk = p / (ND * N);
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < ND; ++b) s += sJ[b * N + jj] * t2[(kk * ND + b) * ND + a];
    t1[p] = s;
  }
  lds_barrier();
  double s = 0.0;
  if (act) {
#pragma unroll
    for (int a = 0; a < ND; ++a) s += sJ[a * N + i] * t1[(k * N + j) * ND + a];
  }
  lds_barrier();
  return s;
}

// ---------------------------------------------------------------------------
// K1: forcing + dealiased convection -> bf (mass weighted)   [UPSTREAM perturb.f advabp /
// advabp_adjoint, convect.f convect_new / convect_adj; sponge: core/utils.f:172-177]
//   bfc = [12][nfine]: 0..2 = w_d J (U . grad xi_a);  3+3c+x = w_d J dU_c/dx_x
//   mtd = [9][nfine]:  a*3+x = w_d J d(xi_a)/d(x_x)          (full equations, mode 2)
// ---------------------------------------------------------------------------
// One component at a time lives on the dealiasing mesh (sf = 1 x lxd^3: 27 KB at lx1 = 10), the pointwise products
// are accumulated in registers across the component loop.
template <int N>
__device__ inline void fine_grad(const double* sDd, const double* f, int a, int b, int cc, double (&g)[3]) {
  constexpr int ND = 3 * N / 2;
  double ur = 0, us = 0, ut = 0;
#pragma unroll
  for (int m = 0; m < ND; ++m) {
    ur += sDd[a * ND + m] * f[(cc * ND + b) * ND + m];
    us += sDd[b * ND + m] * f[(cc * ND + m) * ND + a];
    ut += sDd[cc * ND + m] * f[(m * ND + b) * ND + a];
  }
  g[0] = ur; g[1] = us; g[2] = ut;
}

// lx1 = 10: 1024 threads per workgroup leave 128 registers per lane; the register form below (twelve to twenty-four
// accumulators per lane, four fine-mesh points per lane unrolled) spilled 272 bytes per lane there.  This form keeps the
// three output components (linearised maps) or the three fine-mesh velocity components (full equations) in LDS -- 81 KB next
// to the 60 KB of tiles, one workgroup per CU either way -- and walks a lane's points one at a time: no scratch.
template <int N>
__device__ inline void convect_lds(const Dev& d, const double* __restrict__ uin, double* __restrict__ bf, int adjoint) {
  // base-flow constants: the steady set, or slot `*bstep` of the stored periodic orbit (Floquet, core/matvec.f:200-236)
  const double* __restrict__ bfc = d.bfc + (d.bf_stride ? (size_t)(*d.bstep) * (size_t)d.bf_stride : (size_t)0);
  using C = Cfg<N>;
  constexpr int NN = C::NN, ND = C::ND, NDD = C::NDD, NT = C::NT;
  __shared__ double sJ[ND * N], sDd[ND * ND];
  __shared__ double sf[NDD], t1[N * N * ND], t2[N * ND * ND], so[3 * NDD];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  for (int p = tid; p < ND * N; p += NT) sJ[p] = d.Jd[p];
  for (int p = tid; p < ND * ND; p += NT) sDd[p] = d.Dd[p];
  const size_t nf = (size_t)d.nfine;
  const long long l = e * NN + tid;
  double sb = 0.0;
  if (act) sb = d.spng[l] * d.bm1[l];
  if (adjoint == 2) {                                  // full equations   [UPSTREAM advab]
#pragma unroll 1
    for (int c = 0; c < 3; ++c) to_fine<N>(sJ, uin + c * d.cs + e * NN, so + c * NDD, t1, t2, tid, NT);
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
#pragma unroll 1
      for (int p = tid; p < NDD; p += NT) {
        const int a = p % ND, b = (p / ND) % ND, cc = p / (ND * ND);
        double g[3];
        fine_grad<N>(sDd, so + c * NDD, a, b, cc, g);
        const size_t q = (size_t)e * NDD + p;
        const double u0 = so[p], u1 = so[NDD + p], u2 = so[2 * NDD + p];
        double v = 0.0;
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2)               // convecting field c_a = w_d J (u . grad xi_a)
          v += ((((d.zmask >> (a2 * 3 + 0)) & 1u) ? 0.0 : d.mtd[(a2 * 3 + 0) * nf + q]) * u0 + (((d.zmask >> (a2 * 3 + 1)) & 1u) ? 0.0 : d.mtd[(a2 * 3 + 1) * nf + q]) * u1 +
                (((d.zmask >> (a2 * 3 + 2)) & 1u) ? 0.0 : d.mtd[(a2 * 3 + 2) * nf + q]) * u2) * g[a2];
        sf[p] = v;
      }
      lds_barrier();
      const double s = from_fine<N>(sJ, sf, t1, t2, tid, NT, act, k, j, i);
      if (act) {
        const double kk = sb * d.nl_spng_str;
        bf[c * d.cs + l] = ((kk != 0.0) ? kk * (d.spng_vr[c * d.cs + l] - uin[c * d.cs + l]) : 0.0) - s;
      }
    }
    return;
  }
  for (int p = tid; p < 3 * NDD; p += NT) so[p] = 0.0;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    to_fine<N>(sJ, uin + c * d.cs + e * NN, sf, t1, t2, tid, NT);
#pragma unroll 1
    for (int p = tid; p < NDD; p += NT) {
      const int a = p % ND, b = (p / ND) % ND, cc = p / (ND * ND);
      double g[3];
      fine_grad<N>(sDd, sf, a, b, cc, g);
      const double uf = sf[p];
      const size_t q = (size_t)e * NDD + p;
      const double conv = ld_bfc(d, bfc, 0, nf, q) * g[0] + ld_bfc(d, bfc, 1, nf, q) * g[1] + ld_bfc(d, bfc, 2, nf, q) * g[2];   // (U.grad) u'_c
      so[c * NDD + p] += adjoint ? -conv : conv;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        // direct:  + u'_c dU_x/dx_c   (u'.grad) U ;   adjoint:  + u'_c dU_c/dx_x   (grad U)^T u'
        const double G = ld_bfc(d, bfc, adjoint ? 3 + 3 * c + x : 3 + 3 * x + c, nf, q);
        so[x * NDD + p] += uf * G;
      }
    }
    lds_barrier();
  }
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const double s = from_fine<N>(sJ, so + c * NDD, t1, t2, tid, NT, act, k, j, i);
    if (act) bf[c * d.cs + l] = -(sb * uin[c * d.cs + l] + s);
  }
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_convect(Dev d, const double* __restrict__ uin,
                                                        double* __restrict__ bf, int adjoint) {
  // base-flow constants: the steady set, or slot `*bstep` of the stored periodic orbit (Floquet, core/matvec.f:200-236)
  const double* __restrict__ bfc = d.bfc + (d.bf_stride ? (size_t)(*d.bstep) * (size_t)d.bf_stride : (size_t)0);
  if constexpr (N >= 10) { convect_lds<N>(d, uin, bf, adjoint); return; }
  using C = Cfg<N>;
  constexpr int NN = C::NN, ND = C::ND, NDD = C::NDD, NT = C::NT;
  constexpr int PPT = (NDD + NT - 1) / NT;
  __shared__ double sJ[ND * N], sDd[ND * ND];
  __shared__ double sf[NDD], t1[N * N * ND], t2[N * ND * ND];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  for (int p = tid; p < ND * N; p += NT) sJ[p] = d.Jd[p];
  for (int p = tid; p < ND * ND; p += NT) sDd[p] = d.Dd[p];
  const size_t nf = (size_t)d.nfine;
  double o[PPT][3], ca[PPT][3];
#pragma unroll
  for (int r = 0; r < PPT; ++r) { o[r][0] = o[r][1] = o[r][2] = 0.0; ca[r][0] = ca[r][1] = ca[r][2] = 0.0; }
  if (adjoint == 2) {                                  // full equations: convecting field c_a = w_d J (u . grad xi_a)
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
      to_fine<N>(sJ, uin + c * d.cs + e * NN, sf, t1, t2, tid, NT);
#pragma unroll
      for (int r = 0; r < PPT; ++r) {
        const int p = tid + r * NT;
        if (p < NDD) {
          const size_t q = (size_t)e * NDD + p;
          const double uf = sf[p];
#pragma unroll
          for (int a2 = 0; a2 < 3; ++a2) ca[r][a2] += (((d.zmask >> (a2 * 3 + c)) & 1u) ? 0.0 : d.mtd[(a2 * 3 + c) * nf + q]) * uf;
        }
      }
      lds_barrier();
    }
  }
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    to_fine<N>(sJ, uin + c * d.cs + e * NN, sf, t1, t2, tid, NT);
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int p = tid + r * NT;
      if (p < NDD) {
        const int a = p % ND, b = (p / ND) % ND, cc = p / (ND * ND);
        double g[3];
        fine_grad<N>(sDd, sf, a, b, cc, g);
        const double uf = sf[p];
        const size_t q = (size_t)e * NDD + p;
        if (adjoint == 2) {                             // (u.grad) u_c   [UPSTREAM advab]
          const double v = ca[r][0] * g[0] + ca[r][1] * g[1] + ca[r][2] * g[2];
          if (c == 0) o[r][0] = v; else if (c == 1) o[r][1] = v; else o[r][2] = v;
        } else {
          const double conv = ld_bfc(d, bfc, 0, nf, q) * g[0] + ld_bfc(d, bfc, 1, nf, q) * g[1] + ld_bfc(d, bfc, 2, nf, q) * g[2];   // (U.grad) u'_c
          const double sg = adjoint ? -conv : conv;
          if (c == 0) o[r][0] += sg; else if (c == 1) o[r][1] += sg; else o[r][2] += sg;
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            // direct:  + u'_c dU_x/dx_c   (u'.grad) U ;   adjoint:  + u'_c dU_c/dx_x   (grad U)^T u'
            const double G = ld_bfc(d, bfc, adjoint ? 3 + 3 * c + x : 3 + 3 * x + c, nf, q);
            o[r][x] += uf * G;
          }
        }
      }
    }
    lds_barrier();
  }
  const long long l = e * NN + tid;
  double sb = 0.0;
  if (act) sb = d.spng[l] * d.bm1[l];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int p = tid + r * NT;
      if (p < NDD) sf[p] = o[r][c];
    }
    lds_barrier();
    const double s = from_fine<N>(sJ, sf, t1, t2, tid, NT, act, k, j, i);
    if (act) {
      const double un = uin[c * d.cs + l];
      if (adjoint == 2) {
        const double kk = sb * d.nl_spng_str;
        bf[c * d.cs + l] = ((kk != 0.0) ? kk * (d.spng_vr[c * d.cs + l] - un) : 0.0) - s;
      } else {
        bf[c * d.cs + l] = -(sb * un + s);
      }
    }
  }
}

// base-flow constants of the convection kernel from a state vector (set-up, set_baseflow)
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_baseflow(Dev d, const double* __restrict__ q, double* __restrict__ bfc,
                                                         double*, double*, double*, double*, double*) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, ND = C::ND, NDD = C::NDD, NT = C::NT;
  __shared__ double sJ[ND * N], sDd[ND * ND];
  __shared__ double sf[NDD], t1[N * N * ND], t2[N * ND * ND];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  for (int p = tid; p < ND * N; p += NT) sJ[p] = d.Jd[p];
  for (int p = tid; p < ND * ND; p += NT) sDd[p] = d.Dd[p];
  const size_t nf = (size_t)d.nfine;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    to_fine<N>(sJ, q + c * d.nloc + e * NN, sf, t1, t2, tid, NT);
    for (int p = tid; p < NDD; p += NT) {
      const int a = p % ND, b = (p / ND) % ND, cc = p / (ND * ND);
      const size_t qq = (size_t)e * NDD + p;
      double mt[3][3];
#pragma unroll
      for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
        for (int x = 0; x < 3; ++x) mt[a2][x] = d.mtd[(a2 * 3 + x) * nf + qq];
      const double uf = sf[p];
#pragma unroll
      for (int a2 = 0; a2 < 3; ++a2) {
        const double v = mt[a2][c] * uf;
        bfc[a2 * nf + qq] = (c == 0) ? v : bfc[a2 * nf + qq] + v;
      }
      double g[3];
      fine_grad<N>(sDd, sf, a, b, cc, g);
#pragma unroll
      for (int x = 0; x < 3; ++x) bfc[(3 + 3 * c + x) * nf + qq] = mt[0][x] * g[0] + mt[1][x] * g[1] + mt[2][x] * g[2];
    }
    lds_barrier();
  }
}

// ---------------------------------------------------------------------------
// K2: makextp + makebdfp + lagfieldp + extrapprp + cresvipp  [UPSTREAM perturb.f]
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_rhs(Dev d, StepCoef sc) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  constexpr int RG = (3 * MM + 3 * NMM + 2 * NNM) > 4 * NN ? (3 * MM + 3 * NMM + 2 * NNM) : 4 * NN;
  __shared__ double sD[N * N], sDt[N * N], sJ12[NM], sD12[NM];
  __shared__ double reg[RG];                                                 // D^T p scratch, then the axhelm tiles
  double* sP = reg; double* sC = reg + 3 * MM; double* sE = sC + 3 * NMM;
  double* su = reg; double* st = reg + NN;
  const int tid = threadIdx.x;
  if (d.bf_stride && sc.adjoint != 2 && blockIdx.x == 0 && tid == 0) *d.bstep += 1;     // next step reads the next orbit slot (as the 2-D k_rhs)
  if (d.stepctr && blockIdx.x == 0 && tid == 0) *d.stepctr += 1;                          // per-step iteration record (rec_step_iters)
  const long long e = blockIdx.x;
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  const long long l = e * NN + tid, nl = d.cs;
  load_basis3<N>(d, sD, sDt, sJ12, sD12, tid, NT);
  if (d.nproj_max > 0 && blockIdx.x == 0 && tid == 0) {
    GmresScal* G = d.gsc;
    if (G->st_pending) {
      G->st_pending = 0;
      if (G->st_n > 0.0) {
        if (G->st_slot < 0) {                        // restart of a full space on the latest total solution (k_proj_update)
          G->pn[0] = G->st_n; G->pcnt = 1; G->nproj = 1;
        } else {
          G->pn[G->st_slot] = G->st_n;
          G->pcnt += 1;
          G->nproj = (G->pcnt < d.nproj_max) ? G->pcnt : d.nproj_max;
        }
      } else { G->pcnt = 0; G->nproj = 0; }
    }
    if (d.proj_reset && sc.cls == 0) { G->pcnt = 0; G->nproj = 0; }
  }
  double u[3] = {0, 0, 0}, bfv[3] = {0, 0, 0}, bm = 0, g[6] = {0, 0, 0, 0, 0, 0};
  if (act) {
    bm = d.bm1[l];
    load_g6(d, l, g);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long lc = c * nl + l;
      const double un = d.u[lc];
      u[c] = un + sc.xg[0] * d.dulag[lc] + sc.xg[1] * d.dulag[3 * nl + lc] + sc.xg[2] * d.dulag[6 * nl + lc];
      const double bn = d.bf[lc];
      const double e1 = d.exlag[lc], e2 = d.exlag[3 * nl + lc];
      double b = sc.ab[0] * bn + sc.ab[1] * e1 + sc.ab[2] * e2;      // makextp
      d.exlag[3 * nl + lc] = e1;
      d.exlag[lc] = bn;
      const double l1 = d.ulag[lc], l2 = d.ulag[3 * nl + lc];
      b += bm * (sc.bd[1] * un + sc.bd[2] * l1 + sc.bd[3] * l2) * sc.invdt;   // makebdfp
      d.ulag[3 * nl + lc] = l1;                                      // lagfieldp
      d.ulag[lc] = un;
      bfv[c] = b;
    }
  }
  double pe = 0.0, w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (tid < MM) {                                                    // extrapprp
    const long long q = e * MM + tid;
    const double pn = d.p[q];
    pe = (sc.pxt == 0.0) ? pn : 2.0 * pn - d.plag[q];
    d.plag[q] = pn;
    d.pext[q] = pe;
    load_w2(d, q, w2);
  }
  lds_barrier();
  double gp[3];
  opgradt3<N>(sJ12, sD12, pe, w2, sP, sC, sE, tid, NT, act, k, j, i, gp);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lds_barrier();
    if (act) su[tid] = u[c];
    lds_barrier();
    double au[1];
    axhelm3<N, 1>(sD, sDt, su, st, act, k, j, i, g, au);
    if (act) {
      const double b = bfv[c] + gp[c];                               // rhs of H u* = b
      d.bloc[c * nl + l] = b;
      d.rloc[c * nl + l] = b - (d.nu * au[0] + sc.h2 * bm * u[c]);
    }
  }
}

// ---------------------------------------------------------------------------
// K3: one Jacobi-PCG iteration of H du = dssum(r), three components, single-reduction
// (Chronopoulos-Gear) form  [UPSTREAM hmholtz.f cggo].  The components are processed one after the
// other (one tile set in LDS, low register count => several workgroups per CU: this kernel is HBM-bound
// on 3-D meshes).  Sums of the previous launch come from htot (k_tot2), never from the per-workgroup partials.
//   hscal[par*16 + c*4 + {0:gamma,1:alpha,2:done,3:res}], reference norms at hscal[32 + c]
//   hpart[par][12][nblk] / htot[par*16 + .]: c*3 + {0:(r,z), 1:(z,Az), 2:(r,r)}, 9 + c: (b,b)
// ---------------------------------------------------------------------------
// a wave-uniform double moved to scalar registers
__device__ inline double uniform_f64(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
// The eight-member lists of the element corners, one ENTRY per thread: thread p = (field, corner, member) loads its index with the
// kernel's first loads and its value with the gather; the corner thread adds the eight values from LDS in list order (the sum of
// gs_csr; a missing member is -0.0, the identity of the addition; NaN in member 0 = more than eight members: the CSR lists).
__device__ inline double corner_list_sum(const double* v, const double* __restrict__ f, const Dev& d, long long l) {
  if (v[0] != v[0]) return gs_csr_rolled(f, d, l);
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < 8; ++m) s += v[m];
  return s;
}
// it = 0 of k_helm: r = mask dssum(rhs), (b, b), x = p = s = 0, z = r / diag, A z.  Once per solve: one component after the other.
// Wave-uniform scalars that another kernel left in global memory, fetched with ONE load per wavefront (lane q loads word q) and
// spread with v_readlane: the compiler turns `x = p[3]; y = q[1]; ...` on pointers it cannot prove invariant into vector loads with a
// full wait behind each -- nine dependent trips to L2 in k_helm's preamble, 3.6 us of a workgroup's 15.8 (stamps, config 4's size).
__device__ inline double lane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int N>
__device__ inline void helm_first(const Dev& d, const StepCoef& sc, const double* __restrict__ rhs, double* sD, double* sDt,
                                  double* sz, double* st, double* sred, double* scv) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, NT = C::NT;
  static_assert(NT >= 128 && N * N <= NT, "corner entries and the basis: one per thread");
  const int tid = threadIdx.x;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  const long long l = e * NN + tid, nl = d.cs;
  const double dv = (tid < N * N) ? d.D[tid] : 0.0;
  const int cm = (d.gs_corner && tid < 128) ? d.gs_corner[(size_t)e * 64 + (tid & 63)] : -1;      // tid = field * 64 + corner * 8 + member
  int4 tab = make_int4(0, -1, -1, -1);
  double bm = 0, g[6] = {0, 0, 0, 0, 0, 0}, mk = 0, mi = 0, di = 0;
  if (act) {
    tab = d.gs_tab[l];
    bm = d.bm1[l]; mk = d.mask[l]; mi = d.minv[l];
    load_g6(d, l, g);
    di = d.dinv[(size_t)(sc.k - 1) * d.nloc + l];
  }
  if (tid < N * N) { sD[tid] = dv; sDt[(tid % N) * N + tid / N] = dv; }
  const int cid = act ? corner_id<N>(k, j, i) : -1;
  const bool wide = act && tab.x < 0, from_list = wide && cid >= 0 && d.gs_corner;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const long long lc = c * nl + l;
    const double* f0 = rhs + c * nl;
    const double* f1 = d.bloc + c * nl;
    if (tid < 128) scv[tid] = (cm == -2) ? __builtin_nan("") : ((cm >= 0) ? ((tid < 64) ? f0[cm] : f1[cm]) : -0.0);
    GsVals gv, gb;
    if (act && !wide) { gv = gs_load_o(f0, tab, (unsigned)l); gb = gs_load_o(f1, tab, (unsigned)l); }
    lds_barrier();
    double r = 0.0, z = 0.0, bb = 0.0;
    if (act) {
      if (!wide) { r = mk * (((gv.a + gv.b) + gv.c) + gv.d); bb = mk * (((gb.a + gb.b) + gb.c) + gb.d); }
      else if (from_list) { r = mk * corner_list_sum(scv + cid * 8, f0, d, l); bb = mk * corner_list_sum(scv + 64 + cid * 8, f1, d, l); }
      else { r = mk * gs_csr_rolled(f0, d, l); bb = mk * gs_csr_rolled(f1, d, l); }
      d.hx[lc] = 0.0; d.hp[lc] = 0.0; d.hs[lc] = 0.0; d.hr[lc] = r;
      z = di * r;
      sz[tid] = z;
    }
    lds_barrier();
    double au[1];
    axhelm3<N, 1>(sD, sDt, sz, st, act, k, j, i, g, au);
    double v[4] = {0, 0, 0, 0};
    if (act) {
      const double wl = d.nu * au[0] + sc.h2 * bm * z;
      d.hwl[(size_t)c * nl + l] = wl;                                     // parity 0
      v[0] = r * z * mi; v[1] = z * wl; v[2] = r * r * mi; v[3] = bb * bb * mi;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double x = wave_sum63(v[q]);
      if ((tid & 63) == 63) sred[q * 16 + (tid >> 6)] = x;
    }
    lds_barrier();
    if (tid < 4) {
      double s = 0.0;
      for (int w = 0; w < NT / 64; ++w) s += sred[tid * 16 + w];
      d.hpart[((size_t)(tid < 3 ? c * 3 + tid : 9 + c)) * d.nblk + e] = s;
    }
  }
}
#ifndef NSK_HELM3_WAVES
#define NSK_HELM3_WAVES 4
#endif
// it >= 1: every load of the three components is issued before the first is used -- two round trips to memory per workgroup (the
// arrays addressable from the thread index, then the neighbours' values) where the component loop of rounds 2-4 had five -- and the
// three components' A z then run one after the other on one set of LDS tiles, on the matrix cores (axhelm3_mfma).
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT, NSK_HELM3_WAVES) void k_helm(Dev d, StepCoef sc, int it, const double* rhs) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, NT = C::NT;
  static_assert(NT >= 192, "corner entries: one per thread");
  constexpr bool HEAD_FIRST = NT >= 1024;
  using L = SlimLay<N>;
  constexpr int EXT = L::EXT;
  static_assert(2 * N * N + 3 * NN <= 5 * EXT, "it = 0 borrows the tiles");
  __shared__ double tiles[8 * EXT];
  double* sz = tiles; double* sW = tiles + 3 * EXT; double* sO = tiles + 6 * EXT;
  __shared__ double sred[12 * 16];
  __shared__ double scv[192];
  if (it == 0) { helm_first<N>(d, sc, rhs, sW, sW + N * N, sz, sW + 2 * N * N, sred, scv); return; }
  const int tid = threadIdx.x;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);      // (boff: shards launch their boundary elements first, the interior behind)
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  const long long l = e * NN + tid, nl = d.cs;
  NSK_STAMP(0);
  const int par = it & 1, ppar = par ^ 1;
  double alpha[3] = {0, 0, 0}, beta[3] = {0, 0, 0};
  bool done[3] = {false, false, false};
  int cm = -1;
  int4 tab = make_int4(0, -1, -1, -1);
  AxFrag<N> F;
  {
    // lanes 0..11: htot[ppar][0..11]; 12..27: hscal[ppar][0..15]; 28..30: hscal[32..34]
    const int ln = tid & 63;
    double hv = 0.0;
    if (ln < 12) hv = d.htot[ppar * 16 + ln];
    else if (ln < 28) hv = d.hscal[ppar * 16 + ln - 12];
    else if (ln < 31) hv = d.hscal[32 + ln - 28];
    // lx1 = 10 (one workgroup per CU: nothing hides a trip): the head of the gather chain rides behind the scalars in the same trip.
    // A launch that finds its solve finished has then fetched 16 B per node for nothing (245 us at config 4's size, where the two
    // workgroups per CU hide the trip anyway: 1703 us with the head here, 1719 below): not at lx1 <= 8.
    if constexpr (HEAD_FIRST) {
      cm = (d.gs_corner && tid < 192) ? d.gs_corner[(size_t)e * 64 + (tid & 63)] : -1;      // tid = component * 64 + corner * 8 + member
      if (act) tab = d.gs_tab[l];
      F = ax_frags<N>(d.D, tid & 63);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double gg = lane_f64(hv, c * 3 + 0), del = lane_f64(hv, c * 3 + 1), rr = lane_f64(hv, c * 3 + 2);
      const double o0 = lane_f64(hv, 12 + c * 4 + 0), o1 = lane_f64(hv, 12 + c * 4 + 1), o2 = lane_f64(hv, 12 + c * 4 + 2), o3 = lane_f64(hv, 12 + c * 4 + 3);
      const double res = sqrt(rr / d.vol);
      const double ref = (it == 1) ? sqrt(lane_f64(hv, 9 + c) / d.vol) : lane_f64(hv, 28 + c);
      const double tol = d.tol_relative ? d.tol_helm * ref : d.tol_helm;
      const bool was = (it > 1 && o2 != 0.0);
      done[c] = was || (res <= tol) || !(gg > 0.0);
      if (!done[c]) {
        if (it == 1) { beta[c] = 0.0; alpha[c] = gg / del; }
        else { beta[c] = gg / o0; alpha[c] = gg / (del - beta[c] * gg / o1); }
      }
      if (blockIdx.x == 0 && d.boff == 0 && tid == 0) {
        double* cur = d.hscal + par * 16 + c * 4;
        const double keep = was ? o3 : res;
        cur[0] = gg; cur[1] = alpha[c]; cur[2] = done[c] ? 1.0 : 0.0; cur[3] = keep;
        if (it == 1) d.hscal[32 + c] = ref;
        if (done[c] && !was) {
          if (c == 0) atomicAdd((unsigned long long*)&d.stats->helm_iters, (unsigned long long)(it - 1));
          atomicMax((unsigned long long*)&d.stats->max_helm, (unsigned long long)(it - 1));
          atomicMax((unsigned long long*)&d.stats->max_helm_k[sc.cls], (unsigned long long)(it - 1)); rec_step_iters(d, 0, it - 1);
        }
      }
    }
    if (done[0] && done[1] && done[2]) return;
    // wave-uniform CG scalars go to scalar registers
#pragma unroll
    for (int c = 0; c < 3; ++c) { alpha[c] = uniform_f64(alpha[c]); beta[c] = uniform_f64(beta[c]); done[c] = __builtin_amdgcn_readfirstlane((int)done[c]) != 0; }
  }
  NSK_STAMP(1);
  // ---- everything (else) addressable from the thread index
  if constexpr (!HEAD_FIRST) {
    cm = (d.gs_corner && tid < 192) ? d.gs_corner[(size_t)e * 64 + (tid & 63)] : -1;
    if (act) tab = d.gs_tab[l];
    F = ax_frags<N>(d.D, tid & 63);
  }
  const int tn = k * L::PS + j * L::RS + i;
  double bm = 0, g[6] = {0, 0, 0, 0, 0, 0}, mk = 0, mi = 0, di = 0;
  double ro[3] = {0, 0, 0}, po[3] = {0, 0, 0}, so[3] = {0, 0, 0}, xo[3] = {0, 0, 0};
  if (act) {
    mk = d.mask[l]; di = d.dinv[(size_t)(sc.k - 1) * d.nloc + l];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long lc = c * nl + l;
      ro[c] = d.hr[lc];
      if (!done[c]) { po[c] = d.hp[lc]; so[c] = d.hs[lc]; xo[c] = d.hx[lc]; }
    }
    bm = d.bm1[l]; mi = d.minv[l];
    load_g6(d, l, g);
  }
  const double* wl0 = d.hwl + (size_t)ppar * 3 * nl;
  const bool wide = act && tab.x < 0;
  double cv = -0.0;
  {
    const int cc = tid >> 6;
    const bool cdone = (cc == 0) ? done[0] : ((cc == 1) ? done[1] : done[2]);
    if (tid < 192 && cm >= 0 && !cdone) cv = wl0[(size_t)cc * nl + cm];
  }
  GsVals gv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    if (act && !wide && !done[c]) gv[c] = gs_load_o(wl0 + (size_t)c * nl, tab, (unsigned)l);
  if (tid < 192) scv[tid] = (cm == -2) ? __builtin_nan("") : cv;            // (after the gather is out: the store waits for its value)
  const int cid = act ? corner_id<N>(k, j, i) : -1;
  const bool from_list = wide && cid >= 0 && d.gs_corner;
  NSK_STAMP(2);
  lds_barrier();
  NSK_STAMP(3);
  double rz[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
  if (act) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long lc = c * nl + l;
      double r = ro[c];
      if (!done[c]) {
        const double* wl = wl0 + (size_t)c * nl;
        const double sum = !wide ? (((gv[c].a + gv[c].b) + gv[c].c) + gv[c].d)
                                 : (from_list ? corner_list_sum(scv + c * 64 + cid * 8, wl, d, l) : gs_csr_rolled(wl, d, l));
        const double w = mk * sum;
        const double pn = di * ro[c] + beta[c] * po[c];
        const double sn = w + beta[c] * so[c];
        d.hp[lc] = pn; d.hs[lc] = sn;
        d.hx[lc] = xo[c] + alpha[c] * pn;
        r = ro[c] - alpha[c] * sn;
        d.hr[lc] = r;
      }
      const double z = di * r;
      sz[c * EXT + tn] = z;
      rz[c] = r * z * mi; rr[c] = r * r * mi;
    }
  }
  NSK_STAMP(4);
  lds_barrier();
  NSK_STAMP(5);
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    double z;
    const double au = axhelm3_mfma<N>(F, sz + c * EXT, sW, sO, g, act, tn, tid >> 6, NT / 64, tid & 63, z);
    double v[3] = {0, 0, 0};
    if (act) {
      const double wl = d.nu * au + sc.h2 * bm * z;
      d.hwl[((size_t)par * 3 + c) * nl + l] = wl;
      v[0] = (c == 0) ? rz[0] : ((c == 1) ? rz[1] : rz[2]); v[1] = z * wl; v[2] = (c == 0) ? rr[0] : ((c == 1) ? rr[1] : rr[2]);
    }
    // workgroup sums: wave sums to LDS, rows added (fixed order) by the threads that store them
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double x = wave_sum63(v[q]);
      if ((tid & 63) == 63) sred[(c * 3 + q) * 16 + (tid >> 6)] = x;
    }
    lds_barrier();                                           // (the tiles are rewritten by the next component)
  }
  NSK_STAMP(6);
  if (tid < 12) {
    double s = 0.0;
    if (tid < 9)
      for (int w = 0; w < NT / 64; ++w) s += sred[tid * 16 + w];
    d.hpart[((size_t)par * 12 + tid) * d.nblk + e] = s;      // (rows 9..11, (b, b), are it = 0's)
  }
}


// k_helm (it >= 1) as RESIDENT workgroups that walk their share of the elements, the next element's r, p, s, x (twelve arrays,
// 96 KB at lx1 = 10) arriving in LDS by LDS-DMA (global_load_lds_dwordx4: no register is their destination) while the current
// element's A z runs on the matrix cores.  Why: k_helm<10> is 1024 threads x 128 registers -- the whole register file of a CU, one
// workgroup per CU -- so nothing hides a workgroup's trips to memory behind another's passes (config 5: 8.3 ms per launch, 0.47 of
// 8 TB/s, 58-65 % of a time step; the 512-thread workgroups of lx1 = 8 reach 0.63 with the same code at two per CU), and a second
// workgroup per CU would need 64 registers.  LDS holds what the registers cannot: 67 KB of tiles + 96 KB of
// landing zone = 163 072 of the CU's 163 840 bytes.  Same arithmetic in the same order as k_helm: bit-identical.
//   per element:  A  the arrays addressable from the thread index and the neighbours' A z (gather table in registers since the
//                    previous element's phase C), wait, barrier
//                 B  r, p, s, x from the landing zone; updates, stores; z tiles; barrier
//                 -> LDS-DMA of the next element's r, p, s, x; its gather-table entry and corner-list entry
//                 C  A z of the three components (axhelm3_mfma), partial sums
// `count` elements from d.boff on, XCD-contiguous as in k_helm (a step of gridDim.x, a multiple of 8, stays in the XCD's run).
#ifdef NSK_HP_NT
#define NSK_HP_NT_LD " nt"
#define NSK_HP_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define NSK_HP_NT_LD ""
#define NSK_HP_ST(p, v) (*(p) = (v))
#endif
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT, NSK_HELM3_WAVES) void k_helm_p(Dev d, StepCoef sc, int it, int count) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, NT = C::NT;
  static_assert(NT == 1024 && NN % 2 == 0 && NN / 2 <= 512, "LDS-DMA: two arrays per round, 16 bytes per lane");
  using L = SlimLay<N>;
  constexpr int EXT = L::EXT;
  __shared__ double tiles[8 * EXT];
  double* sz = tiles; double* sW = tiles + 3 * EXT; double* sO = tiles + 6 * EXT;
  __shared__ double sred[9 * 16];
  __shared__ double scv[192];
  __shared__ double sDm[N * N];                                    // the derivative matrix: its fragments are rebuilt per element (36 registers that phases A and B need)
  __shared__ __attribute__((aligned(16))) double pf[12 * NN];      // landing zone [component][r, p, s, x][NN]
  const int tid = threadIdx.x;
  const long long nl = d.cs;
  const int par = it & 1, ppar = par ^ 1;
  const unsigned G = gridDim.x;
  unsigned b = blockIdx.x;
  long long e = d.boff + xcd_element(b, count);
  double alpha[3] = {0, 0, 0}, beta[3] = {0, 0, 0};
  bool done[3] = {false, false, false};
  int cm = -1;
  int4 tab = make_int4(0, -1, -1, -1);
  int dmask = 0;                                                   // finished components (wave-uniform, in a scalar register)
  {
    const bool act = tid < NN;
    // lanes 0..11: htot[ppar][0..11]; 12..27: hscal[ppar][0..15]; 28..30: hscal[32..34]   (as k_helm)
    const int ln = tid & 63;
    double hv = 0.0;
    if (ln < 12) hv = d.htot[ppar * 16 + ln];
    else if (ln < 28) hv = d.hscal[ppar * 16 + ln - 12];
    else if (ln < 31) hv = d.hscal[32 + ln - 28];
    cm = (d.gs_corner && tid < 192) ? d.gs_corner[(size_t)e * 64 + (tid & 63)] : -1;      // tid = component * 64 + corner * 8 + member
    if (act) tab = d.gs_tab[e * NN + tid];
    if (tid < N * N) sDm[tid] = d.D[tid];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double gg = lane_f64(hv, c * 3 + 0), del = lane_f64(hv, c * 3 + 1), rr = lane_f64(hv, c * 3 + 2);
      const double o0 = lane_f64(hv, 12 + c * 4 + 0), o1 = lane_f64(hv, 12 + c * 4 + 1), o2 = lane_f64(hv, 12 + c * 4 + 2), o3 = lane_f64(hv, 12 + c * 4 + 3);
      const double res = sqrt(rr / d.vol);
      const double ref = (it == 1) ? sqrt(lane_f64(hv, 9 + c) / d.vol) : lane_f64(hv, 28 + c);
      const double tol = d.tol_relative ? d.tol_helm * ref : d.tol_helm;
      const bool was = (it > 1 && o2 != 0.0);
      done[c] = was || (res <= tol) || !(gg > 0.0);
      if (!done[c]) {
        if (it == 1) { beta[c] = 0.0; alpha[c] = gg / del; }
        else { beta[c] = gg / o0; alpha[c] = gg / (del - beta[c] * gg / o1); }
      }
      if (blockIdx.x == 0 && d.boff == 0 && tid == 0) {
        double* cur = d.hscal + par * 16 + c * 4;
        const double keep = was ? o3 : res;
        cur[0] = gg; cur[1] = alpha[c]; cur[2] = done[c] ? 1.0 : 0.0; cur[3] = keep;
        if (it == 1) d.hscal[32 + c] = ref;
        if (done[c] && !was) {
          if (c == 0) atomicAdd((unsigned long long*)&d.stats->helm_iters, (unsigned long long)(it - 1));
          atomicMax((unsigned long long*)&d.stats->max_helm, (unsigned long long)(it - 1));
          atomicMax((unsigned long long*)&d.stats->max_helm_k[sc.cls], (unsigned long long)(it - 1)); rec_step_iters(d, 0, it - 1);
        }
      }
    }
    if (done[0] && done[1] && done[2]) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) { alpha[c] = uniform_f64(alpha[c]); beta[c] = uniform_f64(beta[c]); }
    dmask = __builtin_amdgcn_readfirstlane((done[0] ? 1 : 0) | (done[1] ? 2 : 0) | (done[2] ? 4 : 0));
  }
  // LDS-DMA of element en's r, p, s, x: six rounds of two arrays (waves 0-7 / 8-15), lane t of a half moves doubles 2t, 2t + 1;
  // the LDS address is the wave's base + lane * 16 (inactive lanes write nothing).  p, s, x of a finished component are not read.
  const int half = __builtin_amdgcn_readfirstlane(tid >> 9), hw = __builtin_amdgcn_readfirstlane((tid & 511) >> 6), ht = tid & 511;
  auto prefetch = [&](long long en) {
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const int a = 2 * rd + half, c = a >> 2, w = a & 3;
      if (w != 0 && ((dmask >> c) & 1)) continue;
      const double* src = ((w == 0) ? d.hr : ((w == 1) ? d.hp : ((w == 2) ? d.hs : d.hx))) + (size_t)c * nl + en * NN;
      // (inline assembly: the builtin makes hipcc drain the DMA -- s_waitcnt vmcnt(0) -- in front of LDS writes of phase C that
      //  it cannot tell from the landing zone; M0 = the wave's LDS base, written in the statement that reads it)
      const unsigned ldst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(pf + a * NN + hw * 128);
      const double* gsrc = src + 2 * ht;
      unsigned keep;
      if (ht < NN / 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" NSK_HP_NT_LD "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
    }
  };
  prefetch(e);
  const double* wl0 = d.hwl + (size_t)ppar * 3 * nl;
  for (;;) {
    // (the thread index is made opaque per element: otherwise every lane-dependent address and LDS offset below is hoisted out of
    //  the loop, and the loop spills)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const bool act = tl < NN;
    const int k = tl / (N * N), j = (tl / N) % N, i = tl % N;
    const int tn = k * L::PS + j * L::RS + i;
    const int cid = act ? corner_id<N>(k, j, i) : -1;
    const long long l = e * NN + tl;
    // ---- A: everything else addressable from the thread index, and the neighbours' values
    // (unconditional loads from clamped addresses, selects behind them: `cond ? load : 0` is a branch and a wait per load;
    //  uniform bases + 32-bit lane offsets: the global_load saddr form; the gather-table entry -- in flight since the previous
    //  element's phase C -- is touched FIRST, so that its wait stands in front of the loads, not between them)
    const bool wide = act && tab.x < 0;
    const unsigned lown = (unsigned)(e * NN + (act ? tl : 0)) * 8u;
    const unsigned ox = (tab.x >= 0 && act) ? (unsigned)tab.x * 8u : lown, oy = (tab.y >= 0) ? (unsigned)tab.y * 8u : lown,
                   oz = (tab.z >= 0) ? (unsigned)tab.z * 8u : lown, ow = (tab.w >= 0) ? (unsigned)tab.w * 8u : lown;
    const unsigned oc = (tl < 192 && cm >= 0) ? (unsigned)cm * 8u : 0u;
    // a finished component's loads go to a live component's array (cache hits, no branch): their values are never used
    const int live = (dmask & 1) ? ((dmask & 2) ? 2 : 1) : 0;
    const double* __restrict__ w0 = wl0 + (size_t)((dmask & 1) ? live : 0) * nl;
    const double* __restrict__ w1 = wl0 + (size_t)((dmask & 2) ? live : 1) * nl;
    const double* __restrict__ w2 = wl0 + (size_t)((dmask & 4) ? live : 2) * nl;
    GsVals gv[3];
    gv[0].a = ld_boff(w0, ox); gv[0].b = ld_boff(w0, oy); gv[0].c = ld_boff(w0, oz); gv[0].d = ld_boff(w0, ow);
    gv[1].a = ld_boff(w1, ox); gv[1].b = ld_boff(w1, oy); gv[1].c = ld_boff(w1, oz); gv[1].d = ld_boff(w1, ow);
    gv[2].a = ld_boff(w2, ox); gv[2].b = ld_boff(w2, oy); gv[2].c = ld_boff(w2, oz); gv[2].d = ld_boff(w2, ow);
    const int wv = __builtin_amdgcn_readfirstlane(tl >> 6);
    double cv = ld_boff((wv == 0) ? w0 : ((wv == 1) ? w1 : w2), oc);         // waves 0..2: the corner lists of component wv
    const unsigned lo = (unsigned)(act ? tl : 0) * 8u;              // this lane's node of the element (lanes past it: node 0)
    double bm, g[6], mk, mi, di;
    mk = ld_boff(d.mask + e * NN, lo); di = ld_boff(d.dinv + (size_t)(sc.k - 1) * d.nloc + e * NN, lo);
    bm = ld_boff(d.bm1 + e * NN, lo); mi = ld_boff(d.minv + e * NN, lo);
    g[0] = ld_boff(d.g1 + e * NN, lo); g[1] = ld_boff(d.g2 + e * NN, lo); g[2] = ld_boff(d.g3 + e * NN, lo);
    g[3] = (d.zmask & (1u << 9)) ? 0.0 : ld_boff(d.g4 + e * NN, lo);        // (zmask: wave-uniform branches)
    g[4] = (d.zmask & (1u << 10)) ? 0.0 : ld_boff(d.g5 + e * NN, lo);
    g[5] = (d.zmask & (1u << 11)) ? 0.0 : ld_boff(d.g6 + e * NN, lo);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (tab.y < 0) gv[c].b = 0.0;
      if (tab.z < 0) gv[c].c = 0.0;
      if (tab.w < 0) gv[c].d = 0.0;
    }
    if (!(tl < 192 && cm >= 0) || ((dmask >> (wv < 3 ? wv : 0)) & 1)) cv = -0.0;
    if (tl < 192) scv[tl] = (cm == -2) ? __builtin_nan("") : cv;
    const bool from_list = wide && cid >= 0 && d.gs_corner;
    // the landing zone is complete, and every load above (the builtin, not a statement hipcc cannot see: it goes on waiting for
    // loads it believes in flight, and behind the stores of phase B such a wait waits for the stores)
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
    lds_barrier();
    // ---- B
    double rz[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
    if (act) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long long lc = c * nl + l;
        const double ro = pf[(c * 4 + 0) * NN + tl];
        double r = ro;
        if (!((dmask >> c) & 1)) {
          const double po = pf[(c * 4 + 1) * NN + tl], so = pf[(c * 4 + 2) * NN + tl], xo = pf[(c * 4 + 3) * NN + tl];
          const double* wl = wl0 + (size_t)c * nl;
          const double sum = !wide ? (((gv[c].a + gv[c].b) + gv[c].c) + gv[c].d)
                                   : (from_list ? corner_list_sum(scv + c * 64 + cid * 8, wl, d, l) : gs_csr_rolled(wl, d, l));
          const double w = mk * sum;
          const double pn = di * ro + beta[c] * po;
          const double sn = w + beta[c] * so;
          NSK_HP_ST(d.hp + lc, pn); NSK_HP_ST(d.hs + lc, sn);
          NSK_HP_ST(d.hx + lc, xo + alpha[c] * pn);
          r = ro - alpha[c] * sn;
          NSK_HP_ST(d.hr + lc, r);
        }
        const double z = di * r;
        sz[c * EXT + tn] = z;
        rz[c] = r * z * mi; rr[c] = r * r * mi;
      }
    }
    lds_barrier();                                               // (the landing zone and the corner values are read, the z tiles written)
    // ---- the next element: its r, p, s, x by LDS-DMA, its gather-table and corner-list entries
    const unsigned bn = b + G;
    const bool more = bn < (unsigned)count;
    const long long en = d.boff + xcd_element(more ? bn : b, count);
    // (order: hipcc does not count the LDS-DMA statements, so every wait it places BEHIND them for an older load or store also
    //  drains them -- the table loads and the fragment reads, whose waits guard registers, come first)
    int cmn = -1;
    int4 tabn = make_int4(0, -1, -1, -1);
    if (more) {
      cmn = (d.gs_corner && tl < 192) ? d.gs_corner[(size_t)en * 64 + (tl & 63)] : -1;
      if (act) tabn = d.gs_tab[en * NN + tl];
    }
    const AxFrag<N> F = ax_frags<N>(sDm, tl & 63);
#ifndef NSK_HP_SKIP_DMA    // (timing experiment: no prefetch)
    if (more) prefetch(en);
#endif
    // ---- C
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
      double z;
#ifdef NSK_HP_SKIP_C      // (timing experiment: phase C without its passes)
      const double au = g[0] + F.d.a[0][0]; z = sz[c * EXT + tn];
#else
      const double au = axhelm3_mfma<N>(F, sz + c * EXT, sW, sO, g, act, tn, tl >> 6, NT / 64, tl & 63, z);
#endif
      double v[3] = {0, 0, 0};
      if (act) {
        const double wl = d.nu * au + sc.h2 * bm * z;
        d.hwl[((size_t)par * 3 + c) * nl + l] = wl;
        v[0] = (c == 0) ? rz[0] : ((c == 1) ? rz[1] : rz[2]); v[1] = z * wl; v[2] = (c == 0) ? rr[0] : ((c == 1) ? rr[1] : rr[2]);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double x = wave_sum63(v[q]);
        if ((tl & 63) == 63) sred[(c * 3 + q) * 16 + (tl >> 6)] = x;
      }
      lds_barrier();
    }
    if (tl < 12) {
      double s = 0.0;
      if (tl < 9)
        for (int w = 0; w < NT / 64; ++w) s += sred[tl * 16 + w];
      d.hpart[((size_t)par * 12 + tl) * d.nblk + e] = s;      // (rows 9..11, (b, b), are it = 0's)
    }
    if (!more) break;
    e = en; b = bn; tab = tabn; cm = cmn;
  }
}


// ---------------------------------------------------------------------------
// Velocity solve with an ELEMENT-BLOCK FAST-DIAGONALISATION preconditioner (option "helm_fdm"; config 5's wall cells have
// aspect ratios of 29, where the Jacobi-preconditioned CG of k_helm needs 60-105 iterations per time step):
//   M^-1 = sum_e R_e^T W (S_t x S_s x S_r) (h2 + nu (l_r + l_s + l_t))^-1 (S_t x S_s x S_r)^T W R_e,   W = mask / sqrt(multiplicity),
// the exact inverse of the element's own Helmholtz block on an undeformed box (1-D pairs of its GLL lines scaled by its
// lengths, Dirichlet ends at walls; mean lengths on deformed elements).  M^-1 r needs a dssum of its own, so one CG iteration
// (same Chronopoulos-Gear single-reduction recurrences, same scalars, flags and statistics as k_helm) is TWO launches:
//   k_helm_fa(it): sums of iteration it-1 -> alpha, beta, flags; gather A z;  p = z + beta p, s = A z + beta s, x += alpha p,
//                  r -= alpha s;  y_e = W H_e^-1 W r  (six LDS passes), stored unassembled
//   k_helm_fb(it): z = mask dssum(y);  A z (axhelm), stored unassembled;  (r, z), (z, A z), (r, r) partial sums
// 1.4x the bytes of a Jacobi iteration for 3-4x fewer iterations where the cells are that anisotropic (scripts/helm_fdm_study.py).
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT, 4) void k_helm_fa(Dev d, StepCoef sc, int it, const double* rhs) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, NT = C::NT;
  __shared__ double sS[3 * N * N], sL[3 * N];
  __shared__ double sa[NN], sb[NN];
  __shared__ double sred[16];
  __shared__ int sW[NT * 8];
  const int tid = threadIdx.x;
  const long long e = xcd_element(blockIdx.x, gridDim.x);
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  const long long l = e * NN + tid, nl = d.cs;
  const int par = it & 1, ppar = par ^ 1;
  double alpha[3] = {0, 0, 0}, beta[3] = {0, 0, 0};
  bool done[3] = {false, false, false};
  if (it > 0) {                                        // (the same bookkeeping as k_helm)
    const double* ps = d.htot + ppar * 16;
    const double* o = d.hscal + ppar * 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double gg = ps[c * 3 + 0], del = ps[c * 3 + 1], rr = ps[c * 3 + 2];
      const double res = sqrt(rr / d.vol);
      const double ref = (it == 1) ? sqrt(ps[9 + c] / d.vol) : d.hscal[32 + c];
      const double tol = d.tol_relative ? d.tol_helm * ref : d.tol_helm;
      const bool was = (it > 1 && o[c * 4 + 2] != 0.0);
      done[c] = was || (res <= tol) || !(gg > 0.0);
      if (!done[c]) {
        if (it == 1) { beta[c] = 0.0; alpha[c] = gg / del; }
        else { beta[c] = gg / o[c * 4 + 0]; alpha[c] = gg / (del - beta[c] * gg / o[c * 4 + 1]); }
      }
      if (blockIdx.x == 0 && tid == 0) {
        double* cur = d.hscal + par * 16 + c * 4;
        const double keep = was ? o[c * 4 + 3] : res;
        cur[0] = gg; cur[1] = alpha[c]; cur[2] = done[c] ? 1.0 : 0.0; cur[3] = keep;
        if (it == 1) d.hscal[32 + c] = ref;
        if (done[c] && !was) {
          if (c == 0) atomicAdd((unsigned long long*)&d.stats->helm_iters, (unsigned long long)(it - 1));
          atomicMax((unsigned long long*)&d.stats->max_helm, (unsigned long long)(it - 1));
          atomicMax((unsigned long long*)&d.stats->max_helm_k[sc.cls], (unsigned long long)(it - 1)); rec_step_iters(d, 0, it - 1);
        }
      }
    }
    if (done[0] && done[1] && done[2]) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) { alpha[c] = uniform_f64(alpha[c]); beta[c] = uniform_f64(beta[c]); done[c] = __builtin_amdgcn_readfirstlane((int)done[c]) != 0; }
  } else if (blockIdx.x == 0 && tid < 3) {
    d.hscal[par * 16 + tid * 4 + 2] = 0.0;            // launch 0: nothing is converged yet (k_helm_fb reads the flags of its own launch)
  }
  // hybrid: an element of moderate aspect ratio contributes the Jacobi diagonal (multiplicity-weighted, so that all-Jacobi nodes
  // get exactly D^-1 r), an anisotropic one its fast-diagonalisation block; the sum of the pieces stays symmetric positive definite
  const bool blk = d.hfT[e] != 0;
  if (blk) {
    for (int q = tid; q < 3 * N * N; q += NT) sS[q] = d.hfS[(size_t)e * 3 * N * N + q];
    for (int q = tid; q < 3 * N; q += NT) sL[q] = d.hfL[(size_t)e * 3 * N + q];
  }
  int4 tab = make_int4(0, -1, -1, -1);
  double mk = 0, mi = 0, wt = 0, dj = 0;
  if (act) {
    tab = d.gs_tab[l];
    mk = d.mask[l]; mi = d.minv[l];
    wt = mk * sqrt(mi);
    if (!blk) dj = mk * mi * d.dinv[(size_t)(sc.k - 1) * d.nloc + l];
    const int cid = corner_id<N>(k, j, i);
    gs_wide_stage(d, tab, l, sW + tid * 8, (cid >= 0 && d.gs_corner) ? reinterpret_cast<const int4*>(d.gs_corner + ((size_t)e * 8 + cid) * 8) : nullptr);
  }
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    if (it > 0 && done[c]) continue;
    const long long lc = c * nl + l;
    double r = 0.0, bbv = 0.0;
    if (act) {
      if (it == 0) {
        const GsVals gv = gs_load(rhs + c * nl, tab, l), gb = gs_load(d.bloc + c * nl, tab, l);
        r = mk * (tab.x < 0 ? gs_wide_sum(rhs + c * nl, d, l, sW + tid * 8) : gs_sum(gv, rhs + c * nl, d, tab, l));
        const double bb = mk * (tab.x < 0 ? gs_wide_sum(d.bloc + c * nl, d, l, sW + tid * 8) : gs_sum(gb, d.bloc + c * nl, d, tab, l));
        d.hx[lc] = 0.0; d.hp[lc] = 0.0; d.hs[lc] = 0.0; d.hr[lc] = r;
        bbv = bb * bb * mi;                                              // (b, b): the reference of the relative tolerance
      } else {
        const double* wl = d.hwl + ((size_t)ppar * 3 + c) * nl;
        const GsVals gv = gs_load(wl, tab, l);
        const double rold = d.hr[lc], pold = d.hp[lc], sold = d.hs[lc], xold = d.hx[lc], zold = d.hz[lc];
        const double w = mk * (tab.x < 0 ? gs_wide_sum(wl, d, l, sW + tid * 8) : gs_sum(gv, wl, d, tab, l));
        const double pn = zold + beta[c] * pold;
        const double sn = w + beta[c] * sold;
        d.hp[lc] = pn; d.hs[lc] = sn;
        d.hx[lc] = xold + alpha[c] * pn;
        r = rold - alpha[c] * sn;
        d.hr[lc] = r;
      }
      sa[tid] = wt * r;
    }
    if (it == 0) {
      const double x = wave_sum63(bbv);
      if ((tid & 63) == 63) sred[tid >> 6] = x;
    }
    lds_barrier();
    if (it == 0 && tid == 0) {
      double s = 0.0;
      for (int w = 0; w < NT / 64; ++w) s += sred[w];
      d.hpart[((size_t)par * 12 + 9 + c) * d.nblk + e] = s;
    }
    if (!blk) {                                        // Jacobi piece (uniform per workgroup)
      if (act) d.hy[lc] = dj * r;
      lds_barrier();
      continue;
    }
    // forward: S^T along r, s, t; scale; back: S along t, s, r   (S: [pos][mode], B-orthonormal)
    if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[0 * N * N + m * N + i] * sa[(k * N + j) * N + m]; sb[tid] = s; }
    lds_barrier();
    if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[1 * N * N + m * N + j] * sb[(k * N + m) * N + i]; sa[tid] = s; }
    lds_barrier();
    if (act) {
      double s = 0; for (int m = 0; m < N; ++m) s += sS[2 * N * N + m * N + k] * sa[(m * N + j) * N + i];
      sb[tid] = s / (sc.h2 + d.nu * (sL[i] + sL[N + j] + sL[2 * N + k]));
    }
    lds_barrier();
    if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[2 * N * N + k * N + m] * sb[(m * N + j) * N + i]; sa[tid] = s; }
    lds_barrier();
    if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[1 * N * N + j * N + m] * sa[(k * N + m) * N + i]; sb[tid] = s; }
    lds_barrier();
    if (act) {
      double s = 0; for (int m = 0; m < N; ++m) s += sS[0 * N * N + i * N + m] * sb[(k * N + j) * N + m];
      d.hy[lc] = wt * s;
    }
    lds_barrier();
  }
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT, 4) void k_helm_fb(Dev d, StepCoef sc, int it) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, NT = C::NT;
  __shared__ double sD[N * N], sDt[N * N];
  __shared__ double sz[NN], st[3 * NN];
  __shared__ double sred[3 * 16];
  __shared__ int sW[NT * 8];
  const int tid = threadIdx.x;
  const long long e = xcd_element(blockIdx.x, gridDim.x);
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  const long long l = e * NN + tid, nl = d.cs;
  const int par = it & 1;
  bool done[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) done[c] = __builtin_amdgcn_readfirstlane((int)(d.hscal[par * 16 + c * 4 + 2] != 0.0)) != 0;   // written by k_helm_fa(it)
  if (done[0] && done[1] && done[2]) return;
  for (int q = tid; q < N * N; q += NT) { const double v = d.D[q]; sD[q] = v; sDt[(q % N) * N + q / N] = v; }
  int4 tab = make_int4(0, -1, -1, -1);
  double bm = 0, g[6] = {0, 0, 0, 0, 0, 0}, mk = 0, mi = 0;
  if (act) {
    tab = d.gs_tab[l];
    bm = d.bm1[l]; mk = d.mask[l]; mi = d.minv[l];
    load_g6(d, l, g);
    const int cid = corner_id<N>(k, j, i);
    gs_wide_stage(d, tab, l, sW + tid * 8, (cid >= 0 && d.gs_corner) ? reinterpret_cast<const int4*>(d.gs_corner + ((size_t)e * 8 + cid) * 8) : nullptr);
  }
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    if (done[c]) continue;
    const long long lc = c * nl + l;
    double r = 0.0, z = 0.0;
    if (act) {
      const double* yl = d.hy + c * nl;
      const GsVals gv = gs_load(yl, tab, l);
      r = d.hr[lc];
      z = mk * (tab.x < 0 ? gs_wide_sum(yl, d, l, sW + tid * 8) : gs_sum(gv, yl, d, tab, l));
      d.hz[lc] = z;
      sz[tid] = z;
    }
    lds_barrier();
    double au[1];
    axhelm3<N, 1>(sD, sDt, sz, st, act, k, j, i, g, au);
    double v[3] = {0, 0, 0};
    if (act) {
      const double wl = d.nu * au[0] + sc.h2 * bm * z;
      d.hwl[((size_t)par * 3 + c) * nl + l] = wl;
      v[0] = r * z * mi; v[1] = z * wl; v[2] = r * r * mi;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double x = wave_sum63(v[q]);
      if ((tid & 63) == 63) sred[q * 16 + (tid >> 6)] = x;
    }
    lds_barrier();
    if (tid < 3) {
      double s = 0.0;
      for (int w = 0; w < NT / 64; ++w) s += sred[tid * 16 + w];
      d.hpart[((size_t)par * 12 + c * 3 + tid) * d.nblk + e] = s;
    }
  }
}

// ---------------------------------------------------------------------------
// K4: u* = u + du ;  g = -D u*  -> V[0], |g|^2 partials; checks the Helmholtz solve  [UPSTREAM incomprp]
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_pres_rhs(Dev d, StepCoef sc, int helm_par, int check_helm) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[3 * NN], sA[2 * NNM], sB[3 * NMM];
  __shared__ double sred[12 * 16];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  const bool act = tid < NN;
  const long long l = e * NN + tid, nl = d.cs;
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  if (check_helm && blockIdx.x == 0) {
    const double* s = d.htot + helm_par * 16;
    if (tid == 0) {
      double worst = 0.0; int bad = 0;
      for (int c = 0; c < 3; ++c) {
        const double res = sqrt(s[c * 3 + 2] / d.vol);
        const double tol = d.tol_relative ? d.tol_helm * d.hscal[32 + c] : d.tol_helm;
        const bool was = d.hscal[helm_par * 16 + c * 4 + 2] != 0.0;
        const double rr = was ? d.hscal[helm_par * 16 + c * 4 + 3] : res;
        worst = fmax(worst, rr);
        if (!was && !(res <= tol)) bad = 1;
        if (!was) {
          if (c == 0) atomicAdd((unsigned long long*)&d.stats->helm_iters, (unsigned long long)check_helm);
          atomicMax((unsigned long long*)&d.stats->max_helm, (unsigned long long)check_helm);
          atomicMax((unsigned long long*)&d.stats->max_helm_k[sc.cls], (unsigned long long)check_helm); rec_step_iters(d, 0, check_helm);
        }
      }
      d.stats->last_helm_res = worst;
      if (bad) atomicAdd((unsigned long long*)&d.stats->unconverged, 1ull);
    }
  }
  if (act) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long lc = c * nl + l;
      const double l1 = d.dulag[lc], l2 = d.dulag[3 * nl + lc], l3 = d.dulag[6 * nl + lc];
      const double du = sc.xg[0] * l1 + sc.xg[1] * l2 + sc.xg[2] * l3 + d.hx[lc];
      d.dulag[6 * nl + lc] = l2;
      d.dulag[3 * nl + lc] = l1;
      d.dulag[lc] = du;
      const double us = d.u[lc] + du;
      d.u[lc] = us;
      su[c * NN + tid] = us;
    }
  }
  double w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool pact = tid < MM;
  const long long q = e * MM + tid;
  if (pact) load_w2(d, q, w2);
  lds_barrier();
  const double div = opdiv3<N>(sJ12, sD12, su, sA, sB, tid, NT, w2);
  double v[1] = {0.0};
  double g = 0.0;
  if (pact) {
    g = -div;
    d.V[q] = g;
    v[0] = g * g;
  }
  block_reduce<1>(v, sred, tid, NT);
  if (tid == 0) d.gpart[blockIdx.x] = v[0];
  if (!d.has_outflow) {
    double t[1] = {g};
    block_reduce<1>(t, sred, tid, NT);
    if (tid == 0) d.gpart[(size_t)d.nblk + blockIdx.x] = t[0];
  }
  if (d.nproj_max > 0) {
    if (tid == 0) d.ppart[(size_t)MAXPROJ * d.nblk + blockIdx.x] = v[0];
    const int np = d.gsc->nproj;
    __shared__ double sdot[MAXPROJ * 16];
    const int lane = tid & 63, wv = tid >> 6;
    for (int kk = 0; kk < np; ++kk) {
      double t = pact ? g * d.PX[(size_t)kk * d.npr + q] : 0.0;
      t = wave_sum63(t);
      if (lane == 63) sdot[kk * 16 + wv] = t;
    }
    lds_barrier();
    if (tid < np) {
      double t = 0.0;
      for (int ww = 0; ww < NT / 64; ++ww) t += sdot[tid * 16 + ww];
      d.ppart[(size_t)tid * d.nblk + blockIdx.x] = t;
    }
  }
}

// Pieces of the Gram-Schmidt passes over the GMRES basis V[0..j] (one pressure dof per lane, basis vectors 86 MB apart at config
// 4's size).  The trip count is a run-time value, so the plain loops issued ONE load, waited, and used it; here the loads of four
// basis vectors are in flight together (same order of operations; results equal to the last bit or two), and only the wavefronts
// that hold pressure dofs (lx2^3 of the lx1^3 lanes: 4 of 8 at lx1 = 8) take part in the reductions.  Config 4: 3.27 -> 3.17 ms
// per GMRES iteration.
__device__ inline double basis_subtract(const Dev& d, double w, const double* coef, long long q, int j) {
  for (int k0 = 0; k0 <= j; k0 += 4) {
    double v[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = k0 + u <= j;
      v[u] = in ? d.V[(size_t)(k0 + u) * d.ps + q] : 0.0;
      c[u] = in ? coef[k0 + u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) w -= c[u] * v[u];
  }
  return w;
}
// sdot[kk * 16 + wave] = sum over the wavefront of w * V[kk] (kk <= j) and of w * w (kk = j + 1)
template <int N>
__device__ inline void basis_dots(const Dev& d, double w, bool pact, long long q, int j, int lane, int wv, double* sdot) {
  constexpr int PW = (Cfg<N>::MM + 63) / 64;
  if (wv >= PW) return;
  if (!pact) w = 0.0;
  for (int k0 = 0; k0 <= j + 1; k0 += 4) {
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + u;
      x[u] = (pact && kk <= j) ? d.V[(size_t)kk * d.ps + q] : ((pact && kk == j + 1) ? w : 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double y = wave_sum63(w * x[u]);
      if (lane == 63 && k0 + u <= j + 1) sdot[(k0 + u) * 16 + wv] = y;
    }
  }
}

// Second Gram-Schmidt pass of the pressure GMRES.  The hexahedral solves take 20-40 iterations and single-pass
// classical Gram-Schmidt with the Pythagorean norm loses orthogonality there (measured: residual estimate 1e-8 against
// a true residual of 5e-2), so here  w' = w - sum_i h_i v_i  is formed and projected once more:
// gpart2[k] = (w', v_k), k <= j, and gpart2[j+1] = (w', w').   [UPSTREAM navier1.f uzawa_gmres uses modified GS]
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_gmres_reorth(Dev d, int j) {
  using C = Cfg<N>;
  constexpr int MM = C::MM, NT = C::NT;
  __shared__ double sh[MAXMR + 2];
  __shared__ double sdot[(MAXMR + 2) * 16];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  if (d.gsc->done) return;
  if (tid <= j) sh[tid] = d.gtot[tid];
  lds_barrier();
  const bool pact = tid < MM;
  const long long q = e * MM + tid;
  double w = 0.0;
  if (pact) {
    w = basis_subtract(d, d.V[(size_t)(j + 1) * d.ps + q], sh, q, j);
    d.V[(size_t)(j + 1) * d.ps + q] = w;
  }
  const int lane = tid & 63, wv = tid >> 6;
  basis_dots<N>(d, w, pact, q, j, lane, wv, sdot);
  lds_barrier();
  if (tid <= j + 1) {
    double t = 0.0;
#pragma unroll
    for (int ww = 0; ww < (MM + 63) / 64; ++ww) t += sdot[tid * 16 + ww];
    d.gpart2[(size_t)tid * d.nblk + blockIdx.x] = t;
  }
}

// GMRES bookkeeping + next basis vector + element-corner restriction (8 trilinear vertex functions).
// j >= 0: runs after k_gmres_reorth; Hessenberg column = first-pass + second-pass coefficients.
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_gmres_update(Dev d, int j, double scale, int min_iter, int ord) {
  using C = Cfg<N>;
  constexpr int MM = C::MM, NT = C::NT;
  __shared__ double sv[MM];
  __shared__ double sh[MAXMR + 2], sc2[MAXMR + 2], scol[MAXMR + 2];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  GmresScal* G = d.gsc;
  const bool restart = (j == -2);               // open the next GMRES cycle on the residual k_gmres_restart left in V[0]
  if (restart) j = -1;
  if ((j >= 0 || restart) && G->done) return;
  double wnew = 0.0;
  if (tid < MM) wnew = d.V[(size_t)(j + 1) * d.ps + e * MM + tid];
  double hn;
  if (j < 0) {
    hn = sqrt(d.gtot[0]);
  } else {
    if (tid <= j) sh[tid] = d.gtot[tid];
    if (tid <= j + 1) sc2[tid] = d.gtot2[tid];
    lds_barrier();
    double s2 = 0.0;
    for (int q = 0; q <= j; ++q) s2 += sc2[q] * sc2[q];
    const double hn2 = sc2[j + 1] - s2;
    hn = sqrt(hn2 > 0.0 ? hn2 : 0.0);
  }
  const double hinv = (hn > 0.0) ? 1.0 / hn : 0.0;
  if (blockIdx.x == 0 && tid == 0) {
    if (restart) {
      G->nit_prev += G->nit;
      G->beta0 = hn; G->g[0] = hn; G->nit = 0; G->resid = hn * scale;
      G->pending = 0;
      if (!(hn > 0.0)) G->done = 1;
    } else if (j < 0) {
      G->beta0 = hn; G->g[0] = hn; G->nit = 0; G->nit_prev = 0; G->resid = hn * scale;
      G->pending = 0;
      if (d.nproj_max <= 0) G->gnorm0 = hn;
      const double tol0 = d.tol_relative ? fmax(d.tol_pres * G->gnorm0 * scale, d.tol_pres_floor) : d.tol_pres;
      const int dn = (!(hn > 0.0) || (min_iter <= 0 && hn * scale <= tol0)) ? 1 : 0;
      if (dn) d.stats->last_pres_res = hn * scale;
      if (!(hn * 0.0 == 0.0)) d.stats->nonfinite += 1;      // |g| is NaN or Inf: nothing below can repair it (map_finish returns NSK_ENAN)
      G->done = dn;
    } else {
      double* col = scol;
      for (int q = 0; q <= j; ++q) col[q] = sh[q] + sc2[q];
      col[j + 1] = hn;
      for (int q = 0; q < j; ++q) {
        const double cq = G->cs[q], sq = G->sn[q];
        const double t = cq * col[q] + sq * col[q + 1];
        col[q + 1] = -sq * col[q] + cq * col[q + 1];
        col[q] = t;
      }
      const double rho = sqrt(col[j] * col[j] + col[j + 1] * col[j + 1]);
      const double cj = (rho > 0.0) ? col[j] / rho : 1.0, sj = (rho > 0.0) ? col[j + 1] / rho : 0.0;
      G->cs[j] = cj; G->sn[j] = sj;
      col[j] = rho;
      for (int q = 0; q <= j; ++q) G->R[j * MAXMR + q] = col[q];
      const double gj = G->g[j];
      G->g[j] = cj * gj;
      G->g[j + 1] = -sj * gj;
      G->nit = j + 1;
      d.stats->pres_jsum += j;
      const double res = fabs(sj * gj) * scale;
      G->resid = res;
      const double tol = d.tol_relative ? fmax(d.tol_pres * G->gnorm0 * scale, d.tol_pres_floor) : d.tol_pres;
      if ((res <= tol && (j + 1) >= min_iter) || !(hn > 0.0) || (d.pres_cap > 0 && (j + 1) >= d.pres_cap)) {
        atomicAdd((unsigned long long*)&d.stats->pres_iters, (unsigned long long)(G->nit_prev + j + 1));
        atomicMax((unsigned long long*)&d.stats->max_pres, (unsigned long long)(G->nit_prev + j + 1));
        atomicMax((unsigned long long*)&d.stats->max_pres_k[ord], (unsigned long long)(G->nit_prev + j + 1)); rec_step_iters(d, 1, G->nit_prev + j + 1);
        d.stats->last_pres_res = res;
        if (!(res <= tol) && hn > 0.0) {           // ended by the cap, not by its tolerance: counted, never silent
          d.stats->capped_solves += 1;
          if (res / tol > d.stats->worst_cap_ratio) d.stats->worst_cap_ratio = res / tol;
        }
        G->done = 1;
      }
    }
  }
  if (tid < MM) {
    const long long q = e * MM + tid;
    double w = basis_subtract(d, wnew, sc2, q, j);
    w *= hinv;
    d.V[(size_t)(j + 1) * d.ps + q] = w;
    sv[tid] = w;
  }
  lds_barrier();
  for (int c = tid >> 6; c < 8; c += NT / 64) {          // one wavefront per corner
    const int lane = tid & 63;
    double s = 0.0;
    for (int kk = lane; kk < MM; kk += 64) s += d.hat[c * MM + kk] * sv[kk];
    s = wave_sum63(s);
    if (lane == 63) d.ec[e * 8 + c] = s;
  }
}

// ---------------------------------------------------------------------------
// Two-pass classical Gram-Schmidt of the hexahedral pressure GMRES with TWO reads of the basis per column instead of four
// (option "gs_lag", the default on single-rank hexahedral contexts).
//
// The classic sequence (k_divgs dots, k_gmres_reorth subtract + dots, k_gmres_update subtract + normalise) reads V_0..j four
// times per column, 86 MB per vector at config 4's size, with the 216 pressure nodes of an element on 512-thread workgroups
// (42 % of the lanes, four loads in flight each): 0.25-0.3 of the HBM rate.  Here
//   * k_divgs still emits the first-pass dots (read 1);
//   * k_gs_lag streams the basis ONCE (read 2): every lane holds its node's V_0..j in registers, forms
//     w' = w - V h1, stores it, and takes the second-pass dots h2 = V^T w', (w', w') from the same registers;
//   * the second correction v_{j+1} = (w' - V h2) / h_{j+1,j} is NOT applied by a pass of its own: the vector stays
//     stored as w' ("pending", coefficients in GmresScal::pc / phinv) and the NEXT k_gs_lag, which has V_0..j in registers
//     anyway, forms v_{j+1} on the fly and writes it back.  Until then its only readers are the preconditioner
//     (k_schwarz and the corner restriction, which take phinv * w' -- right-preconditioned GMRES is flexible in its
//     z_j, the Arnoldi relation E z_j = V_{j+2} h holds with the FINAL vectors whatever z_j was made from; the
//     difference is the size of the lost orthogonality, 1e-8) and k_divgs' first-pass dot against it, whose value
//     (w, v_{j+1}) follows from (w, w') and the dots against V_0..j by linearity (gs_lag prologue below).
// One wavefront per element (4 rows of 64 lanes over the 216 nodes at lx1 = 8), RB rows in flight together,
// JB = compile-time bound of j: every basis load of a row block is issued before the first use.
// ---------------------------------------------------------------------------
template <int N, int JB, int RB>
__global__ __launch_bounds__(256) void k_gs_lag(Dev d, int j) {
  using C = Cfg<N>;
  constexpr int MM = C::MM, ROWS = (MM + 63) / 64;
  __shared__ double sh[MAXMR + 2], spc[MAXMR + 2];
  __shared__ double shat[8 * MM];
  GmresScal* G = d.gsc;
  if (G->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pend = G->pending;
  const double phinv = pend ? G->phinv : 1.0;
  if (tid < MAXMR + 2) {
    sh[tid] = (tid <= j) ? d.gtot[tid] : 0.0;
    spc[tid] = (pend && tid < j) ? G->pc[tid] : 0.0;
  }
  for (int p = tid; p < 8 * MM; p += 256) shat[p] = d.hat[p];
  lds_barrier();
  if (tid == 0 && pend) {                       // (w, v_j) from (w, w'_{j-1}) and the dots against V_0..j-1
    double s = sh[j];
    for (int k = 0; k < j; ++k) s -= spc[k] * sh[k];
    sh[j] = s * phinv;
  }
  lds_barrier();
  if (blockIdx.x == 0 && tid <= j) G->hcol[tid] = sh[tid];      // first-pass column for k_gmres_col
  const double hj = sh[j];
  const size_t ps = (size_t)d.ps;
  double* __restrict__ Vj = d.V + (size_t)j * ps;
  double* __restrict__ Vw = d.V + (size_t)(j + 1) * ps;
  const long long nw = (long long)gridDim.x * 4;
  for (long long e = (long long)blockIdx.x * 4 + wv; e < d.nel; e += nw) {
    asm volatile("" ::: "memory");               // (the coefficients stay in LDS: hoisted out of this loop they cost 4 registers per basis vector)
    double acc[JB], accj = 0.0, accw = 0.0, cr[8];
#pragma unroll
    for (int k = 0; k < JB; ++k) acc[k] = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) cr[c] = 0.0;
    const unsigned eb = (unsigned)e * (unsigned)MM;
#pragma unroll 1
    for (int r0 = 0; r0 < ROWS; r0 += RB) {
      if (ROWS > RB) asm volatile("" ::: "memory");
      double v[RB][JB], t[RB], x[RB];
      unsigned bo[RB];
      bool in[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int kk = lane + 64 * (r0 + rb);
        in[rb] = (r0 + rb < ROWS) && kk < MM;
        bo[rb] = (eb + (in[rb] ? kk : 0)) * 8u;
      }
#pragma unroll
      for (int k = 0; k < JB; ++k) {
        const double* __restrict__ Vk = d.V + (size_t)k * ps;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) v[rb][k] = (in[rb] && k < j) ? ld_boff(Vk, bo[rb]) : 0.0;
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        t[rb] = in[rb] ? ld_boff(Vj, bo[rb]) : 0.0;
        x[rb] = in[rb] ? ld_boff(Vw, bo[rb]) : 0.0;
      }
      if (pend) {                                // v_j = (w'_{j-1} - sum_k pc[k] v_k) * phinv, formed here and written back
#pragma unroll
        for (int k = 0; k < JB; ++k) {
          const double ck = spc[k];
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) t[rb] -= ck * v[rb][k];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) { t[rb] *= phinv; if (in[rb]) st_boff(Vj, bo[rb], t[rb]); }
      }
#pragma unroll
      for (int k = 0; k < JB; ++k) {             // first pass: w' = w - sum_k h1[k] v_k
        const double ck = sh[k];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) x[rb] -= ck * v[rb][k];
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) { x[rb] -= hj * t[rb]; if (in[rb]) st_boff(Vw, bo[rb], x[rb]); }
#pragma unroll
      for (int k = 0; k < JB; ++k)               // second-pass dots from the same registers
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[k] += x[rb] * v[rb][k];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        accj += x[rb] * t[rb];
        accw += x[rb] * x[rb];
        const int kc = in[rb] ? lane + 64 * (r0 + rb) : 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) cr[c] += shat[c * MM + kc] * x[rb];
      }
    }
#pragma unroll
    for (int k = 0; k < JB; ++k) {
      if (k < j) {
        const double s = wave_sum63(acc[k]);
        if (lane == 63) d.gpart2[(size_t)k * d.nblk + e] = s;
      }
    }
    accj = wave_sum63(accj);
    accw = wave_sum63(accw);
    if (lane == 63) { d.gpart2[(size_t)j * d.nblk + e] = accj; d.gpart2[(size_t)(j + 1) * d.nblk + e] = accw; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const double s = wave_sum63(cr[c]);
      if (lane == 63) d.ec[e * 8 + c] = s;       // restriction of w' (k_coarse_restrict_csr applies phinv)
    }
  }
}

// First-pass dots (w, V_k), k <= j, and (w, w) as a streaming pass of their own (gs_lag = 2: k_divgs then runs without its
// dots, whose loads come from 42 % of its lanes): same wavefront-per-element layout and partial sums as k_gs_lag.
template <int N, int JB, int RB>
__global__ __launch_bounds__(256) void k_gs_dots(Dev d, int j) {
  using C = Cfg<N>;
  constexpr int MM = C::MM, ROWS = (MM + 63) / 64;
  if (d.gsc->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t ps = (size_t)d.ps;
  const double* __restrict__ Vw = d.V + (size_t)(j + 1) * ps;
  const long long nw = (long long)gridDim.x * 4;
  for (long long e = (long long)blockIdx.x * 4 + wv; e < d.nel; e += nw) {
    double acc[JB], accw = 0.0;
#pragma unroll
    for (int k = 0; k < JB; ++k) acc[k] = 0.0;
    const unsigned eb = (unsigned)e * (unsigned)MM;
#pragma unroll 1
    for (int r0 = 0; r0 < ROWS; r0 += RB) {
      double v[RB][JB], x[RB];
      unsigned bo[RB];
      bool in[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int kk = lane + 64 * (r0 + rb);
        in[rb] = (r0 + rb < ROWS) && kk < MM;
        bo[rb] = (eb + (in[rb] ? kk : 0)) * 8u;
      }
#pragma unroll
      for (int k = 0; k < JB; ++k) {
        const double* __restrict__ Vk = d.V + (size_t)k * ps;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) v[rb][k] = (in[rb] && k <= j) ? ld_boff(Vk, bo[rb]) : 0.0;
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) x[rb] = in[rb] ? ld_boff(Vw, bo[rb]) : 0.0;
#pragma unroll
      for (int k = 0; k < JB; ++k)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[k] += x[rb] * v[rb][k];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) accw += x[rb] * x[rb];
    }
#pragma unroll
    for (int k = 0; k < JB; ++k) {
      if (k <= j) {
        const double s = wave_sum63(acc[k]);
        if (lane == 63) d.gpart[(size_t)k * d.nblk + e] = s;
      }
    }
    accw = wave_sum63(accw);
    if (lane == 63) d.gpart[(size_t)(j + 1) * d.nblk + e] = accw;
  }
}

// Hessenberg column j of the lagged scheme: first-pass coefficients (GmresScal::hcol, written by k_gs_lag) + second-pass
// dots (gtot2), h_{j+1,j} from the Pythagorean form, Givens rotation and convergence test as k_gmres_update; leaves the
// pending correction of the new basis vector in GmresScal::pc / phinv.
__global__ __launch_bounds__(64) void k_gmres_col(Dev d, int j, double scale, int min_iter, int ord) {
  __shared__ double col[MAXMR + 2];
  GmresScal* G = d.gsc;
  if (G->done || threadIdx.x != 0) return;
  double s2 = 0.0;
  for (int q = 0; q <= j; ++q) { const double h2 = d.gtot2[q]; s2 += h2 * h2; G->pc[q] = h2; col[q] = G->hcol[q] + h2; }
  const double hn2 = d.gtot2[j + 1] - s2;
  const double hn = sqrt(hn2 > 0.0 ? hn2 : 0.0);
  G->phinv = (hn > 0.0) ? 1.0 / hn : 0.0;
  G->pending = 1;
  col[j + 1] = hn;
  for (int q = 0; q < j; ++q) {
    const double cq = G->cs[q], sq = G->sn[q];
    const double t = cq * col[q] + sq * col[q + 1];
    col[q + 1] = -sq * col[q] + cq * col[q + 1];
    col[q] = t;
  }
  const double rho = sqrt(col[j] * col[j] + col[j + 1] * col[j + 1]);
  const double cj = (rho > 0.0) ? col[j] / rho : 1.0, sj = (rho > 0.0) ? col[j + 1] / rho : 0.0;
  G->cs[j] = cj; G->sn[j] = sj;
  col[j] = rho;
  for (int q = 0; q <= j; ++q) G->R[j * MAXMR + q] = col[q];
  const double gj = G->g[j];
  G->g[j] = cj * gj;
  G->g[j + 1] = -sj * gj;
  G->nit = j + 1;
  d.stats->pres_jsum += j;
  const double res = fabs(sj * gj) * scale;
  G->resid = res;
  const double tol = d.tol_relative ? fmax(d.tol_pres * G->gnorm0 * scale, d.tol_pres_floor) : d.tol_pres;
  if ((res <= tol && (j + 1) >= min_iter) || !(hn > 0.0) || (d.pres_cap > 0 && (j + 1) >= d.pres_cap)) {
    atomicAdd((unsigned long long*)&d.stats->pres_iters, (unsigned long long)(G->nit_prev + j + 1));
    atomicMax((unsigned long long*)&d.stats->max_pres, (unsigned long long)(G->nit_prev + j + 1));
    atomicMax((unsigned long long*)&d.stats->max_pres_k[ord], (unsigned long long)(G->nit_prev + j + 1)); rec_step_iters(d, 1, G->nit_prev + j + 1);
    d.stats->last_pres_res = res;
    if (!(res <= tol) && hn > 0.0) {             // ended by the cap, not by its tolerance: counted, never silent (as k_gmres_update)
      d.stats->capped_solves += 1;
      if (res / tol > d.stats->worst_cap_ratio) d.stats->worst_cap_ratio = res / tol;
    }
    G->done = 1;
  }
}

// z_j = restricted overlapping Schwarz (v_j) + R^T x_c ;  yl = D^T z_j (unassembled)
// Patch = the element's Gauss nodes plus the adjacent Gauss layer of every face neighbour = an lx1^3 tensor grid
// (GLL position n <-> patch position n; own Gauss index a <-> position a+1).  The neighbours' layers come through a
// gather table derived at set-up from the velocity-mesh dssum lists, which resolves their orientation (p_idx).  Local solve by fast diagonalisation:
// fdS = [nel][3][N*N] generalised eigenvectors S_d[pos][mode] of the 1-D pairs (A_d, M_d), fdL = [nel][3][N]:
//   z = (S_t x S_s x S_r) diag(1/(lr+ls+lt)) (S_t x S_s x S_r)^T w,  restricted to the element's own nodes.
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_schwarz(Dev d, const double* __restrict__ vin,
                                                        double* __restrict__ zout, int use_coarse, int check_done) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double sS[3 * N * N], sL[3 * N];
  __shared__ double sa[NN], sb[NN];
#ifdef NSK_NO_MFMA_OPS
  constexpr bool PRE = false;
#else
  constexpr bool PRE = (N <= 8);                                       // all nine metric products in LDS up front (lx1 = 10: the LDS would halve the occupancy)
#endif
  __shared__ double sP[(PRE ? 9 : 3) * MM], sC[3 * NMM], sE[2 * NNM];
  const int tid = threadIdx.x;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);      // XCD-contiguous runs of elements: the neighbours' face lines hit the L2 that streams them
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  if (check_done && d.gsc->done) return;
  NSK_STAMP(0);
  const bool pact = tid < MM;
  const long long q = e * MM + tid;
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  for (int p = tid; p < 3 * N * N; p += NT) sS[p] = d.fdS[(size_t)e * 3 * N * N + p];
  for (int p = tid; p < 3 * N; p += NT) sL[p] = d.fdL[(size_t)e * 3 * N + p];
  double w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, zc = 0.0;
  if (pact) {
    load_w2(d, q, w2);
    if (use_coarse) {
#pragma unroll
      for (int c = 0; c < 8; ++c) zc += d.hat[c * MM + tid] * d.xc[d.evert[e * 8 + c]];
    }
  }
  // (gs_lag: the newest basis vector is stored un-normalised until the next k_gs_lag finalises it)
  const double vsc = (d.gs_lag && d.gsc->pending) ? d.gsc->phinv : 1.0;
  if (act) {                                          // the patch, one thread per position
    const int id = d.p_idx[e * NN + tid];
    sa[tid] = (id >= 0) ? vsc * vin[id] : 0.0;
  }
  NSK_STAMP(1);
  lds_barrier();
  NSK_STAMP(2);
  // forward: S^T along r, s, t
  if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[0 * N * N + m * N + i] * sa[(k * N + j) * N + m]; sb[tid] = s; }
  lds_barrier();
  if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[1 * N * N + m * N + j] * sb[(k * N + m) * N + i]; sa[tid] = s; }
  lds_barrier();
  if (act) {
    double s = 0; for (int m = 0; m < N; ++m) s += sS[2 * N * N + m * N + k] * sa[(m * N + j) * N + i];
    const double lam = sL[i] + sL[N + j] + sL[2 * N + k];
    sb[tid] = (lam > d.fd_eps) ? s / lam : 0.0;
  }
  lds_barrier();
  // back: S along t, s, r
  if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[2 * N * N + k * N + m] * sb[(m * N + j) * N + i]; sa[tid] = s; }
  lds_barrier();
  if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[1 * N * N + j * N + m] * sa[(k * N + m) * N + i]; sb[tid] = s; }
  lds_barrier();
  if (act) { double s = 0; for (int m = 0; m < N; ++m) s += sS[0 * N * N + i * N + m] * sb[(k * N + j) * N + m]; sa[tid] = s; }
  lds_barrier();
  NSK_STAMP(3);
  double z = 0.0;
  if (pact) {
    const int a = tid % M, b = (tid / M) % M, cc = tid / (M * M);
    z = sa[((cc + 1) * N + (b + 1)) * N + (a + 1)] + zc;       // restriction to the element's own nodes
    zout[q] = z;
    if constexpr (PRE) {
#pragma unroll
      for (int m = 0; m < 9; ++m) sP[((m % 3) * 3 + m / 3) * MM + tid] = z * w2[m];    // [component c = m % 3][axis a = m / 3]
    }
  }
  if constexpr (PRE) lds_barrier();
  double gp[3];
#ifdef NSK_NO_MFMA_OPS
  opgradt3<N>(sJ12, sD12, z, w2, sP, sC, sE, tid, NT, act, k, j, i, gp);
#else
  opgradt3_mfma<N, PRE>(sJ12, sD12, z, w2, sP, sC, sE, tid, NT, gp);
#endif
  NSK_STAMP(4);
  if (act) {
    const long long l = e * NN + tid;
    d.yl[l] = gp[0]; d.yl[d.cs + l] = gp[1]; d.yl[2 * d.cs + l] = gp[2];
  }
  NSK_STAMP(5);
}

// k_schwarz as a resident workgroup that walks its share of the elements (b, b + gridDim.x, ...) with the NEXT element's inputs
// in flight while the current one is solved.  One element of k_schwarz<8> at config 4's size: 4.2 us waiting for the patch values
// (gather table -> values, two dependent round trips), 2.9 us in the six fast-diagonalisation passes, 4.9 us in D^T; four
// workgroups per CU is the hardware maximum for 512 threads, so the waiting can only be hidden inside the workgroup.  Here:
//   - the gather-table entry of the element after next, the patch value / S, Lambda entry / coarse-vertex value of the next
//     element are loads issued at the top of an iteration and consumed at its bottom (7 registers);
//   - the six passes run on the matrix cores (fd_forward_mfma, fd_back_mfma);
//   - the D^T p stores go out after the next element's tile is in LDS, so that nothing waits for them.
// `count` elements from d.boff on, XCD-contiguous as in k_schwarz (a step of gridDim.x, a multiple of 8, stays in the XCD's run).
template <int N>
__device__ __forceinline__ void schwarz_p_body(const Dev& d, const double* __restrict__ vin, double* __restrict__ zout,
                                               int use_coarse, int check_done, int count) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  constexpr int NS = 3 * N * N, NSL = NS + 3 * N;
  static_assert(NSL <= NT, "one S / Lambda entry per thread");
  __shared__ double sJ12[NM], sD12[NM];
  using L = PadLay<N>;                                 // bank-conflict-free tile strides (nsk3_mfma_ops.hpp)
  __shared__ double sSL[NSL], sH[2 * M];
  __shared__ double sa[L::FEXT], sb[L::FEXT];
  constexpr bool PRE = (N <= 8);
  __shared__ double sP[(PRE ? 9 : 3) * MM], sC[3 * L::CEXT], sE[2 * L::EEXT];
  const int tid = threadIdx.x;
  if (check_done && d.gsc->done) return;
  const bool act = tid < NN, pact = tid < MM;
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  const double vsc = (d.gs_lag && d.gsc->pending) ? d.gsc->phinv : 1.0;
  const unsigned G = gridDim.x;
  unsigned b = blockIdx.x;
  long long e = d.boff + xcd_element(b, count);
  // pipeline registers: idn = gather-table entry (element after next); pv / svS, svL / xl = patch value, S and Lambda entry, coarse
  // value of vertex (tid & 7) (next element), evl = that vertex' index (element after next).  The loads are unconditional
  // (clamped addresses) and their values are not touched before the bottom of the iteration: a select or a scale at issue
  // time would be a use, and the wait for it would sit right behind the load.
  int idn = 0, evl = 0;
  bool pok = false;
  double pv = 0.0, svS = 0.0, svL = 0.0, xl = 0.0;
  const unsigned tS = min((unsigned)tid, (unsigned)(NS - 1)), tL = min((unsigned)max(tid - NS, 0), (unsigned)(3 * N - 1)), tA = min((unsigned)tid, (unsigned)(NN - 1));
  {
    const int id = d.p_idx[e * NN + tA];
    pv = (act && id >= 0) ? vsc * vin[id < 0 ? 0 : id] : 0.0;
    svS = d.fdS[(size_t)e * NS + tS]; svL = d.fdL[(size_t)e * 3 * N + tL];
    if (use_coarse) xl = d.xc[d.evert[e * 8 + (tid & 7)]];
    const long long e1 = d.boff + xcd_element((b + G < (unsigned)count) ? b + G : b, count);
    idn = d.p_idx[e1 * NN + tA];
    if (use_coarse) evl = d.evert[e1 * 8 + (tid & 7)];
  }
  if (act) sa[(tid / (N * N)) * L::F[0][0] + ((tid / N) % N) * L::F[0][1] + (tid % N) * L::F[0][2]] = pv;
  if (tid < NSL) sSL[tid] = (tid < NS) ? svS : svL;
  double xcur = xl;
  if (tid < 2 * M) sH[tid] = d.hat[8 * MM + tid];      // [2][M]: the 1-D factors of the eight vertex functions
  int it = 0;
  for (;; ++it) {
#define SW_STAMP(i) do { if (it == 2) NSK_STAMP(i); } while (0)
    SW_STAMP(0);
    // (the lane index is made opaque per iteration: otherwise every lane-dependent address and LDS offset of the passes below is
    //  hoisted out of the loop and the kernel needs twice the registers -- half the workgroups per CU)
    unsigned tl = tid;
    asm volatile("" : "+v"(tl));
    const bool pactl = tl < MM, actl = tl < NN;
    lds_barrier();                                     // tile, S / Lambda (and sH) of this element are in LDS
    // uniform base + 32-bit lane offset (global_load saddr form): one address register for the nine metric loads
    double w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (pactl) {
#pragma unroll
      for (int m = 0; m < 9; ++m) w2[m] = ((d.zmask >> m) & 1u) ? 0.0 : ld_boff(d.w2m + (size_t)m * d.npr + e * MM, tl * 8u);      // parked in sP after the forward passes
    }
    double zc = 0.0;
    if (use_coarse) {                                  // R^T x_c: the eight vertex values sit in lanes 0..7 of every wave
      const unsigned a = tl % M, bb = (tl / M) % M, cc = (tl / (M * M)) % M;
      const double hr[2] = {sH[a], sH[M + a]}, hs[2] = {sH[bb], sH[M + bb]}, ht[2] = {sH[cc], sH[M + cc]};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const double xv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xcur), c), __builtin_amdgcn_readlane(__double2loint(xcur), c));
        zc += ((hr[c & 1] * hs[(c >> 1) & 1]) * ht[c >> 2]) * xv;         // = hat[c][tl] * xv, same products as the table
      }
    }
    SW_STAMP(1);
    fd_forward_mfma<N, L>(sSL, sSL + NS, sa, sb, d.fd_eps, (int)tl, NT);
    SW_STAMP(2);
    if constexpr (PRE) {
      if (pactl) {
#pragma unroll
        for (int m = 0; m < 9; ++m) sP[((m % 3) * 3 + m / 3) * MM + tl] = w2[m];      // [component c = m % 3][axis a = m / 3]
      }
    }
    SW_STAMP(3);
    // next element (the last iteration re-reads its own: harmless).  Issued here, behind the only wait of the iteration (the
    // metrics above), and consumed at the bottom: in flight under the backward passes and D^T.
    const unsigned bn = b + G;
    const bool more = bn < (unsigned)count;
    const long long en = d.boff + xcd_element(more ? bn : b, count);
    {
      const unsigned tSl = min(tl, (unsigned)(NS - 1)), tLl = min((unsigned)max((int)tl - NS, 0), (unsigned)(3 * N - 1)), tAl = min(tl, (unsigned)(NN - 1));
      pok = idn >= 0;
      pv = ld_boff(vin, (unsigned)max(idn, 0) * 8u);
      svS = ld_boff(d.fdS + (size_t)en * NS, tSl * 8u);
      svL = ld_boff(d.fdL + (size_t)en * 3 * N, tLl * 8u);
      if (use_coarse) xl = ld_boff(d.xc, (unsigned)evl * 8u);
      const long long e2 = d.boff + xcd_element((bn + G < (unsigned)count) ? bn + G : (more ? bn : b), count);
      idn = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(d.p_idx + e2 * NN) + tAl * 4u);
      if (use_coarse) evl = d.evert[e2 * 8 + (tl & 7)];
    }
    fd_back_mfma<N, L>(sSL, sa, sb, (int)tl, NT);
    SW_STAMP(4);
    double z = 0.0;
    if (pactl) {
      const unsigned a = tl % M, bb = (tl / M) % M, cc = tl / (M * M);
      z = sa[(cc + 1) * L::F[6][0] + (bb + 1) * L::F[6][1] + (a + 1) * L::F[6][2]] + zc;       // restriction to the element's own nodes
      st_boff(zout + e * MM, tl * 8u, z);
      if constexpr (PRE) {
#pragma unroll
        for (int m = 0; m < 9; ++m) sP[m * MM + tl] *= z;
      }
    }
    if constexpr (PRE) lds_barrier();
    SW_STAMP(5);
    double gp[3];
    opgradt3_mfma<N, PRE, L>(sJ12, sD12, z, w2, sP, sC, sE, (int)tl, NT, gp);       // (ends with a barrier: sa / sSL are free)
    SW_STAMP(6);
    if (more) {
      if (actl) sa[(tl / (N * N)) * L::F[0][0] + ((tl / N) % N) * L::F[0][1] + (tl % N) * L::F[0][2]] = pok ? vsc * pv : 0.0;
      if (tl < NSL) sSL[tl] = (tl < NS) ? svS : svL;
      xcur = xl;
    }
    SW_STAMP(7);
    if (actl) {
      double* __restrict__ y = d.yl + e * NN;
      st_boff(y, tl * 8u, gp[0]); st_boff(y + d.cs, tl * 8u, gp[1]); st_boff(y + 2 * d.cs, tl * 8u, gp[2]);
    }
    SW_STAMP(8);
    if (!more) break;
    e = en; b = bn;
  }
#undef SW_STAMP
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_schwarz_p(Dev d, const double* __restrict__ vin, double* __restrict__ zout,
                                                          int use_coarse, int check_done, int count) {
  schwarz_p_body<N>(d, vin, zout, use_coarse, check_done, count);
}
// lx1 = 8: three workgroups per CU (<= 80 registers; left alone the scheduler takes 92 and two workgroups fit)
#ifndef NSK_SWP_WAVES
#define NSK_SWP_WAVES 6
#endif
template <>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NSK_SWP_WAVES, 8))) void k_schwarz_p<8>(
    Dev d, const double* __restrict__ vin, double* __restrict__ zout, int use_coarse, int check_done, int count) {
  schwarz_p_body<8>(d, vin, zout, use_coarse, check_done, count);
}

// k_schwarz with ONE WAVEFRONT PER ELEMENT (lx1 = 8: 512 nodes = 8 per lane).  What the counters said about the workgroup-per-element
// forms above at config 4's size (scripts/pmc_sq_summary.py): waves parked at barriers / waits for 65-75 % of their cycles, instruction
// issue the rest -- every one of the eight waves walks through all fourteen passes, most of them with no tile to compute, and the
// per-pass time (~0.5 us) is the same whether the LDS banks conflict or not.  Here a wave owns its element: no workgroup barrier
// at all (the LDS serves a wave's accesses in order), every tile of a pass is this wave's work and is unrolled with constant LDS
// offsets, the linear phases handle eight nodes per lane.  Twelve elements per CU are in flight (LDS: 12.4 KB per wave), so
// one wave's waiting for its loads is another wave's compute, and the registers of a CU are shared by 8-9 waves instead of 32.
template <int N>
__global__ __launch_bounds__(64) void k_schwarz_w(Dev d, const double* __restrict__ vin, double* __restrict__ zout,
                                                  int use_coarse, int check_done) {
  using C = Cfg<N>;
  using L = PadLay<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NM = N * M, NS = 3 * N * N, NSL = NS + 3 * N;
  constexpr int RN = (NN + 63) / 64, RM = (MM + 63) / 64, RS = (NSL + 63) / 64;
  constexpr int GBUF = GtWave<N, L>::BUF, FBUF = 2 * L::FEXT, BUF = GBUF > FBUF ? GBUF : FBUF;
  __shared__ double sJ12[NM], sD12[NM], sSL[NSL], sH[2 * M];
  __shared__ double buf[BUF];
  double* sa = buf; double* sb = buf + L::FEXT;                    // fast-diagonalisation tiles; afterwards the D^T intermediates (GtWave)
  const int lane = threadIdx.x;
  if (check_done && d.gsc->done) return;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);
  NSK_STAMP(0);
  // ---- every load of the element, issued together
  int id[RN];
#pragma unroll
  for (int r = 0; r < RN; ++r) { const int idx = r * 64 + lane; id[r] = (idx < NN) ? d.p_idx[e * NN + idx] : -1; }
  double sv[RS];
#pragma unroll
  for (int r = 0; r < RS; ++r) {
    const int idx = r * 64 + lane;
    sv[r] = (idx < NS) ? d.fdS[(size_t)e * NS + idx] : ((idx < NSL) ? d.fdL[(size_t)e * 3 * N + (idx - NS)] : 0.0);
  }
  double w2[9][RM];
#pragma unroll
  for (int m = 0; m < 9; ++m)
#pragma unroll
    for (int r = 0; r < RM; ++r) { const int idx = r * 64 + lane; w2[m][r] = (idx < MM && !((d.zmask >> m) & 1u)) ? d.w2m[(size_t)m * d.npr + e * MM + idx] : 0.0; }
  const double xl = use_coarse ? d.xc[d.evert[e * 8 + (lane & 7)]] : 0.0;
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, lane, 64);
  if (lane < 2 * M) sH[lane] = d.hat[8 * MM + lane];
  const double vsc = (d.gs_lag && d.gsc->pending) ? d.gsc->phinv : 1.0;
  double pv[RN];
#pragma unroll
  for (int r = 0; r < RN; ++r) pv[r] = (id[r] >= 0) ? vsc * vin[id[r]] : 0.0;
#pragma unroll
  for (int r = 0; r < RS; ++r) { const int idx = r * 64 + lane; if (idx < NSL) sSL[idx] = sv[r]; }
#pragma unroll
  for (int r = 0; r < RN; ++r) {
    const int idx = r * 64 + lane;
    if (idx < NN) sa[(idx / (N * N)) * L::F[0][0] + ((idx / N) % N) * L::F[0][1] + (idx % N) * L::F[0][2]] = pv[r];
  }
  wave_sync();
  NSK_STAMP(1);
  fd_forward_mfma<N, L, true>(sSL, sSL + NS, sa, sb, d.fd_eps, lane, 64);
  NSK_STAMP(2);
  fd_back_mfma<N, L, true>(sSL, sa, sb, lane, 64);
  NSK_STAMP(3);
  // ---- restriction to the element's own nodes + R^T x_c (the eight vertex values sit in lanes 0..7)
  double xv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) xv[c] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xl), c), __builtin_amdgcn_readlane(__double2loint(xl), c));
  double z[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    const int idx = r * 64 + lane;
    z[r] = 0.0;
    if (idx < MM) {
      const int a = idx % M, bb = (idx / M) % M, cc = idx / (M * M);
      double zc = 0.0;
      if (use_coarse) {
        const double hr[2] = {sH[a], sH[M + a]}, hs[2] = {sH[bb], sH[M + bb]}, ht[2] = {sH[cc], sH[M + cc]};
#pragma unroll
        for (int c = 0; c < 8; ++c) zc += ((hr[c & 1] * hs[(c >> 1) & 1]) * ht[c >> 2]) * xv[c];       // = hat[c][idx] * x_c
      }
      z[r] = sa[(cc + 1) * L::F[6][0] + (bb + 1) * L::F[6][1] + (a + 1) * L::F[6][2]] + zc;
      zout[e * MM + idx] = z[r];
    }
  }
  wave_sync();
  NSK_STAMP(4);
  opgradt3_wave<N, L, RM>(sJ12, sD12, z, w2, buf, lane, d.yl + e * NN, d.cs);
  NSK_STAMP(5);
}

// k_schwarz_w with sixteen elements per CU in flight instead of twelve: the fast-diagonalisation solve in place in ONE tile
// (fd_solve_inplace_wave), its factors inside the D^T buffer, the metrics loaded per component -> 10 KB of LDS and <= 128 registers.
template <int N>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_schwarz_w16(Dev d, const double* __restrict__ vin, double* __restrict__ zout,
                                                                                            int use_coarse, int check_done) {
  using C = Cfg<N>;
  using L = PadLay<N>;
  using S = SlimLay<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NM = N * M, NS = 3 * N * N, NSL = NS + 3 * N;
  constexpr int RN = (NN + 63) / 64, RM = (MM + 63) / 64, RS = (NSL + 63) / 64;
  constexpr int GBUF = GtWave<N, L>::BUF, FBUF = S::EXT + NSL, BUF = GBUF > FBUF ? GBUF : FBUF;
  __shared__ double sJ12[NM], sD12[NM], sH[2 * M];
  __shared__ double buf[BUF];
  double* sa = buf; double* sSL = buf + S::EXT;                    // tile + factors; afterwards the D^T intermediates (GtWave)
  const int lane = threadIdx.x;
  if (check_done && d.gsc->done) return;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);
  // Three trips to memory, not twelve (round 5: the instruction stream had a full wait behind the vertex list, behind the basis, behind
  // the prolongation weights and behind EVERY ONE of the eight patch values -- `cond ? scale * load : 0` compiles to a load and
  // its wait inside the branch -- : half of a wavefront's 31 us).  (1) everything addressable from the lane, the heads of the two
  // dependent chains first; (2) the vertex values and the patch values, unconditionally (index 0 where the patch has no node);
  // (3) nothing: the metrics of D^T go out per component under the passes.
  static_assert(NM <= 64 && 2 * M <= 64, "basis and prolongation weights: one entry per lane");
  const int ev = d.evert[e * 8 + (lane & 7)];            // (unconditional: a branch on use_coarse would hold the load and its wait)
  int id[RN];
#pragma unroll
  for (int r = 0; r < RN; ++r) { const int idx = r * 64 + lane; id[r] = d.p_idx[e * NN + (idx < NN ? idx : 0)]; }
  const int pend = d.gs_lag ? d.gsc->pending : 0;
  const double phinv = d.gs_lag ? d.gsc->phinv : 1.0;
  double sv[RS];
#pragma unroll
  for (int r = 0; r < RS; ++r) {
    const int idx = r * 64 + lane;
    sv[r] = (idx < NS) ? d.fdS[(size_t)e * NS + idx] : ((idx < NSL) ? d.fdL[(size_t)e * 3 * N + (idx - NS)] : 0.0);
  }
  const double bJ = (lane < NM) ? d.J12[lane] : 0.0, bD = (lane < NM) ? d.D12[lane] : 0.0;
  const double hh = (lane < 2 * M) ? d.hat[8 * MM + lane] : 0.0;
  const double xl = d.xc[ev];
  double pv[RN];
#pragma unroll
  for (int r = 0; r < RN; ++r) pv[r] = vin[id[r] >= 0 ? id[r] : 0];
  const double vsc = pend ? phinv : 1.0;
#pragma unroll
  for (int r = 0; r < RS; ++r) { const int idx = r * 64 + lane; if (idx < NSL) sSL[idx] = sv[r]; }
  if (lane < NM) { sJ12[lane] = bJ; sD12[lane] = bD; }
  if (lane < 2 * M) sH[lane] = hh;
#pragma unroll
  for (int r = 0; r < RN; ++r) {
    const int idx = r * 64 + lane;
    if (idx < NN) sa[(idx / (N * N)) * S::PS + ((idx / N) % N) * S::RS + (idx % N)] = (id[r] >= 0) ? vsc * pv[r] : 0.0;
  }
  wave_sync();
  fd_solve_inplace_wave<N>(sSL, sSL + NS, sa, d.fd_eps, lane);
  double xv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) xv[c] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xl), c), __builtin_amdgcn_readlane(__double2loint(xl), c));
  double z[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    const int idx = r * 64 + lane;
    z[r] = 0.0;
    if (idx < MM) {
      const int a = idx % M, bb = (idx / M) % M, cc = idx / (M * M);
      double zc = 0.0;
      if (use_coarse) {
        const double hr[2] = {sH[a], sH[M + a]}, hs[2] = {sH[bb], sH[M + bb]}, ht[2] = {sH[cc], sH[M + cc]};
#pragma unroll
        for (int c = 0; c < 8; ++c) zc += ((hr[c & 1] * hs[(c >> 1) & 1]) * ht[c >> 2]) * xv[c];
      }
      z[r] = sa[(cc + 1) * S::PS + (bb + 1) * S::RS + (a + 1)] + zc;
      zout[e * MM + idx] = z[r];
    }
  }
  wave_sync();
  opgradt3_wave_ld<N, L, RM>(sJ12, sD12, z, d.w2m + e * MM, d.npr, d.zmask, buf, lane, d.yl + e * NN, d.cs);
}

// k_schwarz for lx1 = 10 as a SMALL workgroup per element: four wavefronts (256 threads, four patch nodes and two Gauss nodes per
// thread), 24 KB of LDS -> five or six elements per CU in flight, where k_schwarz<10> / k_schwarz_p<10> are one 1024-thread
// workgroup alone on its CU meeting at twenty-four barriers per element (4.2-4.9 ms per launch at config 5's size: 0.13-0.15 of
// 8 TB/s).  The loads as in k_schwarz_w16 (three trips: everything addressable from the thread, then vertex and patch values,
// the metrics per component under the passes), the six fast-diagonalisation passes and D^T on the matrix cores with the tiles
// of a pass spread over the four waves (fd_forward_mfma / fd_back_mfma, opgradt3_wg_ld), barriers of four waves.
template <int N>
__global__ __launch_bounds__(256) void k_schwarz_q(Dev d, const double* __restrict__ vin, double* __restrict__ zout,
                                                   int use_coarse, int check_done) {
  using C = Cfg<N>;
  using L = PadLay<N>;
  constexpr int NW = 4, NT = 64 * NW;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NM = N * M, NS = 3 * N * N, NSL = NS + 3 * N;
  constexpr int RN = (NN + NT - 1) / NT, RM = (MM + NT - 1) / NT, RS = (NSL + NT - 1) / NT;
  constexpr int GBUF = GtWave<N, L>::BUF, FBUF = 2 * L::FEXT, BUF = GBUF > FBUF ? GBUF : FBUF;
  static_assert(NM <= NT && 2 * M <= NT, "basis and prolongation weights: one entry per thread");
  __shared__ double sJ12[NM], sD12[NM], sSL[NSL], sH[2 * M];
  __shared__ double buf[BUF];
  double* sa = buf; double* sb = buf + L::FEXT;                    // fast-diagonalisation tiles; afterwards the D^T intermediates (GtWave)
  const int tid = threadIdx.x;
  if (check_done && d.gsc->done) return;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);
  // (1) everything addressable from the thread, the heads of the two dependent chains first; unconditional loads from clamped addresses
  const int ev = d.evert[e * 8 + (tid & 7)];
  int id[RN];
#pragma unroll
  for (int r = 0; r < RN; ++r) { const int idx = r * NT + tid; id[r] = d.p_idx[e * NN + (idx < NN ? idx : 0)]; }
  const int pend = d.gs_lag ? d.gsc->pending : 0;
  const double phinv = d.gs_lag ? d.gsc->phinv : 1.0;
  double sv[RS];
#pragma unroll
  for (int r = 0; r < RS; ++r) {
    const int idx = r * NT + tid;
    sv[r] = (idx < NS) ? d.fdS[(size_t)e * NS + idx] : ((idx < NSL) ? d.fdL[(size_t)e * 3 * N + (idx - NS)] : 0.0);
  }
  const double bJ = (tid < NM) ? d.J12[tid] : 0.0, bD = (tid < NM) ? d.D12[tid] : 0.0;
  const double hh = (tid < 2 * M) ? d.hat[8 * MM + tid] : 0.0;
  // (2) the vertex values and the patch values (index 0 where the patch has no node)
  const double xl = d.xc[ev];
  double pv[RN];
#pragma unroll
  for (int r = 0; r < RN; ++r) pv[r] = vin[id[r] >= 0 ? id[r] : 0];
  const double vsc = pend ? phinv : 1.0;
#pragma unroll
  for (int r = 0; r < RS; ++r) { const int idx = r * NT + tid; if (idx < NSL) sSL[idx] = sv[r]; }
  if (tid < NM) { sJ12[tid] = bJ; sD12[tid] = bD; }
  if (tid < 2 * M) sH[tid] = hh;
#pragma unroll
  for (int r = 0; r < RN; ++r) {
    const int idx = r * NT + tid;
    if (idx < NN) sa[(idx / (N * N)) * L::F[0][0] + ((idx / N) % N) * L::F[0][1] + (idx % N) * L::F[0][2]] = (id[r] >= 0) ? vsc * pv[r] : 0.0;
  }
  lds_barrier();
  fd_forward_mfma<N, L>(sSL, sSL + NS, sa, sb, d.fd_eps, tid, NT);
  fd_back_mfma<N, L>(sSL, sa, sb, tid, NT);
  // restriction to the element's own nodes + R^T x_c (the eight vertex values sit in lanes 0..7 of every wave)
  double xv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) xv[c] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xl), c), __builtin_amdgcn_readlane(__double2loint(xl), c));
  double z[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    const int idx = r * NT + tid;
    z[r] = 0.0;
    if (idx < MM) {
      const int a = idx % M, bb = (idx / M) % M, cc = idx / (M * M);
      double zc = 0.0;
      if (use_coarse) {
        const double hr[2] = {sH[a], sH[M + a]}, hs[2] = {sH[bb], sH[M + bb]}, ht[2] = {sH[cc], sH[M + cc]};
#pragma unroll
        for (int c = 0; c < 8; ++c) zc += ((hr[c & 1] * hs[(c >> 1) & 1]) * ht[c >> 2]) * xv[c];       // = hat[c][idx] * x_c
      }
      z[r] = sa[(cc + 1) * L::F[6][0] + (bb + 1) * L::F[6][1] + (a + 1) * L::F[6][2]] + zc;
      zout[e * MM + idx] = z[r];
    }
  }
  lds_barrier();                                         // (the tile is read: D^T reuses the buffer)
  opgradt3_wg_ld<N, L, RM, NW>(sJ12, sD12, z, d.w2m + e * MM, d.npr, d.zmask, buf, tid, d.yl + e * NN, d.cs);
}

// yl = D^T p for an arbitrary pressure vector
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_gradt(Dev d, const double* __restrict__ pin, double* __restrict__ yl) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double sP[3 * MM], sC[3 * NMM], sE[2 * NNM];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  double w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, z = 0.0;
  if (tid < MM) { z = pin[e * MM + tid]; load_w2(d, e * MM + tid, w2); }
  lds_barrier();
  double gp[3];
  opgradt3<N>(sJ12, sD12, z, w2, sP, sC, sE, tid, NT, act, k, j, i, gp);
  if (act) {
    const long long l = e * NN + tid;
    yl[l] = gp[0]; yl[d.cs + l] = gp[1]; yl[2 * d.cs + l] = gp[2];
  }
}

// w = D ( B^-1 mask dssum(yl) ) ; optional dots (w, V_i), i <= j, and (w,w)
// Two round trips to memory per workgroup: (1) everything addressable from the thread index -- gather table, 1/B, the nine metric
// terms of the pressure node, the basis, and the element corners' lists, one list ENTRY per thread (3 components x 8 corners x 8
// members = 192 threads); (2) the neighbours' values, one per corner-list entry.  The corner threads then add their eight values
// from LDS in list order (the left-to-right sum of gs_csr; a missing member is -0.0, the identity of the addition).
// (Until round 5 the basis was a round trip of its own ahead of the table, the metric terms went out after the gather and a
//  corner thread walked its three components' lists one after the other: five round trips, 9 of a workgroup's 14 us.)
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_divgs(Dev d, const double* __restrict__ yl,
                                                      double* __restrict__ wout, int j, int check_done) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  constexpr int CW = 192, NCS = (CW + NT - 1) / NT;
  static_assert(NM <= NT, "basis: one entry per thread");
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[3 * NN], sA[2 * NNM], sB[3 * NMM];
  __shared__ double scv[CW];
  __shared__ double sdot[(MAXMR + 2) * 16];
  const int tid = threadIdx.x;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);      // XCD-contiguous runs of elements: the neighbours' face lines hit the L2 that streams them
  const bool act = tid < NN;
  if (check_done && d.gsc->done) return;
  NSK_STAMP(0);
  const long long l = e * NN + tid;
  const bool pact = tid < MM;
  const long long q = e * MM + tid;
  // ---- round trip 1
  int cm[NCS];
#pragma unroll
  for (int s = 0; s < NCS; ++s) {
    const int p = tid + s * NT;
    cm[s] = (d.gs_corner && p < CW) ? d.gs_corner[(size_t)e * 64 + (p & 63)] : -1;
  }
  int4 tab = make_int4(0, -1, -1, -1);
  double bi = 0.0;
  if (act) { tab = d.gs_tab[l]; bi = d.binv[l]; }
  double w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (pact) load_w2(d, q, w2);
  // ---- round trip 2
  double cv[NCS];
#pragma unroll
  for (int s = 0; s < NCS; ++s) {
    const int p = tid + s * NT;
    cv[s] = (cm[s] >= 0) ? yl[(size_t)(p >> 6) * d.cs + cm[s]] : -0.0;
  }
  const int cid = act ? corner_id<N>(tid / (N * N), (tid / N) % N, tid % N) : -1;
  const bool wide = act && tab.x < 0;
  const bool from_list = wide && cid >= 0 && d.gs_corner;
  GsVals g0, g1, g2;
  if (act && !wide) { g0 = gs_load_o(yl, tab, (unsigned)l); g1 = gs_load_o(yl + d.cs, tab, (unsigned)l); g2 = gs_load_o(yl + 2 * d.cs, tab, (unsigned)l); }
  double bJ = 0.0, bD = 0.0;
  if (tid < NM) { bJ = d.J12[tid]; bD = d.D12[tid]; }
  if (act) {
    if (!wide) {
      su[tid] = bi * (((g0.a + g0.b) + g0.c) + g0.d);
      su[NN + tid] = bi * (((g1.a + g1.b) + g1.c) + g1.d);
      su[2 * NN + tid] = bi * (((g2.a + g2.b) + g2.c) + g2.d);
    } else if (!from_list) {                                  // more than four co-located nodes away from the corners: the CSR lists
#pragma unroll 1
      for (int c = 0; c < 3; ++c) su[c * NN + tid] = bi * gs_csr_rolled(yl + (size_t)c * d.cs, d, l);
    }
  }
#pragma unroll
  for (int s = 0; s < NCS; ++s) {
    const int p = tid + s * NT;
    if (p < CW) scv[p] = (cm[s] == -2) ? __builtin_nan("") : cv[s];
  }
  if (tid < NM) { sJ12[tid] = bJ; sD12[tid] = bD; }
  NSK_STAMP(1);
  lds_barrier();
  if (from_list) {
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
      const double* v = scv + c * 64 + cid * 8;
      double sum = 0.0;
      if (v[0] != v[0]) sum = gs_csr_rolled(yl + (size_t)c * d.cs, d, l);       // more than eight members (or a NaN in the data: same result)
      else {
#pragma unroll
        for (int m = 0; m < 8; ++m) sum += v[m];
      }
      su[c * NN + tid] = bi * sum;
    }
  }
  lds_barrier();
  NSK_STAMP(2);
  const double w = opdiv3<N>(sJ12, sD12, su, sA, sB, tid, NT, w2);
  NSK_STAMP(3);
  if (pact) wout[q] = w;
  if (j >= 0) {
    const int lane = tid & 63, wv = tid >> 6;
    basis_dots<N>(d, w, pact, q, j, lane, wv, sdot);
    lds_barrier();
    if (tid <= j + 1) {
      double t = 0.0;
#pragma unroll
      for (int ww = 0; ww < (MM + 63) / 64; ++ww) t += sdot[tid * 16 + ww];
      d.gpart[(size_t)tid * d.nblk + e] = t;               // slot = element: the order of the sums does not depend on how a launch is split or mapped
    }
  }
}

// k_divgs with the three components' pass chains side by side (opdiv3_mfma_c3): 5 workgroup barriers after the gather instead of 12.
// Same two round trips as k_divgs (the metrics go out first, in the accumulator order of the wavefront's last-pass tiles).
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_divgs_c3(Dev d, const double* __restrict__ yl,
                                                         double* __restrict__ wout, int j, int check_done) {
  using C = Cfg<N>;
  using W = DvWave<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M;
  static_assert(NT >= 192 && NM <= NT, "corner entries and the basis: one per thread");
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double buf[3 * W::BUF];
  __shared__ double scv[192];
  static_assert((MAXMR + 2) * 16 <= 3 * W::BUF, "the dot partials reuse the buffer");
  double* sdot = buf;
  const int tid = threadIdx.x;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);
  const bool act = tid < NN;
  if (check_done && d.gsc->done) return;
  const long long l = e * NN + tid;
  // ---- round trip 1
  const int cm = (d.gs_corner && tid < 192) ? d.gs_corner[(size_t)e * 64 + (tid & 63)] : -1;
  int4 tab = make_int4(0, -1, -1, -1);
  double bi = 0.0;
  if (act) { tab = d.gs_tab[l]; bi = d.binv[l]; }
  Dv3Met<N> mt;
  dv3_metrics<N>(d, e, tid, NT, mt);
  // ---- round trip 2
  const double cv = (cm >= 0) ? yl[(size_t)(tid >> 6) * d.cs + cm] : -0.0;
  const int cid = act ? corner_id<N>(tid / (N * N), (tid / N) % N, tid % N) : -1;
  const bool wide = act && tab.x < 0;
  const bool from_list = wide && cid >= 0 && d.gs_corner;
  GsVals g0, g1, g2;
  if (act && !wide) { g0 = gs_load_o(yl, tab, (unsigned)l); g1 = gs_load_o(yl + d.cs, tab, (unsigned)l); g2 = gs_load_o(yl + 2 * d.cs, tab, (unsigned)l); }
  double bJ = 0.0, bD = 0.0;
  if (tid < NM) { bJ = d.J12[tid]; bD = d.D12[tid]; }
  if (act) {
    if (!wide) {
      buf[W::oU + tid] = bi * (((g0.a + g0.b) + g0.c) + g0.d);
      buf[W::BUF + W::oU + tid] = bi * (((g1.a + g1.b) + g1.c) + g1.d);
      buf[2 * W::BUF + W::oU + tid] = bi * (((g2.a + g2.b) + g2.c) + g2.d);
    } else if (!from_list) {
#pragma unroll 1
      for (int c = 0; c < 3; ++c) buf[c * W::BUF + W::oU + tid] = bi * gs_csr_rolled(yl + (size_t)c * d.cs, d, l);
    }
  }
  if (tid < 192) scv[tid] = (cm == -2) ? __builtin_nan("") : cv;
  if (tid < NM) { sJ12[tid] = bJ; sD12[tid] = bD; }
  lds_barrier();
  if (from_list) {
#pragma unroll 1
    for (int c = 0; c < 3; ++c) buf[c * W::BUF + W::oU + tid] = bi * corner_list_sum(scv + c * 64 + cid * 8, yl + (size_t)c * d.cs, d, l);
  }
  const bool pact = tid < MM;
  const long long q = e * MM + tid;
  lds_barrier();
  const double w = opdiv3_mfma_c3<N>(mt, sJ12, sD12, buf, tid, NT);
  if (pact) wout[q] = w;
  if (j >= 0) {
    const int lane = tid & 63, wv = tid >> 6;
    lds_barrier();                                         // (the partial sums above are read; sdot reuses the buffer)
    basis_dots<N>(d, w, pact, q, j, lane, wv, sdot);
    lds_barrier();
    if (tid <= j + 1) {
      double t = 0.0;
#pragma unroll
      for (int ww = 0; ww < (MM + 63) / 64; ++ww) t += sdot[tid * 16 + ww];
      d.gpart[(size_t)tid * d.nblk + e] = t;
    }
  }
}

// k_divgs (without the Gram-Schmidt dots: the lagged Gram-Schmidt has them in k_gs_dots) with ONE WAVEFRONT PER ELEMENT, as
// k_schwarz_w: no workgroup barrier, eight nodes per lane in the gather, the three passes of a component unrolled with constant
// LDS offsets, the last pass combined with the metrics in the matrix-core accumulators (no staging tile), 11-12 elements per CU.
#ifndef NSK_DVW_WAVES
#define NSK_DVW_WAVES 3
#endif
template <int N>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NSK_DVW_WAVES))) void k_divgs_w(Dev d, const double* __restrict__ yl, double* __restrict__ wout, int check_done) {
  using C = Cfg<N>;
  using W = DvWave<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NM = N * M, KQ = (N + 3) / 4, RN = (NN + 63) / 64;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double buf[W::BUF];
  const int lane = threadIdx.x;
  if (check_done && d.gsc->done) return;
  const long long e = d.boff + xcd_element(blockIdx.x, gridDim.x);
  int4 tab[RN];
#pragma unroll
  for (int r = 0; r < RN; ++r) {
    const int idx = r * 64 + lane;
    const unsigned lo = (unsigned)(idx < NN ? idx : 0);
    tab[r] = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(d.gs_tab + e * NN) + lo * 16u);
  }
  // element corners (wide nodes): planes k = 0 and k = N - 1, lanes (j, i) in the corners of the 8 x 8 lane grid (N = 8; other
  // orders: found by corner_id)
  static_assert(N * N <= 64 || N == 10, "corner planes");
  CornerList CL0, CL1;                                 // planes k = 0 and k = N - 1 (r = 0 and r = RN - 1 at N = 8)
  {
    const int i0 = lane, i1 = (RN - 1) * 64 + lane;
    if (i0 < N * N) corner_issue(d, e, corner_id<N>(0, (i0 / N) % N, i0 % N), CL0); else CL0.id[0] = -2;
    if (i1 < NN && i1 / (N * N) == N - 1) corner_issue(d, e, corner_id<N>(N - 1, (i1 / N) % N, i1 % N), CL1); else CL1.id[0] = -2;
  }
  CornerList CLx; CLx.id[0] = -2;                      // every other node: the general list walk if its table entry says "wide"
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, lane, 64);
  wave_sync();
  const int m16 = lane & 15, kq = lane >> 4;
  double aDJ[KQ], aJ[KQ], aD[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = 4 * q + kq;
    const bool kok = k < N;
    aJ[q] = (m16 < M && kok) ? sJ12[m16 * N + k] : 0.0;
    aD[q] = (m16 < M && kok) ? sD12[m16 * N + k] : 0.0;
    aDJ[q] = !kok ? 0.0 : ((m16 < M) ? sD12[m16 * N + k] : ((m16 < 2 * M) ? sJ12[(m16 - M) * N + k] : 0.0));
  }
  double div[W::NT3][W::RQ];
#pragma unroll
  for (int t = 0; t < W::NT3; ++t)
#pragma unroll
    for (int r = 0; r < W::RQ; ++r) div[t][r] = 0.0;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    // (lane index opaque per component: otherwise every lane-dependent offset below is hoisted out of this loop and spills)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const double* __restrict__ f = yl + c * d.cs;
    // (two half-batches of RN / 2 planes: eight nodes per lane in flight at once is 64 registers of values alone)
    constexpr int RH = (RN + 1) / 2;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      GsVals g[RH];
      double bi[RH];                                     // (B^-1 mask: re-read per component, from the caches, rather than held)
#pragma unroll
      for (int rr = 0; rr < RH; ++rr) {
        const int r = hb * RH + rr, idx = r * 64 + ln;
        if (r < RN) {
          g[rr] = gs_load_o(f, tab[r], (unsigned)(e * NN) + (unsigned)(idx < NN ? idx : 0));
          bi[rr] = ld_boff(d.binv + e * NN, (unsigned)(idx < NN ? idx : 0) * 8u);
        }
      }
#pragma unroll
      for (int rr = 0; rr < RH; ++rr) {
        const int r = hb * RH + rr, idx = r * 64 + ln;
        if (r < RN && idx < NN) buf[W::oU + idx] = bi[rr] * gs_sum3(g[rr], f, d, tab[r], e * NN + idx, (r == 0) ? CL0 : ((r == RN - 1) ? CL1 : CLx));
      }
      asm volatile("" ::: "memory");
    }
    wave_sync();
    opdiv3_wave_comp<N>(aDJ, aJ, aD, buf, ln, d.w2m + (size_t)(0 * 3 + c) * d.npr + e * MM, d.w2m + (size_t)(1 * 3 + c) * d.npr + e * MM,
                        d.w2m + (size_t)(2 * 3 + c) * d.npr + e * MM, div);
  }
#pragma unroll
  for (int t = 0; t < W::NT3; ++t)
#pragma unroll
    for (int r = 0; r < W::RQ; ++r) {
      const int n = t * 16 + m16, cc = kq + 4 * r;
      if (n < M * M && cc < M) wout[e * MM + cc * M * M + n] = div[t][r];
    }
}

// after GMRES: dp = h2 * sum_i y_i Z_i (+ projected part) ; p = p* + dp ; yl = D^T dp
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_pres_update(Dev d, StepCoef sc) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double sP[3 * MM], sC[3 * NMM], sE[2 * NNM];
  __shared__ double sy[MAXMR], sg[MAXMR], sR[MAXMR * MAXMR];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  const bool act = tid < NN;
  const int k = tid / (N * N), j = (tid / N) % N, i = tid % N;
  const GmresScal* G = d.gsc;
  const int nit = G->nit;
  for (int p = tid; p < nit * nit; p += NT) { const int cc = p / nit, rr = p % nit; sR[cc * MAXMR + rr] = G->R[cc * MAXMR + rr]; }
  if (tid < nit) sg[tid] = G->g[tid];
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  lds_barrier();
  if (tid == 0) {
    for (int q = nit - 1; q >= 0; --q) {
      double s = sg[q];
      for (int kk = q + 1; kk < nit; ++kk) s -= sR[kk * MAXMR + q] * sy[kk];
      sy[q] = s / sR[q * MAXMR + q];
    }
  }
  lds_barrier();
  double dp = 0.0, w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (tid < MM) {
    const long long q = e * MM + tid;
    double x = (G->nit_prev > 0) ? d.xacc[q] : 0.0;  // completed GMRES cycles of a restarted solve
    for (int kk = 0; kk < nit; ++kk) x += sy[kk] * d.Z[(size_t)kk * d.npr + q];
    if (d.nproj_max > 0) {
      d.PD[q] = x;
      const int np = G->nproj;
      for (int kk = 0; kk < np; ++kk) x += G->pa[kk] * d.PX[(size_t)kk * d.npr + q];
    }
    dp = sc.h2 * x;
    d.p[q] = d.pext[q] + dp;
    load_w2(d, q, w2);
  }
  lds_barrier();
  double gp[3];
  opgradt3<N>(sJ12, sD12, dp, w2, sP, sC, sE, tid, NT, act, k, j, i, gp);
  if (act) {
    const long long l = e * NN + tid;
    d.yl[l] = gp[0]; d.yl[d.cs + l] = gp[1]; d.yl[2 * d.cs + l] = gp[2];
  }
  if (blockIdx.x == 0 && tid == 0 && !G->done) atomicAdd((unsigned long long*)&d.stats->unconverged, 1ull);
}

// ---------------------------------------------------------------------------
// The once-per-step linear algebra over the GMRES basis Z (<= 48 vectors) and the projection space PX / PEX (<= 32 vectors
// each, 86 MB per vector at config 4's size) as STREAMING passes (option "flat_proj", default on single-rank hexahedral
// contexts): inside the element kernels k_pres_update / k_vel_update_proj those loops ran on 216 of 512 lanes with one load
// in flight (2.1 + 2.2 ms per step at config 4); here one wavefront per element walks its 216 nodes in rows of 64 with every
// vector of a row loaded before the first use.
//   k_pres_comb:  x = xacc + Z y,  PD = x,  x += PX a,  dp = h2 x,  p = p* + dp,  dpw = dp     (then k_gradt: yl = D^T dp)
//   k_proj_dots:  E dp (raw, left in PED by k_vel_update_proj) -= PEX a;  (PD, PEX_k), (PD, E dp) partial sums
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void k_pres_comb(Dev d, StepCoef sc) {
  using C = Cfg<N>;
  constexpr int MM = C::MM, ROWS = (MM + 63) / 64;
  __shared__ double sy[MAXMR], sg[MAXMR], sR[MAXMR * MAXMR], spa[MAXPROJ];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const GmresScal* G = d.gsc;
  const int nit = G->nit;
  const int np = (d.nproj_max > 0) ? G->nproj : 0;
  for (int p = tid; p < nit * nit; p += 256) { const int cc = p / nit, rr = p % nit; sR[cc * MAXMR + rr] = G->R[cc * MAXMR + rr]; }
  if (tid < nit) sg[tid] = G->g[tid];
  if (tid < MAXPROJ) spa[tid] = (tid < np) ? G->pa[tid] : 0.0;
  if (tid < MAXMR && tid >= nit) sy[tid] = 0.0;
  lds_barrier();
  if (tid == 0) {
    for (int q = nit - 1; q >= 0; --q) {
      double s = sg[q];
      for (int kk = q + 1; kk < nit; ++kk) s -= sR[kk * MAXMR + q] * sy[kk];
      sy[q] = s / sR[q * MAXMR + q];
    }
  }
  lds_barrier();
  if (blockIdx.x == 0 && tid == 0 && !G->done) atomicAdd((unsigned long long*)&d.stats->unconverged, 1ull);
  const bool acc0 = G->nit_prev > 0;
  const size_t npr = (size_t)d.npr;
  const long long nw = (long long)gridDim.x * 4;
  for (long long e = (long long)blockIdx.x * 4 + wv; e < d.nel; e += nw) {
    asm volatile("" ::: "memory");
    const unsigned eb = (unsigned)e * (unsigned)MM;
#pragma unroll 1
    for (int r = 0; r < ROWS; ++r) {
      if (ROWS > 1) asm volatile("" ::: "memory");
      const int kk = lane + 64 * r;
      const bool in = kk < MM;
      const unsigned bo = (eb + (in ? kk : 0)) * 8u;
      double z[MAXMR], px[MAXPROJ];
#pragma unroll
      for (int k = 0; k < MAXMR; ++k) z[k] = (in && k < nit) ? ld_boff(d.Z + (size_t)k * npr, bo) : 0.0;
#pragma unroll
      for (int k = 0; k < MAXPROJ; ++k) px[k] = (in && k < np) ? ld_boff(d.PX + (size_t)k * npr, bo) : 0.0;
      const double pe = in ? ld_boff(d.pext, bo) : 0.0;
      double x = (in && acc0) ? ld_boff(d.xacc, bo) : 0.0;
#pragma unroll
      for (int k = 0; k < MAXMR; ++k) x += sy[k] * z[k];
      if (d.nproj_max > 0) {
        if (in) st_boff(d.PD, bo, x);
#pragma unroll
        for (int k = 0; k < MAXPROJ; ++k) x += spa[k] * px[k];
      }
      const double dp = sc.h2 * x;
      if (in) { st_boff(d.p, bo, pe + dp); st_boff(d.dpw, bo, dp); }
    }
  }
}

template <int N>
__global__ __launch_bounds__(256) void k_proj_dots(Dev d) {
  using C = Cfg<N>;
  constexpr int MM = C::MM, ROWS = (MM + 63) / 64;
  __shared__ double spa[MAXPROJ];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const GmresScal* G = d.gsc;
  if (G->nit == 0) return;
  const int np = G->nproj;
  if (tid < MAXPROJ) spa[tid] = (tid < np) ? G->pa[tid] : 0.0;
  lds_barrier();
  const size_t npr = (size_t)d.npr;
  const long long nw = (long long)gridDim.x * 4;
  for (long long e = (long long)blockIdx.x * 4 + wv; e < d.nel; e += nw) {
    asm volatile("" ::: "memory");
    double acc[MAXPROJ], accn = 0.0;
#pragma unroll
    for (int k = 0; k < MAXPROJ; ++k) acc[k] = 0.0;
    const unsigned eb = (unsigned)e * (unsigned)MM;
#pragma unroll 1
    for (int r = 0; r < ROWS; ++r) {
      if (ROWS > 1) asm volatile("" ::: "memory");
      const int kk = lane + 64 * r;
      const bool in = kk < MM;
      const unsigned bo = (eb + (in ? kk : 0)) * 8u;
      double pex[MAXPROJ];
#pragma unroll
      for (int k = 0; k < MAXPROJ; ++k) pex[k] = (in && k < np) ? ld_boff(d.PEX + (size_t)k * npr, bo) : 0.0;
      double edel = in ? ld_boff(d.PED, bo) : 0.0;
      const double del = in ? ld_boff(d.PD, bo) : 0.0;
#pragma unroll
      for (int k = 0; k < MAXPROJ; ++k) edel -= spa[k] * pex[k];
      if (in) st_boff(d.PED, bo, edel);
#pragma unroll
      for (int k = 0; k < MAXPROJ; ++k) acc[k] += del * pex[k];
      accn += del * edel;
    }
#pragma unroll
    for (int k = 0; k < MAXPROJ; ++k) {
      if (k < np) {
        const double s = wave_sum63(acc[k]);
        if (lane == 63) d.ppart[(size_t)k * d.nblk + e] = s;
      }
    }
    accn = wave_sum63(accn);
    if (lane == 63) d.ppart[(size_t)np * d.nblk + e] = accn;
  }
}

// velocity correction fused with E*dp for the projection space
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_vel_update_proj(Dev d, StepCoef sc) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[3 * NN], sA[2 * NNM], sB[3 * NMM];
  const int tid = threadIdx.x;
  const long long e = xcd_element(blockIdx.x, gridDim.x);
  const bool act = tid < NN;
  const GmresScal* G = d.gsc;
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  if (act) {
    const long long l = e * NN + tid;
    CornerList CL;
    corner_issue(d, e, corner_id<N>(tid / (N * N), (tid / N) % N, tid % N), CL);
    const int4 tab = d.gs_tab[l];
    const double bi = d.binv[l];
    const GsVals g0 = gs_load(d.yl, tab, l), g1 = gs_load(d.yl + d.cs, tab, l), g2 = gs_load(d.yl + 2 * d.cs, tab, l);
    const double v0 = bi * gs_sum3(g0, d.yl, d, tab, l, CL), v1 = bi * gs_sum3(g1, d.yl + d.cs, d, tab, l, CL), v2 = bi * gs_sum3(g2, d.yl + 2 * d.cs, d, tab, l, CL);
    d.u[l] += v0 / sc.h2; d.u[d.cs + l] += v1 / sc.h2; d.u[2 * d.cs + l] += v2 / sc.h2;
    su[tid] = v0; su[NN + tid] = v1; su[2 * NN + tid] = v2;
  }
  if (G->nit == 0) return;
  const bool pact = tid < MM;
  const long long q = e * MM + tid;
  double w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (pact) load_w2(d, q, w2);
  lds_barrier();
  const double w = opdiv3<N>(sJ12, sD12, su, sA, sB, tid, NT, w2);
  if (d.flat_proj) {                         // E dp raw; k_proj_dots subtracts the projected part and takes the dots
    if (pact) d.PED[q] = w / sc.h2;
    return;
  }
  const int np = G->nproj;
  double del = 0.0, edel = 0.0;
  if (pact) {
    edel = w / sc.h2;
    for (int kk = 0; kk < np; ++kk) edel -= G->pa[kk] * d.PEX[(size_t)kk * d.npr + q];
    del = d.PD[q];
    d.PED[q] = edel;
  }
  __shared__ double sdot[(MAXPROJ + 1) * 16];
  const int lane = tid & 63, wv = tid >> 6;
  for (int kk = 0; kk <= np; ++kk) {
    double t = 0.0;
    if (pact) t = (kk < np) ? del * d.PEX[(size_t)kk * d.npr + q] : del * edel;
    t = wave_sum63(t);
    if (lane == 63) sdot[kk * 16 + wv] = t;
  }
  lds_barrier();
  if (tid <= np) {
    double t = 0.0;
    for (int ww = 0; ww < NT / 64; ++ww) t += sdot[tid * 16 + ww];
    d.ppart[(size_t)tid * d.nblk + e] = t;
  }
}

// ---- test kernels ----
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_axhelm_test(Dev d, const double* __restrict__ u, double h1, double h2,
                                                            double* __restrict__ out) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, NT = C::NT;
  __shared__ double sD[N * N], sDt[N * N];
  __shared__ double sz[NN], st[3 * NN];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  const bool act = tid < NN;
  const long long l = e * NN + tid;
  load_basis3<N>(d, sD, sDt, nullptr, nullptr, tid, NT);
  double z = 0, g[6] = {0, 0, 0, 0, 0, 0};
  if (act) {
    z = u[l]; sz[tid] = z;
    load_g6(d, l, g);
  }
  lds_barrier();
  double au[1];
  axhelm3<N, 1>(sD, sDt, sz, st, act, tid / (N * N), (tid / N) % N, tid % N, g, au);
  if (act) out[l] = h1 * au[0] + h2 * d.bm1[l] * z;
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_opdiv_test(Dev d, const double* __restrict__ u, double* __restrict__ out) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, NT = C::NT, NM = N * M, NNM = C::NNM, NMM = C::NMM;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[3 * NN], sA[2 * NNM], sB[3 * NMM];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  load_basis3<N>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  if (tid < NN)
    for (int c = 0; c < 3; ++c) su[c * NN + tid] = u[c * d.cs + e * NN + tid];
  double w2[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (tid < MM) load_w2(d, e * MM + tid, w2);
  lds_barrier();
  const double w = opdiv3<N>(sJ12, sD12, su, sA, sB, tid, NT, w2);
  if (tid < MM) out[e * MM + tid] = w;
}

}  // namespace k3
}  // namespace nsk
