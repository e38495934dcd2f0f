// HIP kernels (gfx950) of the linearised PnPn-2 time step and the Krylov vector
// algebra.  One thread per GLL node, EPB elements per workgroup (N=8: one
// element = one 64-lane wavefront), D-hat / interpolation matrices and element
// tiles staged in LDS, E-contiguous coalesced HBM access.  dssum is a *gather*
// over a precomputed CSR of co-located nodes fused into the consuming kernel,
// which makes it deterministic (fixed summation order) and removes the separate
// gather-scatter launch.  All reductions are two-level with a fixed order.
#pragma once
#include "nsk_dev.hpp"
#include "nsk_crtrig.hpp"

namespace nsk {

__device__ int g_dbg = 0;
#ifdef NSK_STAMPS
// diagnostic build only: wall-clock stamps (100 MHz s_memrealtime) of thread 0 of every workgroup
#define NSK_STAMP(i) do { if (threadIdx.x == 0 && d.dbg) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); d.dbg[(size_t)blockIdx.x * 16 + (i)] = t_; } } while (0)
#else
#define NSK_STAMP(i) do { } while (0)
#endif     // developer ablation switches (bit mask), 0 in production

namespace k2 {
#ifdef NSK_NT_STORES
#define NSK_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define NSK_ST(p, v) (*(p) = (v))
#endif
template <int N>
struct Cfg {
  static constexpr int NN = N * N, M = N - 2, MM = M * M, ND = 3 * N / 2, NDD = ND * ND;
  static constexpr int EPB = (256 / NN) > 0 ? (256 / NN) : 1;
  static constexpr int NT = ((EPB * NN + 63) / 64) * 64;
  static constexpr int NTD = ((NDD + 63) / 64) * 64;
};
}  // namespace k2


// Workgroup barrier that orders LDS traffic only: global loads issued earlier stay in
// flight across it (hipcc's __syncthreads() drains vmcnt as well, which serialises every
// dependent-latency chain in these small kernels).
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Element tiles are private to one wavefront when an element has exactly 64 nodes (lx1 = 8): LDS operations
// of one wave execute in issue order, so the write -> read hand-off inside a tile helper needs no s_barrier.
template <int NN>
__device__ inline void tile_barrier() {
  if constexpr (NN == 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  else { lds_barrier(); }
}

// Workgroup -> element map.  Workgroups are handed to the 8 XCDs round-robin (blockIdx % 8), each XCD
// with its own L2; consecutive elements are almost always face neighbours (97 % of the r-faces of the
// reference's cylinder mesh are e, e+1), and the dssum gather reads the neighbours' face nodes.  Giving
// every XCD one contiguous run of elements makes those reads hit the L2 that streams the neighbour's
// own tile at about the same time, instead of fetching the same lines again from memory.
__device__ inline long long xcd_element(unsigned b, unsigned n) {
  const unsigned x = b & 7u, q = b >> 3, base = n >> 3, rem = n & 7u;
  return (long long)(x * base + (x < rem ? x : rem) + q);
}

// ---------------------------------------------------------------------------
// wave64 sum with DPP row shifts / row broadcasts (total lands in lane 63): ~6 dependent
// v_mov_dpp+v_add_f64 steps instead of 6 ds_bpermute round trips through the LDS crossbar.
// ---------------------------------------------------------------------------
template <int CTRL, int RM>
__device__ inline double dpp_get(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, RM, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, RM, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum63(double x) {
  x += dpp_get<0x111, 0xf>(x);      // row_shr:1
  x += dpp_get<0x112, 0xf>(x);      // row_shr:2
  x += dpp_get<0x114, 0xf>(x);      // row_shr:4
  x += dpp_get<0x118, 0xf>(x);      // row_shr:8   -> lane 15 of every row = row sum
  x += dpp_get<0x142, 0xa>(x);      // row_bcast:15 into rows 1,3
  x += dpp_get<0x143, 0xc>(x);      // row_bcast:31 into rows 2,3 -> lane 63 = wave sum
  return x;
}

// ---------------------------------------------------------------------------
// deterministic block reduction of NV values; every thread gets the result
// ---------------------------------------------------------------------------
template <int NV>
__device__ inline void block_reduce(double (&v)[NV], double* sred, int tid, int nthreads) {
  const int lane = tid & 63, w = tid >> 6, nw = (nthreads + 63) >> 6;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const double x = wave_sum63(v[q]);
    if (lane == 63) sred[q * 16 + w] = x;
  }
  lds_barrier();
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double s = 0.0;
    for (int k = 0; k < nw; ++k) s += sred[q * 16 + k];
    v[q] = s;
  }
  lds_barrier();
}

// sum NV arrays of `n` per-block partials (stride `n`), fixed order, all threads get it
template <int NV>
__device__ inline void sum_partials(const double* part, int n, double (&out)[NV], double* sred,
                                    int tid, int nthreads) {
  double v[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    double s = 0.0;
    for (int k = tid; k < n; k += nthreads) s += part[(size_t)q * n + k];
    v[q] = s;
  }
  block_reduce<NV>(v, sred, tid, nthreads);
#pragma unroll
  for (int q = 0; q < NV; ++q) out[q] = v[q];
}

// sums nq arrays of n per-block partials into sh[0..nq): waves take different q, fixed order
__device__ inline void sum_partials_multi(const double* part, int n, int nq, double* sh, int tid, int nthreads) {
  const int lane = tid & 63, w = tid >> 6, nw = nthreads >> 6;
  for (int q = w; q < nq; q += nw) {
    double s = 0.0;
    for (int k = lane; k < n; k += 64) s += part[(size_t)q * n + k];
    s = wave_sum63(s);
    if (lane == 63) sh[q] = s;
  }
  lds_barrier();
}

// the same sums with every load issued before the first wait: R rows per wavefront (rows w, w + 4, ...) of n <= 512 partials
// live in registers; rows beyond 4 R fall back to the loop.  Same summation order as sum_partials_multi (bit-identical sums).
// Rows q < nq_issue are LOADED (a bound known from the kernel arguments), rows q < nq are summed into sh.
template <int R, bool UNC = false>
struct PartialRows {
  double p[R][8];
  // UNC (the roles of k_schwarz_uc): unconditional loads from clamped addresses + a select -- `cond ? load : 0` costs a branch per
  // load; the kernels of rounds 2-5 keep the conditional form they were tuned with (k_proj_update measured 13.4 -> 18.5 us in the
  // trace when it was switched)
  __device__ inline void issue(const double* part, int n, int nq_issue, int tid, int row0 = 0) {
    const int lane = tid & 63, w = tid >> 6;
    if constexpr (!UNC) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k)
          p[r][k] = (row0 + w + 4 * r < nq_issue && lane + 64 * k < n) ? part[(size_t)(row0 + w + 4 * r) * n + lane + 64 * k] : 0.0;
      return;
    }
    if (nq_issue <= row0) {                     // (uniform) nothing to load
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) p[r][k] = 0.0;
      return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + w + 4 * r;
      const double* prow = part + (size_t)(row < nq_issue ? row : row0) * n;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int col = lane + 64 * k;
        const double v = prow[col < n ? col : 0];
        p[r][k] = (row < nq_issue && col < n) ? v : 0.0;
      }
    }
  }
  // sums rows row0 .. row0 + 4 R - 1 (those below nq) from the registers; `tail`: the rows behind them by the loop, then a barrier
  __device__ inline void reduce(const double* part, int n, int nq, double* sh, int tid, int row0 = 0, bool tail = true) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row0 + w + 4 * r < nq) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += p[r][k];
        s = wave_sum63(s);
        if (lane == 63) sh[row0 + w + 4 * r] = s;
      }
    }
    if (!tail) return;
    for (int q = row0 + w + 4 * R; q < nq; q += 4) {
      double s = 0.0;
      for (int k = lane; k < n; k += 64) s += part[(size_t)q * n + k];
      s = wave_sum63(s);
      if (lane == 63) sh[q] = s;
    }
    lds_barrier();
  }
};

// CSR fallback of the gather (valence > 4: hexahedral vertices, irregular 2-D vertices).  Same
// left-to-right sum as a plain loop, but the index and value loads of up to eight entries are
// issued together: three dependent round trips instead of two per entry.
__device__ inline double gs_csr(const double* __restrict__ f, const Dev& d, long long l) {
  const int o0 = d.gs_off[l], o1 = d.gs_off[l + 1];
  double s = 0.0;
  for (int k0 = o0; k0 < o1; k0 += 8) {
    int id[8];
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) id[q] = (k0 + q < o1) ? d.gs_idx[k0 + q] : -1;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (id[q] >= 0) ? f[id[q]] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (id[q] >= 0) s += v[q];
  }
  return s;
}

// dssum as a gather: co-located local nodes (self included, ascending => every copy sums in
// the same order) in a 4-wide table; valence > 4 falls back to the CSR lists.
__device__ inline double gs_gather(const double* __restrict__ f, const Dev& d, long long l) {
  const int4 t = d.gs_tab[l];
  if (t.y < 0 && t.x >= 0) return f[l];
  if (t.x >= 0) {
    double s = f[t.x] + f[t.y];
    if (t.z >= 0) s += f[t.z];
    if (t.w >= 0) s += f[t.w];
    return s;
  }
  return gs_csr(f, d, l);
}

// two-phase form: issue the value loads as soon as the table entry is known, sum later
struct GsVals { double a, b, c, d; };
__device__ inline GsVals gs_load(const double* __restrict__ f, const int4 t, long long l) {
  GsVals v;
  v.a = f[t.x >= 0 ? t.x : l];
  v.b = (t.y >= 0) ? f[t.y] : 0.0;
  v.c = (t.z >= 0) ? f[t.z] : 0.0;
  v.d = (t.w >= 0) ? f[t.w] : 0.0;
  return v;
}
__device__ inline double gs_sum(const GsVals& v, const double* __restrict__ f, const Dev& d, const int4 t, long long l) {
  if (t.x < 0) return gs_csr(f, d, l);
  return ((v.a + v.b) + v.c) + v.d;
}

// ---------------------------------------------------------------------------
// element-local building blocks on LDS tiles
// ---------------------------------------------------------------------------
namespace k2 {
// D^T G D on NC component tiles su[c][EPB][NN]  [UPSTREAM hmholtz.f axhelm]
template <int N, int EPB, int NC>
__device__ inline void axhelm_tiles(const double* sD, const double* sDt, const double* su,
                                    double* st1, double* st2, bool act, int el, int j, int i,
                                    double g1, double g2, double g4, double (&out)[NC]) {
  constexpr int NN = N * N;
  if (act) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double* z = su + (c * EPB + el) * NN;
      double ur = 0.0, us = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        ur += sDt[k * N + i] * z[j * N + k];
        us += sDt[k * N + j] * z[k * N + i];
      }
      st1[(c * EPB + el) * NN + j * N + i] = g1 * ur + g4 * us;
      st2[(c * EPB + el) * NN + j * N + i] = g2 * us + g4 * ur;
    }
  }
  tile_barrier<N * N>();
  if (act) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double* t1 = st1 + (c * EPB + el) * NN;
      const double* t2 = st2 + (c * EPB + el) * NN;
      double au = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) au += sD[k * N + i] * t1[j * N + k] + sD[k * N + j] * t2[k * N + i];
      out[c] = au;
    }
  }
}

// weak divergence GLL -> Gauss  [UPSTREAM navier1.f opdiv/multd]; su = [2][EPB][NN]
// sA scratch [4][EPB][N*M]; returns the value for Gauss node nd (< MM)
template <int N, int EPB>
__device__ inline double opdiv_tiles(const double* sJ12, const double* sD12, const double* su,
                                     double* sA, bool act, int el, int nd, const Dev& d, long long e) {
  constexpr int NN = N * N, M = N - 2, MM = M * M, NM = N * M;
  if (act && nd < NM) {
    const int j = nd / M, a = nd % M;
    const double* u = su + (0 * EPB + el) * NN + j * N;
    const double* v = su + (1 * EPB + el) * NN + j * N;
    double a1u = 0, a2u = 0, a1v = 0, a2v = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double dd = sD12[a * N + i], jj = sJ12[a * N + i];
      a1u += dd * u[i]; a2u += jj * u[i];
      a1v += dd * v[i]; a2v += jj * v[i];
    }
    sA[(0 * EPB + el) * NM + nd] = a1u;
    sA[(1 * EPB + el) * NM + nd] = a2u;
    sA[(2 * EPB + el) * NM + nd] = a1v;
    sA[(3 * EPB + el) * NM + nd] = a2v;
  }
  tile_barrier<N * N>();
  double div = 0.0;
  if (act && nd < MM) {
    const int b = nd / M, a = nd % M;
    double ur = 0, us = 0, vr = 0, vs = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double jj = sJ12[b * N + j], dd = sD12[b * N + j];
      ur += jj * sA[(0 * EPB + el) * NM + j * M + a];
      us += dd * sA[(1 * EPB + el) * NM + j * M + a];
      vr += jj * sA[(2 * EPB + el) * NM + j * M + a];
      vs += dd * sA[(3 * EPB + el) * NM + j * M + a];
    }
    const long long q = e * MM + nd;
    div = d.w2rx[q] * ur + d.w2sx[q] * us + d.w2ry[q] * vr + d.w2sy[q] * vs;
  }
  return div;
}

// D^T p  [UPSTREAM navier1.f opgradt/cdtp]; sP = [4][EPB][MM] holds p*w2rx, p*w2sx, p*w2ry, p*w2sy
// sB scratch [4][EPB][M*N]
template <int N, int EPB>
__device__ inline void opgradt_tiles(const double* sJ12, const double* sD12, const double* sP,
                                     double* sB, bool act, int el, int nd, double& gx, double& gy) {
  constexpr int M = N - 2, MM = M * M, NM = N * M;
  if (act && nd < NM) {
    const int b = nd / N, i = nd % N;
    double b1x = 0, b2x = 0, b1y = 0, b2y = 0;
#pragma unroll
    for (int a = 0; a < M; ++a) {
      const double dd = sD12[a * N + i], jj = sJ12[a * N + i];
      b1x += sP[(0 * EPB + el) * MM + b * M + a] * dd;
      b2x += sP[(1 * EPB + el) * MM + b * M + a] * jj;
      b1y += sP[(2 * EPB + el) * MM + b * M + a] * dd;
      b2y += sP[(3 * EPB + el) * MM + b * M + a] * jj;
    }
    sB[(0 * EPB + el) * NM + nd] = b1x;
    sB[(1 * EPB + el) * NM + nd] = b2x;
    sB[(2 * EPB + el) * NM + nd] = b1y;
    sB[(3 * EPB + el) * NM + nd] = b2y;
  }
  tile_barrier<N * N>();
  gx = 0.0; gy = 0.0;
  if (act) {
    const int j = nd / N, i = nd % N;
#pragma unroll
    for (int b = 0; b < M; ++b) {
      const double jj = sJ12[b * N + j], dd = sD12[b * N + j];
      gx += jj * sB[(0 * EPB + el) * NM + b * N + i] + dd * sB[(1 * EPB + el) * NM + b * N + i];
      gy += jj * sB[(2 * EPB + el) * NM + b * N + i] + dd * sB[(3 * EPB + el) * NM + b * N + i];
    }
  }
}

template <int N, int EPB>
__device__ inline void load_basis(const Dev& d, double* sD, double* sDt, double* sJ12, double* sD12,
                                  int tid, int nt) {
  constexpr int NN = N * N, M = N - 2;
  for (int k = tid; k < NN; k += nt) {
    if (sD) sD[k] = d.D[k];
    if (sDt) sDt[(k % N) * N + k / N] = d.D[k];
  }
  if (sJ12)
    for (int k = tid; k < M * N; k += nt) { sJ12[k] = d.J12[k]; sD12[k] = d.D12[k]; }
}

// the same in two halves: `issue` puts this thread's entries in registers (the loads go out with the kernel's other
// front-loaded loads), `commit` stores them to LDS just before the first barrier -- a plain load_basis at the top of a kernel
// waits for its own loads before the kernel's main loads are issued (one more round trip)
template <int N>
struct BasisRegs {
  static constexpr int NN = N * N, NM = N * (N - 2);
  double dv = 0.0, j12 = 0.0, d12 = 0.0;
  __device__ inline void issue(const Dev& d, int tid, bool want_d, bool want_j) {
    if (want_d && tid < NN) dv = d.D[tid];
    if (want_j && tid < NM) { j12 = d.J12[tid]; d12 = d.D12[tid]; }
  }
  __device__ inline void commit(double* sD, double* sDt, double* sJ12, double* sD12, int tid) const {
    if (sD && tid < NN) { sD[tid] = dv; if (sDt) sDt[(tid % N) * N + tid / N] = dv; }
    if (sJ12 && tid < NM) { sJ12[tid] = j12; sD12[tid] = d12; }
  }
};

// ---------------------------------------------------------------------------
// K1: forcing + dealiased convection  -> bf (mass weighted)
//   makeufp + advabp / advabp_adjoint  [UPSTREAM perturb.f], sponge term of
//   nekStab_forcing (core/utils.f:172-177).  One element per workgroup, one
//   thread per dealiasing node.
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(Cfg<N>::NTD) void k_convect(Dev d, const double* __restrict__ uin,
                                                         double* __restrict__ bf, int adjoint) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, ND = C::ND, NDD = C::NDD, NT = C::NTD;
  __shared__ double sJ[ND * N], sDd[ND * ND];
  __shared__ double su[2][NN], st[2][N * ND], sf[2][NDD], so[2][NDD], sq[2][ND * N];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  // the field first (the first barrier waits for it; loads return in issue order), then the pointwise operands of the
  // fine-mesh product and of the sponge term, which are used after two and four barriers and depend on nothing
  double u0 = 0, u1 = 0;
  if (tid < NN) { u0 = uin[e * NN + tid]; u1 = uin[d.cs + e * NN + tid]; }
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, sp0 = 0, bm0 = 0;
  if (tid < NDD) {
    const long long q = e * NDD + tid;
    if (adjoint == 2) { c0 = d.rxd[q]; c1 = d.ryd[q]; c2 = d.sxd[q]; c3 = d.syd[q]; }
    else {
      const long long qb = q + (d.bf_stride ? (long long)(*d.bstep) * d.bf_stride : 0);
      c0 = d.cUr[qb]; c1 = d.cUs[qb]; c2 = d.GUx[qb]; c3 = d.GUy[qb]; c4 = d.GVx[qb]; c5 = d.GVy[qb];
    }
  }
  if (tid < NN) { sp0 = d.spng[e * NN + tid]; bm0 = d.bm1[e * NN + tid]; }     // (multiplied where they are used: no wait here)
  for (int k = tid; k < ND * N; k += NT) sJ[k] = d.Jd[k];
  for (int k = tid; k < NDD; k += NT) sDd[k] = d.Dd[k];
  if (tid < NN) { su[0][tid] = u0; su[1][tid] = u1; }
  __syncthreads();
  // interpolate in r: st[c][j][a] = sum_i Jd[a][i] u[j][i]
  if (tid < N * ND) {
    const int j = tid / ND, a = tid % ND;
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double w = sJ[a * N + i];
      s0 += w * su[0][j * N + i];
      s1 += w * su[1][j * N + i];
    }
    st[0][tid] = s0; st[1][tid] = s1;
  }
  __syncthreads();
  const int b = tid / ND, a = tid % ND;
  const bool fact = tid < NDD;
  if (fact) {
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double w = sJ[b * N + j];
      s0 += w * st[0][j * ND + a];
      s1 += w * st[1][j * ND + a];
    }
    sf[0][tid] = s0; sf[1][tid] = s1;
  }
  __syncthreads();
  if (fact) {
    double ur = 0, us = 0, vr = 0, vs = 0;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const double dr = sDd[a * ND + k], ds = sDd[b * ND + k];
      ur += dr * sf[0][b * ND + k]; us += ds * sf[0][k * ND + a];
      vr += dr * sf[1][b * ND + k]; vs += ds * sf[1][k * ND + a];
    }
    const double uf = sf[0][tid], vf = sf[1][tid];
    double ox, oy;
    if (adjoint == 2) {      // full equations: (u.grad) u   [UPSTREAM advab], newton_krylov's nonlinear map
      const double cr = c0 * uf + c1 * vf, cs = c2 * uf + c3 * vf;           // rxd, ryd, sxd, syd
      ox = cr * ur + cs * us;
      oy = cr * vr + cs * vs;
    } else {
      // base-flow constants (steady, or slot `*bstep` of the stored periodic orbit: Floquet, core/matvec.f:200-236):
      // c0..c5 = cUr, cUs, GUx, GUy, GVx, GVy
      const double conv_u = c0 * ur + c1 * us, conv_v = c0 * vr + c1 * vs;   // (U.grad) u'
      if (!adjoint) {          // + (u'.grad) U
        ox = conv_u + uf * c2 + vf * c3;
        oy = conv_v + uf * c4 + vf * c5;
      } else {                 // (grad U)^T u' - (U.grad) u'
        ox = uf * c2 + vf * c4 - conv_u;
        oy = uf * c3 + vf * c5 - conv_v;
      }
    }
    so[0][tid] = ox; so[1][tid] = oy;
  }
  __syncthreads();
  // project back: sq[c][b][i] = sum_a Jd[a][i] so[b][a]
  if (tid < ND * N) {
    const int bb = tid / N, i = tid % N;
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int aa = 0; aa < ND; ++aa) {
      const double w = sJ[aa * N + i];
      s0 += w * so[0][bb * ND + aa];
      s1 += w * so[1][bb * ND + aa];
    }
    sq[0][tid] = s0; sq[1][tid] = s1;
  }
  __syncthreads();
  if (tid < NN) {
    const int j = tid / N, i = tid % N;
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int bb = 0; bb < ND; ++bb) {
      const double w = sJ[bb * N + j];
      s0 += w * sq[0][bb * N + i];
      s1 += w * sq[1][bb * N + i];
    }
    const long long l = e * NN + tid;
    const double sb = sp0 * bm0;
    if (adjoint == 2) {      // DNS sponge: spng_fun (u_ref - u) spng_str   (core/utils.f:165-170)
      const double k = sb * d.nl_spng_str;
      bf[l] = ((k != 0.0) ? k * (d.spng_vr[l] - su[0][tid]) : 0.0) - s0;
      bf[d.cs + l] = ((k != 0.0) ? k * (d.spng_vr[d.cs + l] - su[1][tid]) : 0.0) - s1;
    } else {
      bf[l] = -(sb * su[0][tid] + s0);
      bf[d.cs + l] = -(sb * su[1][tid] + s1);
    }
  }
}

// New linearisation point: the base-flow-dependent dealiasing-mesh constants from a state vector
// (newton_krylov passes the current Newton iterate as base flow, core/newton_krylov.f:371-372)
template <int N>
__global__ __launch_bounds__(Cfg<N>::NTD) void k_baseflow(Dev d, const double* __restrict__ q, double* cUr, double* cUs,
                                                          double* GUx, double* GUy, double* GVx, double* GVy) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, ND = C::ND, NDD = C::NDD, NT = C::NTD;
  __shared__ double sJ[ND * N], sDd[ND * ND];
  __shared__ double su[2][NN], st[2][N * ND], sf[2][NDD];
  const int tid = threadIdx.x;
  const long long e = blockIdx.x;
  for (int k = tid; k < ND * N; k += NT) sJ[k] = d.Jd[k];
  for (int k = tid; k < NDD; k += NT) sDd[k] = d.Dd[k];
  if (tid < NN) { su[0][tid] = q[e * NN + tid]; su[1][tid] = q[d.nloc + e * NN + tid]; }
  __syncthreads();
  if (tid < N * ND) {
    const int j = tid / ND, a = tid % ND;
    double s0 = 0, s1 = 0;
    for (int i = 0; i < N; ++i) { const double w = sJ[a * N + i]; s0 += w * su[0][j * N + i]; s1 += w * su[1][j * N + i]; }
    st[0][tid] = s0; st[1][tid] = s1;
  }
  __syncthreads();
  const int b = tid / ND, a = tid % ND;
  const bool fact = tid < NDD;
  if (fact) {
    double s0 = 0, s1 = 0;
    for (int j = 0; j < N; ++j) { const double w = sJ[b * N + j]; s0 += w * st[0][j * ND + a]; s1 += w * st[1][j * ND + a]; }
    sf[0][tid] = s0; sf[1][tid] = s1;
  }
  __syncthreads();
  if (fact) {
    double Ur = 0, Us = 0, Vr = 0, Vs = 0;
    for (int k = 0; k < ND; ++k) {
      const double dr = sDd[a * ND + k], ds = sDd[b * ND + k];
      Ur += dr * sf[0][b * ND + k]; Us += ds * sf[0][k * ND + a];
      Vr += dr * sf[1][b * ND + k]; Vs += ds * sf[1][k * ND + a];
    }
    const long long qq = e * NDD + tid;
    const double rx = d.rxd[qq], ry = d.ryd[qq], sx = d.sxd[qq], sy = d.syd[qq], Uf = sf[0][tid], Vf = sf[1][tid];
    cUr[qq] = rx * Uf + ry * Vf; cUs[qq] = sx * Uf + sy * Vf;
    GUx[qq] = rx * Ur + sx * Us; GUy[qq] = ry * Ur + sy * Us;
    GVx[qq] = rx * Vr + sx * Vs; GVy[qq] = ry * Vr + sy * Vs;
  }
}

// ---------------------------------------------------------------------------
// K2: makextp + makebdfp + lagfieldp + extrapprp + cresvipp  [UPSTREAM perturb.f]
//   r_loc = EXT(bf) + BDF lags + D^T p* - H u^n   (unassembled)
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_rhs(Dev d, StepCoef sc) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  __shared__ double sD[NN], sDt[NN], sJ12[NM], sD12[NM];
  __shared__ double su[2 * EPB * NN], st1[2 * EPB * NN], st2[2 * EPB * NN];
  __shared__ double sP[4 * EPB * MM], sB[4 * EPB * NM];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  const int j = nd / N, i = nd % N;
  const long long l = e * NN + nd, nl = d.cs;
  static_assert(NT >= NN, "one basis entry per thread");
  BasisRegs<N> br;
  br.issue(d, tid, true, true);
  if (d.bf_stride && sc.adjoint != 2 && blockIdx.x == 0 && tid == 0) *d.bstep += 1;     // next step reads the next orbit slot
  if (d.stepctr && blockIdx.x == 0 && tid == 0) *d.stepctr += 1;                          // per-step iteration record (rec_step_iters)
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
      } else {              // degenerate direction: drop the whole space and start again
        G->pcnt = 0; G->nproj = 0;
      }
    }
    if (d.proj_reset && sc.cls == 0) { G->pcnt = 0; G->nproj = 0; }      // first step of a map: the space of the last map is stale
  }
  double u[2] = {0, 0}, bfv[2] = {0, 0}, bm = 0, g1 = 0, g2 = 0, g4 = 0;
  if (act) {
    // every load first, then the arithmetic and the lag shifts: a store between two loads pins their order (the compiler must
    // assume the arrays alias), and each pinned load is one more round trip
    const bool pl = nd < MM;
    const long long q = e * MM + nd;
    double un[2], dl1[2], dl2[2], dl3[2], bn[2], e1[2], e2[2], l1[2], l2[2];
    double pn = 0, plg = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    bm = d.bm1[l]; g1 = d.g1[l]; g2 = d.g2[l]; g4 = d.g4[l];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const long long lc = c * nl + l;
      un[c] = d.u[lc];
      dl1[c] = d.dulag[lc]; dl2[c] = d.dulag[2 * nl + lc]; dl3[c] = d.dulag[4 * nl + lc];
      bn[c] = d.bf[lc];
      e1[c] = d.exlag[lc]; e2[c] = d.exlag[2 * nl + lc];
      l1[c] = d.ulag[lc]; l2[c] = d.ulag[2 * nl + lc];
    }
    if (pl) { pn = d.p[q]; plg = d.plag[q]; m0 = d.w2rx[q]; m1 = d.w2sx[q]; m2 = d.w2ry[q]; m3 = d.w2sy[q]; }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const long long lc = c * nl + l;
      u[c] = un[c] + sc.xg[0] * dl1[c] + sc.xg[1] * dl2[c] + sc.xg[2] * dl3[c];   // u^n + du0 (extrapolated guess)
      su[(c * EPB + el) * NN + nd] = u[c];
      double b = sc.ab[0] * bn[c] + sc.ab[1] * e1[c] + sc.ab[2] * e2[c];      // makextp
      d.exlag[2 * nl + lc] = e1[c];
      d.exlag[lc] = bn[c];
      b += bm * (sc.bd[1] * un[c] + sc.bd[2] * l1[c] + sc.bd[3] * l2[c]) * sc.invdt;   // makebdfp
      d.ulag[2 * nl + lc] = l1[c];                                   // lagfieldp
      d.ulag[lc] = un[c];
      bfv[c] = b;
    }
    if (pl) {                                                        // extrapprp
      const double pe = (sc.pxt == 0.0) ? pn : ((sc.pxt == 1.0) ? 2.0 * pn - plg : pn + sc.pxt * (pn - plg));
      d.plag[q] = pn;
      d.pext[q] = pe;
      sP[(0 * EPB + el) * MM + nd] = pe * m0;
      sP[(1 * EPB + el) * MM + nd] = pe * m1;
      sP[(2 * EPB + el) * MM + nd] = pe * m2;
      sP[(3 * EPB + el) * MM + nd] = pe * m3;
    }
  }
  br.commit(sD, sDt, sJ12, sD12, tid);
  __syncthreads();
  double gx, gy;
  opgradt_tiles<N, EPB>(sJ12, sD12, sP, sB, act, el, nd, gx, gy);
  double au[2];
  axhelm_tiles<N, EPB, 2>(sD, sDt, su, st1, st2, act, el, j, i, g1, g2, g4, au);
  if (act) {
    const double bx = bfv[0] + gx, by = bfv[1] + gy;                 // rhs of H u* = b
    d.bloc[l] = bx;
    d.bloc[nl + l] = by;
    d.rloc[l] = bx - (d.nu * au[0] + sc.h2 * bm * u[0]);
    d.rloc[nl + l] = by - (d.nu * au[1] + sc.h2 * bm * u[1]);
  }
}

// ---------------------------------------------------------------------------
// K3: one Jacobi-preconditioned CG iteration of H du = dssum(r), both velocity
// components, in the single-reduction (Chronopoulos-Gear) form so that ONE
// kernel = gather-dssum of the previous A z + vector updates + next local A z +
// the three dot products.  (z, A z) is summed from unassembled element
// contributions, which is exact for continuous z.   [UPSTREAM hmholtz.f cggo]
//   hscal[par][c*4 + {0:gamma,1:alpha,2:done,3:res}]
// ---------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void helm_body(const Dev& d, const StepCoef& sc, int it, const double* rhs, const unsigned bx_, const unsigned gx_) {
  // (shards with halo / interior overlap launch the boundary workgroups first, the rest with an offset; XCD-contiguous runs of
  //  element blocks: the neighbours' edge values are L2 hits where the grid is larger than the caches, config 3)
  const int bid = d.boff + (int)xcd_element(bx_, gx_);
  using C = Cfg<N>;
  constexpr int NN = C::NN, EPB = C::EPB, NT = C::NT;
  __shared__ double sD[NN], sDt[NN];
  __shared__ double sz[2 * EPB * NN], st1[2 * EPB * NN], st2[2 * EPB * NN];
  __shared__ double sred[8 * 16];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)bid * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  const int j = nd / N, i = nd % N;
  const long long l = e * NN + nd, nl = d.cs;
  const int par = it & 1, ppar = par ^ 1;
  NSK_STAMP(0);
  // the previous iteration's scalars and -- with them, so that the neighbour gathers below can go out together with the
  // rest of phase A instead of one round trip later -- this node's gather-table entry (16 B per thread: what a launch that
  // finds the solve finished pays on top of the flag)
  double o[8] = {0, 0, 0, 0, 0, 0, 0, 0}, refn[2] = {0, 0};
  int4 tab = make_int4(0, -1, -1, -1);
  if (it > 1) {
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = d.hscal[ppar * 8 + q];
  }
  if (act) tab = d.gs_tab[l];
  if (it > 2 && o[2] != 0.0 && o[6] != 0.0) {   // finished earlier: cheapest exit
    if (bid == 0 && tid < 8) d.hscal[par * 8 + tid] = o[tid & 7];
    return;
  }
  // ---- phase A: issue every independent global load before anything waits
  if (it > 1) { refn[0] = d.hscal[16]; refn[1] = d.hscal[17]; }
  double bm = 0, g1 = 0, g2 = 0, g4 = 0, mk = 0, mi = 0, di = 0;
  double rold[2] = {0, 0}, pold[2] = {0, 0}, sold[2] = {0, 0}, xold[2] = {0, 0};
  if (act) {
    bm = d.bm1[l]; g1 = d.g1[l]; g2 = d.g2[l]; g4 = d.g4[l]; mk = d.mask[l]; mi = d.minv[l];
    di = d.dinv[(size_t)(sc.k - 1) * d.nloc + l];
    if (it > 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const long long lc = c * nl + l;
        rold[c] = d.hr[lc]; pold[c] = d.hp[lc]; sold[c] = d.hs[lc]; xold[c] = d.hx[lc];
      }
    }
  }
  // the previous kernel's dot-product partials: two per row and thread up to 2 NT workgroups, all in flight at once (a
  // `for` over them would wait for each round of loads and hold back every load behind it)
  double ps[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ps1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (it > 0) {
    if (d.nranks > 1 || d.use_tot) {    // totals: all-reduced over ranks, or summed once by k_tot2 (many workgroups)
#pragma unroll
      for (int q = 0; q < 8; ++q) ps[q] = (tid == 0) ? d.htot[ppar * 8 + q] : 0.0;
    } else {
      const double* part = d.hpart + (size_t)ppar * 8 * d.nblk;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        ps[q] = (tid < d.nblk) ? part[(size_t)q * d.nblk + tid] : 0.0;
        ps1[q] = (tid + NT < d.nblk) ? part[(size_t)q * d.nblk + tid + NT] : 0.0;
      }
    }
  }
  const double dreg = (tid < NN) ? d.D[tid] : 0.0;
  if (it > 1 && o[2] != 0.0 && o[6] != 0.0) {        // both components finished earlier
    if (bid == 0 && tid < 8) d.hscal[par * 8 + tid] = o[tid];
    return;
  }
  NSK_STAMP(1);
  // ---- phase B: gathers of the neighbours' unassembled values (need the table entry)
  GsVals gv[2], gb[2];
  if (act) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (it == 0) { gv[c] = gs_load(rhs + c * nl, tab, l); gb[c] = gs_load(d.bloc + c * nl, tab, l); }
      else gv[c] = gs_load(d.hwl + ((size_t)ppar * 2 + c) * nl, tab, l);
    }
  }
  NSK_STAMP(2);
  // ---- phase C: scalars of this iteration from the previous kernel's partials
  double alpha[2] = {0, 0}, beta[2] = {0, 0};
  bool done[2] = {false, false};
  if (it > 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) ps[q] += ps1[q];
    if (!(d.nranks > 1 || d.use_tot)) {
      const double* part = d.hpart + (size_t)ppar * 8 * d.nblk;
      for (int k = tid + 2 * NT; k < d.nblk; k += NT) {     // more than 2 NT workgroups without the totals path: not a built configuration
#pragma unroll
        for (int q = 0; q < 8; ++q) ps[q] += part[(size_t)q * d.nblk + k];
      }
    }
    block_reduce<8>(ps, sred, tid, NT);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const double g = ps[c * 3 + 0], del = ps[c * 3 + 1], rr = ps[c * 3 + 2];
      const double res = sqrt(rr / d.vol);
      const double ref = (it == 1) ? sqrt(ps[6 + c] / d.vol) : refn[c];   // ||b||: H u* = b
      const double tol = d.tol_relative ? d.tol_helm * ref : d.tol_helm;
      const bool was = (it > 1 && o[c * 4 + 2] != 0.0);
      done[c] = was || (res <= tol) || !(g > 0.0);
      if (!done[c]) {
        if (it == 1) { beta[c] = 0.0; alpha[c] = g / del; }
        else { beta[c] = g / o[c * 4 + 0]; alpha[c] = g / (del - beta[c] * g / o[c * 4 + 1]); }
      }
      if (bid == 0 && tid == 0) {
        double* cur = d.hscal + par * 8 + c * 4;
        cur[0] = g; cur[1] = alpha[c]; cur[2] = done[c] ? 1.0 : 0.0;
        cur[3] = was ? o[c * 4 + 3] : res;
        if (it == 1) d.hscal[16 + c] = ref;
        if (done[c] && !was) {
          if (c == 0) atomicAdd((unsigned long long*)&d.stats->helm_iters, (unsigned long long)(it - 1));
          atomicMax((unsigned long long*)&d.stats->max_helm, (unsigned long long)(it - 1));
          atomicMax((unsigned long long*)&d.stats->max_helm_k[sc.cls], (unsigned long long)(it - 1)); rec_step_iters(d, 0, it - 1);
        }
      }
    }
    if (done[0] && done[1]) return;
  }
  NSK_STAMP(3);
  // ---- phase D: vector updates, next local A z, dot-product partials
  if (tid < NN) { sD[tid] = dreg; sDt[(tid % N) * N + tid / N] = dreg; }
  double r[2] = {0, 0}, z[2] = {0, 0}, bb[2] = {0, 0};
  if (act) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const long long lc = c * nl + l;
      if (it == 0) {
        r[c] = mk * gs_sum(gv[c], rhs + c * nl, d, tab, l);
        bb[c] = mk * gs_sum(gb[c], d.bloc + c * nl, d, tab, l);
        d.hx[lc] = 0.0; d.hp[lc] = 0.0; d.hs[lc] = 0.0; d.hr[lc] = r[c];
      } else if (!done[c]) {
        const double w = mk * gs_sum(gv[c], d.hwl + ((size_t)ppar * 2 + c) * nl, d, tab, l);
        const double pn = di * rold[c] + beta[c] * pold[c];
        const double sn = w + beta[c] * sold[c];
        NSK_ST(d.hp + lc, pn); NSK_ST(d.hs + lc, sn);
        NSK_ST(d.hx + lc, xold[c] + alpha[c] * pn);
        r[c] = rold[c] - alpha[c] * sn;
        NSK_ST(d.hr + lc, r[c]);
      } else {
        r[c] = rold[c];
      }
      z[c] = di * r[c];
      sz[(c * EPB + el) * NN + nd] = z[c];
    }
  }
  NSK_STAMP(4);
  lds_barrier();
  double au[2];
  axhelm_tiles<N, EPB, 2>(sD, sDt, sz, st1, st2, act, el, j, i, g1, g2, g4, au);
  NSK_STAMP(5);
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (act) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const double wl = d.nu * au[c] + sc.h2 * bm * z[c];
      NSK_ST(d.hwl + ((size_t)par * 2 + c) * nl + l, wl);
      v[c * 3 + 0] = r[c] * z[c] * mi;
      v[c * 3 + 1] = z[c] * wl;
      v[c * 3 + 2] = r[c] * r[c] * mi;
      v[6 + c] = bb[c] * bb[c] * mi;
    }
  }
  block_reduce<8>(v, sred, tid, NT);
  NSK_STAMP(6);
  if (tid < 8) d.hpart[((size_t)par * 8 + tid) * d.nblk + bid] = v[tid];
  NSK_STAMP(7);
}
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_helm(Dev d, StepCoef sc, int it, const double* rhs) {
  helm_body<N>(d, sc, it, rhs, blockIdx.x, gridDim.x);
}


// ---------------------------------------------------------------------------
// K4: u* = u + du ;  g = -D u*  -> V[0] (unnormalised), |g|^2 partials
//     also verifies that the Helmholtz solve converged.   [UPSTREAM incomprp]
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_pres_rhs(Dev d, StepCoef sc, int helm_par, int check_helm) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[2 * EPB * NN], sA[4 * EPB * NM];
  __shared__ double sred[8 * 16];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  const long long l = e * NN + nd, nl = d.cs;
  // stored pressure solutions at this thread's Gauss node, for the projection dots at the end: independent of everything
  // else in the launch, so they are issued first (the first PXPRE vectors; slots >= nproj are never used)
  constexpr int PXPRE = MAXPROJ;
  double pxr[PXPRE];
  {
    const bool pl = act && nd < MM && d.nproj_max > 0;
    const int npre = d.nproj_max < PXPRE ? d.nproj_max : PXPRE;
#pragma unroll
    for (int k = 0; k < PXPRE; ++k) pxr[k] = (pl && k < npre) ? d.PX[(size_t)k * d.npr + e * MM + nd] : 0.0;
  }
  BasisRegs<N> br;
  br.issue(d, tid, false, true);
  // velocity increments and their lags: loaded before the block-0 bookkeeping and before any store (see k_rhs)
  double un_[2] = {0, 0}, l1_[2] = {0, 0}, l2_[2] = {0, 0}, l3_[2] = {0, 0}, hx_[2] = {0, 0};
  if (act) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const long long lc = c * nl + l;
      l1_[c] = d.dulag[lc]; l2_[c] = d.dulag[2 * nl + lc]; l3_[c] = d.dulag[4 * nl + lc];
      hx_[c] = d.hx[lc]; un_[c] = d.u[lc];
    }
  }
  if (check_helm && blockIdx.x == 0) {       // last partials -> final residual of the velocity solve
    double s[8];
    if (d.nranks > 1 || d.use_tot) {
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] = d.htot[helm_par * 8 + q];
    } else {
      sum_partials<8>(d.hpart + (size_t)helm_par * 8 * d.nblk, d.nblk, s, sred, tid, NT);
    }
    if (tid == 0) {
      double worst = 0.0; int bad = 0;
      for (int c = 0; c < 2; ++c) {
        const double res = sqrt(s[c * 3 + 2] / d.vol);
        const double tol = d.tol_relative ? d.tol_helm * d.hscal[16 + c] : d.tol_helm;
        const bool was = d.hscal[helm_par * 8 + c * 4 + 2] != 0.0;
        const double rr = was ? d.hscal[helm_par * 8 + c * 4 + 3] : res;
        worst = fmax(worst, rr);
        if (!was && !(res <= tol)) bad = 1;
        if (!was) { if (c == 0) atomicAdd((unsigned long long*)&d.stats->helm_iters, (unsigned long long)check_helm); atomicMax((unsigned long long*)&d.stats->max_helm, (unsigned long long)check_helm); atomicMax((unsigned long long*)&d.stats->max_helm_k[sc.cls], (unsigned long long)check_helm); rec_step_iters(d, 0, check_helm); }
      }
      d.stats->last_helm_res = worst;
      if (bad) atomicAdd((unsigned long long*)&d.stats->unconverged, 1ull);
    }
  }
  if (act) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const long long lc = c * nl + l;
      const double l1 = l1_[c], l2 = l2_[c], l3 = l3_[c];
      const double du = sc.xg[0] * l1 + sc.xg[1] * l2 + sc.xg[2] * l3 + hx_[c];     // guess + CG correction
      d.dulag[4 * nl + lc] = l2;
      d.dulag[2 * nl + lc] = l1;
      d.dulag[lc] = du;
      const double us = un_[c] + du;
      d.u[lc] = us;
      su[(c * EPB + el) * NN + nd] = us;
    }
  }
  br.commit(nullptr, nullptr, sJ12, sD12, tid);
  __syncthreads();
  const double div = opdiv_tiles<N, EPB>(sJ12, sD12, su, sA, act, el, nd, d, e);
  double v[1] = {0.0};
  double g = 0.0;
  const bool pact = act && nd < MM;
  if (pact) {
    g = -div;
    d.V[e * MM + nd] = g;
    v[0] = g * g;
  }
  block_reduce<1>(v, sred, tid, NT);
  if (tid == 0) d.gpart[blockIdx.x] = v[0];
  if (!d.has_outflow) {                        // sum(g) for `ortho` (pressure null space)
    double t[1] = {g};
    block_reduce<1>(t, sred, tid, NT);
    if (tid == 0) d.gpart[(size_t)d.nblk + blockIdx.x] = t[0];
  }
  if (d.nproj_max > 0) {                       // (x_i, g) for the stored solutions
    if (tid == 0) d.ppart[(size_t)MAXPROJ * d.nblk + blockIdx.x] = v[0];
    const int np = d.gsc->nproj;
    __shared__ double sdot[MAXPROJ * 4];
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int k = 0; k < PXPRE; ++k) {
      if (k < np) {
        double t = pact ? g * pxr[k] : 0.0;
        t = wave_sum63(t);
        if (lane == 63) sdot[k * 4 + wv] = t;
      }
    }
#pragma unroll 4
    for (int k = PXPRE; k < np; ++k) {
      double t = pact ? g * d.PX[(size_t)k * d.npr + e * MM + nd] : 0.0;
      t = wave_sum63(t);
      if (lane == 63) sdot[k * 4 + wv] = t;
    }
    lds_barrier();
    if (tid < np) {
      double t = 0.0;
      for (int ww = 0; ww < NT / 64; ++ww) t += sdot[tid * 4 + ww];
      d.ppart[(size_t)tid * d.nblk + blockIdx.x] = t;
    }
  }
}

}  // namespace k2

// all-Dirichlet / periodic velocity => E has the constant null space: remove the mean of the
// right-hand side  [UPSTREAM navier1.f ortho]; recomputes the |g|^2 partials
__global__ __launch_bounds__(256) void k_ortho(Dev d) {
  __shared__ double sred[16];
  const int tid = threadIdx.x;
  double sm[1];
  if (d.nranks > 1 || d.use_tot) sm[0] = d.gtot[1];
  else sum_partials<1>(d.gpart + d.nblk, d.nblk, sm, sred, tid, 256);
  const double mean = sm[0] / (double)d.npr_glob;
  double v[1] = {0.0};
  for (long long q = (long long)blockIdx.x * 256 + tid; q < d.npr; q += (long long)gridDim.x * 256) {
    const double g = d.V[q] - mean;
    d.V[q] = g;
    v[0] += g * g;
  }
  block_reduce<1>(v, sred, tid, 256);
  if (tid == 0) d.gpart[blockIdx.x] = v[0];
}

// g' = g - sum_i a_i E x_i,  a_i = (x_i,g)/n_i   [UPSTREAM navier4.f setrhsp / projh]
__global__ __launch_bounds__(256) void k_proj_apply(Dev d) {
  __shared__ double sh[MAXPROJ + 1];
  __shared__ double sred[16];
  const int tid = threadIdx.x;
  GmresScal* G = d.gsc;
  const int np = G->nproj;
  // every load of this launch is independent of every other: issue them all before the first wait (the space holds at most
  // nproj_max <= MAXPROJ vectors; the first PRE of them go to registers for this thread's entry -- one entry per thread on
  // every mesh this runs on; slots >= nproj hold finite stale data and get a zero coefficient)
  constexpr int PRE = MAXPROJ;
  const long long q0 = (long long)blockIdx.x * 256 + tid, stride = (long long)gridDim.x * 256;
  const bool has = q0 < d.npr;
  const int npre = d.nproj_max < PRE ? d.nproj_max : PRE;
  const double g0 = has ? d.V[q0] : 0.0;
  double pe[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) pe[k] = (has && k < npre) ? d.PEX[(size_t)k * d.npr + q0] : 0.0;
  const double pnv = (tid < MAXPROJ) ? G->pn[tid] : 1.0;
  if (d.nranks > 1 || d.use_tot) {      // totals: all-reduced over ranks, or summed once by k_tot2
    if (tid < np) sh[tid] = d.ptot[tid];
    if (blockIdx.x == 0 && tid == 0) G->gnorm0 = sqrt(d.ptot[MAXPROJ]);
    __syncthreads();
  } else {
    PartialRows<4> pr;                  // 16 rows per pass: a second pass for spaces of more than 16 vectors
    pr.issue(d.ppart, d.nblk, d.nblk <= 512 ? d.nproj_max : 0, tid);
    if (d.nblk <= 512) {
      pr.reduce(d.ppart, d.nblk, np, sh, tid, 0, np <= 16);
      if (np > 16) { pr.issue(d.ppart, d.nblk, np, tid, 16); pr.reduce(d.ppart, d.nblk, np, sh, tid, 16); }
    } else sum_partials_multi(d.ppart, d.nblk, np, sh, tid, 256);
    if (blockIdx.x == 0) {
      double gg[1];
      sum_partials<1>(d.ppart + (size_t)MAXPROJ * d.nblk, d.nblk, gg, sred, tid, 256);
      if (tid == 0) G->gnorm0 = sqrt(gg[0]);
    }
  }
  if (tid < np) sh[tid] = sh[tid] / pnv;
  __syncthreads();
  if (blockIdx.x == 0 && tid < np) G->pa[tid] = sh[tid];
  double v[1] = {0.0};
  if (has) {
    double g = g0;
#pragma unroll
    for (int k = 0; k < PRE; ++k) if (k < np) g -= sh[k] * pe[k];
    for (int k = PRE; k < np; ++k) g -= sh[k] * d.PEX[(size_t)k * d.npr + q0];
    d.V[q0] = g;
    v[0] += g * g;
  }
  for (long long q = q0 + stride; q < d.npr; q += stride) {
    double g = d.V[q];
    for (int k = 0; k < np; ++k) g -= sh[k] * d.PEX[(size_t)k * d.npr + q];
    d.V[q] = g;
    v[0] += g * g;
  }
  block_reduce<1>(v, sred, tid, 256);
  if (tid == 0) d.gpart[blockIdx.x] = v[0];
}

// ---------------------------------------------------------------------------
// pressure GMRES pieces.  Right-preconditioned, M^-1 = restricted additive
// Schwarz (dense patch inverses, fp32 storage) + vertex coarse solve (dense
// inverse).  [UPSTREAM navier1.f uzawa_gmres, hsmg.f -- the preconditioner is
// this build's own; only the converged solution is part of the discretisation]
// ---------------------------------------------------------------------------

// One GMRES bookkeeping kernel per iteration: every workgroup re-sums the dot-product
// partials (fixed order => identical values everywhere), workgroup 0 additionally advances
// the Givens / least-squares state and the convergence flag, and all workgroups form
//   v_{j+1} = (w - sum_i h_i v_i) / h_{j+1,j}      (j = -1: v_0 = g'/|g'|)
// plus the element-corner restriction ec[e][c] = sum_k hat_c(k) v(e,k) for the coarse solve.
namespace k2 {
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_gmres_update(Dev d, int j, double scale, int min_iter, int ord) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, MM = C::MM, EPB = C::EPB, NT = C::NT;
  __shared__ double sv[EPB * MM];
  __shared__ double sh[MAXMR + 2];
  __shared__ double shat[4 * MM];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  GmresScal* G = d.gsc;
  const bool restart = (j == -2);               // open the next GMRES cycle on the residual k_gmres_restart left in V[0]
  if (restart) j = -1;
  if ((j >= 0 || restart) && G->done) return;
  double hatv[(4 * MM + NT - 1) / NT];
#pragma unroll
  for (int r = 0; r < (4 * MM + NT - 1) / NT; ++r) hatv[r] = (tid + r * NT < 4 * MM) ? d.hat[tid + r * NT] : 0.0;
  double wnew = 0.0;
  if (act && nd < MM) wnew = d.wraw ? d.wraw[e * MM + nd] : d.V[(size_t)(j + 1) * d.ps + e * MM + nd];
  const int nv = (j < 0) ? 1 : j + 2;
  const bool two = d.gs2 && j >= 0;             // k_gmres_reorth ran for this column: V[j+1] already holds w - sum h_i v_i
  __shared__ double sc2[MAXMR + 2];
  if (d.nranks > 1 || d.use_tot) {
    if (tid < nv) sh[tid] = d.gtot[tid];
    if (two && tid < nv) sc2[tid] = d.gtot2[tid];
    lds_barrier();
  } else {
    sum_partials_multi(d.gpart, d.nblk, nv, sh, tid, NT);
    if (two) sum_partials_multi(d.gpart2, d.nblk, nv, sc2, tid, NT);
  }
  double hn;
  if (j < 0) hn = sqrt(sh[0]);
  else {
    const double* pc = two ? sc2 : sh;          // coefficients of the pass that left the current w
    double s2 = 0.0;
    for (int q = 0; q <= j; ++q) s2 += pc[q] * pc[q];
    const double hn2 = pc[j + 1] - s2;
    hn = sqrt(hn2 > 0.0 ? hn2 : 0.0);
  }
  const double hinv = (hn > 0.0) ? 1.0 / hn : 0.0;
  __shared__ double scs[MAXMR], ssn[MAXMR], scol[MAXMR + 2];
  __shared__ double sgj;
  if (blockIdx.x == 0 && j >= 0) {        // Givens history -> LDS with parallel loads
    if (tid < j) { scs[tid] = G->cs[tid]; ssn[tid] = G->sn[tid]; }
    if (tid == 0) sgj = G->g[j];
    lds_barrier();
  }
  if (blockIdx.x == 0 && tid == 0) {
    if (restart) {
      G->nit_prev += G->nit;
      G->beta0 = hn; G->g[0] = hn; G->gpre[0] = hn; G->nit = 0; G->resid = hn * scale;
      if (!(hn > 0.0)) G->done = 1;
    } else if (j < 0) {
      G->beta0 = hn; G->g[0] = hn; G->gpre[0] = hn; G->nit = 0; G->nit_prev = 0; G->resid = hn * scale;
      if (d.nproj_max <= 0) G->gnorm0 = hn;
      const double tol0 = d.tol_relative ? fmax(d.tol_pres * G->gnorm0 * scale, d.tol_pres_floor) : d.tol_pres;
      const int dn = (!(hn > 0.0) || (min_iter <= 0 && hn * scale <= tol0)) ? 1 : 0;
      if (dn) d.stats->last_pres_res = hn * scale;
      if (!(hn * 0.0 == 0.0)) d.stats->nonfinite += 1;      // |g| is NaN or Inf: nothing below can repair it (map_finish returns NSK_ENAN)
      G->done = dn;
    } else {
      double* col = scol;
      for (int q = 0; q <= j; ++q) col[q] = two ? sh[q] + sc2[q] : sh[q];
      col[j + 1] = hn;
      for (int q = 0; q < j; ++q) {
        const double t = scs[q] * col[q] + ssn[q] * col[q + 1];
        col[q + 1] = -ssn[q] * col[q] + scs[q] * col[q + 1];
        col[q] = t;
      }
      const double rho = sqrt(col[j] * col[j] + col[j + 1] * col[j + 1]);
      const double cj = (rho > 0.0) ? col[j] / rho : 1.0, sj = (rho > 0.0) ? col[j + 1] / rho : 0.0;
      G->cs[j] = cj; G->sn[j] = sj;
      col[j] = rho;
      for (int q = 0; q <= j; ++q) G->R[j * MAXMR + q] = col[q];
      const double gj = sgj;
      G->g[j] = cj * gj;
      G->g[j + 1] = -sj * gj; G->gpre[j + 1] = -sj * gj;
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
#pragma unroll
  for (int r = 0; r < (4 * MM + NT - 1) / NT; ++r) if (tid + r * NT < 4 * MM) shat[tid + r * NT] = hatv[r];
  if (act && nd < MM) {
    const long long q = e * MM + nd;
    double w = wnew;
    const double* pc = two ? sc2 : sh;
#pragma unroll 8
    for (int k = 0; k <= j; ++k) w -= pc[k] * d.V[(size_t)k * d.ps + q];
    w *= hinv;
    d.V[(size_t)(j + 1) * d.ps + q] = w;
    sv[el * MM + nd] = w;
  }
  lds_barrier();
  if (act && nd < 4) {
    double s = 0.0;
#pragma unroll 6
    for (int k = 0; k < MM; ++k) s += shat[nd * MM + k] * sv[el * MM + k];
    d.ec[e * 4 + nd] = s;
    if (d.ecv) d.ecv[d.ecslot[e * 4 + nd]] = s;
  }
}

}  // namespace k2

// GMRES restart (Nek5000 restarts at lgmres = 30; here the cycle is MAXMR = 48): after m iterations without convergence
//   xacc (+)= Z y,   r = V_{m+1} s  with  s = Omega_0^T ... Omega_{m-1}^T (g_m e_{m+1})   (the residual in the Krylov basis),
// r overwrites V[0] and its |r|^2 partials go to gpart row 0; k_gmres_update(j = -2) then normalises it and opens the next
// cycle.  Same launch shape as k_proj_apply (d.nblk workgroups, grid-stride), works for both kernel sets.
__global__ __launch_bounds__(256) void k_gmres_restart(Dev d, int m) {
  __shared__ double sy[MAXMR], sg[MAXMR + 1], ss[MAXMR + 1], scs[MAXMR], ssn[MAXMR], sR[MAXMR * MAXMR];
  __shared__ double sred[16];
  const GmresScal* G = d.gsc;
  if (G->done) return;
  const int tid = threadIdx.x;
  for (int p = tid; p < m * m; p += 256) { const int cc = p / m, rr = p % m; sR[cc * MAXMR + rr] = G->R[cc * MAXMR + rr]; }
  if (tid <= m) sg[tid] = G->g[tid];
  if (tid < m) { scs[tid] = G->cs[tid]; ssn[tid] = G->sn[tid]; }
  __syncthreads();
  if (tid == 0) {
    for (int q = m - 1; q >= 0; --q) {
      double s = sg[q];
      for (int k = q + 1; k < m; ++k) s -= sR[k * MAXMR + q] * sy[k];
      sy[q] = s / sR[q * MAXMR + q];
    }
    for (int i = 0; i < m; ++i) ss[i] = 0.0;
    ss[m] = sg[m];
    for (int i = m - 1; i >= 0; --i) {
      const double a = ss[i], b = ss[i + 1];
      ss[i] = scs[i] * a - ssn[i] * b;
      ss[i + 1] = ssn[i] * a + scs[i] * b;
    }
    if (G->pending) {                            // V[m] is stored as w' (k_gs_lag): v_m = (w' - sum_k pc[k] v_k) * phinv
      const double f = ss[m] * G->phinv;
      for (int i = 0; i < m; ++i) ss[i] -= f * G->pc[i];
      ss[m] = f;
    }
  }
  __syncthreads();
  const bool first = (G->nit_prev == 0);
  double v[1] = {0.0};
  for (long long q = (long long)blockIdx.x * 256 + tid; q < d.npr; q += (long long)gridDim.x * 256) {
    double x = 0.0;
    for (int k = 0; k < m; ++k) x += sy[k] * d.Z[(size_t)k * d.npr + q];
    d.xacc[q] = first ? x : d.xacc[q] + x;
    double r = 0.0;
    for (int i = 0; i <= m; ++i) r += ss[i] * d.V[(size_t)i * d.ps + q];
    d.V[q] = r;
    v[0] += r * r;
  }
  block_reduce<1>(v, sred, tid, 256);
  if (tid == 0) d.gpart[blockIdx.x] = v[0];
}

namespace k2 {
// Second Gram-Schmidt pass (as the hexahedral set's): w' = w - sum_i h_i v_i is formed and projected once more,
// gpart2[k] = (w', v_k), k <= j, gpart2[j+1] = (w', w').  OFF by default on quadrilaterals (option "gs2_from" = first
// iteration of a cycle that gets it; default MAXMR = never): measured on the closed backward-facing step, whose solves take
// 20-45 iterations, the single classical pass keeps the residual estimate honest (same iterates with and without this pass,
// true residual = estimate: scripts/dbg_proj.py) -- what had looked like lost orthogonality there was the projection
// space's merge policy (k_proj_update).  Kept for meshes where one pass is not enough, as the hexahedral solves showed.
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_gmres_reorth(Dev d, int j) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, MM = C::MM, EPB = C::EPB, NT = C::NT;
  __shared__ double sh[MAXMR + 2];
  __shared__ double sdot[(MAXMR + 2) * (NT / 64)];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool pact = (el < EPB) && (e < d.nel) && nd < MM;
  if (d.gsc->done) return;
  if (d.nranks > 1 || d.use_tot) {
    if (tid <= j) sh[tid] = d.gtot[tid];
    lds_barrier();
  } else {
    sum_partials_multi(d.gpart, d.nblk, j + 1, sh, tid, NT);
  }
  const long long q = e * MM + nd;
  double w = 0.0;
  if (pact) {
    w = d.V[(size_t)(j + 1) * d.ps + q];
    for (int kk = 0; kk <= j; ++kk) w -= sh[kk] * d.V[(size_t)kk * d.ps + q];
    d.V[(size_t)(j + 1) * d.ps + q] = w;
  }
  const int lane = tid & 63, wv = tid >> 6;
  for (int kk = 0; kk <= j + 1; ++kk) {
    double x = 0.0;
    if (pact) x = w * ((kk <= j) ? d.V[(size_t)kk * d.ps + q] : w);
    x = wave_sum63(x);
    if (lane == 63) sdot[kk * (NT / 64) + wv] = x;
  }
  lds_barrier();
  if (tid <= j + 1) {
    double t = 0.0;
    for (int ww = 0; ww < NT / 64; ++ww) t += sdot[tid * (NT / 64) + ww];
    d.gpart2[(size_t)tid * d.nblk + blockIdx.x] = t;
  }
}
}  // namespace k2

// coarse solve: r_c = gather of element-corner restrictions (padded vertex table);
// x_c = Aci r_c with one wavefront per CROWS_W rows, lanes striding the (symmetric) row.
constexpr int CVT = 8;            // table width = max elements around a vertex
constexpr int CROWS_W = 2;        // rows per wavefront, processed together
// Acif: row-major fp32, leading dimension lda = nvert rounded up to 256 (zero padded) so that
// every lane streams float4 (4 columns) per load and a wave covers 256 columns per step.
__global__ __launch_bounds__(256) void k_coarse(Dev d) {
  extern __shared__ double srcv[];            // lda
  const int tid = threadIdx.x;
  if (d.gsc->done) return;
  const int nv = d.nvert, lda = d.coarse_lda;
  const int lane = tid & 63, w = tid >> 6;
  const int row0 = (blockIdx.x * 4 + w) * CROWS_W;
  const bool rok = row0 < nv, r1 = row0 + 1 < nv;
  // issue the matrix loads first (they do not depend on anything)
  constexpr int MAXIT = 12;                   // lda <= 3072
  const int nit = lda / 256;
  float4 a0[MAXIT], a1[MAXIT];
  const float4* A0 = reinterpret_cast<const float4*>(d.Acif + (size_t)(rok ? row0 : 0) * lda) + lane;
  const float4* A1 = reinterpret_cast<const float4*>(d.Acif + (size_t)(r1 ? row0 + 1 : 0) * lda) + lane;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i)
    if (i < nit) { a0[i] = A0[i * 64]; a1[i] = A1[i * 64]; }
  for (int v = tid; v < lda; v += 256) {
    double sv = 0.0;
    if (v < nv) {
      const int4 a = reinterpret_cast<const int4*>(d.vtab)[2 * v], b = reinterpret_cast<const int4*>(d.vtab)[2 * v + 1];
      const double e0 = d.ec[a.x];
      const double e1 = (a.y >= 0) ? d.ec[a.y] : 0.0, e2 = (a.z >= 0) ? d.ec[a.z] : 0.0, e3 = (a.w >= 0) ? d.ec[a.w] : 0.0;
      const double e4 = (b.x >= 0) ? d.ec[b.x] : 0.0, e5 = (b.y >= 0) ? d.ec[b.y] : 0.0, e6 = (b.z >= 0) ? d.ec[b.z] : 0.0;
      const double e7 = (b.w >= 0) ? d.ec[b.w] : 0.0;
      sv = ((((((e0 + e1) + e2) + e3) + e4) + e5) + e6) + e7;
    }
    srcv[v] = sv;
  }
  lds_barrier();
  double s0 = 0, s1 = 0;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i)
    if (i < nit) {
      const double* x = srcv + i * 256 + lane * 4;
      s0 += (double)a0[i].x * x[0] + (double)a0[i].y * x[1] + (double)a0[i].z * x[2] + (double)a0[i].w * x[3];
      s1 += (double)a1[i].x * x[0] + (double)a1[i].y * x[1] + (double)a1[i].z * x[2] + (double)a1[i].w * x[3];
    }
  s0 = wave_sum63(s0); s1 = wave_sum63(s1);
  if (lane == 63) {
    if (rok) d.xc[row0] = s0;
    if (r1) d.xc[row0 + 1] = s1;
  }
}

// GMRES bookkeeping of column j-1 and the coarse solve of iteration j in ONE kernel (quadrilateral set, dense in-LDS coarse
// solve): the chain of a GMRES iteration is 3 dependent kernels instead of 4.  Two independent strands per workgroup:
//   (a) matrix rows (issued first) + gather of the corner restrictions k_divgs wrote for the RAW w  ->  s = A_c^-1 R w
//   (b) sum of the dot-product partials of k_divgs(j-1)  ->  Hessenberg column, rotation, convergence (same arithmetic in every
//       workgroup => the same decision everywhere; workgroup 0 records it)
// and they meet at the end by LINEARITY:  x_c(v_j) = (s - sum_i h_i x_c(v_i)) / h_{j,j-1},  x_c(v_i) from the history `rch`
// (every wavefront appends its two rows), together with the pointwise  v_j = (w - sum_i h_i v_i) / h_{j,j-1}  (in place).
// j = 0: v_0 and its restriction come from k_gmres_update(j = -1).  The column of the LAST launched iteration is closed by
// k_gmres_update<N>(j = np-1) (a no-op launch when the solve is done).
// MAXIT >= coarse_lda / 256 (3, 6, 9 or 12) sizes every per-thread table; UC_ROWS rows per wavefront (measured on config 2:
// 2 rows / 263 workgroups beat 3 rows / 175 workgroups by 5 % of a matvec).
constexpr int UC_ROWS = 2;
// Closes GMRES column jj from the dot-product sums sh[0..jj+1] (one lane; the same arithmetic in every workgroup => the same
// decision everywhere): Givens rotations, residual, convergence.  sbc = {1 / h_{jj+1,jj}, converged}; `record`: this workgroup
// writes the solve's state (GmresScal, statistics).
__device__ __forceinline__ void uc_rotate(const Dev& d, GmresScal* G, int jj, double gj, double scale, int min_iter, int ord, bool record,
                                          const double* sh, const double* scs, const double* ssn, double* scol, double* sbc) {
    double s2 = 0.0;
    for (int q = 0; q <= jj; ++q) s2 += sh[q] * sh[q];
    const double hn2 = sh[jj + 1] - s2;
    const double hn = sqrt(hn2 > 0.0 ? hn2 : 0.0);
    double* col = scol;
    for (int q = 0; q <= jj; ++q) col[q] = sh[q];
    col[jj + 1] = hn;
    for (int q = 0; q < jj; ++q) {
      const double t = scs[q] * col[q] + ssn[q] * col[q + 1];
      col[q + 1] = -ssn[q] * col[q] + scs[q] * col[q + 1];
      col[q] = t;
    }
    const double rho = sqrt(col[jj] * col[jj] + col[jj + 1] * col[jj + 1]);
    const double cj = (rho > 0.0) ? col[jj] / rho : 1.0, sj = (rho > 0.0) ? col[jj + 1] / rho : 0.0;
    col[jj] = rho;
    const double res = fabs(sj * gj) * scale;
    const double tol = d.tol_relative ? fmax(d.tol_pres * G->gnorm0 * scale, d.tol_pres_floor) : d.tol_pres;
    const bool conv = (res <= tol && (jj + 1) >= min_iter) || !(hn > 0.0) || (d.pres_cap > 0 && (jj + 1) >= d.pres_cap);
    sbc[0] = (hn > 0.0) ? 1.0 / hn : 0.0;
    sbc[1] = conv ? 1.0 : 0.0;
    sbc[2] = cj * gj; sbc[3] = -sj * gj;          // g_jj, g_{jj+1} after the rotation (k_pres_update's folded close)
    if (record) {
      G->cs[jj] = cj; G->sn[jj] = sj;
      for (int q = 0; q <= jj; ++q) G->R[jj * MAXMR + q] = col[q];
      G->g[jj] = cj * gj;
      G->g[jj + 1] = -sj * gj; G->gpre[jj + 1] = -sj * gj;
      G->nit = jj + 1;
      G->resid = res;
      d.stats->pres_jsum += jj;
      if (conv) {
        atomicAdd((unsigned long long*)&d.stats->pres_iters, (unsigned long long)(G->nit_prev + jj + 1));
        atomicMax((unsigned long long*)&d.stats->max_pres, (unsigned long long)(G->nit_prev + jj + 1));
        atomicMax((unsigned long long*)&d.stats->max_pres_k[ord], (unsigned long long)(G->nit_prev + jj + 1)); rec_step_iters(d, 1, G->nit_prev + jj + 1);
        d.stats->last_pres_res = res;
        if (!(res <= tol) && hn > 0.0) {
          d.stats->capped_solves += 1;
          if (res / tol > d.stats->worst_cap_ratio) d.stats->worst_cap_ratio = res / tol;
        }
        G->done = 1;
      }
    }
}

// The START of a solve inside A_0 (Dev::uc_start, set per launch): what k_gmres_update(j = -1) does -- |g'| from the partials
// the (element-aligned) projection kernel left in row 0, the solve's state, the convergence flag -- by one lane of every
// workgroup from sh[0] = |g'|^2; sbc = {1 / |g'|, done}.  The raw g' is in Dev::Wr, its corner restrictions in Dev::ecv.
__device__ __forceinline__ void uc_start_solve(const Dev& d, GmresScal* G, double scale, int min_iter, bool record, const double* sh, double* sbc) {
  const double hn = sqrt(sh[0]);
  const double gn0 = (d.nproj_max <= 0) ? hn : G->gnorm0;
  const double tol0 = d.tol_relative ? fmax(d.tol_pres * gn0 * scale, d.tol_pres_floor) : d.tol_pres;
  const int dn = (!(hn > 0.0) || (min_iter <= 0 && hn * scale <= tol0)) ? 1 : 0;
  sbc[0] = (hn > 0.0) ? 1.0 / hn : 0.0;
  sbc[1] = dn ? 1.0 : 0.0;
  if (record) {
    G->beta0 = hn; G->g[0] = hn; G->gpre[0] = hn; G->nit = 0; G->nit_prev = 0; G->resid = hn * scale;
    if (d.nproj_max <= 0) G->gnorm0 = hn;
    if (dn) d.stats->last_pres_res = hn * scale;
    if (!(hn * 0.0 == 0.0)) d.stats->nonfinite += 1;      // |g| is NaN or Inf (map_finish returns NSK_ENAN)
    G->done = dn;
  }
}

// LEAN (the persistent pressure tail, which must stay below 256 registers to be resident at two workgroups per CU): the corner
// restrictions and the matrix rows are loaded in chunks of three 256-column blocks where they are used instead of all at the
// top; the same operations in the same order, so both forms return the same bits.
// NOV (round 6, k_schwarz_uc): the pointwise update of v_j is NOT done here (the Schwarz workgroups of the same launch form v_j
// where they need it); everything else -- column, coarse solve by linearity, history -- as before.
// LEAN_AM (default = LEAN): the same choice for the matrix rows alone.  k_schwarz_uc takes <LEAN = true, NOV, LEAN_AM = false>: the
// corner values in chunks (they are consumed first), ALL matrix loads in flight behind the first chunk (stamps, round 6: the
// chunked product was 3.9 of the coarse role's 10.8 us).
template <int MAXIT, bool LEAN = false, bool NOV = false, bool LEAN_AM = LEAN>
__device__ __forceinline__ void update_coarse_body(const Dev& d, int j, double scale, int min_iter, int ord, const unsigned bx_, const unsigned gx_) {
  extern __shared__ double srcv[];            // lda
  __shared__ double sh[MAXMR + 2], scs[MAXMR], ssn[MAXMR], scol[MAXMR + 2], sbc[4];
  const int tid = threadIdx.x;
  GmresScal* G = d.gsc;
  if (G->done) return;                        // (first: a launch that finds its solve done must stay cheap)
  NSK_STAMP(1);
  const int nv = d.nvert, lda = d.coarse_lda;
  const int lane = tid & 63, w = tid >> 6;
  const int row0 = (bx_ * 4 + w) * UC_ROWS;
  const int nit = lda / 256;
  const int jj = j - 1;                       // the column this launch closes (j > 0)
  // Loads return in issue order: the short dependent chains go first (vertex tables -> corner values; partials of the first
  // row per wavefront), the 24 independent matrix loads last -- they are needed at the product only.
  // corner restrictions of this thread's vertices, vertex-major slots (Dev::ecv): 64 contiguous bytes per vertex
  constexpr int CVN = LEAN ? 1 : MAXIT;
  double2 cv[CVN][4];
  if constexpr (!LEAN) {
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {
      const int v = tid + i * 256;
      const double2* E = reinterpret_cast<const double2*>(d.ecv) + (size_t)(i < nit && v < nv ? v : 0) * 4;
      if (i < nit) { cv[i][0] = E[0]; cv[i][1] = E[1]; cv[i][2] = E[2]; cv[i][3] = E[3]; }
    }
  }
  double gj = 0.0;
  double pr[8], pr2[8];                       // partials of rows `w` and `w + 4` (the first two this wavefront sums): d.nblk <= 512 here
  const bool prow = j > 0 && w < jj + 2 && d.nblk <= 512, prow2 = j > 0 && w + 4 < jj + 2 && d.nblk <= 512;
#pragma unroll
  for (int k = 0; k < 8; ++k) pr[k] = (prow && lane + 64 * k < d.nblk) ? d.gpart[(size_t)w * d.nblk + lane + 64 * k] : 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) pr2[k] = (prow2 && lane + 64 * k < d.nblk) ? d.gpart[(size_t)(w + 4) * d.nblk + lane + 64 * k] : 0.0;
  // history of the coarse solutions for this wavefront's rows: lane k holds x_c(v_k)[row] (MAXMR <= 64 lanes)
  double rh[UC_ROWS];
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) rh[r] = (lane < j && row0 + r < nv) ? d.rch[(size_t)lane * lda + row0 + r] : 0.0;
  // this thread's entry of v_j and of the basis vectors it is orthogonalised against (the first 8; more in the loop below)
  const long long q0 = (long long)bx_ * 256 + tid;
  double vq = 0.0, vk[8];
  if constexpr (!NOV) {
    if (j > 0 && q0 < d.npr) vq = d.V[(size_t)j * d.ps + q0];
#pragma unroll
    for (int k = 0; k < 8; ++k) vk[k] = (k < j && q0 < d.npr) ? d.V[(size_t)k * d.ps + q0] : 0.0;
  }
  if (j > 0) {
    if (tid < jj) { scs[tid] = G->cs[tid]; ssn[tid] = G->sn[tid]; }
    gj = G->gpre[jj];
  }
  constexpr int AMN = LEAN_AM ? 1 : MAXIT;
  float4 am[UC_ROWS][AMN];
  if constexpr (!LEAN_AM) {
#pragma unroll
    for (int r = 0; r < UC_ROWS; ++r) {
      const float4* A = reinterpret_cast<const float4*>(d.Acif + (size_t)(row0 + r < nv ? row0 + r : 0) * lda) + lane;
#pragma unroll
      for (int i = 0; i < MAXIT; ++i)
        if (i < nit) am[r][i] = A[i * 64];
    }
  }
  if constexpr (!LEAN) {
#pragma unroll
    for (int i = 0; i < MAXIT; ++i) {         // R w (raw): the corner restrictions of a vertex in the order of vtab (unused slots are zero)
      const int v = tid + i * 256;
      if (i < nit) srcv[v] = (v < nv) ? ((((((cv[i][0].x + cv[i][0].y) + cv[i][1].x) + cv[i][1].y) + cv[i][2].x) + cv[i][2].y) + cv[i][3].x) + cv[i][3].y : 0.0;
    }
  } else {
#pragma unroll 3
    for (int i = 0; i < MAXIT; ++i) {
      const int v = tid + i * 256;
      if (i < nit) {
        const double2* E = reinterpret_cast<const double2*>(d.ecv) + (size_t)(v < nv ? v : 0) * 4;
        const double2 c0 = E[0], c1 = E[1], c2 = E[2], c3 = E[3];
        srcv[v] = (v < nv) ? ((((((c0.x + c0.y) + c1.x) + c1.y) + c2.x) + c2.y) + c3.x) + c3.y : 0.0;
      }
    }
  }
  NSK_STAMP(2);
  if (j > 0) {
    if (d.nblk <= 512) {                      // first row per wavefront from the registers, the rest as sum_partials_multi
      if (prow) {
        double sacc = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc += pr[k];
        sacc = wave_sum63(sacc);
        if (lane == 63) sh[w] = sacc;
      }
      if (prow2) {
        double sacc = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc += pr2[k];
        sacc = wave_sum63(sacc);
        if (lane == 63) sh[w + 4] = sacc;
      }
      for (int q = w + 8; q < jj + 2; q += 4) {
        double sacc = 0.0;
        for (int k = lane; k < d.nblk; k += 64) sacc += d.gpart[(size_t)q * d.nblk + k];
        sacc = wave_sum63(sacc);
        if (lane == 63) sh[q] = sacc;
      }
      lds_barrier();
    } else {
      sum_partials_multi(d.gpart, d.nblk, jj + 2, sh, tid, 256);   // ends with an LDS barrier
    }
  } else {
    lds_barrier();
  }
  NSK_STAMP(3);
  if (j > 0 && tid == 0) uc_rotate(d, G, jj, gj, scale, min_iter, ord, bx_ == 0, sh, scs, ssn, scol, sbc);
  NSK_STAMP(4);   // one lane rotates the column while the others start on the matrix product
  double sr[UC_ROWS];
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) sr[r] = 0.0;
  if constexpr (!LEAN_AM) {
#pragma unroll
    for (int i = 0; i < MAXIT; ++i)
      if (i < nit) {
        const double* x = srcv + i * 256 + lane * 4;
#pragma unroll
        for (int r = 0; r < UC_ROWS; ++r)
          sr[r] += (double)am[r][i].x * x[0] + (double)am[r][i].y * x[1] + (double)am[r][i].z * x[2] + (double)am[r][i].w * x[3];
      }
  } else {
    const float4* A0 = reinterpret_cast<const float4*>(d.Acif + (size_t)(row0 < nv ? row0 : 0) * lda) + lane;
    const float4* A1 = reinterpret_cast<const float4*>(d.Acif + (size_t)(row0 + 1 < nv ? row0 + 1 : 0) * lda) + lane;
    static_assert(UC_ROWS == 2, "lean form: two rows per wavefront");
#pragma unroll 3
    for (int i = 0; i < MAXIT; ++i)
      if (i < nit) {
        const float4 a0 = A0[i * 64], a1 = A1[i * 64];
        const double* x = srcv + i * 256 + lane * 4;
        sr[0] += (double)a0.x * x[0] + (double)a0.y * x[1] + (double)a0.z * x[2] + (double)a0.w * x[3];
        sr[1] += (double)a1.x * x[0] + (double)a1.y * x[1] + (double)a1.z * x[2] + (double)a1.w * x[3];
      }
  }
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) sr[r] = wave_sum63(sr[r]);
  double hinv = 1.0;
  NSK_STAMP(5);
  if (j > 0) {
    lds_barrier();                            // column rotated
    NSK_STAMP(6);
    hinv = sbc[0];
    if (sbc[1] != 0.0) return;                // converged: nothing of iteration j is needed
    if constexpr (!NOV) {
    if (q0 < d.npr) {                         // first entry: operands already in registers
      double x = vq;
#pragma unroll
      for (int k = 0; k < 8; ++k) if (k < j) x -= sh[k] * vk[k];
      for (int k = 8; k < j; ++k) x -= sh[k] * d.V[(size_t)k * d.ps + q0];
      d.V[(size_t)j * d.ps + q0] = x * hinv;
    }
    for (long long q = q0 + (long long)gx_ * 256; q < d.npr; q += (long long)gx_ * 256) {
      double x = d.V[(size_t)j * d.ps + q];
#pragma unroll 4
      for (int k = 0; k < j; ++k) x -= sh[k] * d.V[(size_t)k * d.ps + q];
      d.V[(size_t)j * d.ps + q] = x * hinv;
    }
    }
  }
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) {         // x_c(v_j)[row] = (s - sum_k h_k x_c(v_k)[row]) / h_{j,j-1}: lane k holds term k
    double t = (lane < j) ? sh[lane] * rh[r] : 0.0;
    t = wave_sum63(t);
    sr[r] = (sr[r] - t) * hinv;
  }
  if (lane == 63) {
#pragma unroll
    for (int r = 0; r < UC_ROWS; ++r)
      if (row0 + r < nv) { d.xc[row0 + r] = sr[r]; d.rch[(size_t)j * lda + row0 + r] = sr[r]; }
  }
  NSK_STAMP(7);
}
template <int MAXIT>
__global__ __launch_bounds__(256) void k_update_coarse(Dev d, int j, double scale, int min_iter, int ord) {
  update_coarse_body<MAXIT>(d, j, scale, min_iter, ord, blockIdx.x, gridDim.x);
}


// large coarse spaces (nvert > 3072, e.g. the 2x2-refined mesh): streaming variant, the vertex
// restriction r_c is built once per launch in global memory by a separate tiny kernel.
__global__ __launch_bounds__(256) void k_coarse_restrict(Dev d, double* __restrict__ rc) {
  if (d.gsc->done) return;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= d.coarse_lda) return;
  double sv = 0.0;
  if (v < d.nvert) {
    const int4 a = reinterpret_cast<const int4*>(d.vtab)[2 * v], b = reinterpret_cast<const int4*>(d.vtab)[2 * v + 1];
    const double e0 = (a.x >= 0) ? d.ec[a.x] : 0.0;     // a vertex may have no element on this rank
    const double e1 = (a.y >= 0) ? d.ec[a.y] : 0.0, e2 = (a.z >= 0) ? d.ec[a.z] : 0.0, e3 = (a.w >= 0) ? d.ec[a.w] : 0.0;
    const double e4 = (b.x >= 0) ? d.ec[b.x] : 0.0, e5 = (b.y >= 0) ? d.ec[b.y] : 0.0, e6 = (b.z >= 0) ? d.ec[b.z] : 0.0;
    const double e7 = (b.w >= 0) ? d.ec[b.w] : 0.0;
    sv = ((((((e0 + e1) + e2) + e3) + e4) + e5) + e6) + e7;
  }
  rc[v] = sv;
}
__global__ __launch_bounds__(256) void k_coarse_big(Dev d, const double* __restrict__ rc) {
  if (d.gsc->done) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= d.nvert) return;
  const float4* A = reinterpret_cast<const float4*>(d.Acif + (size_t)row * d.coarse_lda) + lane;
  const int nit = d.coarse_lda / 256;
  double s = 0.0;
#pragma unroll 8
  for (int i = 0; i < nit; ++i) {
    const float4 a = A[i * 64];
    const double* x = rc + i * 256 + lane * 4;
    s += (double)a.x * x[0] + (double)a.y * x[1] + (double)a.z * x[2] + (double)a.w * x[3];
  }
  s = wave_sum63(s);
  if (lane == 63) d.xc[row] = s;
}

namespace k2 {
// z_j = RAS(v_j) + R^T x_c ;  yl = D^T z_j  (unassembled velocity-space)
// patch tables have a fixed stride PS per element: idx[e*PS + k] (-1 padded), inverse [e][k][MM] fp32
template <int N>
__device__ __forceinline__ void schwarz_body(const Dev& d, const double* __restrict__ vin, double* __restrict__ zout, int use_coarse, int check_done, const unsigned bx_, const unsigned gx_) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  constexpr int MAXP = (M + 8) * (M + 8);
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double sP[4 * EPB * MM], sB[4 * EPB * NM];
  __shared__ double sr[EPB * MAXP];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const int bid = d.boff + (int)xcd_element(bx_, gx_);
  const long long e = (long long)bid * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  if (check_done && d.gsc->done) return;
  const int PS = d.p_stride;
  // front-load: patch indices (2 per thread is enough for PS <= 2*NN), metrics, coarse values, basis
  // (measured: issuing the patch inverse -- the largest stream -- before the barrier as well makes the kernel SLOWER, 8 -> 14 us)
  int i0 = -1, i1 = -1;
  if (act) {
    if (nd < PS) i0 = d.p_idx[e * PS + nd];
    if (nd + NN < PS) i1 = d.p_idx[e * PS + nd + NN];
  }
  const bool pact = act && nd < MM;
  const long long q = e * MM + nd;
  double m0 = 0, m1 = 0, m2 = 0, m3 = 0, zc = 0;
  int4 ev = make_int4(0, 0, 0, 0);
  if (pact) {
    m0 = d.w2rx[q]; m1 = d.w2sx[q]; m2 = d.w2ry[q]; m3 = d.w2sy[q];
    if (use_coarse) ev = reinterpret_cast<const int4*>(d.evert)[e];
  }
  double j12a = 0, d12a = 0;
  if (tid < NM) { j12a = d.J12[tid]; d12a = d.D12[tid]; }
  const double v0 = (i0 >= 0) ? vin[i0] : 0.0, v1 = (i1 >= 0) ? vin[i1] : 0.0;
  if (pact && use_coarse)
    zc = d.hat[0 * MM + nd] * d.xc[ev.x] + d.hat[1 * MM + nd] * d.xc[ev.y] + d.hat[2 * MM + nd] * d.xc[ev.z] + d.hat[3 * MM + nd] * d.xc[ev.w];
  if (tid < NM) { sJ12[tid] = j12a; sD12[tid] = d12a; }
  if (act) {
    if (nd < PS) sr[el * MAXP + nd] = v0;
    if (nd + NN < PS) sr[el * MAXP + nd + NN] = v1;
  }
  lds_barrier();
  if (pact) {
    // inverse stored [e][k/4][own row][4] fp32: one float4 per lane per 4 patch dofs
    const float4* A = reinterpret_cast<const float4*>(d.p_inv + (size_t)e * PS * MM) + nd;
    const double* r = sr + el * MAXP;
    double z0 = 0, z1 = 0, z2 = 0, z3 = 0;
    const int nq = PS / 4;                                        // PS is a multiple of 4
#pragma unroll 8
    for (int k4 = 0; k4 < nq; ++k4) {
      const float4 a = A[(size_t)k4 * MM];
      z0 += (double)a.x * r[4 * k4 + 0];
      z1 += (double)a.y * r[4 * k4 + 1];
      z2 += (double)a.z * r[4 * k4 + 2];
      z3 += (double)a.w * r[4 * k4 + 3];
    }
    const double z = ((z0 + z1) + (z2 + z3)) + zc;
    zout[q] = z;
    sP[(0 * EPB + el) * MM + nd] = z * m0;
    sP[(1 * EPB + el) * MM + nd] = z * m1;
    sP[(2 * EPB + el) * MM + nd] = z * m2;
    sP[(3 * EPB + el) * MM + nd] = z * m3;
  }
  lds_barrier();
  double gx, gy;
  opgradt_tiles<N, EPB>(sJ12, sD12, sP, sB, act, el, nd, gx, gy);
  if (act) {
    const long long l = e * NN + nd;
    d.yl[l] = gx;
    d.yl[d.cs + l] = gy;
  }
}
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_schwarz(Dev d, const double* __restrict__ vin,
                                                        double* __restrict__ zout, int use_coarse,
                                                        int check_done) {
  schwarz_body<N>(d, vin, zout, use_coarse, check_done, blockIdx.x, gridDim.x);
}


// yl = D^T p for an arbitrary pressure vector (setup probes, tests, projection)
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_gradt(Dev d, const double* __restrict__ pin,
                                                      double* __restrict__ yl) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double sP[4 * EPB * MM], sB[4 * EPB * NM];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  load_basis<N, EPB>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  if (act && nd < MM) {
    const long long q = e * MM + nd;
    const double z = pin[q];
    sP[(0 * EPB + el) * MM + nd] = z * d.w2rx[q];
    sP[(1 * EPB + el) * MM + nd] = z * d.w2sx[q];
    sP[(2 * EPB + el) * MM + nd] = z * d.w2ry[q];
    sP[(3 * EPB + el) * MM + nd] = z * d.w2sy[q];
  }
  __syncthreads();
  double gx, gy;
  opgradt_tiles<N, EPB>(sJ12, sD12, sP, sB, act, el, nd, gx, gy);
  if (act) {
    const long long l = e * NN + nd;
    yl[l] = gx;
    yl[d.cs + l] = gy;
  }
}

// w = D ( B^-1 mask dssum(yl) ) ; optional dots (w, V_i), i <= j, and (w,w)
// check_done: 0 = always run, 1 = leave when the solve is done, 2 = the same and write the corner restriction of w to d.ec
template <int N>
__device__ __forceinline__ void divgs_body(const Dev& d, const double* __restrict__ yl, double* __restrict__ wout, int j, int check_done, const unsigned bx_, const unsigned gx_) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[2 * EPB * NN], sA[4 * EPB * NM];
  __shared__ double sdot[(MAXMR + 2) * 4];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const int bid = d.boff + (int)xcd_element(bx_, gx_);
  const long long e = (long long)bid * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  NSK_STAMP(0);
  if (check_done && d.gsc->done) return;
  NSK_STAMP(1);
  const long long l = e * NN + nd;
  int4 tab = make_int4(0, -1, -1, -1);
  double bi = 0;
  if (act) { tab = d.gs_tab[l]; bi = d.binv[l]; }
  int ecs = 0;
  if (check_done == 2 && d.ecv && act && nd < 4) ecs = d.ecslot[e * 4 + nd];      // (with the first loads: the store at the end waits for nothing)
  double j12a = 0, d12a = 0;
  if (tid < NM) { j12a = d.J12[tid]; d12a = d.D12[tid]; }
  GsVals g0, g1;
  if (act) { g0 = gs_load(yl, tab, l); g1 = gs_load(yl + d.cs, tab, l); }
  if (tid < NM) { sJ12[tid] = j12a; sD12[tid] = d12a; }
  if (act) {
    su[(0 * EPB + el) * NN + nd] = bi * gs_sum(g0, yl, d, tab, l);
    su[(1 * EPB + el) * NN + nd] = bi * gs_sum(g1, yl + d.cs, d, tab, l);
  }
  NSK_STAMP(2);
  lds_barrier();
  const double w = opdiv_tiles<N, EPB>(sJ12, sD12, su, sA, act, el, nd, d, e);
  NSK_STAMP(3);
  const bool pact = act && nd < MM;
  const long long q = e * MM + nd;
  if (pact) wout[q] = w;
  if (check_done == 2) {              // merged bookkeeping (k_update_coarse): element-corner restriction of the raw w
    __shared__ double swr[EPB * MM], shat[4 * MM];
    for (int k = tid; k < 4 * MM; k += NT) shat[k] = d.hat[k];
    if (pact) swr[el * MM + nd] = w;
    lds_barrier();
    if (act && nd < 4) {
      double s = 0.0;
#pragma unroll 6
      for (int k = 0; k < MM; ++k) s += shat[nd * MM + k] * swr[el * MM + k];
      d.ec[e * 4 + nd] = s;
      if (d.ecv) d.ecv[ecs] = s;
    }
  }
  if (j >= 0) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll 4
    for (int k = 0; k <= j + 1; ++k) {
      double x = 0.0;
      if (pact) x = w * ((k <= j) ? d.V[(size_t)k * d.ps + q] : w);
      x = wave_sum63(x);
      if (lane == 63) sdot[k * 4 + wv] = x;
    }
    lds_barrier();
    constexpr int NW = NT / 64;
    if (tid <= j + 1) {
      double t = 0.0;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) t += sdot[tid * 4 + ww];
      d.gpart[(size_t)tid * d.nblk + bid] = t;
    }
    NSK_STAMP(4);
  }
}
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_divgs(Dev d, const double* __restrict__ yl,
                                                      double* __restrict__ wout, int j, int check_done) {
  divgs_body<N>(d, yl, wout, j, check_done, blockIdx.x, gridDim.x);
}


// k_proj_apply on the ELEMENT-ALIGNED thread map of the pressure kernels (thread -> (element, Gauss node), EPB elements per
// workgroup, grid = nblk), for the two-launch GMRES iteration: g' = g - sum_i a_i E x_i goes RAW to Dev::Wr (A_0 normalises it
// on the fly: the Schwarz workgroups read it across element boundaries), its |g'|^2 partials to row 0, and -- what the flat map
// of k_proj_apply cannot do -- the element-corner restrictions of g' to Dev::ec / ecv, so that the solve can start INSIDE A_0
// (Dev::uc_start) and k_gmres_update(j = -1) is not launched.  Same coefficients a_i as k_proj_apply (same partial sums).
template <int N>
__global__ __launch_bounds__(256) void k_proj_apply_e(Dev d) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, MM = C::MM, EPB = C::EPB;
  __shared__ double sh[MAXPROJ + 1];
  __shared__ double sred[16];
  __shared__ double sv[EPB * MM], shat[4 * MM];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  const bool pact = act && nd < MM;
  GmresScal* G = d.gsc;
  const int np = G->nproj;
  const long long q = (act ? e : 0) * MM + (nd < MM ? nd : 0);
  const int npre = d.nproj_max < MAXPROJ ? d.nproj_max : MAXPROJ;
  const double g0 = d.V[q];
  double pe[MAXPROJ];
#pragma unroll
  for (int k = 0; k < MAXPROJ; ++k) pe[k] = d.PEX[(size_t)(k < npre ? k : 0) * d.npr + q];      // (slots >= nproj: finite stale data, zero coefficient)
  const double pnv = (tid < MAXPROJ) ? G->pn[tid] : 1.0;
  const int ecs = d.ecslot[(act ? e : 0) * 4 + (nd & 3)];
  double hatv[(4 * MM + 255) / 256];
#pragma unroll
  for (int r = 0; r < (4 * MM + 255) / 256; ++r) hatv[r] = d.hat[(tid + r * 256 < 4 * MM) ? tid + r * 256 : 0];
  PartialRows<4> pr;                  // 16 rows per pass: a second pass for spaces of more than 16 vectors
  pr.issue(d.ppart, d.nblk, d.nblk <= 512 ? d.nproj_max : 0, tid);
  if (d.nblk <= 512) {
    pr.reduce(d.ppart, d.nblk, np, sh, tid, 0, np <= 16);
    if (np > 16) { pr.issue(d.ppart, d.nblk, np, tid, 16); pr.reduce(d.ppart, d.nblk, np, sh, tid, 16); }
  } else sum_partials_multi(d.ppart, d.nblk, np, sh, tid, 256);
  if (blockIdx.x == 0) {
    double gg[1];
    sum_partials<1>(d.ppart + (size_t)MAXPROJ * d.nblk, d.nblk, gg, sred, tid, 256);
    if (tid == 0) G->gnorm0 = sqrt(gg[0]);
  }
  if (tid < np) sh[tid] = sh[tid] / pnv;
#pragma unroll
  for (int r = 0; r < (4 * MM + 255) / 256; ++r) if (tid + r * 256 < 4 * MM) shat[tid + r * 256] = hatv[r];
  __syncthreads();
  if (blockIdx.x == 0 && tid < np) G->pa[tid] = sh[tid];
  double v[1] = {0.0};
  if (pact) {
    double g = g0;
#pragma unroll
    for (int k = 0; k < MAXPROJ; ++k) if (k < np) g -= sh[k] * pe[k];
    d.Wr[q] = g;
    sv[el * MM + nd] = g;
    v[0] = g * g;
  }
  block_reduce<1>(v, sred, tid, 256);            // (two LDS barriers: sv is complete behind them)
  if (tid == 0) d.gpart[blockIdx.x] = v[0];
  if (act && nd < 4) {
    double s = 0.0;
#pragma unroll 6
    for (int k = 0; k < MM; ++k) s += shat[nd * MM + k] * sv[el * MM + k];
    d.ec[e * 4 + nd] = s;
    d.ecv[ecs] = s;
  }
}

// ---------------------------------------------------------------------------
// Round 6: the merged GMRES iteration in TWO launches instead of three (core/matvec.f:216-233 spends its time here).
//   A_j = k_schwarz_uc (grid = nsw Schwarz workgroups + cgrid coarse workgroups, 256 threads each):
//         every workgroup sums the dot-product partials of B_{j-1} and closes column j-1 (same arithmetic everywhere);
//         Schwarz workgroups form  v_j = (w - sum_k h_k v_k) / h_{j,j-1}  ON THE FLY at their patch nodes from the raw w
//         (Dev::Wr) and the basis, store their own nodes of it to V[j], apply the patch inverses and D^T:  yl = D^T RAS(v_j);
//         coarse workgroups: x_c(v_j) by linearity as k_update_coarse (update_coarse_body<.., NOV>), next to them.
//   B_j = k_divgs_t:  w = D B^-1 dssum(yl) + Tc x_c  (the coarse part of z_j through its precomputed image, Dev::Tc),
//         Z_j += R^T x_c, dots against V_0..j, corner restrictions; raw w -> Dev::Wr.
// The additive preconditioner is what allows it: Schwarz(v_j) and coarse(v_j) are independent given the column.  Same Krylov
// method, same h and v_j bits as the three-launch form; w differs from it by rounding (the coarse part is summed separately).
// ---------------------------------------------------------------------------
// The coarse role of k_schwarz_uc: update_coarse_body<.., NOV> with its loads in the order that fits three workgroups per CU
// (168 registers) WITHOUT chunking the matrix rows -- partial sums and the first chunk of corner values in the first trip; the
// matrix rows (72 registers) go out when the partial-sum registers are free, behind them the other chunks of corner values, so
// that the product finds its operands waiting (stamps of the chunked form: 4.5 us corner sums + 3.9 us product of 10.8).
// The same operations in the same order as update_coarse_body: the same bits (the persistent tail runs that one).
template <int MAXIT>
__device__ __forceinline__ void uc_coarse_role(const Dev& d, int j, double scale, int min_iter, int ord, const unsigned bx_) {
  extern __shared__ double srcv[];            // lda
  __shared__ double sh[MAXMR + 2], scs[MAXMR], ssn[MAXMR], scol[MAXMR + 2], sbc[4];
  const int tid = threadIdx.x;
  GmresScal* G = d.gsc;
  NSK_STAMP(1);
  const int nv = d.nvert, lda = d.coarse_lda;
  const int lane = tid & 63, w = tid >> 6;
  const int row0 = (bx_ * 4 + w) * UC_ROWS;
  constexpr int nit = MAXIT;                  // the host pads coarse_lda to 256 MAXIT (a multiple of 768): no run-time bounds, no branches between the loads
  const int jj = j - 1;
  constexpr int CH = 3;
  const bool start = d.uc_start != 0;          // (j = 0 only) the solve starts here: row 0 of the partials = |g'|^2
  const int nrow = (j > 0) ? jj + 2 : (start ? 1 : 0);
  PartialRows<2, true> pr;
  pr.issue(d.gpart, d.nblk, d.nblk <= 512 ? nrow : 0, tid);
  double rh[UC_ROWS];
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) rh[r] = (lane < j && row0 + r < nv) ? d.rch[(size_t)lane * lda + row0 + r] : 0.0;
  double gj = 0.0;
  if (j > 0) {
    if (tid < jj) { scs[tid] = G->cs[tid]; ssn[tid] = G->sn[tid]; }
    gj = G->gpre[jj];
  }
  double2 c0[CH][4];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int v = tid + i * 256;
    const double2* E = reinterpret_cast<const double2*>(d.ecv) + (size_t)(i < nit && v < nv ? v : 0) * 4;
    if (i < nit) { c0[i][0] = E[0]; c0[i][1] = E[1]; c0[i][2] = E[2]; c0[i][3] = E[3]; }
  }
  if (nrow > 0) {
    if (d.nblk <= 512) pr.reduce(d.gpart, d.nblk, nrow, sh, tid);               // ends with an LDS barrier
    else sum_partials_multi(d.gpart, d.nblk, nrow, sh, tid, 256);
  }
  NSK_STAMP(2);
  // (compiler fences between the phases: hoisting every load to the top costs 226 registers; the order below needs ~150)
#define UC_FENCE() asm volatile("" ::: "memory")
#pragma unroll
  for (int i = 0; i < CH; ++i) {              // R w (raw): the corner restrictions of a vertex in the order of vtab (unused slots are zero)
    const int v = tid + i * 256;
    if (i < nit) srcv[v] = (v < nv) ? ((((((c0[i][0].x + c0[i][0].y) + c0[i][1].x) + c0[i][1].y) + c0[i][2].x) + c0[i][2].y) + c0[i][3].x) + c0[i][3].y : 0.0;
  }
  UC_FENCE();
  float4 am[UC_ROWS][MAXIT];
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) {
    const float4* A = reinterpret_cast<const float4*>(d.Acif + (size_t)(row0 + r < nv ? row0 + r : 0) * lda) + lane;
#pragma unroll
    for (int i = 0; i < MAXIT; ++i)
      if (i < nit) am[r][i] = A[i * 64];
  }
#pragma unroll
  for (int ic = CH; ic < MAXIT; ic += CH) {
    double2 c1[CH][4];
#pragma unroll
    for (int ii = 0; ii < CH; ++ii) {
      const int i = ic + ii, v = tid + i * 256;
      const double2* E = reinterpret_cast<const double2*>(d.ecv) + (size_t)(i < nit && v < nv ? v : 0) * 4;
      if (i < MAXIT && i < nit) { c1[ii][0] = E[0]; c1[ii][1] = E[1]; c1[ii][2] = E[2]; c1[ii][3] = E[3]; }
    }
#pragma unroll
    for (int ii = 0; ii < CH; ++ii) {
      const int i = ic + ii, v = tid + i * 256;
      if (i < MAXIT && i < nit) srcv[v] = (v < nv) ? ((((((c1[ii][0].x + c1[ii][0].y) + c1[ii][1].x) + c1[ii][1].y) + c1[ii][2].x) + c1[ii][2].y) + c1[ii][3].x) + c1[ii][3].y : 0.0;
    }
    UC_FENCE();
  }
  NSK_STAMP(3);
  lds_barrier();
  if (j > 0 && tid == 0) uc_rotate(d, G, jj, gj, scale, min_iter, ord, bx_ == 0, sh, scs, ssn, scol, sbc);
  if (start && tid == 0) uc_start_solve(d, G, scale, min_iter, bx_ == 0, sh, sbc);
  NSK_STAMP(4);
  double sr[UC_ROWS];
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) sr[r] = 0.0;
#pragma unroll
  for (int i = 0; i < MAXIT; ++i)
    if (i < nit) {
      const double* x = srcv + i * 256 + lane * 4;
#pragma unroll
      for (int r = 0; r < UC_ROWS; ++r)
        sr[r] += (double)am[r][i].x * x[0] + (double)am[r][i].y * x[1] + (double)am[r][i].z * x[2] + (double)am[r][i].w * x[3];
    }
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) sr[r] = wave_sum63(sr[r]);
  double hinv = 1.0;
  NSK_STAMP(5);
  if (nrow > 0) {
    lds_barrier();                            // column rotated (start: |g'| known)
    NSK_STAMP(6);
    hinv = sbc[0];
    if (sbc[1] != 0.0) return;                // converged: nothing of iteration j is needed
  }
#pragma unroll
  for (int r = 0; r < UC_ROWS; ++r) {         // x_c(v_j)[row] = (s - sum_k h_k x_c(v_k)[row]) / h_{j,j-1}: lane k holds term k
    double t = (lane < j) ? sh[lane] * rh[r] : 0.0;
    t = wave_sum63(t);
    sr[r] = (sr[r] - t) * hinv;
  }
  if (lane == 63) {
#pragma unroll
    for (int r = 0; r < UC_ROWS; ++r)
      if (row0 + r < nv) { d.xc[row0 + r] = sr[r]; d.rch[(size_t)j * lda + row0 + r] = sr[r]; }
  }
  NSK_STAMP(7);
}

template <int N>
struct UcPatch { static constexpr int M = N - 2, PS = (((M + 4) * (M + 4) + 3) / 4) * 4; };   // patch stride at two overlap layers
template <int N>
__device__ __forceinline__ void uc_schwarz_role(const Dev& d, int j, double scale, int min_iter, int ord, const unsigned bx_, const unsigned gx_) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NM = N * M;
  constexpr int MAXP = (M + 8) * (M + 8);
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double sP[4 * EPB * MM], sB[4 * EPB * NM];
  __shared__ double sr[EPB * MAXP];
  __shared__ double sh[MAXMR + 2], scs[MAXMR], ssn[MAXMR], scol[MAXMR + 2], sbc[4];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const int bid = d.boff + (int)xcd_element(bx_, gx_);
  const long long e = (long long)bid * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  GmresScal* G = d.gsc;
  constexpr int PS = UcPatch<N>::PS;            // = d.p_stride (checked on the host: two overlap layers)
  const int jj = j - 1;
  int i0 = -1, i1 = -1;
  {                                            // (unconditional loads from clamped addresses, then a select: no branch per load)
    const long long es = act ? e : 0;
    const int a0 = d.p_idx[es * PS + (nd < PS ? nd : 0)], a1 = d.p_idx[es * PS + (nd + NN < PS ? nd + NN : 0)];
    if (act && nd < PS) i0 = a0;
    if (act && nd + NN < PS) i1 = a1;
  }
  NSK_STAMP(1);
  const bool start = d.uc_start != 0;          // (j = 0 only) the solve starts here: g' raw in Wr, |g'|^2 in row 0 of the partials
  const int nrow = (j > 0) ? jj + 2 : (start ? 1 : 0);
  PartialRows<2, true> pr;                    // rows w, w + 4 of the partials of B_{j-1} (d.nblk <= 512: the merged range)
  pr.issue(d.gpart, d.nblk, d.nblk <= 512 ? nrow : 0, tid);
  double gj = 0.0;
  if (j > 0) {
    if (tid < jj) { scs[tid] = G->cs[tid]; ssn[tid] = G->sn[tid]; }
    gj = G->gpre[jj];
  }
  const bool pact = act && nd < MM;
  const long long q = e * MM + nd;
  double j12a = 0, d12a = 0;
  if (tid < NM) { j12a = d.J12[tid]; d12a = d.D12[tid]; }
  // raw w (j = 0: V[0], normalised by k_gmres_update(-1)) and the basis at this thread's (up to) two patch nodes
  const double* W = (j > 0 || start) ? d.Wr : d.V;
  const int i0s = i0 >= 0 ? i0 : 0, i1s = i1 >= 0 ? i1 : 0;
  double v0 = W[i0s], v1 = W[i1s];
  if (i0 < 0) v0 = 0.0;
  if (i1 < 0) v1 = 0.0;
  constexpr int VK = 8;
  double vk0[VK], vk1[VK];
#pragma unroll
  for (int k = 0; k < VK; ++k) { vk0[k] = 0.0; vk1[k] = 0.0; }
  if (j > 0) {                                 // basis vectors in groups of four behind ONE uniform branch each (index clamped inside a group)
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int kk = k < j ? k : j - 1; vk0[k] = d.V[(size_t)kk * d.ps + i0s]; vk1[k] = d.V[(size_t)kk * d.ps + i1s]; }
  }
  if (j > 4) {
#pragma unroll
    for (int k = 4; k < 8; ++k) { const int kk = k < j ? k : j - 1; vk0[k] = d.V[(size_t)kk * d.ps + i0s]; vk1[k] = d.V[(size_t)kk * d.ps + i1s]; }
  }
  if (tid < NM) { sJ12[tid] = j12a; sD12[tid] = d12a; }
  NSK_STAMP(2);
  // The patch inverse (PS / 4 float4 per lane, the largest stream of the launch; from the Infinity Cache) in TWO prefetch stages:
  // stamps showed the unroll-8 loop behind the column as four dependent trips, 8.9 of this role's 14.7 us.  Stage 1 goes out
  // when the partial-sum registers are free, stage 2 when the basis registers are: both BEHIND the loads the column waits for
  // (loads return in issue order).
  constexpr int nq = PS / 4;                                    // PS is a multiple of 4
  constexpr int PQ1 = nq < 26 ? (nq + 1) / 2 : 13, PQ2 = nq < 26 ? nq - PQ1 : 12;      // (lx1 = 8: 13 + 12 = all 25)
  const float4* PA = reinterpret_cast<const float4*>(d.p_inv + (size_t)(act ? e : 0) * PS * MM) + (nd < MM ? nd : 0);
  float4 pa1[PQ1], pa2[PQ2];
#pragma unroll
  for (int k4 = 0; k4 < PQ1; ++k4) if (k4 < nq) pa1[k4] = PA[(size_t)k4 * MM];      // (measured: here 14.43, behind the partial sums 14.33 matvecs/s)
  if (nrow > 0) {
    if (d.nblk <= 512) pr.reduce(d.gpart, d.nblk, nrow, sh, tid);               // ends with an LDS barrier
    else sum_partials_multi(d.gpart, d.nblk, nrow, sh, tid, 256);
    NSK_STAMP(3);
    if (tid == 0) { if (j > 0) uc_rotate(d, G, jj, gj, scale, min_iter, ord, false, sh, scs, ssn, scol, sbc); else uc_start_solve(d, G, scale, min_iter, false, sh, sbc); }
    lds_barrier();
    NSK_STAMP(4);
    const double hinv = sbc[0];
    if (sbc[1] != 0.0) return;                // column j-1 closed the solve
#pragma unroll
    for (int k = 0; k < VK; ++k) if (k < j) { v0 -= sh[k] * vk0[k]; v1 -= sh[k] * vk1[k]; }
    for (int k = VK; k < j; ++k) {
      if (i0 >= 0) v0 -= sh[k] * d.V[(size_t)k * d.ps + i0];
      if (i1 >= 0) v1 -= sh[k] * d.V[(size_t)k * d.ps + i1];
    }
    v0 *= hinv; v1 *= hinv;
    if (pact) d.V[(size_t)j * d.ps + q] = v0;  // the first MM patch entries are the element's own nodes (p_idx[e][k] = e MM + k)
  }
  asm volatile("" ::: "memory");              // (stage 2 not before the basis registers are free: 168 registers, three workgroups per CU)
#pragma unroll
  for (int k4 = 0; k4 < PQ2; ++k4) if (PQ1 + k4 < nq) pa2[k4] = PA[(size_t)(PQ1 + k4) * MM];
  double m0 = 0, m1 = 0, m2 = 0, m3 = 0;      // (needed behind the patch solve only)
  if (pact) { m0 = d.w2rx[q]; m1 = d.w2sx[q]; m2 = d.w2ry[q]; m3 = d.w2sy[q]; }
  if (act) {
    if (nd < PS) sr[el * MAXP + nd] = v0;
    if (nd + NN < PS) sr[el * MAXP + nd + NN] = v1;
  }
  lds_barrier();
  NSK_STAMP(5);
  if (pact) {
    const double* r = sr + el * MAXP;
    double z0 = 0, z1 = 0, z2 = 0, z3 = 0;
#pragma unroll
    for (int k4 = 0; k4 < PQ1; ++k4) if (k4 < nq) {
      const float4 a = pa1[k4];
      z0 += (double)a.x * r[4 * k4 + 0];
      z1 += (double)a.y * r[4 * k4 + 1];
      z2 += (double)a.z * r[4 * k4 + 2];
      z3 += (double)a.w * r[4 * k4 + 3];
    }
#pragma unroll
    for (int k4 = 0; k4 < PQ2; ++k4) if (PQ1 + k4 < nq) {
      const float4 a = pa2[k4];
      z0 += (double)a.x * r[4 * (PQ1 + k4) + 0];
      z1 += (double)a.y * r[4 * (PQ1 + k4) + 1];
      z2 += (double)a.z * r[4 * (PQ1 + k4) + 2];
      z3 += (double)a.w * r[4 * (PQ1 + k4) + 3];
    }
#pragma unroll 8
    for (int k4 = PQ1 + PQ2; k4 < nq; ++k4) {                     // (lx1 >= 10: patches of more than 100 dofs)
      const float4 a = PA[(size_t)k4 * MM];
      z0 += (double)a.x * r[4 * k4 + 0];
      z1 += (double)a.y * r[4 * k4 + 1];
      z2 += (double)a.z * r[4 * k4 + 2];
      z3 += (double)a.w * r[4 * k4 + 3];
    }
    const double z = (z0 + z1) + (z2 + z3);  // the Schwarz part of z_j; B_j adds R^T x_c
    NSK_STAMP(6);
    d.Z[(size_t)j * d.npr + q] = z;
    sP[(0 * EPB + el) * MM + nd] = z * m0;
    sP[(1 * EPB + el) * MM + nd] = z * m1;
    sP[(2 * EPB + el) * MM + nd] = z * m2;
    sP[(3 * EPB + el) * MM + nd] = z * m3;
  }
  lds_barrier();
  double gx, gy;
  opgradt_tiles<N, EPB>(sJ12, sD12, sP, sB, act, el, nd, gx, gy);
  if (act) {
    const long long l = e * NN + nd;
    d.yl[l] = gx;
    d.yl[d.cs + l] = gy;
  }
  NSK_STAMP(7);
}
// three workgroups per CU (12 wavefronts): nsw + cgrid = 762 workgroups on config 2 are resident at once on 256 CUs
template <int N, int MAXIT>
__global__ __launch_bounds__(256, 3) void k_schwarz_uc(Dev d, int j, double scale, int min_iter, int ord, unsigned nsw, unsigned cgrid) {
  NSK_STAMP(0);
  if (!d.uc_start && d.gsc->done) return;      // (uc_start: the flag still belongs to the previous solve; this launch resets it)
  if (blockIdx.x < nsw) uc_schwarz_role<N>(d, j, scale, min_iter, ord, blockIdx.x, nsw);
    else uc_coarse_role<MAXIT>(d, j, scale, min_iter, ord, blockIdx.x - nsw);
}

// B_j: see above.  All NVL = 20 columns of Tc (fp32 copy: T32) and the first eight basis vectors live in registers.  Every load
// of the first trip is UNCONDITIONAL (lanes without a node load a valid address and drop the value; basis vectors in groups of
// four behind one uniform branch each): `cond ? load : 0` costs a branch per load, and a conversion inside such a branch a
// full wait per load (seen in the instruction stream of the first cut: twenty dependent trips for the fp32 copy).
constexpr int UC_NVL = 20;
template <int N, bool T32, bool SLIM = false>
__device__ __forceinline__ void divgs_t_body(const Dev& d, int j, const unsigned bx_, const unsigned gx_) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NM = N * M;
  constexpr int NVL = UC_NVL;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[2 * EPB * NN], sA[4 * EPB * NM];
  __shared__ double sdot[(MAXMR + 2) * 4];
  __shared__ double swr[EPB * MM], shat[4 * MM], sxc[EPB * NVL];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const int bid = d.boff + (int)xcd_element(bx_, gx_);
  const long long e = (long long)bid * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  NSK_STAMP(0);
  if (d.gsc->done) return;
  NSK_STAMP(1);
  const bool pact = act && nd < MM;
  // safe addresses for the lanes without an element / a pressure node
  const long long es = act ? e : 0;
  const long long l = es * NN + nd;
  const int nds = nd < MM ? nd : 0;
  const long long q = es * MM + nds;
  // ---- first trip: everything addressable from the thread index
  const int4 tab = d.gs_tab[l];
  const double bi = d.binv[l];
  const int ecs = d.ecslot[es * 4 + (nd & 3)];
  const int iv = d.evl[es * NVL + (nd < NVL ? nd : 0)];
  const int4 ev = reinterpret_cast<const int4*>(d.evert)[es];
  const double mw0 = d.w2rx[q], mw1 = d.w2sx[q], mw2 = d.w2ry[q], mw3 = d.w2sy[q];
  const double zq = d.Z[(size_t)j * d.npr + q];
  float ttf[T32 ? NVL : 1];
  double ttd[T32 ? 1 : NVL];
  if constexpr (!SLIM) {
    if constexpr (T32) {
      const float* T = d.Tc32 + ((size_t)es * NVL) * MM + nds;
#pragma unroll
      for (int s = 0; s < NVL; ++s) ttf[s] = T[(size_t)s * MM];
    } else {
      const double* T = d.Tc + ((size_t)es * NVL) * MM + nds;
#pragma unroll
      for (int s = 0; s < NVL; ++s) ttd[s] = T[(size_t)s * MM];
    }
  }
  double vk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) vk[k] = 0.0;
  if constexpr (!SLIM) {
    {                                           // basis vectors 0..3 (j >= 0 always: V_0 exists)
#pragma unroll
      for (int k = 0; k < 4; ++k) vk[k] = d.V[(size_t)(k <= j ? k : j) * d.ps + q];
    }
    if (j >= 4) {
#pragma unroll
      for (int k = 4; k < 8; ++k) vk[k] = d.V[(size_t)(k <= j ? k : j) * d.ps + q];
    }
  }
  double j12a = 0, d12a = 0;
  if (tid < NM) { j12a = d.J12[tid]; d12a = d.D12[tid]; }
  double hatv[(4 * MM + 255) / 256];
#pragma unroll
  for (int r = 0; r < (4 * MM + 255) / 256; ++r) hatv[r] = d.hat[(tid + r * 256 < 4 * MM) ? tid + r * 256 : 0];
  NSK_STAMP(2);
  // ---- second trip: the neighbours' values of yl, the coarse solution at this element's vertices
  const GsVals g0 = gs_load(d.yl, tab, l), g1 = gs_load(d.yl + d.cs, tab, l);
  const double xcv = d.xc[iv];
  const double x0 = d.xc[ev.x], x1 = d.xc[ev.y], x2 = d.xc[ev.z], x3 = d.xc[ev.w];
  if (tid < NM) { sJ12[tid] = j12a; sD12[tid] = d12a; }
#pragma unroll
  for (int r = 0; r < (4 * MM + 255) / 256; ++r) if (tid + r * 256 < 4 * MM) shat[tid + r * 256] = hatv[r];
  if (act) {
    su[(0 * EPB + el) * NN + nd] = bi * gs_sum(g0, d.yl, d, tab, l);
    su[(1 * EPB + el) * NN + nd] = bi * gs_sum(g1, d.yl + d.cs, d, tab, l);
    if (nd < NVL) sxc[el * NVL + nd] = xcv;
  }
  NSK_STAMP(3);
  lds_barrier();
  NSK_STAMP(4);
  // weak divergence (opdiv_tiles with the metrics already in registers)
  if (act && nd < NM) {
    const int jr = nd / M, a = nd % M;
    const double* u = su + (0 * EPB + el) * NN + jr * N;
    const double* v = su + (1 * EPB + el) * NN + jr * N;
    double a1u = 0, a2u = 0, a1v = 0, a2v = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double dd = sD12[a * N + i], jv = sJ12[a * N + i];
      a1u += dd * u[i]; a2u += jv * u[i];
      a1v += dd * v[i]; a2v += jv * v[i];
    }
    sA[(0 * EPB + el) * NM + nd] = a1u;
    sA[(1 * EPB + el) * NM + nd] = a2u;
    sA[(2 * EPB + el) * NM + nd] = a1v;
    sA[(3 * EPB + el) * NM + nd] = a2v;
  }
  tile_barrier<N * N>();
  double w = 0.0;
  if (pact) {
    const int b = nd / M, a = nd % M;
    double ur = 0, us = 0, vr = 0, vs = 0;
#pragma unroll
    for (int jr = 0; jr < N; ++jr) {
      const double jv = sJ12[b * N + jr], dd = sD12[b * N + jr];
      ur += jv * sA[(0 * EPB + el) * NM + jr * M + a];
      us += dd * sA[(1 * EPB + el) * NM + jr * M + a];
      vr += jv * sA[(2 * EPB + el) * NM + jr * M + a];
      vs += dd * sA[(3 * EPB + el) * NM + jr * M + a];
    }
    w = mw0 * ur + mw1 * us + mw2 * vr + mw3 * vs;
    // + E R^T x_c through its precomputed image
    const double* xl = sxc + el * NVL;
    double wc = 0.0;
    if constexpr (SLIM) {
#pragma unroll 4
      for (int s = 0; s < NVL; ++s) wc += (T32 ? (double)d.Tc32[((size_t)e * NVL + s) * MM + nd] : d.Tc[((size_t)e * NVL + s) * MM + nd]) * xl[s];
    } else {
#pragma unroll
      for (int s = 0; s < NVL; ++s) wc += (T32 ? (double)ttf[s] : ttd[s]) * xl[s];
    }
    w += wc;
    d.Wr[q] = w;
    const double h0 = shat[0 * MM + nd], h1 = shat[1 * MM + nd], h2 = shat[2 * MM + nd], h3 = shat[3 * MM + nd];
    d.Z[(size_t)j * d.npr + q] = zq + (h0 * x0 + h1 * x1 + h2 * x2 + h3 * x3);
    swr[el * MM + nd] = w;
  }
  NSK_STAMP(5);
  lds_barrier();
  if (act && nd < 4) {                 // element-corner restriction of the raw w (the coarse workgroups of A_{j+1} read it)
    double s = 0.0;
#pragma unroll 6
    for (int k = 0; k < MM; ++k) s += shat[nd * MM + k] * swr[el * MM + k];
    d.ec[e * 4 + nd] = s;
    d.ecv[ecs] = s;
  }
  {
    const int lane = tid & 63, wv = tid >> 6;
    if constexpr (!SLIM) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k <= j) {
          double x = pact ? w * vk[k] : 0.0;
          x = wave_sum63(x);
          if (lane == 63) sdot[k * 4 + wv] = x;
        }
      }
    }
#pragma unroll 4
    for (int k = SLIM ? 0 : 8; k <= j; ++k) {
      double x = pact ? w * d.V[(size_t)k * d.ps + q] : 0.0;
      x = wave_sum63(x);
      if (lane == 63) sdot[k * 4 + wv] = x;
    }
    {
      double x = pact ? w * w : 0.0;
      x = wave_sum63(x);
      if (lane == 63) sdot[(j + 1) * 4 + wv] = x;
    }
    lds_barrier();
    if (tid <= j + 1) {
      double t = 0.0;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) t += sdot[tid * 4 + ww];
      d.gpart[(size_t)tid * d.nblk + bid] = t;
    }
  }
  NSK_STAMP(6);
}
template <int N, bool T32>
__global__ __launch_bounds__(256, 2) void k_divgs_t(Dev d, int j) {
  divgs_t_body<N, T32>(d, j, blockIdx.x, gridDim.x);
}

// after GMRES: dp = h2 * sum_i y_i Z_i (+ projected part) ; p = p* + dp ; yl = D^T dp
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_pres_update(Dev d, StepCoef sc) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double sP[4 * EPB * MM], sB[4 * EPB * NM];
  __shared__ double sy[MAXMR], sg[MAXMR], sR[MAXMR * MAXMR];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  const GmresScal* G = d.gsc;
  const int nit = G->nit;
  // this thread's operands first: the preconditioned basis z_k (the first ZPRE; slots >= nit hold finite stale data and are
  // not used), the stored solutions x_k, the extrapolated pressure and the metrics -- independent of the small triangular solve
  constexpr int ZPRE = 12, PXPRE = MAXPROJ;
  __shared__ double spa[MAXPROJ];
  const bool pl = act && nd < MM;
  const long long q = e * MM + nd;
  double zr[ZPRE], pxr[PXPRE], pe = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  {
    const int npre = d.nproj_max < PXPRE ? d.nproj_max : PXPRE;
#pragma unroll
    for (int k = 0; k < ZPRE; ++k) zr[k] = pl ? d.Z[(size_t)k * d.npr + q] : 0.0;
#pragma unroll
    for (int k = 0; k < PXPRE; ++k) pxr[k] = (pl && k < npre) ? d.PX[(size_t)k * d.npr + q] : 0.0;
    if (pl) { pe = d.pext[q]; m0 = d.w2rx[q]; m1 = d.w2sx[q]; m2 = d.w2ry[q]; m3 = d.w2sy[q]; }
  }
  if (d.nproj_max > 0 && tid < MAXPROJ) spa[tid] = G->pa[tid];
  for (int k = tid; k < nit * nit; k += NT) { const int cc = k / nit, rr = k % nit; sR[cc * MAXMR + rr] = G->R[cc * MAXMR + rr]; }
  if (tid < nit) sg[tid] = G->g[tid];
  load_basis<N, EPB>(d, nullptr, nullptr, sJ12, sD12, tid, NT);     // (after the operand loads above: nothing waits on it before them)
  lds_barrier();
  if (tid == 0) {                      // back substitution R y = g (nit <= MAXMR), from LDS
    for (int q = nit - 1; q >= 0; --q) {
      double s = sg[q];
      for (int k = q + 1; k < nit; ++k) s -= sR[k * MAXMR + q] * sy[k];
      sy[q] = s / sR[q * MAXMR + q];
    }
  }
  lds_barrier();
  if (pl) {
    double x = (G->nit_prev > 0) ? d.xacc[q] : 0.0;  // completed GMRES cycles of a restarted solve
#pragma unroll
    for (int k = 0; k < ZPRE; ++k) if (k < nit) x += sy[k] * zr[k];
    for (int k = ZPRE; k < nit; ++k) x += sy[k] * d.Z[(size_t)k * d.npr + q];
    if (d.nproj_max > 0) {
      d.PD[q] = x;                                   // GMRES correction delta
      const int np = G->nproj;
#pragma unroll
      for (int k = 0; k < PXPRE; ++k) if (k < np) x += spa[k] * pxr[k];
      for (int k = PXPRE; k < np; ++k) x += spa[k] * d.PX[(size_t)k * d.npr + q];
    }
    const double dp = sc.h2 * x;
    d.p[q] = pe + dp;
    sP[(0 * EPB + el) * MM + nd] = dp * m0;
    sP[(1 * EPB + el) * MM + nd] = dp * m1;
    sP[(2 * EPB + el) * MM + nd] = dp * m2;
    sP[(3 * EPB + el) * MM + nd] = dp * m3;
  }
  __syncthreads();
  double gx, gy;
  opgradt_tiles<N, EPB>(sJ12, sD12, sP, sB, act, el, nd, gx, gy);
  if (act) {
    const long long l = e * NN + nd;
    d.yl[l] = gx;
    d.yl[d.cs + l] = gy;
  }
  if (blockIdx.x == 0 && tid == 0 && !G->done) atomicAdd((unsigned long long*)&d.stats->unconverged, 1ull);
}

// velocity correction fused with E*dp for the projection space:
//   v = B^-1 mask dssum(yl);  u += v/h2;  E delta = D v / h2 - sum a_i E x_i;  dots (E x_i, delta), (delta, E delta)
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_vel_update_proj(Dev d, StepCoef sc) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[2 * EPB * NN], sA[4 * EPB * NM];
  __shared__ double sred[16];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const int bid = (int)xcd_element(blockIdx.x, gridDim.x);             // (as k_helm: XCD-contiguous runs of element blocks)
  const long long e = (long long)bid * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  const GmresScal* G = d.gsc;
  // E-images of the stored solutions and the GMRES correction at this thread's Gauss node: used after the divergence below,
  // independent of it -- issued first (the first PEPRE vectors)
  constexpr int PEPRE = MAXPROJ;
  __shared__ double spa[MAXPROJ];
  double per[PEPRE], del0 = 0.0;
  {
    const bool pl = act && nd < MM;
    const int npre = d.nproj_max < PEPRE ? d.nproj_max : PEPRE;
#pragma unroll
    for (int k = 0; k < PEPRE; ++k) per[k] = (pl && k < npre) ? d.PEX[(size_t)k * d.npr + e * MM + nd] : 0.0;
    if (pl) del0 = d.PD[e * MM + nd];
    if (tid < MAXPROJ) spa[tid] = G->pa[tid];
  }
  BasisRegs<N> br;
  br.issue(d, tid, false, true);
  if (act) {
    const long long l = e * NN + nd;
    const double bi = d.binv[l];
    const double u0 = d.u[l], u1 = d.u[d.cs + l];                 // loads before the stores below (see k_rhs)
    const double vx = bi * gs_gather(d.yl, d, l);
    const double vy = bi * gs_gather(d.yl + d.cs, d, l);
    d.u[l] = u0 + vx / sc.h2;
    d.u[d.cs + l] = u1 + vy / sc.h2;
    su[(0 * EPB + el) * NN + nd] = vx;
    su[(1 * EPB + el) * NN + nd] = vy;
  }
  if (G->nit == 0) return;                      // nothing new to absorb
  br.commit(nullptr, nullptr, sJ12, sD12, tid);
  __syncthreads();
  const double w = opdiv_tiles<N, EPB>(sJ12, sD12, su, sA, act, el, nd, d, e);
  const bool pact = act && nd < MM;
  const long long q = e * MM + nd;
  const int np = G->nproj;
  double del = 0.0, edel = 0.0;
  if (pact) {
    edel = w / sc.h2;
#pragma unroll
    for (int k = 0; k < PEPRE; ++k) if (k < np) edel -= spa[k] * per[k];
    for (int k = PEPRE; k < np; ++k) edel -= spa[k] * d.PEX[(size_t)k * d.npr + q];
    del = del0;
    d.PED[q] = edel;
  }
  __shared__ double sdot[(MAXPROJ + 1) * 4];
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int k = 0; k < PEPRE; ++k) {
    if (k < np) {
      double t = pact ? del * per[k] : 0.0;
      t = wave_sum63(t);
      if (lane == 63) sdot[k * 4 + wv] = t;
    }
  }
#pragma unroll 4
  for (int k = PEPRE; k < np; ++k) {
    double t = pact ? del * d.PEX[(size_t)k * d.npr + q] : 0.0;
    t = wave_sum63(t);
    if (lane == 63) sdot[k * 4 + wv] = t;
  }
  {
    double t = pact ? del * edel : 0.0;
    t = wave_sum63(t);
    if (lane == 63) sdot[np * 4 + wv] = t;
  }
  lds_barrier();
  if (tid <= np) {
    double t = 0.0;
    for (int ww = 0; ww < NT / 64; ++ww) t += sdot[tid * 4 + ww];
    d.ppart[(size_t)tid * d.nblk + bid] = t;
  }
}

}  // namespace k2

// absorb the newest solution into the E-orthogonal projection space   [UPSTREAM navier4.f gensolnp / updtseth / projh]
// The GMRES correction delta is E-orthogonal to the space by construction, so while the space has room it is APPENDED as a
// new basis vector (after removing what the inexact solve left along the old ones): span = the solutions since the last
// restart, exactly (Fischer 1998, what Nek5000 does).  When the space is full:
//   proj_restart = 1 (default): restart it on the latest TOTAL solution  x_0 = delta + sum_i a_i x_i  (Nek5000's rule);
//   proj_restart = 0 (rounds 1-2): overwrite the oldest slot s with  a_s x_s + delta.  Measured (scripts/dbg_proj.py): that
//     merge perturbs the stored history by the innovation of every step, and the projection then removes 97 % of the
//     right-hand side on the cylinder and 91 % on the closed backward-facing step where the solutions since a restart remove
//     99.99 % / 99.95 % (exact arithmetic on the oracle, same sequences).
__global__ __launch_bounds__(256) void k_proj_update(Dev d) {
  __shared__ double sh[MAXPROJ + 1];
  __shared__ double cf[MAXPROJ];
  const int tid = threadIdx.x;
  GmresScal* G = d.gsc;
  const int nit = G->nit, np = G->nproj, nmax = d.nproj_max, pcnt = G->pcnt;
  // as k_proj_apply: all loads first (this thread's entry of delta, E delta and of the first PRE vectors of the space and of
  // their E-images; partial sums; the scalars), then the arithmetic
  constexpr int PRE = MAXPROJ;
  const long long q0 = (long long)blockIdx.x * 256 + tid, stride = (long long)gridDim.x * 256;
  const bool has = q0 < d.npr;
  const int npre = nmax < PRE ? nmax : PRE;
  double x0 = 0.0, ex0 = 0.0, px[PRE], pex[PRE];
  if (has) { x0 = d.PD[q0]; ex0 = d.PED[q0]; }
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    px[k] = (has && k < npre) ? d.PX[(size_t)k * d.npr + q0] : 0.0;
    pex[k] = (has && k < npre) ? d.PEX[(size_t)k * d.npr + q0] : 0.0;
  }
  const double pnv = (tid < MAXPROJ) ? G->pn[tid] : 1.0;
  const double pav = (tid < MAXPROJ) ? G->pa[tid] : 0.0;
  const bool totals = d.nranks > 1 || d.use_tot;
  PartialRows<5> pr;
  pr.issue(d.ppart, d.nblk, (!totals && d.nblk <= 512) ? nmax + 1 : 0, tid);
  if (nit == 0) return;
  const bool full = np >= nmax;
  const bool restart = full && d.proj_restart;
  const int s = restart ? 0 : (full ? pcnt % nmax : np);          // slot written
  const double as = (!restart && s < np) ? G->pa[s] : 0.0;
  if (totals) { if (tid <= np) sh[tid] = d.ptot[tid]; __syncthreads(); }
  else if (d.nblk <= 512) {
    pr.reduce(d.ppart, d.nblk, np + 1, sh, tid, 0, np + 1 <= 20);
    if (np + 1 > 20) { pr.issue(d.ppart, d.nblk, np + 1, tid, 20); pr.reduce(d.ppart, d.nblk, np + 1, sh, tid, 20); }
  }
  else sum_partials_multi(d.ppart, d.nblk, np + 1, sh, tid, 256);
  // coefficient of x_k in the new vector: -c_k / n_k (what the inexact solve left along x_k), + a_k on a restart
  if (tid < np) cf[tid] = ((!restart && tid == s) ? 0.0 : sh[tid] / pnv) - (restart ? pav : 0.0);
  __syncthreads();
  if (has) {
    double x = x0, ex = ex0, xs = 0.0, exs = 0.0;
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      if (k < np && (restart || k != s)) { x -= cf[k] * px[k]; ex -= cf[k] * pex[k]; }
      if (k == s) { xs = px[k]; exs = pex[k]; }
    }
    for (int k = PRE; k < np; ++k) {
      if (!restart && k == s) continue;
      x -= cf[k] * d.PX[(size_t)k * d.npr + q0];
      ex -= cf[k] * d.PEX[(size_t)k * d.npr + q0];
    }
    if (!restart && s < np) {
      if (s >= PRE) { xs = d.PX[(size_t)s * d.npr + q0]; exs = d.PEX[(size_t)s * d.npr + q0]; }
      x += as * xs;
      ex += as * exs;
    }
    d.PX[(size_t)s * d.npr + q0] = x;
    d.PEX[(size_t)s * d.npr + q0] = ex;
  }
  for (long long q = q0 + stride; q < d.npr; q += stride) {
    double x = d.PD[q], ex = d.PED[q];
    for (int k = 0; k < np; ++k) {
      if (!restart && k == s) continue;
      x -= cf[k] * d.PX[(size_t)k * d.npr + q];
      ex -= cf[k] * d.PEX[(size_t)k * d.npr + q];
    }
    if (!restart && s < np) {
      x += as * d.PX[(size_t)s * d.npr + q];
      ex += as * d.PEX[(size_t)s * d.npr + q];
    }
    d.PX[(size_t)s * d.npr + q] = x;
    d.PEX[(size_t)s * d.npr + q] = ex;
  }
  if (blockIdx.x == 0 && tid == 0) {
    double nn = sh[np];                                   // (delta, E delta)
    for (int k = 0; k < np; ++k) if (restart || k != s) nn -= sh[k] * sh[k] / G->pn[k];
    if (restart) { for (int k = 0; k < np; ++k) nn += G->pa[k] * G->pa[k] * G->pn[k]; }
    else if (s < np) nn += as * as * G->pn[s] + 2.0 * as * sh[s];
    G->st_n = nn; G->st_slot = restart ? -1 : s; G->st_pending = 1;
  }
}

// u^{n+1} = u* + (h2 B)^-1 mask dssum(D^T dp)      [UPSTREAM opbinv]
__global__ void k_vel_update(Dev d, StepCoef sc) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= d.nloc) return;
  const double f = d.binv[l] / sc.h2;
  for (int c = 0; c < d.ndim; ++c) d.u[c * d.cs + l] += f * gs_gather(d.yl + c * d.cs, d, l);
}

// ---------------------------------------------------------------------------
// Krylov vector algebra (core/krylov_subspace.f).  State = [vx | vy | pr].
// ---------------------------------------------------------------------------
// partial dots of f with nq vectors, bm1s-weighted, velocity only  (krylov_inner_product)
// (+ the scalar fields theta_1..nscal behind the pressure, same weights: core/krylov_subspace.f:46-50)
__global__ __launch_bounds__(256) void k_dots(const double* __restrict__ f, const double* const* __restrict__ Q,
                                              int nq, const double* __restrict__ w, long long nloc,
                                              double* __restrict__ part, int nblk, int ndim, long long toff, int nscal) {
  __shared__ double sred[16];
  const int tid = threadIdx.x;
  for (int k = 0; k < nq; ++k) {
    const double* q = Q[k];
    double v[1] = {0.0};
    for (long long l = (long long)blockIdx.x * 256 + tid; l < nloc; l += (long long)nblk * 256) {
      const double ww = w[l];
      double t = f[l] * q[l] + f[nloc + l] * q[nloc + l];
      if (ndim == 3) t += f[2 * nloc + l] * q[2 * nloc + l];
      for (int m = 0; m < nscal; ++m) t += f[toff + m * nloc + l] * q[toff + m * nloc + l];
      v[0] += ww * t;
    }
    block_reduce<1>(v, sred, tid, 256);
    if (tid == 0) part[(size_t)k * nblk + blockIdx.x] = v[0];
  }
}

__global__ void k_reduce_final(const double* __restrict__ part, int nq, int nblk, double* __restrict__ out) {
  __shared__ double sred[16];
  const int tid = threadIdx.x;
  for (int k = 0; k < nq; ++k) {
    double v[1];
    sum_partials<1>(part + (size_t)k * nblk, nblk, v, sred, tid, blockDim.x);
    if (tid == 0) out[k] = v[0];
  }
}

// f -= sum_k h_k Q_k  over the whole state (velocity and pressure), h on device
__global__ void k_project_out(double* __restrict__ f, const double* const* __restrict__ Q, int nq,
                              const double* __restrict__ h, long long n) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  double s = f[l];
  for (int k = 0; k < nq; ++k) s -= h[k] * Q[k][l];
  f[l] = s;
}

// the same, and the coefficient vector is accumulated on the device (pass 0 sets, pass 1 adds): update_hessenberg_matrix
// needs h = h_pass1 + h_pass2, and keeping it here removes the host round trip between the passes
__global__ void k_project_out_acc(double* __restrict__ f, const double* const* __restrict__ Q, int nq,
                                  const double* __restrict__ h, long long n, double* __restrict__ acc, int pass) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0)
    for (int k = threadIdx.x; k < nq; k += blockDim.x) acc[k] = (pass ? acc[k] : 0.0) + h[k];
  if (l >= n) return;
  double s = f[l];
  for (int k = 0; k < nq; ++k) s -= h[k] * Q[k][l];
  f[l] = s;
}

__global__ void k_axpby(double* __restrict__ y, double a, const double* __restrict__ x, double b, long long n) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l < n) y[l] = a * x[l] + b * y[l];
}

// scale by 1/sqrt(*nrm2) read from device memory (krylov_normalize without a host round trip)
__global__ void k_scale_rsqrt(double* __restrict__ y, const double* __restrict__ nrm2, long long n) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l < n) y[l] *= 1.0 / sqrt(*nrm2);
}

// add_noise seed (core/utils.f:344-408) with mth_rand (:457-469): one thread per GLL node.  The chain
// 1e3 sin(1e3 sin(r)) amplifies a last-bit difference in r by 1e6, so the operations follow the host mirror
// (nekstab_amd/seed.py) one by one and fused multiply-adds are off: both then differ by the rounding of sin/cos only.
__global__ void k_seed_rand(const double* __restrict__ xyz, long long nloc, int ndim, int N, double* __restrict__ out) {
#pragma clang fp contract(off)
  // mth_rand (core/utils.f:457-469) in the reference's operation order, every operation rounded once, with the correctly rounded
  // sin / cos of nsk_crtrig.hpp: bit for bit the host mirror nekstab_amd/seed.py (tests/test_kernels_gpu.py)
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nloc) return;
  const int per = (ndim == 3) ? N * N * N : N * N;
  const double ieg = (double)(l / per + 1);                        // LGLEL: global element id, 1-based
  const int nd = (int)(l % per);
  const double ix = (double)(nd % N + 1), iy = (double)((nd / N) % N + 1), iz = (double)(nd / (N * N) + 1);
  const double x = xyz[l], y = xyz[nloc + l], z = (ndim == 3) ? xyz[2 * nloc + l] : 0.0;
  const double FC[3][3] = {{3.0e4, -1.5e3, 0.5e5}, {2.3e4, 2.3e3, -2.0e5}, {2.0e4, 1.0e3, 1.0e5}};
  const double siny = crtrig::sin_cr(y);
  for (int c = 0; c < ndim; ++c) {
    const double t0 = x * siny;
    const double t1 = FC[c][0] * (ieg + t0);
    const double t2 = (FC[c][1] * ix) * iy;
    const double t3 = FC[c][2] * ix;
    double r = (t1 + t2) + t3;
    if (ndim == 3) {
      const double u0 = z * crtrig::sin_cr(r);
      const double u1 = FC[c][0] * (ieg + u0);
      const double u2 = (FC[c][1] * iz) * ix;
      const double u3 = FC[c][2] * iz;
      r = (u1 + u2) + u3;
    }
    r = 1.0e3 * crtrig::sin_cr(r);
    r = 1.0e3 * crtrig::sin_cr(r);
    out[(size_t)c * nloc + l] = crtrig::cos_cr(r);
  }
}
// face averaging of add_noise: pass 0: t = (dssum(q) / mult) / mult ; pass 1: q = mask * dssum(t)   (opdssum, opcolv(vmult), dsavg, bcdirvc)
__global__ void k_seed_avg(Dev d, const double* __restrict__ in, double* __restrict__ out, int pass) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= d.nloc) return;
  for (int c = 0; c < d.ndim; ++c) {
    const double s = gs_gather(in + (size_t)c * d.nloc, d, l);
    out[(size_t)c * d.nloc + l] = pass ? d.mask[l] * s : (s * d.minv[l]) * d.minv[l];
  }
}

// out_c = sum_k Q_k * Z[k][c]  for c < nc   (basis rotation / mode assembly)
__global__ void k_basis_comb(const double* const* __restrict__ Q, int k, const double* __restrict__ Z, int ldz,
                             int c0, int nc, double* const* __restrict__ out, long long n) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  for (int c = 0; c < nc; ++c) {
    double s = 0.0;
    for (int q = 0; q < k; ++q) s += Q[q][l] * Z[(size_t)(c0 + c) * ldz + q];
    out[c][l] = s;
  }
}

// Basis rotation  Q(:,1:k) <- Q(:,1:k) Z  (schur_condensation, core/eigensolvers.f:466-474) on the fp64
// matrix cores: per wavefront  D(16 x 16) += A(16 x 4) B(4 x 16)  with  A = Z^T tile (new vector c x old
// vector q), B = Q^T tile (old vector q x 16 consecutive state entries), so the 16 state entries sit on
// the lanes (coalesced loads and stores) and the accumulator rows are the new vectors.  In place: a wave
// reads all k old values of its 16 entries before it writes any.   v_mfma_f64_16x16x4_f64 layouts:
// A[lane&15][lane>>4], B[lane>>4][lane&15], D: col = lane&15, row = (lane>>4) + 4*reg.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
template <int CT>
__global__ __launch_bounds__(256) void k_basis_gemm_mfma(double* const* __restrict__ Q, int k,
                                                         const double* __restrict__ Z, int ldz, long long n) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long i0 = ((long long)blockIdx.x * 4 + wv) * 16;
  if (i0 >= n) return;
  const int n16 = lane & 15, kk = lane >> 4;
  const long long i = i0 + n16;
  const bool iok = i < n;
  mfma_d4 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) acc[ct] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
  for (int q0 = 0; q0 < k; q0 += 4) {
    const int q = q0 + kk;
    const double b = (q < k && iok) ? Q[q][i] : 0.0;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int c = ct * 16 + n16;
      const double a = (c < k && q < k) ? Z[(size_t)c * ldz + q] : 0.0;      // Z is column-major: Z(q,c)
      acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[ct], 0, 0, 0);
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = ct * 16 + kk + 4 * r;
      if (c < k && iok) Q[c][i] = acc[ct][r];
    }
}

// ---------------------------------------------------------------------------
// element sharding: halo of the gather-scatter and of the Schwarz overlap, rank-level sums
// ---------------------------------------------------------------------------
// One message per peer holding every component: buffer = [peer][component][entry], i.e. halo slot k (seg[k] = {first slot,
// slot count} of its peer) of component c sits at  ncomp * seg.x + c * seg.y + (k - seg.x).
// sendbuf[..] = sum of this rank's copies of shared node k (one entry per (peer, shared global node))
__global__ void k_halo_pack(const double* __restrict__ f, long long cstride, int ncomp, const int* __restrict__ off,
                            const int* __restrict__ idx, const int2* __restrict__ seg, int n, double* __restrict__ sendbuf) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int2 sg = seg[k];
  for (int c = 0; c < ncomp; ++c) {
    double s = 0.0;
    for (int i = off[k]; i < off[k + 1]; ++i) s += f[c * cstride + idx[i]];
    sendbuf[(size_t)ncomp * sg.x + (size_t)c * sg.y + (k - sg.x)] = s;
  }
}
// ghost slots of every component <- received partial sums
__global__ void k_halo_unpack(double* __restrict__ f, long long cstride, long long nloc, int ncomp, const int2* __restrict__ seg, int n,
                              const double* __restrict__ recvbuf) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int2 sg = seg[k];
  for (int c = 0; c < ncomp; ++c) f[c * cstride + nloc + k] = recvbuf[(size_t)ncomp * sg.x + (size_t)c * sg.y + (k - sg.x)];
}
// pressure-vector halo (Schwarz overlap into neighbouring ranks' elements): plain gather / copy
__global__ void k_phalo_pack(const double* __restrict__ v, const int* __restrict__ idx, int n, double* __restrict__ sendbuf) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) sendbuf[k] = v[idx[k]];
}
// one workgroup per row of per-workgroup partials: tot[q] = sum_k part[q][k]  (fixed order)
__global__ __launch_bounds__(256) void k_tot2(const double* __restrict__ part, int nblk, double* __restrict__ tot, const int* gate) {
  __shared__ double sred[16];
  const int q = blockIdx.x, tid = threadIdx.x;
  if (gate && *gate) return;                     // GMRES rows of a solve that has converged
  double v[1] = {0.0};
  const double* row = part + (size_t)q * nblk;
  int k = tid;
  for (; k + 7 * 256 < nblk; k += 8 * 256) {          // eight loads in flight, added in the same order as a plain loop
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = row[k + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[0] += a[u];
  }
  for (; k < nblk; k += 256) v[0] += row[k];
  block_reduce<1>(v, sred, tid, 256);
  if (tid == 0) tot[q] = v[0];
}
// loop-back all-reduce for virtual ranks living in one process: every buffer <- sum of all (rank order)
struct LoopPack { double* p[16]; };
// Do all ranks hold the SAME bits after an all-reduce?  (Device-side convergence flags and launch budgets assume it; a
// collective whose reduction order depends on the rank would break that silently.)  k_ar_chunks splits the first n <= 8
// doubles of the result into four 16-bit chunks c (exact small integers as doubles) and appends c^2; the sums of both over
// the ranks are exact in any order, so EVERY rank evaluates  W sum c^2 - (sum c)^2  identically: zero iff all ranks agree.
__global__ void k_ar_chunks(const double* __restrict__ x, int n, double* __restrict__ out) {
  const int t = threadIdx.x;
  if (t >= 4 * n) return;
  const unsigned long long b = (unsigned long long)__double_as_longlong(x[t >> 2]);
  const double c = (double)((b >> (16 * (t & 3))) & 0xffffull);
  out[t] = c;
  out[4 * n + t] = c * c;
}
__global__ void k_ar_check(const double* __restrict__ sums, int n, int world, Stats* st) {
  const int t = threadIdx.x;
  bool bad = false;
  if (t < 4 * n) { const double s = sums[t], s2 = sums[4 * n + t]; bad = ((double)world * s2 - s * s) != 0.0; }
  if (__syncthreads_or(bad ? 1 : 0) && t == 0) st->allred_mismatch += 1;
}
__global__ void k_loop_allreduce_pack(LoopPack pk, int nr, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double s = 0.0;
  for (int r = 0; r < nr; ++r) s += pk.p[r][k];
  for (int r = 0; r < nr; ++r) pk.p[r][k] = s;
}
// gather whole elements out of an element-major array (shard creation)
template <class T>
__global__ void k_slice_elems(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ elems, int nel, int per) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)nel * per) return;
  dst[t] = src[(size_t)elems[t / per] * per + t % per];
}

namespace k2 {
// local axhelm for tests
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_axhelm_test(Dev d, const double* __restrict__ u, double h1,
                                                            double h2, double* __restrict__ out) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, EPB = C::EPB, NT = C::NT;
  __shared__ double sD[NN], sDt[NN];
  __shared__ double sz[EPB * NN], st1[EPB * NN], st2[EPB * NN];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  const long long l = e * NN + nd;
  load_basis<N, EPB>(d, sD, sDt, nullptr, nullptr, tid, NT);
  double z = 0, g1 = 0, g2 = 0, g4 = 0;
  if (act) { z = u[l]; g1 = d.g1[l]; g2 = d.g2[l]; g4 = d.g4[l]; sz[el * NN + nd] = z; }
  __syncthreads();
  double au[1];
  axhelm_tiles<N, EPB, 1>(sD, sDt, sz, st1, st2, act, el, nd / N, nd % N, g1, g2, g4, au);
  if (act) out[l] = h1 * au[0] + h2 * d.bm1[l] * z;
}

}  // namespace k2

__global__ void k_dssum_test(Dev d, const double* __restrict__ u, double* __restrict__ out) {
  const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (l < d.nloc) out[l] = gs_gather(u, d, l);
}

namespace k2 {
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_opdiv_test(Dev d, const double* __restrict__ u,
                                                           double* __restrict__ out) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  __shared__ double sJ12[NM], sD12[NM];
  __shared__ double su[2 * EPB * NN], sA[4 * EPB * NM];
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  load_basis<N, EPB>(d, nullptr, nullptr, sJ12, sD12, tid, NT);
  if (act) {
    su[(0 * EPB + el) * NN + nd] = u[e * NN + nd];
    su[(1 * EPB + el) * NN + nd] = u[d.cs + e * NN + nd];
  }
  __syncthreads();
  const double w = opdiv_tiles<N, EPB>(sJ12, sD12, su, sA, act, el, nd, d, e);
  if (act && nd < MM) out[e * MM + nd] = w;
}

}  // namespace k2

// copy probe responses into the block-sparse E storage (setup)
__global__ void k_collect_probe(const double* __restrict__ w, const int* __restrict__ col_el, int ncol,
                                const int* __restrict__ nb_off, const int* __restrict__ nb_idx,
                                double* __restrict__ Eblk, const long long* __restrict__ blk_off, int MM, int k) {
  // grid.x over (element of this colour), block threads over (slot, row)
  const int ce = blockIdx.x;
  if (ce >= ncol) return;
  const int e = col_el[ce];
  const int n0 = nb_off[e], nn = nb_off[e + 1] - n0;
  for (int t = threadIdx.x; t < nn * MM; t += blockDim.x) {
    const int s = t / MM, r = t % MM;
    const int b = nb_idx[n0 + s];
    Eblk[blk_off[e] + ((size_t)s * MM + k) * MM + r] = w[(size_t)b * MM + r];
  }
}

__global__ void k_set_probe(double* __restrict__ v, const int* __restrict__ col_el, int ncol, int MM, int k, double val) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ncol) v[(size_t)col_el[t] * MM + k] = val;
}

}  // namespace nsk
