// Persistent velocity solve of one time step (quadrilaterals): ONE launch runs
//   K2  makextp + makebdfp + lagfieldp + extrapprp + cresvipp          (was k_rhs)
//   K3  the whole Jacobi-PCG solve of H du = dssum(r), both components   (was one k_helm launch per iteration)
//   K4  u* = u + du, g = -D u*, projection dots                          (was k_pres_rhs)
// with the CG state (x, r, p, s, geometry factors, Jacobi diagonal, gather table) in registers for the whole
// solve and a device-side grid barrier per CG iteration instead of a kernel boundary.  What crosses workgroups per
// iteration is only the unassembled A z of the element-boundary nodes (dssum as a gather from the neighbours' tiles)
// and 8 dot-product partials per workgroup; both are published write-through (sc1 stores) and read after one
// agent-scope acquire (cdna_hip_programming.md, Guideline 16, recipe R1).  The solve ends on the device: no launch is
// spent on iterations that find the solve converged (17 % of a config-2 step with the launch-per-iteration form).
//
// Arithmetic, summation orders and convergence rule are those of k_rhs / k_helm / k_pres_rhs, so the two forms give
// bit-identical fields (tests/test_persistent_gpu.py).  Requires every workgroup of the grid to be resident at once
// (checked on the host against the occupancy query with a margin); every spin is bounded and a time-out is
// reported through Stats::sync_timeouts.
#pragma once
#include "nsk_kernels.hpp"

namespace nsk {

constexpr int SYNC_GROUPS = 8;                 // arrival counters, one 128-B line each, blocks sharded by blockIdx % 8
constexpr int SYNC_WORDS = (2 * SYNC_GROUPS + 2) * 32;      // cnt[8] | top | fail | gen[8], every word on a line of its own
constexpr unsigned SYNC_SPIN_LIMIT = 400000u;  // polls (~1 us each with s_sleep) before a barrier gives up
#ifndef NSK_BARRIER_ACQUIRE
#define NSK_BARRIER_ACQUIRE 1                  // 0 (experiment only): measured WRONG on MI355X at 2 workgroups per CU, the acquire stays
#endif

// Publishing form.  1 (default): the architecturally defined one -- plain stores, every storing wave drains, the workgroup
// meets, ONE lane executes an agent-scope RELEASE fence (buffer_wbl2 sc1) before it arrives; consumers: agent-scope ACQUIRE
// + plain loads.  0: write-through (sc1) stores and no release fence (Guideline 16, R1): faster by ~1 us per barrier, but
// measured on MI355X with two workgroups per CU it is NOT reliable here (scripts/fused_stress.py: a solve that stagnates
// on stale values in 1 of ~10 maps at lx1 = 8), so it is an experiment switch only.
#ifndef NSK_PUBLISH_RELEASE
#define NSK_PUBLISH_RELEASE 1
#endif
#if NSK_PUBLISH_RELEASE
__device__ inline void st_sc1(double* p, double v) { *p = v; }
#else
__device__ inline void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
// Loads of bytes another workgroup wrote in this launch: PLAIN loads behind the barrier's agent-scope acquire.  Measured on
// MI355X (tests/test_persistent_gpu.py): 8-byte sc1 loads (`global_load_dwordx2 sc1`) return stale values here, with or
// without the acquire -- the form is outside the table of MI355X_MICROARCH.md (section visibility) -- while plain loads
// behind the acquire reproduce the launch-per-iteration results bit for bit.
#ifdef NSK_XLD_SC1
__device__ inline double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
__device__ inline double ld_sc1(const double* p) { return *p; }
#endif
__device__ inline GsVals gs_load_x(const double* f, const int4 t, long long l) {
  GsVals v;
  v.a = ld_sc1(f + (t.x >= 0 ? t.x : l));
  v.b = (t.y >= 0) ? ld_sc1(f + t.y) : 0.0;
  v.c = (t.z >= 0) ? ld_sc1(f + t.z) : 0.0;
  v.d = (t.w >= 0) ? ld_sc1(f + t.w) : 0.0;
  return v;
}
__device__ inline double gs_csr_x(const double* f, const Dev& d, long long l) {
  const int o0 = d.gs_off[l], o1 = d.gs_off[l + 1];
  double s = 0.0;
  for (int k0 = o0; k0 < o1; k0 += 8) {
    int id[8];
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) id[q] = (k0 + q < o1) ? d.gs_idx[k0 + q] : -1;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (id[q] >= 0) ? ld_sc1(f + id[q]) : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (id[q] >= 0) s += v[q];
  }
  return s;
}
__device__ inline double gs_sum_x(const GsVals& v, const double* f, const Dev& d, const int4 t, long long l) {
  if (t.x < 0) return gs_csr_x(f, d, l);
  return ((v.a + v.b) + v.c) + v.d;
}

// Grid barrier for a fully resident grid, two levels (blocks grouped by blockIdx % 8: with round-robin placement one
// group = one XCD; only speed depends on that).  Payload stores before it must be sc1 (write-through): every storing
// wave drains them, the workgroup meets, one lane arrives on its group's counter.  The last arriver of a group is the
// group's leader for this epoch: it arrives on the top counter, polls it, and then releases its group through the group's
// generation word; everybody else polls that word only (62 pollers per line instead of 499 on one).  One agent-scope
// acquire per workgroup, after which plain loads may read the other workgroups' bytes.
// `epoch` = 1, 2, ... counts barriers within the launch (all words are zeroed by a memset node before it).
__device__ inline bool grid_barrier(unsigned* sync, unsigned epoch, int nblk, int* s_fail) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (g_dbg & 1) return true;                   // timing ablation only (results are wrong): no grid barrier
  if (threadIdx.x == 0 && !*s_fail) {
#if NSK_PUBLISH_RELEASE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // after the fence, always (ROCm 7.2 can drop the fence's own wait: Guideline 16, pitfall 12)
#endif
    const int g = blockIdx.x % SYNC_GROUPS;
    const unsigned gsize = (unsigned)((nblk - g + SYNC_GROUPS - 1) / SYNC_GROUPS);
    const unsigned ngroups = (unsigned)(nblk < SYNC_GROUPS ? nblk : SYNC_GROUPS);
    unsigned* top = sync + SYNC_GROUPS * 32;
    unsigned* failw = top + 32;
    unsigned* gen = sync + (SYNC_GROUPS + 2 + g) * 32;
    const unsigned old = __hip_atomic_fetch_add(sync + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool leader = (old + 1u == epoch * gsize);
    if (leader) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned* word = leader ? top : gen;
    const unsigned want = leader ? epoch * ngroups : epoch;
    unsigned spins = 0;
    bool bad = false;
    unsigned seen;
    while ((seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > SYNC_SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load(failw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { bad = true; break; }
    }
    if (bad) __hip_atomic_store(failw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // tell everybody; stop waiting for good
    // A leader that timed out still releases its group (below), so a member can leave the poll loop on `gen` without ever
    // having looked at `failw`: every workgroup reloads it once after the loop and treats non-zero as its own failure
    // (otherwise such a member continues on unsynchronised data and the time-out goes unreported).
    if (bad || (!leader && (seen & 0x80000000u)) || __hip_atomic_load(failw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) *s_fail = 1;
    if (leader) __hip_atomic_store(gen, epoch | (bad ? 0x80000000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // release the group (also after a time-out: marked)
#if NSK_BARRIER_ACQUIRE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  __syncthreads();
  return *s_fail == 0;
}

namespace k2 {

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT, 2) void k_helm_fused(Dev d, StepCoef sc, int max_it, unsigned* sync) {
  using C = Cfg<N>;
  constexpr int NN = C::NN, M = C::M, MM = C::MM, EPB = C::EPB, NT = C::NT, NM = N * M;
  __shared__ double sD[NN], sDt[NN], sJ12[NM], sD12[NM];
  __shared__ double sz[2 * EPB * NN], st1[2 * EPB * NN], st2[2 * EPB * NN];
  __shared__ double sP[4 * EPB * MM], sB[4 * EPB * NM];
  __shared__ double sred[8 * 16];
  __shared__ int s_fail;
  const int tid = threadIdx.x, el = tid / NN, nd = tid % NN;
  const long long e = (long long)blockIdx.x * EPB + el;
  const bool act = (el < EPB) && (e < d.nel);
  const int j = nd / N, i = nd % N;
  const long long l = e * NN + nd, nl = d.cs;
  if (tid == 0) s_fail = 0;
  load_basis<N, EPB>(d, sD, sDt, sJ12, sD12, tid, NT);

  // ================= K2 (k_rhs) =================
  if (d.bf_stride && sc.adjoint != 2 && blockIdx.x == 0 && tid == 0) *d.bstep += 1;
  if (d.stepctr && blockIdx.x == 0 && tid == 0) *d.stepctr += 1;
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
      } else {
        G->pcnt = 0; G->nproj = 0;
      }
    }
    if (d.proj_reset && sc.cls == 0) { G->pcnt = 0; G->nproj = 0; }
  }
  int4 tab = make_int4(0, -1, -1, -1);
  double bm = 0, g1 = 0, g2 = 0, g4 = 0, mk = 0, mi = 0, di = 0;
  double un[2] = {0, 0}, du0[2] = {0, 0};
  {
    double u[2] = {0, 0}, bfv[2] = {0, 0};
    if (act) {
      tab = d.gs_tab[l];
      bm = d.bm1[l]; g1 = d.g1[l]; g2 = d.g2[l]; g4 = d.g4[l]; mk = d.mask[l]; mi = d.minv[l];
      di = d.dinv[(size_t)(sc.k - 1) * d.nloc + l];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const long long lc = c * nl + l;
        un[c] = d.u[lc];
        const double l1 = d.dulag[lc], l2 = d.dulag[2 * nl + lc], l3 = d.dulag[4 * nl + lc];
        du0[c] = sc.xg[0] * l1 + sc.xg[1] * l2 + sc.xg[2] * l3;                 // extrapolated guess of the increment
        d.dulag[4 * nl + lc] = l2; d.dulag[2 * nl + lc] = l1;                  // (slot 0 <- du at the end of the solve)
        u[c] = un[c] + sc.xg[0] * l1 + sc.xg[1] * l2 + sc.xg[2] * l3;           // u^n + du0, summed as k_rhs does
        sz[(c * EPB + el) * NN + nd] = u[c];
        const double bn = d.bf[lc];
        const double e1 = d.exlag[lc], e2 = d.exlag[2 * nl + lc];
        double b = sc.ab[0] * bn + sc.ab[1] * e1 + sc.ab[2] * e2;              // makextp
        d.exlag[2 * nl + lc] = e1;
        d.exlag[lc] = bn;
        const double v1 = d.ulag[lc], v2 = d.ulag[2 * nl + lc];
        b += bm * (sc.bd[1] * un[c] + sc.bd[2] * v1 + sc.bd[3] * v2) * sc.invdt;   // makebdfp
        d.ulag[2 * nl + lc] = v1;                                               // lagfieldp
        d.ulag[lc] = un[c];
        bfv[c] = b;
      }
      if (nd < MM) {                                                            // extrapprp
        const long long q = e * MM + nd;
        const double pn = d.p[q];
        const double pe = (sc.pxt == 0.0) ? pn : 2.0 * pn - d.plag[q];
        d.plag[q] = pn;
        d.pext[q] = pe;
        sP[(0 * EPB + el) * MM + nd] = pe * d.w2rx[q];
        sP[(1 * EPB + el) * MM + nd] = pe * d.w2sx[q];
        sP[(2 * EPB + el) * MM + nd] = pe * d.w2ry[q];
        sP[(3 * EPB + el) * MM + nd] = pe * d.w2sy[q];
      }
    }
    __syncthreads();
    double gx, gy;
    opgradt_tiles<N, EPB>(sJ12, sD12, sP, sB, act, el, nd, gx, gy);
    double au[2];
    axhelm_tiles<N, EPB, 2>(sD, sDt, sz, st1, st2, act, el, j, i, g1, g2, g4, au);
    if (act) {
      const double bx = bfv[0] + gx, by = bfv[1] + gy;                          // rhs of H u* = b
      st_sc1(d.bloc + l, bx);
      st_sc1(d.bloc + nl + l, by);
      st_sc1(d.rloc + l, bx - (d.nu * au[0] + sc.h2 * bm * u[0]));
      st_sc1(d.rloc + nl + l, by - (d.nu * au[1] + sc.h2 * bm * u[1]));
    }
  }
  unsigned epoch = 1;
  bool ok = grid_barrier(sync, epoch++, d.nblk, &s_fail);

  // ================= K3 (k_helm, all iterations) =================
  double r[2] = {0, 0}, p[2] = {0, 0}, s[2] = {0, 0}, x[2] = {0, 0};
  double gprev[2] = {0, 0}, aprev[2] = {0, 0}, refn[2] = {0, 0}, resrec[2] = {0, 0};
  bool fin[2] = {false, false};
  int used = 0;
  bool unconv = false;
  for (int it = 0; ok; ++it) {
    const int par = it & 1, ppar = par ^ 1;
    double alpha[2] = {0, 0}, beta[2] = {0, 0};
    bool done[2] = {false, false};
    GsVals gv[2], gb[2];
    if (act) {                                                                  // gathers first: they only need the table
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (it == 0) { gv[c] = gs_load_x(d.rloc + c * nl, tab, l); gb[c] = gs_load_x(d.bloc + c * nl, tab, l); }
        else gv[c] = gs_load_x(d.hwl + ((size_t)ppar * 2 + c) * nl, tab, l);
      }
    }
    if (it > 0) {
      double ps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const double* part = d.hpart + (size_t)ppar * 8 * d.nblk;
      if (!(g_dbg & 2))                          // (timing ablation: no partial-sum loads)
      for (int k = tid; k < d.nblk; k += NT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ps[q] += ld_sc1(part + (size_t)q * d.nblk + k);
      }
      block_reduce<8>(ps, sred, tid, NT);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double g = ps[c * 3 + 0], del = ps[c * 3 + 1], rr = ps[c * 3 + 2];
        const double res = sqrt(rr / d.vol);
        if (it == 1) refn[c] = sqrt(ps[6 + c] / d.vol);                        // ||b||: H u* = b
        const double tol = d.tol_relative ? d.tol_helm * refn[c] : d.tol_helm;
        const bool was = fin[c];
        done[c] = was || (res <= tol) || !(g > 0.0);
        if (!done[c]) {
          if (it == 1) { beta[c] = 0.0; alpha[c] = g / del; }
          else { beta[c] = g / gprev[c]; alpha[c] = g / (del - beta[c] * g / aprev[c]); }
        }
        gprev[c] = g; aprev[c] = alpha[c];
        if (!was) resrec[c] = res;
        fin[c] = done[c];
      }
      if (done[0] && done[1]) { used = it - 1; break; }
      if (it >= max_it) { used = it - 1; unconv = true; break; }
    }
    double zz[2] = {0, 0}, bb[2] = {0, 0};
    if (act) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (it == 0) {
          r[c] = mk * gs_sum_x(gv[c], d.rloc + c * nl, d, tab, l);
          bb[c] = mk * gs_sum_x(gb[c], d.bloc + c * nl, d, tab, l);
        } else if (!done[c]) {
          const double w = mk * gs_sum_x(gv[c], d.hwl + ((size_t)ppar * 2 + c) * nl, d, tab, l);
          const double pn = di * r[c] + beta[c] * p[c];
          const double sn = w + beta[c] * s[c];
          p[c] = pn; s[c] = sn;
          x[c] = x[c] + alpha[c] * pn;
          r[c] = r[c] - alpha[c] * sn;
        }
        zz[c] = di * r[c];
        sz[(c * EPB + el) * NN + nd] = zz[c];
      }
    }
    lds_barrier();
    double au[2];
    axhelm_tiles<N, EPB, 2>(sD, sDt, sz, st1, st2, act, el, j, i, g1, g2, g4, au);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (act) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double wl = d.nu * au[c] + sc.h2 * bm * zz[c];
        st_sc1(d.hwl + ((size_t)par * 2 + c) * nl + l, wl);
        v[c * 3 + 0] = r[c] * zz[c] * mi;
        v[c * 3 + 1] = zz[c] * wl;
        v[c * 3 + 2] = r[c] * r[c] * mi;
        v[6 + c] = bb[c] * bb[c] * mi;
      }
    }
    block_reduce<8>(v, sred, tid, NT);
    if (tid < 8) st_sc1(d.hpart + ((size_t)par * 8 + tid) * d.nblk + blockIdx.x, v[tid]);
    ok = grid_barrier(sync, epoch++, d.nblk, &s_fail);
  }
  // a time-out anywhere in the grid (this workgroup's own, or one it has not looked at since its last barrier) ends the
  // step here: nothing below may overwrite u / dulag / V on unsynchronised data, and EVERY workgroup that sees it counts it
  if (ok && tid == 0 && __hip_atomic_load(sync + (SYNC_GROUPS + 1) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) s_fail = 1;
  __syncthreads();
  ok = ok && (s_fail == 0);
  if (!ok && tid == 0) atomicAdd((unsigned long long*)&d.stats->sync_timeouts, 1ull);
  if (blockIdx.x == 0 && tid == 0) {
    if (ok) {
      atomicAdd((unsigned long long*)&d.stats->helm_iters, (unsigned long long)used);
      atomicMax((unsigned long long*)&d.stats->max_helm, (unsigned long long)used);
      atomicMax((unsigned long long*)&d.stats->max_helm_k[sc.cls], (unsigned long long)used); rec_step_iters(d, 0, used);
      d.stats->last_helm_res = fmax(resrec[0], resrec[1]);
      if (unconv) atomicAdd((unsigned long long*)&d.stats->unconverged, 1ull);
    }
  }
  if (!ok) return;

  // ================= K4 (k_pres_rhs) =================
  if (act) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const long long lc = c * nl + l;
      const double du = du0[c] + x[c];                                          // guess + CG correction
      d.dulag[lc] = du;
      const double us = un[c] + du;
      d.u[lc] = us;
      sz[(c * EPB + el) * NN + nd] = us;
    }
  }
  __syncthreads();
  const double div = opdiv_tiles<N, EPB>(sJ12, sD12, sz, sB, act, el, nd, d, e);
  double v1[1] = {0.0};
  double g = 0.0;
  const bool pact = act && nd < MM;
  if (pact) {
    g = -div;
    d.V[e * MM + nd] = g;
    v1[0] = g * g;
  }
  block_reduce<1>(v1, sred, tid, NT);
  if (tid == 0) d.gpart[blockIdx.x] = v1[0];
  if (!d.has_outflow) {
    double t[1] = {g};
    block_reduce<1>(t, sred, tid, NT);
    if (tid == 0) d.gpart[(size_t)d.nblk + blockIdx.x] = t[0];
  }
  if (d.nproj_max > 0) {
    if (tid == 0) d.ppart[(size_t)MAXPROJ * d.nblk + blockIdx.x] = v1[0];
    const int np = d.gsc->nproj;
    double* sdot = sP;                                                          // [MAXPROJ * 4] <= 4 * EPB * MM doubles
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll 4
    for (int k = 0; k < np; ++k) {
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

// ---------------------------------------------------------------------------
// Persistent TAILS of the two inner solves (round 5).  A step's captured graph cannot know how many iterations a solve will
// take, so rounds 1-4 budgeted launches: every launch beyond a solve's own count still costs its dispatch and one cold flag
// load (4.4 us in the graph-mode trace of config 2; 19-39 % of the kernel time of a map).  Now the graph holds a HEAD of
// launches -- the median count of this time step over the last maps: almost all of them do work -- and ONE persistent launch
// that runs the rest of the solve, however long, as a loop over the SAME kernel bodies with a grid barrier where a kernel
// boundary was: bit-identical arithmetic, no launch for an iteration that is not needed, no budget that can overflow (the loop
// runs to the solver's caps: no redone maps).  A tail iteration costs more than a launched one (a grid barrier with agent-scope
// release / acquire is ~9 us against a ~4.4 us launch floor), so the head takes what is predictable; measured on config 2 the
// tail runs 0.6 iterations per solve on average.  Requirements as k_helm_fused: every workgroup of the grid resident at once
// (checked on the host with a margin), bounded spins, time-outs reported through Stats::sync_timeouts.
// Barrier words: two sets; a tail uses one and zeroes the other for the next persistent launch (velocity tail: set 0, pressure
// tail: set 1: they alternate within every step), so no memset node is spent on them.
// ---------------------------------------------------------------------------
__device__ inline void zero_sync(unsigned* w, int tid, int nt) {
  if (blockIdx.x == 0) for (int k = tid; k < SYNC_WORDS; k += nt) w[k] = 0u;
}

namespace k2 {

// CG iterations it0 .. it_end-1 of the velocity solve (launch indices as k_helm's `it`); leaves the state exactly as the launches
// it0 .. it_end-1 would: k_pres_rhs follows with helm_par = (it_end-1) & 1, check_helm = it_end-1.
template <int N>
__global__ __launch_bounds__(Cfg<N>::NT, 2) void k_helm_tail(Dev d, StepCoef sc, int it0, int it_end, const double* rhs, unsigned* sync, unsigned* sync_other) {
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  if (tid == 0) s_fail = 0;
  zero_sync(sync_other, tid, Cfg<N>::NT);
  __syncthreads();
  if (d.stats->sync_timeouts != 0) return;       // an earlier tail of this map timed out: the map is redone with launch budgets (do not spin again)
  unsigned epoch = 1;
  bool ok = true, fin = false;
  int it = it0;
  for (; it < it_end; ++it) {
    if (it > 1) {                                // both components finished in an earlier launch (flags written by launch it-1)
      const double f0 = __hip_atomic_load(d.hscal + ((it - 1) & 1) * 8 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double f1 = __hip_atomic_load(d.hscal + ((it - 1) & 1) * 8 + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (f0 != 0.0 && f1 != 0.0) { fin = true; break; }
    }
    helm_body<N>(d, sc, it, rhs, blockIdx.x, gridDim.x);
    ok = grid_barrier(sync, epoch++, (int)gridDim.x, &s_fail);
    if (!ok) break;
  }
  if (!ok) { if (tid == 0) atomicAdd((unsigned long long*)&d.stats->sync_timeouts, 1ull); return; }
  if (!fin && it > 1) {                          // the last launch of the loop may have been the one that found the solve finished
    const double f0 = __hip_atomic_load(d.hscal + ((it - 1) & 1) * 8 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double f1 = __hip_atomic_load(d.hscal + ((it - 1) & 1) * 8 + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fin = f0 != 0.0 && f1 != 0.0;
  }
  // finished at launch it-1: what the remaining launches would have done is to hand the flags (and the recorded residuals) on to
  // the other parity; k_pres_rhs reads the parity of launch it_end-1
  if (fin && blockIdx.x == 0 && tid < 8) d.hscal[(it & 1) * 8 + tid] = d.hscal[((it - 1) & 1) * 8 + tid];
}

// merged pressure GMRES iterations j0 .. j1-1 (k_update_coarse, k_schwarz, k_divgs of each); the closing k_gmres_update(j1-1)
// launch follows as behind the launched form.  skip_a: k_update_coarse(j0) has been LAUNCHED in front of this kernel (it closes
// column j0-1: a solve of exactly j0 iterations -- the predicted count -- then ends there and this launch finds nothing to do).
template <int N, int MAXIT>
__global__ __launch_bounds__(Cfg<N>::NT, 2) void k_pres_tail(Dev d, int j0, int j1, double scale, int min_iter, int ord, unsigned cgrid, int skip_a, unsigned* sync, unsigned* sync_other) {
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  if (tid == 0) s_fail = 0;
  zero_sync(sync_other, tid, Cfg<N>::NT);
  __syncthreads();
  if (d.stats->sync_timeouts != 0) return;       // (as k_helm_tail)
  unsigned epoch = 1;
  bool ok = true;
  for (int j = j0; j < j1 && ok; ++j) {
    if (__hip_atomic_load(&d.gsc->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (!(skip_a && j == j0)) {
      if (blockIdx.x < cgrid) update_coarse_body<MAXIT, true>(d, j, scale, min_iter, ord, blockIdx.x, cgrid);
      ok = grid_barrier(sync, epoch++, (int)gridDim.x, &s_fail);
      if (!ok || __hip_atomic_load(&d.gsc->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;      // column j-1 closed the solve
    }
    schwarz_body<N>(d, (const double*)(d.V + (size_t)j * d.ps), d.Z + (size_t)j * d.npr, 1, 1, blockIdx.x, gridDim.x);
    ok = grid_barrier(sync, epoch++, (int)gridDim.x, &s_fail);
    if (!ok) break;
    divgs_body<N>(d, (const double*)d.yl, d.V + (size_t)(j + 1) * d.ps, j, 2, blockIdx.x, gridDim.x);
    ok = grid_barrier(sync, epoch++, (int)gridDim.x, &s_fail);
  }
  if (!ok && tid == 0) atomicAdd((unsigned long long*)&d.stats->sync_timeouts, 1ull);
}

// The same for the two-launch form of round 6 (k_schwarz_uc, k_divgs_t): iterations j0 .. j1-1 as
//   [A_j: coarse role on workgroups < cgrid, then the Schwarz role on all]  barrier  [B_j]  barrier
// two grid barriers per iteration instead of three.  skip_a: A_{j0} has been LAUNCHED in front of this kernel (it closes column
// j0-1: a solve of exactly j0 iterations ends there).  256 threads per workgroup whatever lx1 (the coarse role is written for
// four wavefronts); the coarse role loops over cgrid, so grids smaller than cgrid are covered too.
template <int N, int MAXIT>
__global__ __launch_bounds__(256, 2) void k_pres_tail2(Dev d, int j0, int j1, double scale, int min_iter, int ord, unsigned cgrid, int skip_a, unsigned* sync, unsigned* sync_other) {
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  if (tid == 0) s_fail = 0;
  zero_sync(sync_other, tid, 256);
  __syncthreads();
  if (d.stats->sync_timeouts != 0) return;       // an earlier tail of this map timed out: the map is redone with launch budgets anyway
  unsigned epoch = 1;
  bool ok = true;
  for (int j = j0; j < j1 && ok; ++j) {
    if (__hip_atomic_load(&d.gsc->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (!(skip_a && j == j0)) {
      for (unsigned bx = blockIdx.x; bx < cgrid; bx += gridDim.x) {
        uc_coarse_role<MAXIT>(d, j, scale, min_iter, ord, bx);       // (the SAME role functions as the launched form: same bits)
        __syncthreads();                         // (its LDS is reused by the next pass / the Schwarz role)
      }
      uc_schwarz_role<N>(d, j, scale, min_iter, ord, blockIdx.x, gridDim.x);
      ok = grid_barrier(sync, epoch++, (int)gridDim.x, &s_fail);
      if (!ok || __hip_atomic_load(&d.gsc->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;      // column j-1 closed the solve
    }
    divgs_t_body<N, false>(d, j, blockIdx.x, gridDim.x);             // (the fp64 coarse image whatever option "tc32" says: one body, no spills)
    ok = grid_barrier(sync, epoch++, (int)gridDim.x, &s_fail);
  }
  if (!ok && tid == 0) atomicAdd((unsigned long long*)&d.stats->sync_timeouts, 1ull);
}

}  // namespace k2
}  // namespace nsk
