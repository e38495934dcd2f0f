// Device-side data layout shared by the kernels and the host driver.
// All fields live in HBM, element-major (Nek layout): GLL node l = e*N*N + j*N + i,
// pressure (Gauss) node = e*M*M + b*M + a, dealiasing node = e*ND*ND + b*ND + a.
#pragma once
#include <hip/hip_runtime.h>

namespace nsk {

constexpr int MAXMR = 48;        // GMRES basis cap (Nek: lgmres = 30, SIZE:34)
constexpr int MAXPROJ = 32;      // pressure projection space cap (Nek: mxprev = 20)

struct StepCoef {                // order k = min(istep,3)  [UPSTREAM setordbd/setbd/setabbd]
  double bd[4];
  double ab[3];
  double h2;                     // bd0/dt
  double invdt;
  int k;
  int adjoint;
  double xg[3];                  // Helmholtz initial guess  du0 = xg0*du^{n-1} + xg1*du^{n-2} + xg2*du^{n-3}
  int cls;                       // step_class(istep): one captured graph + budget each
  double pxt;                    // pressure extrapolation p* = p^n + pxt (p^n - p^{n-1}): 0 for k < 3, 1 for k = 3  [UPSTREAM extrapprp]
};

struct GmresScal {               // device-resident small state of one pressure solve
  double beta0;                  // ||g||
  double g[MAXMR + 1];           // rotated rhs
  double cs[MAXMR], sn[MAXMR];
  double R[MAXMR * MAXMR];       // upper triangular, column j at R[j*MAXMR + i]
  double hcol[MAXMR + 2];        // projection coefficients of the current column
  double hinv;                   // 1 / h(j+1,j)
  double resid;                  // current residual estimate, Nek norm
  double y[MAXMR];
  double gnorm0;                 // ||g|| before projection (reference of the relative tolerance)
  double pa[MAXPROJ];            // projection coefficients a_i = (x_i, g)/n_i
  double pn[MAXPROJ];            // n_i = (x_i, E x_i)
  double st_n;                   // staged update of the projection space (committed by the next k_rhs)
  int st_slot, st_pending;
  int pcnt;                      // solutions absorbed so far
  int done;
  int nit;
  int nproj;                     // vectors currently in the projection space
  int nit_prev;                  // iterations of the completed GMRES cycles of this solve (restarts)
  double gpre[MAXMR + 1];        // g[j] BEFORE the rotation of column j (never rewritten: k_update_coarse reads it in every workgroup)
  // lagged second Gram-Schmidt correction (hexahedral GMRES, k_gs_lag): the newest basis vector is stored as
  // w' (first pass applied, not normalised); v = (w' - sum_k pc[k] v_k) * phinv is formed by the next k_gs_lag
  double pc[MAXMR + 2];
  double phinv;
  int pending;
};

// Step classes: one captured hipGraph and one launch budget each.  Time steps 1, 2, 3 differ in BDF/EXT order and
// in how much of the input's divergence they remove; steps 4-6 and 7-16 still carry that transient.  Cutting the
// tail (>= 17) further does not pay: its per-class iteration maxima are the same (measured with 8 classes).
constexpr int NCLS = 6;
__host__ __device__ inline int step_class(int istep) {
  return istep <= 3 ? istep - 1 : (istep <= 6 ? 3 : (istep <= 16 ? 4 : 5));
}

struct Stats {
  long long helm_iters, pres_iters, unconverged, steps;
  long long max_helm, max_pres;
  long long max_helm_k[NCLS], max_pres_k[NCLS];   // per step class (separate graphs, separate budgets)
  double last_helm_res, last_pres_res;
  long long capped_solves;                        // pressure solves ended by `pres_cap` above their tolerance
  double worst_cap_ratio;                         // largest residual / tolerance among them
  long long sync_timeouts;                        // grid barriers of the persistent kernels that gave up (never in a healthy run)
  long long pres_jsum;                            // sum over the GMRES columns of their basis index j (the Gram-Schmidt bytes are proportional to it)
  long long allred_mismatch;                      // verified all-reduces whose result was NOT bit-identical on every rank (sharded runs: NSK_ECOMM)
  long long nonfinite;                            // pressure right-hand sides whose norm was NaN / Inf: the map's input or the solver state was not finite (map_finish: NSK_ENAN + state reset)
};

struct Dev {
  int nel, nblk, nvert;
  int boff;                      // set per launch: first workgroup of this launch (k_helm on shards: boundary elements first, interior behind)
  int gs2;                       // set per launch: the GMRES column of this k_gmres_update had a second Gram-Schmidt pass (k_gmres_reorth)
  int flat_proj;                 // set per launch: the linear combinations / dots over the GMRES and projection bases run in the streaming kernels (k_pres_comb, k_proj_dots)
  int gs_lag;                    // set per launch: the basis vector read here may still carry its pending scale GmresScal::phinv (k_gs_lag)
  long long nloc, npr;
  long long cs, ps;              // component stride of velocity-mesh arrays / stride of the GMRES basis V
                                 // (= nloc, npr on one rank; + ghost slots when elements are sharded)
  long long npr_glob;            // pressure dofs over all ranks
  double nu, dt, vol, tol_helm, tol_pres;
  int tol_relative, max_mr, has_outflow, nproj_max;
  int proj_reset;                // 1: every map starts with an empty pressure projection space
  int proj_restart;              // 1: a full projection space restarts on the latest total solution (Nek5000); 0: merge into the oldest slot
  int pres_cap;                  // > 0: a pressure solve stops after this many GMRES iterations whatever its residual (set per launch)
  double tol_pres_floor;         // relative pressure tolerance never asks for less than this (units of GmresScal::resid); 0 = off
  // bases
  const double *D, *J12, *D12, *Jd, *Dd, *hat;
  // per GLL node
  const double *g1, *g2, *g4, *bm1, *mask, *minv, *binv, *spng, *bm1s, *dinv;
  // per Gauss (pressure) node, premultiplied by the Gauss weights
  const double *w2rx, *w2ry, *w2sx, *w2sy;
  // per dealiasing node (base-flow dependent, constant in time)
  const double *cUr, *cUs, *GUx, *GUy, *GVx, *GVy;
  const double *rxd, *ryd, *sxd, *syd;     // Jacobian-scaled metrics x weights on the dealiasing mesh (nonlinear convection)
  const double *spng_vr;                   // sponge reference field (DNS branch of nekStab_forcing), [2][cs]
  double nl_spng_str;
  int* bstep;                              // time-periodic base flow (Floquet): device step counter and the
  long long bf_stride;                     // stride between the stored per-step base-flow constants (0: steady)
  // gather-scatter (dssum) as a gather: CSR of co-located local nodes, ascending
  const int *gs_off, *gs_idx;
  const int4* gs_tab;
  const int* gs_corner;          // hexahedra: [nel][8 corners][8] co-located local nodes of the element corners (ascending, -1 padded;
                                 // first entry -2: more than 8, use the CSR lists): addressable WITHOUT the gs_tab / gs_off round trips
  // time-stepper state
  double *u, *p, *plag, *pext, *ulag, *exlag, *bf, *rloc, *bloc, *dulag;
  // Helmholtz CG (both components advance together)
  double *hx, *hr, *hp, *hs, *hwl, *hpart, *hscal;
  // pressure GMRES
  double *V, *Z, *yl, *ec, *xc, *gpart;
  double *xacc;                  // solution accumulated over the completed GMRES cycles (restarted solves)
  double *dpw;                   // hexahedra: the pressure increment dp of the step (k_pres_comb -> k_gradt)
  // hexahedra, option "helm_fdm": element-block fast-diagonalisation preconditioner of the velocity solves -- per element and
  // direction the generalised eigenvectors / values of the 1-D pair (stiffness, mass) of its GLL line scaled by the element's
  // length (Dirichlet ends at walls): hfS [nel][3][N*N], hfL [nel][3][N]; hz = z = M^-1 r (assembled), hy = its unassembled
  // element contributions ([3][cs] each)
  int helm_fdm;
  const double *hfS, *hfL;
  const int* hfT;                // [nel] 1: this element takes its fast-diagonalisation block, 0: it is isotropic enough for the Jacobi diagonal
  double *hz, *hy;
  double *rch;                   // [MAXMR][coarse_lda] coarse solutions x_c(v_i) of the GMRES basis vectors (k_update_coarse)
  // element-corner restrictions in VERTEX-major slots (quadrilateral merged coarse solve): ecv[8 v + k] = the value of the k-th
  // element corner at vertex v (the order of vtab; unused slots stay zero), ecslot[4 e + c] = the slot of corner c of element e.
  // The restriction R w of vertex v is then 64 contiguous bytes at an address known from v alone: k_update_coarse, which every
  // workgroup runs over ALL vertices, reads it with coalesced 16-byte loads in its first round trip instead of two dependent
  // rounds of scattered 8-byte gathers through vtab (round 5: 14.9 -> see DESIGN.md section 5.1)
  double *ecv;
  const int *ecslot;
  // Round 6, the merged GMRES iteration in two launches (k_schwarz_uc, k_divgs_t): the coarse part of the preconditioner enters
  // the E-apply through its precomputed image  Tc = D B^-1 dssum D^T R^T  (block sparse: evl[e][nvl] = the vertices of element
  // e and of its node-sharing neighbours, ascending, padded with vertex 0 / zero columns; Tc[e][nvl][MM]), so that the coarse
  // solve can run NEXT TO the Schwarz solves instead of in front of them.  Wr: the raw w = E z_j of the last iteration (the
  // Schwarz workgroups read it across element boundaries, so it cannot be normalised in place).  wraw (set per launch):
  // k_gmres_update takes the raw w from here instead of V[j+1].
  const int* evl;
  const double* Tc;
  const float* Tc32;             // fp32 copy of Tc: taken by solves whose tolerance is far above 1e-7 (tc32, set per launch); E z_j = w_j then
  int tc32;                      // holds to ~1e-8 of the coarse part instead of rounding -- invisible at a relative tolerance >= 1e-5
  int nvl;
  double* Wr;
  const double* wraw;
  int uc_start;                  // set per launch (A_0 only): the solve starts inside k_schwarz_uc (g' raw in Wr, written by k_proj_apply_e)
  GmresScal* gsc;
  // projection onto previous pressure solutions (E-orthonormal)
  double *PX, *PEX, *PD, *PED, *ppart;
  // coarse space
  const int *v_off, *v_ent, *evert, *vtab;
  const double* Aci;
  const float* Acif;
  // restricted additive Schwarz patches
  const int *p_off, *p_idx;
  int p_stride;
  int coarse_lda;
  const float* p_inv;
  const long long* p_invoff;
  Stats* stats;
  unsigned long long* dbg;       // diagnostic stamps (NSK_STAMPS builds)
  // element sharding (one shard per rank): nranks > 1 => the dot-product sums come from the
  // all-reduced totals below instead of the local per-workgroup partials, and the ghost slots
  // behind every velocity-mesh array / GMRES basis vector hold the neighbours' contributions
  int nranks, rank;
  // hexahedral elements (ndim = 3): the remaining G factors, the nine Gauss-mesh metrics [a*3+c][npr] (x Gauss
  // weights), base-flow / metric constants on the dealiasing mesh, element-wise fast-diagonalisation factors
  int ndim;
  long long nfine;
  const double *g3, *g5, *g6, *w2m, *bfc, *mtd, *fdS, *fdL;
  // Arrays that are zero on every node are not loaded (nsk3_setup.inc: structural zeros of the mapping; nsk_set_baseflow: of the
  // base flow).  zmask bit q < 9: metric term q of w2m / mtd; bit 9 + m: g4, g5, g6.  bfmask bit q < 12: constant q of bfc.
  unsigned zmask, bfmask;
  double fd_eps;
  double* gpart2;                // partial sums of the second Gram-Schmidt pass (3-D GMRES)
  // hexahedral meshes have one workgroup per element (10^3-10^5 of them): every set of per-workgroup partials is
  // summed once by k_tot2 into these totals instead of being re-summed by every consumer workgroup (O(nblk^2))
  int use_tot;
  double *gtot2, *ptot;
  double *htot;                  // [2 parities][8]  Helmholtz sums over all ranks
  double *gtot;                  // [MAXMR + 2]      GMRES sums over all ranks
  // per-TIME-STEP iteration record of the running map: stepctr = time steps started (k_rhs), step_iters[2 s] / [2 s + 1] =
  // CG iterations / GMRES iterations of step s (0-based): what the per-step launch budgets follow (nsk.hip: step_budgets)
  int *stepctr, *step_iters;
  int step_cap;
};

// iteration count of the current time step's velocity (which = 0) or pressure (1) solve into the per-step record
__device__ inline void rec_step_iters(const Dev& d, int which, int count) {
  if (!d.step_iters) return;
  const int s = *d.stepctr - 1;
  if (s >= 0 && s < d.step_cap) atomicMax(&d.step_iters[2 * s + which], count);
}

}  // namespace nsk
