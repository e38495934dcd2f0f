/* CPU PORT OF THE HOT PATH -- TEST / BASELINE INFRASTRUCTURE ONLY (oracle/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (nekstab_amd + libnekstab_hip.so) never does.
 *
 * What it is: the linearised PnPn-2 time step of nekStab's matvec (core/matvec.f:163-243: load q,
 * nsteps x nek_advance in perturbation mode, read f) restated in C + OpenMP with the *same iterative
 * algorithms* as the HIP path -- Jacobi-preconditioned CG for the velocity Helmholtz systems
 * ([UPSTREAM hmholtz.f cggo]), right-preconditioned GMRES on E = D B^-1 D^T with restricted additive
 * Schwarz patches + a vertex coarse space ([UPSTREAM navier1.f uzawa_gmres]; the preconditioner is
 * this build's own, see nekstab_amd/csrc/nsk_kernels.hpp), extrapolated Helmholtz initial guess --
 * so that `bench.py` can time whole matvecs on the host cores next to the GPU ("kind": "port").
 * The discretisation follows oracle/linns.py line by line; set-up data (geometry,
 * patch inverses, coarse inverse) come from oracle/cpu_port.py, which builds them from the oracle's
 * assembled operators.  Direct map, domains with an outflow boundary (configs 1 and 2).
 * Pressure projection space ([UPSTREAM navier4.f setrhsp / gensolnp], Fischer 1998) as the HIP path builds it
 * (nsk_kernels.hpp: k_proj_apply / k_proj_update): E-orthogonal solutions since the last restart, appended while there is
 * room, restarted on the latest total solution when full; kept from matvec to matvec (cpu_proj_reset empties it).
 *
 * Threading: one `omp parallel for` over elements per phase, static schedule; dssum is a gather over
 * the CSR of co-located nodes (no atomics).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define MAXN 12
#define MAXND 18
#define MAXMR 48
#define MAXPROJ 32

typedef struct {
  int nel, N, M, ND, nvert, PS, max_helm, max_pres, min_pres, tol_relative, helm_guess, nproj;   /* nproj: size of the pressure projection space (0 = none) */
  const double *D, *J12, *D12, *Jd, *Dd, *hat;
  const double *g1, *g2, *g4, *bm1, *mask, *minv, *binv, *spng, *dinv;
  const double *w2rx, *w2sx, *w2ry, *w2sy;
  const double *cUr, *cUs, *GUx, *GUy, *GVx, *GVy;
  const int *gs_off, *gs_idx;
  const int *p_idx;
  const float *p_inv;          /* [nel][MM][PS] restricted patch inverse (own rows), fp32 like the device copy */
  const float *Aci;            /* [nvert][nvert] coarse inverse, fp32 like the device copy */
  const int *evert, *v_off, *v_ent;
  double nu, dt, vol, tol_helm, tol_pres, early_pres_mul;
} cpu_case;

typedef struct { long long steps, helm_iters, pres_iters, unconverged; double last_helm_res, last_pres_res; long long last_pres_iters; } cpu_stats;

static const double BD[3][4] = {{1.0, 1.0, 0.0, 0.0}, {1.5, 2.0, -0.5, 0.0}, {11.0 / 6.0, 3.0, -1.5, 1.0 / 3.0}};
static const double AB[3][3] = {{1.0, 0.0, 0.0}, {2.0, -1.0, 0.0}, {3.0, -3.0, 1.0}};
static const double XG[4][3] = {{0, 0, 0}, {1, 0, 0}, {2, -1, 0}, {3, -3, 1}};

void cpu_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
/* how many OpenMP parallel regions a run opened (bench.py: regions per time step x the measured cost of an empty region = the
   fork-join floor of this port at a given thread count: why more than ~32 threads do not pay on 127 744 points per field) */
static long long g_regions = 0;
long long cpu_regions(void) { return g_regions; }
void cpu_regions_reset(void) { g_regions = 0; }
/* microseconds per EMPTY parallel-for region (static schedule, n iterations of one double store) at the current thread count */
double cpu_empty_region_us(int n, int reps) {
  static double sink[4096];
  if (n > 4096) n = 4096;
  for (int r = 0; r < 10; ++r) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) sink[i] = (double)i;
  }
  const double t0 = omp_get_wtime();
  for (int r = 0; r < reps; ++r) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) sink[i] = (double)(i + r);
  }
  return 1e6 * (omp_get_wtime() - t0) / reps;
}
int cpu_max_threads(void) { return omp_get_max_threads(); }

/* out = dssum(f): sum over co-located local nodes, ascending (same order as the device gather) */
static void dssum(const cpu_case* c, const double* f, double* out, long long nloc) {
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long l = 0; l < nloc; ++l) {
    double s = 0.0;
    for (int k = c->gs_off[l]; k < c->gs_off[l + 1]; ++k) s += f[c->gs_idx[k]];
    out[l] = s;
  }
}

/* w = D^T G D u (one element, no h1/h2)   [UPSTREAM hmholtz.f axhelm] */
static inline void axhelm_e(const cpu_case* c, long long e, const double* u, double* w) {
  const int N = c->N, NN = N * N;
  const double *D = c->D, *g1 = c->g1 + e * NN, *g2 = c->g2 + e * NN, *g4 = c->g4 + e * NN;
  double t1[MAXN * MAXN], t2[MAXN * MAXN];
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {
      double ur = 0, us = 0;
      for (int k = 0; k < N; ++k) { ur += D[i * N + k] * u[j * N + k]; us += D[j * N + k] * u[k * N + i]; }
      t1[j * N + i] = g1[j * N + i] * ur + g4[j * N + i] * us;
      t2[j * N + i] = g2[j * N + i] * us + g4[j * N + i] * ur;
    }
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {
      double s = 0;
      for (int k = 0; k < N; ++k) s += D[k * N + i] * t1[j * N + k] + D[k * N + j] * t2[k * N + i];
      w[j * N + i] = s;
    }
}

/* weak divergence GLL -> Gauss of (u,v) on one element   [UPSTREAM navier1.f opdiv/multd] */
static inline void opdiv_e(const cpu_case* c, long long e, const double* u, const double* v, double* out) {
  const int N = c->N, M = c->M, MM = M * M;
  const double *J = c->J12, *D12 = c->D12;
  double a1u[MAXN * MAXN], a2u[MAXN * MAXN], a1v[MAXN * MAXN], a2v[MAXN * MAXN];   /* [j][a] */
  for (int j = 0; j < N; ++j)
    for (int a = 0; a < M; ++a) {
      double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
      for (int i = 0; i < N; ++i) {
        s1 += D12[a * N + i] * u[j * N + i]; s2 += J[a * N + i] * u[j * N + i];
        s3 += D12[a * N + i] * v[j * N + i]; s4 += J[a * N + i] * v[j * N + i];
      }
      a1u[j * M + a] = s1; a2u[j * M + a] = s2; a1v[j * M + a] = s3; a2v[j * M + a] = s4;
    }
  const double *rx = c->w2rx + e * MM, *sx = c->w2sx + e * MM, *ry = c->w2ry + e * MM, *sy = c->w2sy + e * MM;
  for (int b = 0; b < M; ++b)
    for (int a = 0; a < M; ++a) {
      double ur = 0, us = 0, vr = 0, vs = 0;
      for (int j = 0; j < N; ++j) {
        ur += J[b * N + j] * a1u[j * M + a]; us += D12[b * N + j] * a2u[j * M + a];
        vr += J[b * N + j] * a1v[j * M + a]; vs += D12[b * N + j] * a2v[j * M + a];
      }
      const int q = b * M + a;
      out[q] = rx[q] * ur + sx[q] * us + ry[q] * vr + sy[q] * vs;
    }
}

/* D^T p on one element   [UPSTREAM navier1.f opgradt/cdtp] */
static inline void opgradt_e(const cpu_case* c, long long e, const double* p, double* gx, double* gy) {
  const int N = c->N, M = c->M, MM = M * M;
  const double *J = c->J12, *D12 = c->D12;
  const double *rx = c->w2rx + e * MM, *sx = c->w2sx + e * MM, *ry = c->w2ry + e * MM, *sy = c->w2sy + e * MM;
  double b1x[MAXN * MAXN], b2x[MAXN * MAXN], b1y[MAXN * MAXN], b2y[MAXN * MAXN];   /* [b][i] */
  for (int b = 0; b < M; ++b)
    for (int i = 0; i < N; ++i) {
      double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
      for (int a = 0; a < M; ++a) {
        const double pv = p[b * M + a];
        s1 += pv * rx[b * M + a] * D12[a * N + i]; s2 += pv * sx[b * M + a] * J[a * N + i];
        s3 += pv * ry[b * M + a] * D12[a * N + i]; s4 += pv * sy[b * M + a] * J[a * N + i];
      }
      b1x[b * N + i] = s1; b2x[b * N + i] = s2; b1y[b * N + i] = s3; b2y[b * N + i] = s4;
    }
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {
      double sx_ = 0, sy_ = 0;
      for (int b = 0; b < M; ++b) {
        sx_ += J[b * N + j] * b1x[b * N + i] + D12[b * N + j] * b2x[b * N + i];
        sy_ += J[b * N + j] * b1y[b * N + i] + D12[b * N + j] * b2y[b * N + i];
      }
      gx[j * N + i] = sx_; gy[j * N + i] = sy_;
    }
}

/* sponge forcing + dealiased direct convection, mass weighted: bf = -(B spng u' + J^T[w (U.grad u' + u'.grad U)])
 * (nekStab_forcing, core/utils.f:172-177; [UPSTREAM perturb.f advabp, convect.f convect_new]) */
static inline void convect_e(const cpu_case* c, long long e, const double* u, const double* v, double* bx, double* by) {
  const int N = c->N, ND = c->ND, NN = N * N, NDD = ND * ND;
  const double *Jd = c->Jd, *Dd = c->Dd;
  double t0[MAXN * MAXND], t1[MAXN * MAXND], f0[MAXND * MAXND], f1[MAXND * MAXND], o0[MAXND * MAXND], o1[MAXND * MAXND];
  for (int j = 0; j < N; ++j)
    for (int a = 0; a < ND; ++a) {
      double s0 = 0, s1 = 0;
      for (int i = 0; i < N; ++i) { s0 += Jd[a * N + i] * u[j * N + i]; s1 += Jd[a * N + i] * v[j * N + i]; }
      t0[j * ND + a] = s0; t1[j * ND + a] = s1;
    }
  for (int b = 0; b < ND; ++b)
    for (int a = 0; a < ND; ++a) {
      double s0 = 0, s1 = 0;
      for (int j = 0; j < N; ++j) { s0 += Jd[b * N + j] * t0[j * ND + a]; s1 += Jd[b * N + j] * t1[j * ND + a]; }
      f0[b * ND + a] = s0; f1[b * ND + a] = s1;
    }
  const double *cUr = c->cUr + e * NDD, *cUs = c->cUs + e * NDD, *GUx = c->GUx + e * NDD, *GUy = c->GUy + e * NDD,
               *GVx = c->GVx + e * NDD, *GVy = c->GVy + e * NDD;
  for (int b = 0; b < ND; ++b)
    for (int a = 0; a < ND; ++a) {
      double ur = 0, us = 0, vr = 0, vs = 0;
      for (int k = 0; k < ND; ++k) {
        ur += Dd[a * ND + k] * f0[b * ND + k]; us += Dd[b * ND + k] * f0[k * ND + a];
        vr += Dd[a * ND + k] * f1[b * ND + k]; vs += Dd[b * ND + k] * f1[k * ND + a];
      }
      const int q = b * ND + a;
      o0[q] = cUr[q] * ur + cUs[q] * us + f0[q] * GUx[q] + f1[q] * GUy[q];
      o1[q] = cUr[q] * vr + cUs[q] * vs + f0[q] * GVx[q] + f1[q] * GVy[q];
    }
  for (int b = 0; b < ND; ++b)
    for (int i = 0; i < N; ++i) {
      double s0 = 0, s1 = 0;
      for (int a = 0; a < ND; ++a) { s0 += Jd[a * N + i] * o0[b * ND + a]; s1 += Jd[a * N + i] * o1[b * ND + a]; }
      t0[b * N + i] = s0; t1[b * N + i] = s1;      /* reuse: [b][i], ND*N <= MAXN*MAXND */
    }
  const double *bm = c->bm1 + e * NN, *sp = c->spng + e * NN;
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {
      double s0 = 0, s1 = 0;
      for (int b = 0; b < ND; ++b) { s0 += Jd[b * N + j] * t0[b * N + i]; s1 += Jd[b * N + j] * t1[b * N + i]; }
      const int l = j * N + i;
      bx[l] = -(sp[l] * bm[l] * u[l] + s0);
      by[l] = -(sp[l] * bm[l] * v[l] + s1);
    }
}

typedef struct {
  long long nloc, npr;
  double *u, *p, *plag, *pext, *ulag, *exlag, *bf, *rloc, *bloc, *dulag;   /* velocity arrays [2][nloc] (lags [2 lags][2][nloc]) */
  double *hx, *hr, *hp, *hz, *hw, *tmp;                                   /* CG work [2][nloc] */
  double *V, *Z, *yl, *vv, *ec, *rc, *xc, *wp, *pd, *ped;                 /* GMRES; projection space: correction and its E-image */
} work_t;

static double* dz(size_t n) { return (double*)calloc(n > 0 ? n : 1, sizeof(double)); }

/* Jacobi-PCG on H du = mask dssum(rloc) for both components (separately converged)  [UPSTREAM hmholtz.f cggo] */
static int helm_solve(const cpu_case* c, work_t* w, int k, double h2, cpu_stats* st) {
  const int N = c->N, NN = N * N, nel = c->nel;
  const long long nl = w->nloc;
  const double* di = c->dinv + (size_t)(k - 1) * nl;
  int worst = 0;
  dssum(c, w->rloc, w->hr, nl); dssum(c, w->rloc + nl, w->hr + nl, nl);
  dssum(c, w->bloc, w->tmp, nl); dssum(c, w->bloc + nl, w->tmp + nl, nl);
  for (int cc = 0; cc < 2; ++cc) {
    double *r = w->hr + cc * nl, *x = w->hx + cc * nl, *p = w->hp + cc * nl, *z = w->hz + cc * nl, *q = w->hw + cc * nl, *bb = w->tmp + cc * nl;
    double rr = 0, ref = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : rr, ref)
    for (long long l = 0; l < nl; ++l) {
      r[l] *= c->mask[l]; x[l] = 0.0; p[l] = 0.0;
      const double b = bb[l] * c->mask[l];
      rr += r[l] * r[l] * c->minv[l]; ref += b * b * c->minv[l];
    }
    ref = sqrt(ref / c->vol);
    const double tol = c->tol_relative ? c->tol_helm * ref : c->tol_helm;
    double rz_old = 0.0;
    int it = 0;
    for (; it < c->max_helm; ++it) {
      if (sqrt(rr / c->vol) <= tol) break;
      double rz = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : rz)
      for (long long l = 0; l < nl; ++l) { z[l] = di[l] * r[l]; rz += r[l] * z[l] * c->minv[l]; }
      if (!(rz > 0.0)) break;
      const double beta = (it == 0) ? 0.0 : rz / rz_old;
      rz_old = rz;
g_regions++;
#pragma omp parallel for schedule(static)
      for (long long e = 0; e < nel; ++e) {
        double* pe = p + e * NN;
        for (int l = 0; l < NN; ++l) pe[l] = z[e * NN + l] + beta * pe[l];
        double a[MAXN * MAXN];
        axhelm_e(c, e, pe, a);
        for (int l = 0; l < NN; ++l) w->yl[e * NN + l] = c->nu * a[l] + h2 * c->bm1[e * NN + l] * pe[l];
      }
      dssum(c, w->yl, q, nl);
      double pq = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : pq)
      for (long long l = 0; l < nl; ++l) { q[l] *= c->mask[l]; pq += p[l] * q[l] * c->minv[l]; }
      const double alpha = rz / pq;
      rr = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : rr)
      for (long long l = 0; l < nl; ++l) { x[l] += alpha * p[l]; r[l] -= alpha * q[l]; rr += r[l] * r[l] * c->minv[l]; }
    }
    if (sqrt(rr / c->vol) > tol) st->unconverged++;
    st->last_helm_res = sqrt(rr / c->vol);
    if (it > worst) worst = it;
  }
  st->helm_iters += worst;
  return 0;
}

/* z = RAS(v) + R^T Aci R v   (preconditioner of the pressure GMRES) */
static void precond(const cpu_case* c, work_t* w, const double* v, double* z) {
  const int M = c->M, MM = M * M, nel = c->nel, PS = c->PS, nv = c->nvert;
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long e = 0; e < nel; ++e)
    for (int cn = 0; cn < 4; ++cn) {
      double s = 0;
      for (int k = 0; k < MM; ++k) s += c->hat[cn * MM + k] * v[e * MM + k];
      w->ec[e * 4 + cn] = s;
    }
g_regions++;
#pragma omp parallel for schedule(static)
  for (int vtx = 0; vtx < nv; ++vtx) {
    double s = 0;
    for (int k = c->v_off[vtx]; k < c->v_off[vtx + 1]; ++k) s += w->ec[c->v_ent[k]];
    w->rc[vtx] = s;
  }
g_regions++;
#pragma omp parallel for schedule(static)
  for (int r = 0; r < nv; ++r) {
    const float* a = c->Aci + (size_t)r * nv;
    double s = 0;
    for (int q = 0; q < nv; ++q) s += (double)a[q] * w->rc[q];
    w->xc[r] = s;
  }
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long e = 0; e < nel; ++e) {
    double rloc[4 * MAXN * MAXN];
    const int* idx = c->p_idx + (size_t)e * PS;
    for (int k = 0; k < PS; ++k) rloc[k] = idx[k] >= 0 ? v[idx[k]] : 0.0;
    const float* A = c->p_inv + (size_t)e * MM * PS;
    const int* ev = c->evert + e * 4;
    for (int r = 0; r < MM; ++r) {
      double s = 0;
      for (int k = 0; k < PS; ++k) s += (double)A[(size_t)r * PS + k] * rloc[k];
      s += c->hat[0 * MM + r] * w->xc[ev[0]] + c->hat[1 * MM + r] * w->xc[ev[1]] + c->hat[2 * MM + r] * w->xc[ev[2]] + c->hat[3 * MM + r] * w->xc[ev[3]];
      z[e * MM + r] = s;
    }
  }
}

/* wout = D B^-1 mask dssum(D^T z) */
static void eapply(const cpu_case* c, work_t* w, const double* z, double* wout) {
  const int N = c->N, NN = N * N, MM = c->M * c->M, nel = c->nel;
  const long long nl = w->nloc;
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long e = 0; e < nel; ++e) opgradt_e(c, e, z + e * MM, w->yl + e * NN, w->yl + nl + e * NN);
  dssum(c, w->yl, w->vv, nl); dssum(c, w->yl + nl, w->vv + nl, nl);
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long e = 0; e < nel; ++e) {
    double a[MAXN * MAXN], b[MAXN * MAXN];
    for (int l = 0; l < NN; ++l) { a[l] = c->binv[e * NN + l] * w->vv[e * NN + l]; b[l] = c->binv[e * NN + l] * w->vv[nl + e * NN + l]; }
    opdiv_e(c, e, a, b, wout + e * MM);
  }
}

/* right-preconditioned GMRES for E0 y = g (E0 = D B^-1 D^T), g in V[0]; returns y in w->wp   [UPSTREAM navier1.f uzawa_gmres] */
static int pres_solve(const cpu_case* c, work_t* w, double h2, double tol_mul, double gnorm0, cpu_stats* st) {
  const long long np = w->npr;
  double H[(MAXMR + 1) * MAXMR], cs[MAXMR], sn[MAXMR], g[MAXMR + 1], y[MAXMR];
  const double scale = 1.0 / (h2 * sqrt(c->vol));
  double* V = w->V;
  double b2 = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : b2)
  for (long long q = 0; q < np; ++q) b2 += V[q] * V[q];
  const double beta0 = sqrt(b2);
  double tp = c->tol_pres;
  if (c->tol_relative) { tp = c->tol_pres * tol_mul; const double lo = c->tol_pres < 1e-4 ? c->tol_pres : 1e-4; if (tp < lo) tp = lo; }
  const double tol = c->tol_relative ? tp * (gnorm0 >= 0.0 ? gnorm0 : beta0) * scale : c->tol_pres;    /* relative to |g| BEFORE the projection */
  memset(w->wp, 0, np * sizeof(double));
  st->last_pres_iters = 0;
  if (!(beta0 > 0.0) || (c->min_pres <= 0 && beta0 * scale <= tol)) { st->last_pres_res = beta0 * scale; return 0; }
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long q = 0; q < np; ++q) V[q] /= beta0;
  g[0] = beta0;
  int j = 0, conv = 0;
  for (; j < c->max_pres; ++j) {
    double* vj = V + (size_t)j * np, *zj = w->Z + (size_t)j * np, *wn = V + (size_t)(j + 1) * np;
    precond(c, w, vj, zj);
    eapply(c, w, zj, wn);
    for (int i = 0; i <= j; ++i) {                       /* modified Gram-Schmidt */
      const double* vi = V + (size_t)i * np;
      double h = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : h)
      for (long long q = 0; q < np; ++q) h += wn[q] * vi[q];
g_regions++;
#pragma omp parallel for schedule(static)
      for (long long q = 0; q < np; ++q) wn[q] -= h * vi[q];
      H[i * MAXMR + j] = h;
    }
    double hn = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : hn)
    for (long long q = 0; q < np; ++q) hn += wn[q] * wn[q];
    hn = sqrt(hn);
    if (hn > 0.0) {
g_regions++;
#pragma omp parallel for schedule(static)
      for (long long q = 0; q < np; ++q) wn[q] /= hn;
    }
    H[(j + 1) * MAXMR + j] = hn;
    for (int i = 0; i < j; ++i) {
      const double t = cs[i] * H[i * MAXMR + j] + sn[i] * H[(i + 1) * MAXMR + j];
      H[(i + 1) * MAXMR + j] = -sn[i] * H[i * MAXMR + j] + cs[i] * H[(i + 1) * MAXMR + j];
      H[i * MAXMR + j] = t;
    }
    const double rho = hypot(H[j * MAXMR + j], H[(j + 1) * MAXMR + j]);
    cs[j] = rho > 0 ? H[j * MAXMR + j] / rho : 1.0; sn[j] = rho > 0 ? H[(j + 1) * MAXMR + j] / rho : 0.0;
    H[j * MAXMR + j] = rho;
    g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j];
    const double res = fabs(g[j + 1]) * scale;
    st->last_pres_res = res;
    if ((res <= tol && j + 1 >= c->min_pres) || !(hn > 0.0)) { conv = 1; ++j; break; }
  }
  if (!conv) st->unconverged++;
  const int nit = j;
  st->pres_iters += nit; st->last_pres_iters = nit;
  for (int q = nit - 1; q >= 0; --q) {
    double s = g[q];
    for (int k = q + 1; k < nit; ++k) s -= H[q * MAXMR + k] * y[k];
    y[q] = s / H[q * MAXMR + q];
  }
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long q = 0; q < np; ++q) {
    double s = 0;
    for (int k = 0; k < nit; ++k) s += y[k] * w->Z[(size_t)k * np + q];
    w->wp[q] = s;
  }
  return 0;
}

/* ---- pressure projection space: X_i, E X_i, n_i = (X_i, E X_i); persists across cpu_matvec calls like the device's ---- */
typedef struct { int n, pcnt, cap; long long np; double *X, *EX; double nn[MAXPROJ], a[MAXPROJ]; } proj_t;
static proj_t g_proj;
void cpu_proj_reset(void) { free(g_proj.X); free(g_proj.EX); memset(&g_proj, 0, sizeof(g_proj)); }
static void proj_ensure(int cap, long long np) {
  if (g_proj.cap == cap && g_proj.np == np && g_proj.X) return;
  cpu_proj_reset();
  g_proj.cap = cap; g_proj.np = np;
  g_proj.X = (double*)calloc((size_t)cap * np, sizeof(double)); g_proj.EX = (double*)calloc((size_t)cap * np, sizeof(double));
}
/* g' = g - sum_i a_i E x_i,  a_i = (x_i, g) / n_i ; returns |g| */
static double proj_apply(double* g, long long np) {
  proj_t* P = &g_proj;
  double gg = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : gg)
  for (long long q = 0; q < np; ++q) gg += g[q] * g[q];
  for (int i = 0; i < P->n; ++i) {
    const double* x = P->X + (size_t)i * np;
    double s = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (long long q = 0; q < np; ++q) s += x[q] * g[q];
    P->a[i] = s / P->nn[i];
  }
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long q = 0; q < np; ++q) {
    double t = g[q];
    for (int i = 0; i < P->n; ++i) t -= P->a[i] * P->EX[(size_t)i * np + q];
    g[q] = t;
  }
  return sqrt(gg);
}
/* absorb the GMRES correction delta (E-image edel) : nsk_kernels.hpp k_proj_update, proj_restart = 1 */
static void proj_update(const double* delta, const double* edel, long long np) {
  proj_t* P = &g_proj;
  const int n = P->n, full = n >= P->cap, s = full ? 0 : n;
  double c[MAXPROJ], cf[MAXPROJ], dd = 0;
  for (int k = 0; k < n; ++k) {
    const double* ex = P->EX + (size_t)k * np;
    double t = 0;
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : t)
    for (long long q = 0; q < np; ++q) t += delta[q] * ex[q];
    c[k] = t;
  }
g_regions++;
#pragma omp parallel for schedule(static) reduction(+ : dd)
  for (long long q = 0; q < np; ++q) dd += delta[q] * edel[q];
  double nn = dd;
  for (int k = 0; k < n; ++k) { nn -= c[k] * c[k] / P->nn[k]; cf[k] = c[k] / P->nn[k] - (full ? P->a[k] : 0.0); }
  if (full) for (int k = 0; k < n; ++k) nn += P->a[k] * P->a[k] * P->nn[k];
  if (!(nn > 0.0)) { P->n = 0; P->pcnt = 0; return; }     /* degenerate direction: drop the space */
  double* xs = P->X + (size_t)s * np; double* exs = P->EX + (size_t)s * np;
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long q = 0; q < np; ++q) {
    double x = delta[q], ex = edel[q];
    for (int k = 0; k < n; ++k) { x -= cf[k] * P->X[(size_t)k * np + q]; ex -= cf[k] * P->EX[(size_t)k * np + q]; }
    xs[q] = x; exs[q] = ex;                                  /* (slot 0 on a restart: every thread has read its own entry of it above) */
  }
  if (full) { P->nn[0] = nn; P->n = 1; P->pcnt = 1; }
  else { P->nn[s] = nn; P->n = n + 1; P->pcnt += 1; }
}

static int step(const cpu_case* c, work_t* w, int istep, cpu_stats* st) {
  const int N = c->N, NN = N * N, M = c->M, MM = M * M, nel = c->nel;
  const long long nl = w->nloc;
  const int k = istep < 3 ? istep : 3;
  const double* bd = BD[k - 1]; const double* ab = AB[k - 1];
  const double h2 = bd[0] / c->dt, invdt = 1.0 / c->dt;
  const double* xg = XG[c->helm_guess ? (istep < 4 ? istep : 4) - 1 : 0];
  /* makefp + makextp + makebdfp + lagfieldp + extrapprp + cresvipp  (oracle/linns.py step()) */
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long e = 0; e < nel; ++e) {
    double bx[MAXN * MAXN], by[MAXN * MAXN], ug[2][MAXN * MAXN], au[MAXN * MAXN], gx[MAXN * MAXN], gy[MAXN * MAXN], pe[MAXN * MAXN];
    convect_e(c, e, w->u + e * NN, w->u + nl + e * NN, bx, by);
    for (int q = 0; q < MM; ++q) {
      const long long qq = e * MM + q;
      const double pn = w->p[qq];
      pe[q] = (k < 3) ? pn : 2.0 * pn - w->plag[qq];
      w->plag[qq] = pn; w->pext[qq] = pe[q];
    }
    opgradt_e(c, e, pe, gx, gy);
    for (int cc = 0; cc < 2; ++cc) {
      const double* bn = cc ? by : bx; const double* gg = cc ? gy : gx;
      for (int l = 0; l < NN; ++l) {
        const long long lc = cc * nl + e * NN + l;
        const double un = w->u[lc];
        const double e1 = w->exlag[lc], e2 = w->exlag[2 * nl + lc];
        double b = ab[0] * bn[l] + ab[1] * e1 + ab[2] * e2;
        w->exlag[2 * nl + lc] = e1; w->exlag[lc] = bn[l];
        const double l1 = w->ulag[lc], l2 = w->ulag[2 * nl + lc];
        b += c->bm1[e * NN + l] * (bd[1] * un + bd[2] * l1 + bd[3] * l2) * invdt;
        w->ulag[2 * nl + lc] = l1; w->ulag[lc] = un;
        ug[cc][l] = un + xg[0] * w->dulag[lc] + xg[1] * w->dulag[2 * nl + lc] + xg[2] * w->dulag[4 * nl + lc];
        w->bloc[lc] = b + gg[l];
      }
      axhelm_e(c, e, ug[cc], au);
      for (int l = 0; l < NN; ++l) {
        const long long lc = cc * nl + e * NN + l;
        w->rloc[lc] = w->bloc[lc] - (c->nu * au[l] + h2 * c->bm1[e * NN + l] * ug[cc][l]);
      }
    }
  }
  helm_solve(c, w, k, h2, st);
  /* u* = u + du0 + dx ; g = -D u*   (incomprp) */
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long e = 0; e < nel; ++e) {
    double us[2][MAXN * MAXN], dv[MAXN * MAXN];
    for (int cc = 0; cc < 2; ++cc)
      for (int l = 0; l < NN; ++l) {
        const long long lc = cc * nl + e * NN + l;
        const double l1 = w->dulag[lc], l2 = w->dulag[2 * nl + lc], l3 = w->dulag[4 * nl + lc];
        const double du = xg[0] * l1 + xg[1] * l2 + xg[2] * l3 + w->hx[lc];
        w->dulag[4 * nl + lc] = l2; w->dulag[2 * nl + lc] = l1; w->dulag[lc] = du;
        us[cc][l] = w->u[lc] + du;
        w->u[lc] = us[cc][l];
      }
    opdiv_e(c, e, us[0], us[1], dv);
    for (int q = 0; q < MM; ++q) w->V[e * MM + q] = -dv[q];
  }
  double gnorm0 = -1.0;
  const int useproj = c->nproj > 0;
  if (useproj) { proj_ensure(c->nproj < MAXPROJ ? c->nproj : MAXPROJ, w->npr); gnorm0 = proj_apply(w->V, w->npr); }
  pres_solve(c, w, h2, istep <= 3 ? c->early_pres_mul : 1.0, gnorm0, st);
  if (useproj) {                      /* delta = GMRES correction (kept in w->pd); total solution = delta + sum a_i x_i */
    memcpy(w->pd, w->wp, w->npr * sizeof(double));
g_regions++;
#pragma omp parallel for schedule(static)
    for (long long q = 0; q < w->npr; ++q) {
      double t = w->wp[q];
      for (int i = 0; i < g_proj.n; ++i) t += g_proj.a[i] * g_proj.X[(size_t)i * w->npr + q];
      w->wp[q] = t;
    }
  }
  /* p = p* + h2 y ; u += (h2 B)^-1 mask dssum(D^T dp) */
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long e = 0; e < nel; ++e) {
    double dp[MAXN * MAXN];
    for (int q = 0; q < MM; ++q) { dp[q] = h2 * w->wp[e * MM + q]; w->p[e * MM + q] = w->pext[e * MM + q] + dp[q]; }
    opgradt_e(c, e, dp, w->yl + e * NN, w->yl + nl + e * NN);
  }
  dssum(c, w->yl, w->vv, nl); dssum(c, w->yl + nl, w->vv + nl, nl);
  if (useproj) {                      /* E delta = D B^-1 dssum(D^T dp) / h2 - sum a_i E x_i, from the arrays of the velocity correction */
g_regions++;
#pragma omp parallel for schedule(static)
    for (long long e = 0; e < nel; ++e) {
      double a[MAXN * MAXN], b[MAXN * MAXN], dv[MAXN * MAXN];
      for (int l = 0; l < NN; ++l) { a[l] = c->binv[e * NN + l] * w->vv[e * NN + l]; b[l] = c->binv[e * NN + l] * w->vv[nl + e * NN + l]; }
      opdiv_e(c, e, a, b, dv);
      for (int q = 0; q < MM; ++q) {
        double t = dv[q] / h2;
        for (int i = 0; i < g_proj.n; ++i) t -= g_proj.a[i] * g_proj.EX[(size_t)i * w->npr + e * MM + q];
        w->ped[e * MM + q] = t;
      }
    }
  }
g_regions++;
#pragma omp parallel for schedule(static)
  for (long long l = 0; l < nl; ++l) {
    const double f = c->binv[l] / h2;
    w->u[l] += f * w->vv[l]; w->u[nl + l] += f * w->vv[nl + l];
  }
  if (useproj && st->last_pres_iters > 0) proj_update(w->pd, w->ped, w->npr);
  st->steps++;
  return 0;
}

/* f = Phi_T q, direct map: q, f = [vx | vy | pr]   (core/matvec.f:163-243) */
int cpu_matvec(const cpu_case* c, const double* q, double* f, int nsteps, cpu_stats* st) {
  if (!c || !q || !f || c->N > MAXN || c->ND > MAXND || c->PS > 4 * MAXN * MAXN || c->max_pres > MAXMR) return -1;
  work_t w;
  const long long nl = (long long)c->nel * c->N * c->N, np = (long long)c->nel * c->M * c->M;
  w.nloc = nl; w.npr = np;
  w.u = dz(2 * nl); w.p = dz(np); w.plag = dz(np); w.pext = dz(np); w.ulag = dz(4 * nl); w.exlag = dz(4 * nl); w.bf = dz(2 * nl);
  w.rloc = dz(2 * nl); w.bloc = dz(2 * nl); w.dulag = dz(6 * nl); w.hx = dz(2 * nl); w.hr = dz(2 * nl); w.hp = dz(2 * nl);
  w.hz = dz(2 * nl); w.hw = dz(2 * nl); w.tmp = dz(2 * nl); w.V = dz((size_t)(MAXMR + 1) * np); w.Z = dz((size_t)MAXMR * np);
  w.yl = dz(2 * nl); w.vv = dz(2 * nl); w.ec = dz((size_t)c->nel * 4); w.rc = dz(c->nvert); w.xc = dz(c->nvert); w.wp = dz(np); w.pd = dz(np); w.ped = dz(np);
  memcpy(w.u, q, 2 * nl * sizeof(double)); memcpy(w.p, q + 2 * nl, np * sizeof(double));
  cpu_stats local; memset(&local, 0, sizeof(local));
  for (int istep = 1; istep <= nsteps; ++istep) step(c, &w, istep, &local);
  memcpy(f, w.u, 2 * nl * sizeof(double)); memcpy(f + 2 * nl, w.p, np * sizeof(double));
  if (st) *st = local;
  double* all[] = {w.u, w.p, w.plag, w.pext, w.ulag, w.exlag, w.bf, w.rloc, w.bloc, w.dulag, w.hx, w.hr, w.hp, w.hz, w.hw, w.tmp, w.V, w.Z, w.yl, w.vv, w.ec, w.rc, w.xc, w.wp, w.pd, w.ped};
  for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) free(all[i]);
  return local.unconverged ? 1 : 0;
}
