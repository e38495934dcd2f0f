/* libnekstab_hip.so -- C-ABI of the MI355X-native Arnoldi hot path for nekStab.
 *
 * The reference (nekStab, Fortran on Nek5000) has no FFI layer; its seam is the
 * Fortran call interface  `matvec(f,q)`  ("all subroutines need to have the
 * same interface", core/matvec.f:64-68)  plus the `krylov_*` vector algebra
 * (core/krylov_subspace.f:24-258).  Every entry point below names the
 * reference routine it replaces.  Plain pointers and sizes only; vectors are
 * opaque device-resident handles so the Krylov basis never leaves HBM.
 *
 * Conventions: every function returns 0 on success, a negative nsk_status
 * otherwise (the reference aborts through nek_end/exitt instead);
 * one context per process; calls are serialised by the caller.
 * Host arrays are Nek element-major:  index = i + lx1*(j + lx1*e).
 */
#ifndef NEKSTAB_HIP_H
#define NEKSTAB_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsk_ctx nsk_ctx;
typedef void* nsk_vec;               /* device state vector [vx | vy | pr]  (3-D: [vx | vy | vz | pr]) */

enum nsk_status {
  NSK_OK = 0,
  NSK_EINVAL = -1,      /* bad argument / unsupported lx1 */
  NSK_EHIP = -2,        /* HIP runtime error */
  NSK_ENAN = -3,        /* NaN in an inner product (core/krylov_subspace.f:53 aborts) */
  NSK_ENOCONV = -4,     /* an inner Helmholtz / pressure solve hit its iteration cap */
  NSK_ENOMEM = -5,
  NSK_ECOMM = -6        /* the ranks of a sharded run received DIFFERENT results from one all-reduce (checked bit for bit on the first
                           all-reduces of a context): device-side convergence flags would diverge; the map is refused */
};

enum nsk_mode {          /* `evop`, core/matvec.f:124-151 */
  NSK_DIRECT = 0,        /* 'd'  forward_linearized_map   core/matvec.f:163-243 */
  NSK_ADJOINT = 1,       /* 'a'  adjoint_linearized_map   core/matvec.f:249-326 */
  NSK_DIRECT_ADJOINT = 2,/* 'p'  transient_growth_map     core/matvec.f:332-349 */
  NSK_NEWTON = 3,        /* 'n'  newton_linearized_map    core/matvec.f:381-428 (exp(LT)-I) */
  NSK_FORCE_SENSITIVITY = 4 /*  ts_force_sensitivity_map   core/matvec.f:357-374 (uparam(1) = 4: (I - exp(L^+ T)) q, adjoint map) */
};

/* Case description = what Nek5000 holds in COMMON when nekStab_init runs
 * (core/usr_extra.f:72-132): geometry, numbering, masks, base flow, sponge. */
typedef struct {
  int ndim;                 /* 2: quadrilaterals; 3: hexahedra (arrays [nel*lx1^3], z / wb / 8 vertices per element) */
  int nel;                  /* elements on this rank */
  int lx1;                  /* GLL points per direction: 6, 8, 10 or 12 (SIZE:13) */
  int lxd;                  /* dealiasing points, 3*lx1/2 (SIZE:14) */
  long long nglob;          /* number of distinct global GLL nodes */
  const double* x;          /* [nel*lx1*lx1] GLL coordinates (xm1) */
  const double* y;
  const long long* gid;     /* [nel*lx1*lx1] 0-based global node id (gslib numbering) */
  const double* mask;       /* [nel*lx1*lx1] velocity Dirichlet mask v1mask (1 free / 0 fixed) */
  const double* ub;         /* [nel*lx1*lx1] base flow ubase (core/NEKSTAB:26) */
  const double* vb;
  const double* spng;       /* [nel*lx1*lx1] spng_fun (core/utils.f:283-318) */
  const long long* vert;    /* [nel*4] 0-based vertex ids, lexicographic corners (.ma2) */
  long long nvert;
  double re;                /* Reynolds number; viscosity = 1/re (1cyl.par:37) */
  double endtime;           /* sampling period T = param(10) */
  double cfl;               /* target CFL = param(26) (0.5) */
  int has_outflow;          /* 0: all-Dirichlet/periodic => pressure null space (`ortho`) */
  double tol_helm;          /* [VELOCITY] residualTol, Nek norm (1cyl.par:34) */
  double tol_pres;          /* [PRESSURE] residualTol, Nek norm (1cyl.par:29) */
  int tol_relative;         /* 0: absolute tolerances as Nek (param(21/22)>0); 1: relative to the initial residual */
  int schwarz_layers;       /* overlap (GL-node layers) of the pressure Schwarz patches */
  int max_helm_iter;        /* iteration caps per solve */
  int max_pres_iter;
  int nproj;                /* pressure projection space (residualProj; mxprev, SIZE:33); 0 = off */
  const double* z;          /* ndim = 3: zm1 */
  const double* wb;         /* ndim = 3: third base-flow component (wbase, core/NEKSTAB:26) */
} nsk_case;

/* nekStab_init + prepare_linearized_solver (core/usr_extra.f:72-132, core/matvec.f:1-52):
 * uploads the case, builds operators/preconditioners, derives dt and nsteps. */
int nsk_init(const nsk_case* c, nsk_ctx** out);
int nsk_finalize(nsk_ctx* ctx);
const char* nsk_last_error(void);

/* dt, nsteps (core/matvec.f:30-34), state length S = 2*P + P2, P, P2 */
int nsk_get_info(nsk_ctx* ctx, double* dt, int* nsteps, long long* nstate,
                 long long* nvel, long long* npres);
int nsk_set_nsteps(nsk_ctx* ctx, int nsteps);   /* test hook: shorten the map (dt unchanged) */
int nsk_set_tolerances(nsk_ctx* ctx, double tol_helm, double tol_pres, int relative);
/* run-time switches: "use_graph" (hipGraph replay of the step classes, default 1), "min_pres_iter" (at least this many
 * GMRES iterations per pressure solve, default 0), "pres_cap" (at most this many in time steps >= 4, default 0 = none),
 * "helm_guess" (extrapolated Helmholtz initial guess, default 1), "early_pres_mul" (pressure tolerance factor of
 * time steps 1-3 of every map, default 0.01: they project out the divergence of the input vector),
 * "proj_reset" (1: every map starts with an empty pressure projection space: default on hexahedra; 0: the space carries over: default on quadrilaterals),
 * "pres_floor" (absolute floor under the relative pressure tolerance, in the scaled units of nsk_stats.last_pres_res; 0 = off),
 * "budget_helm" / "budget_pres" (launch budgets), "fused" (persistent velocity solve: right-hand side, all CG iterations and
 * the pressure right-hand side in one launch with device-side grid barriers; default 0: not faster on config 2),
 * "endtime" (the sampling period T = param(10); used by the next nsk_set_baseflow / nsk_set_orbit),
 * "gmres_cycle" (the pressure GMRES restarts after this many iterations, default and maximum 48; `max_pres_iter` may be up to 4 x 48),
 * "mfma_convect" (hexahedra, lx1 = 8: convection contractions on v_mfma_f64_16x16x4_f64, default 1),
 * "merged_update" / "merged_iters" (quadrilaterals with the dense in-LDS coarse solve: the GMRES column bookkeeping runs inside
 * the coarse-solve kernel for the first `merged_iters` iterations of a solve, default 1 / 24; same iteration counts and results to
 * rounding as the classic four-kernel iteration), "shard_graph" / "shard_hostcheck" / "halo_overlap" (shard contexts, see
 * nsk_shard_release_parent and nsk_shard_elems below), "hostcheck" (full-mesh contexts: eager time steps in which the host reads
 * the device's convergence flags and stops issuing solver iterations -- no launch budgets, no redone maps; -1 = default: yes on
 * hexahedral meshes of >= 8192 elements, where a launch that only finds its solve converged costs 25-140 us; 0 / 1; bit-identical),
 * "proj_restart" (1, default: a full projection space restarts on the latest total solution -- Fischer's / Nek5000's rule; 0: the
 * rounds-1-2 policy of merging into the oldest slot, kept for A/B runs), "gs2_from" (quadrilaterals: GMRES columns from this
 * iteration of a cycle on get a second Gram-Schmidt pass, default 48 = never; hexahedra always), "orth_overlap" (RCCL ranks: nsk_orth
 * all-reduces its coefficients in chunks on a second stream while the next chunk's dots are computed, default 1),
 * "gs_lag" (hexahedra: lagged second Gram-Schmidt correction of the pressure GMRES, two basis reads per column instead of four;
 * -1 = default: on for single-rank contexts; 0 / 1 / 2), "flat_proj" (hexahedra: once-per-step sums over the bases as streaming
 * kernels; -1 = default: on for single-rank contexts), "eapply_pipe" (hexahedra, form of the Schwarz + D^T kernel: 0 = one
 * workgroup per element, 1 = resident workgroups, 2 = one wavefront per element, 3 = 2 + the divergence kernel in that form,
 * 4 = default: one wavefront per element, sixteen per CU, at lx1 <= 8 and form 5 at lx1 = 10; 5 = four wavefronts per element,
 * lx1 = 10 only; same results to rounding: DESIGN.md section 4.3, 4.36),
 * "divgs_c3" (hexahedra: divergence kernel with its components side by side; -1 = default: at lx1 = 10, where it runs two
 * workgroups per CU and the plain form one; 0 / 1), "helm_pf" (hexahedra, lx1 = 10: the CG iteration as resident workgroups with
 * the next element's vectors arriving in LDS by LDS-DMA, default 1; bit-identical to 0), "helm_pf_grid" (its resident workgroups;
 * default: what the device holds), "mfma_convect" (hexahedra, lx1 = 8 and 10: convection kernel on the matrix cores, default 1),
 * "helm_fdm" (hexahedra:
 * element-block fast-diagonalisation preconditioner of the velocity solves, default 0: measured slower), "graph_steps" (time steps
 * per captured graph of the last step class, default 1),
 * "budget_freeze" / "budget_add_helm" / "budget_add_pres" (measurement switches of scripts/noop_cost.py),
 * "dbg_max_order" / "dbg_ab2" / "dbg_pext" (time-scheme sensitivity switches of scripts/wake_bisect.py; defaults = SURVEY App. A),
 * "dbg" (developer ablation mask),
 * "step_budgets" (1, default: launch budgets per TIME STEP from the per-step iteration record of the last maps instead of one
 *   per step class; graph-replayed single-rank contexts), "tail" (persistent tail kernels of the two inner solves behind the launches
 *   of a solve: -1 (DEFAULT) = as a safety net behind the per-time-step budgets (mode 2) wherever every workgroup of the grid is
 *   resident, else none; 0 = never; 1 = heads (median counts) + tail; 2 = budgets + safety net; with "tail_off_h" / "tail_off_p"
 *   = offsets of the heads from the median count), "rccl_fuse" (1, default: on RCCL ranks the all-reduce of an iteration rides
 *   in the group of that iteration's halo messages), "allred_verify" (24, default: that many all-reduces of a sharded context
 *   are verified bit for bit across the ranks; a mismatch makes the map return NSK_ECOMM),
 * "fuse2" (1, default; round 6: the merged pressure GMRES iteration in TWO launches -- k_schwarz_uc: Schwarz workgroups and
 *   coarse-solve workgroups side by side, k_divgs_t: the coarse part through its precomputed image under E; 0 = the three
 *   launches of rounds 3-5), "fuse2_start" (1: the solve starts inside the first of them, no k_gmres_update(-1) launch),
 *   "tc32" (0, default; 1: fp32 copy of the coarse image, moves a map by 2e-8),
 * "zero_metrics" (hexahedra; 1, default: arrays of the mapping and of the base-flow constants that are zero on every node -- the
 *   cross terms Nek5000 skips on its undeformed elements, hmholtz.f axhelm / ifdfrm -- are not loaded; 0: every array is loaded,
 *   same bits.  The set-up sets derivatives of the mapping that are rounding noise on every node to zero; NSK_ZERO_METRICS=0 in
 *   the environment of nsk_init keeps the noise, as rounds 1-4 did) */
int nsk_set_option(nsk_ctx* ctx, const char* name, double value);

/* allocate(Q(k_dim+1)) (core/eigensolvers.f:170) */
int nsk_vec_alloc(nsk_ctx* ctx, int n, nsk_vec* out);
int nsk_vec_free(nsk_ctx* ctx, int n, nsk_vec* v);
/* nopcopy / outpost / load_files (core/utils.f:471-550, core/IO.f:15-60) */
int nsk_vec_upload(nsk_ctx* ctx, nsk_vec v, const double* vx, const double* vy, const double* pr);
int nsk_vec_download(nsk_ctx* ctx, nsk_vec v, double* vx, double* vy, double* pr);
/* krylov_vector%theta(:, m) (core/krylov_subspace.f:13): after nsk_set_option(ctx, "nscal", ldimt) -- set it before the first
 * nsk_vec_alloc -- every vector carries `nscal` scalar fields of P entries behind the pressure; the inner product adds their
 * bm1s-weighted products (:46-50), the vector operations scale / add / rotate them, and the time steppers pass them through
 * unchanged, as the reference does when ifheat = .false. (no case in scope solves a scalar equation) */
int nsk_vec_upload_scalar(nsk_ctx* ctx, nsk_vec v, int m, const double* theta);
int nsk_vec_download_scalar(nsk_ctx* ctx, nsk_vec v, int m, double* theta);
/* hexahedral contexts (krylov_vector carries vz, core/krylov_subspace.f:9-11) */
int nsk_vec_upload3(nsk_ctx* ctx, nsk_vec v, const double* vx, const double* vy, const double* vz, const double* pr);
int nsk_vec_download3(nsk_ctx* ctx, nsk_vec v, double* vx, double* vy, double* vz, double* pr);

/* matvec(f,q) (core/matvec.f:64-154) */
int nsk_matvec(nsk_ctx* ctx, int mode, nsk_vec f, nsk_vec q);

/* Newton-Krylov support (core/newton_krylov.f): the full nonlinear map Phi_T(q) [optionally minus q] and
 * the change of linearisation point (base flow := q, new dt / nsteps from the CFL rule). */
int nsk_nonlinear_map(nsk_ctx* ctx, nsk_vec f, nsk_vec q, int subtract_q);
int nsk_set_baseflow(nsk_ctx* ctx, nsk_vec q);
/* Floquet (uparam(1)=3.11): time-periodic base flow over one period T = endtime, stored per step on the
 * device (quadrilaterals: six arrays per step; hexahedra: the twelve dealiasing-mesh constants per step, 12 * nel * lxd^3 doubles);
 * `end` (optional) receives Phi_T(q0) for a periodicity check. */
int nsk_set_orbit(nsk_ctx* ctx, nsk_vec q0, double spng_str, nsk_vec end);

/* krylov_inner_product / norm / cmult / add2,sub2 / copy / zero
 * (core/krylov_subspace.f:24-212) */
int nsk_dot(nsk_ctx* ctx, nsk_vec p, nsk_vec q, double* alpha);
int nsk_norm(nsk_ctx* ctx, nsk_vec p, double* alpha);
int nsk_scal(nsk_ctx* ctx, nsk_vec p, double alpha);
int nsk_axpy(nsk_ctx* ctx, nsk_vec p, double alpha, nsk_vec q);   /* p += alpha*q */
int nsk_copy(nsk_ctx* ctx, nsk_vec dst, nsk_vec src);
int nsk_zero(nsk_ctx* ctx, nsk_vec p);

/* update_hessenberg_matrix (core/krylov_decomposition.f:116-202): two
 * projection passes against Q(1:j) + normalisation; h[0..j-1] = H(1:j,j),
 * *beta = H(j+1,j).  f is overwritten with the new unit Krylov vector. */
int nsk_orth(nsk_ctx* ctx, nsk_vec f, const nsk_vec* Q, int j, double* h, double* beta);

/* Q(:,1:k) <- Q(:,1:k) * Z   (schur_condensation, core/eigensolvers.f:455-474) */
int nsk_basis_gemm(nsk_ctx* ctx, nsk_vec* Q, int k, const double* Z, int ldz);
/* krylov_matmul / eigenmode assembly  re + i im = Q (y_re + i y_im)
 * (core/krylov_subspace.f:214-258, core/eigensolvers.f:607-615) */
int nsk_basis_gemv(nsk_ctx* ctx, const nsk_vec* Q, int k, const double* y_re,
                   const double* y_im, nsk_vec re, nsk_vec im);

/* add_noise seed (core/utils.f:344-408): deterministic pseudo-noise, dssum-averaged, masked */
int nsk_seed_noise(nsk_ctx* ctx, nsk_vec v);

/* ---- lanes: independent maps in flight at once on one GPU ----
 * nsk_clone gives a context a second LANE: its own stream, time-stepper state, solver work arrays and projection space; geometry,
 * operators and preconditioner are shared (quadrilateral full-mesh contexts).  nsk_matvec_batch(lanes, b, mode, f, q) runs
 * f[k] = map(q[k]) on lane k, all b maps queued before any is waited for: a single map is a chain of dependent 5-15 us kernels and
 * leaves the launch pipeline ~40 % idle on BASELINE config 2; two lanes reach ~1.6x the matvecs/s of one.  The reference has no
 * counterpart (one MPI job = one map); users: band Arnoldi (nekstab_amd/krylov.py: band_arnoldi), sweeps.  Vectors allocated on
 * any lane are usable on all of them; finalize clones before the context they were cloned from. */
int nsk_clone(nsk_ctx* ctx, nsk_ctx** lane);
int nsk_matvec_batch(nsk_ctx** lanes, int b, int mode, nsk_vec* f, nsk_vec* q);

/* ---- statistics of the last nsk_matvec (define the algorithmic bytes) ---- */
typedef struct {
  long long steps;
  long long helm_iters;     /* summed over steps (both components advance together) */
  long long pres_iters;
  long long unconverged;    /* solves that hit the cap */
  double last_helm_res, last_pres_res;
  long long max_helm_iter, max_pres_iter;   /* worst single solve */
  long long budget_helm, budget_pres;       /* iterations currently launched per solve */
  long long recaptures, retries;            /* graph re-captures / maps redone with larger budgets (since init) */
  long long capped_solves;                  /* pressure solves ended by "pres_cap" above their tolerance (last matvec) */
  double worst_cap_ratio;                   /* largest residual / tolerance among them */
  long long total_capped_solves;            /* the same two since nsk_init */
  double total_worst_cap_ratio;
  long long total_helm_iters, total_pres_iters, total_steps;   /* since nsk_init: what bytes_per_matvec is computed from */
  double recapture_seconds;                 /* host time spent (re)capturing and instantiating the step graphs since nsk_init */
  long long total_pres_jsum;                /* since nsk_init: sum over all GMRES columns of their basis index j (Gram-Schmidt bytes) */
  double coarse_bytes_per_solve;            /* operator bytes ONE coarse solve reads (dense / block-circulant inverse, or degree x sparse rows) */
  /* per-time-step launch budgets (option "step_budgets"): maps run on them since init and the launches they budgeted per
   * time step on average (velocity: k_helm launches; pressure: GMRES iterations), 0 while none ran */
  long long step_budget_maps;
  double step_budget_helm_mean, step_budget_pres_mean;
  long long tail_maps;                      /* maps whose solves ended in the persistent tail kernels (option "tail"): the per-step numbers above are then HEADS */
  long long zero_arrays;                    /* hexahedra: bit mask of the arrays that are zero on every node and therefore not loaded: bits 0-8 the
                                               nine metric terms, 9-11 the G factors g4 g5 g6 (set-up), 12-23 the twelve base-flow constants of the
                                               convection kernel (nsk_set_baseflow); 0 on quadrilaterals, on deformed meshes, with option zero_metrics = 0 */
} nsk_stats;
int nsk_get_stats(nsk_ctx* ctx, nsk_stats* s);

/* ---- element sharding: one shard per rank, cut out of a full-mesh context --------
 * part[e] = owning rank of global element e.  The parent's set-up (geometry, Jacobi diagonal,
 * Schwarz patches, coarse inverse) is replicated; state, halos and reductions are per rank.
 * Vectors of a shard hold its own elements only ([vx | vy | pr], local element order: nsk_shard_elems).  Ranks that live in one process ("virtual ranks", what the single-GPU tests
 * use) advance in lock-step through nsk_group_matvec with loop-back copies as transport; ranks in
 * separate processes use RCCL (build with -DNSK_WITH_RCCL, nsk_comm_init_rccl). */
int nsk_shard_create(nsk_ctx* parent, const int* part, int rank, int nranks, nsk_ctx** out);
/* Local element order of a shard: its BOUNDARY elements (a node shared with another rank) first, the interior behind, each
 * group by ascending global id; out[le] = global id of local element le (nsk_vec_upload / _download of a shard move element
 * blocks in this order).  Option "halo_overlap" (nsk_set_option on the shard, eager steps): the velocity solve and the
 * pressure iteration launch the boundary workgroups apart from the interior ones and move the halos on a second stream while
 * the interior workgroups run; the all-reduces wait for both (events).  Bit-identical to the serial order of the same shard. */
int nsk_shard_elems(nsk_ctx* shard, long long* out);
/* The exchange plan of a shard, per peer rank p (arrays of nranks entries, 0 where p is no neighbour): vel[p] = global nodes
 * whose partial sums travel in one dssum message to AND from p (one double per node and component); pres_send[p] / pres_recv[p]
 * = pressure dofs of a GMRES basis vector sent to / received from p (the Schwarz patch layers).  Two ranks' plans must agree:
 * vel is symmetric and pres_send[p] on rank r = pres_recv[r] on rank p (tests/test_local_setup_gpu.py checks it on 8 ranks
 * against the host-side derivation nekstab_amd.sharded.velocity_halo_plan). */
int nsk_shard_halo_counts(nsk_ctx* shard, int* vel, int* pres_send, int* pres_recv);
int nsk_group_matvec(nsk_ctx** shards, int n, int mode, nsk_vec* f, nsk_vec* q);    /* every mode of nsk_matvec */
/* The rest of the operator interface on shards, for the ranks living in this process (the reference runs all of it under MPI):
 *   nsk_group_nonlinear_map  nonlinear_forward_map, core/newton_krylov.f:336-378 (subtract_q != 0: Phi_T(q) - q)
 *   nsk_group_set_baseflow   new linearisation point + prepare_linearized_solver: dt / nsteps from the CFL maximum over ALL ranks
 *   nsk_group_set_orbit      time-periodic base flow (uparam(1) = 3.11 / 3.21, core/matvec.f:191-236), stored per rank
 * Singular pressure operators (adjoint runs, closed domains) shard like the others: `ortho` takes its mean over all ranks.
 * Shards keep the parent's pressure projection space size (nsk_case.nproj), so the sharded operator is the single-rank one. */
int nsk_group_nonlinear_map(nsk_ctx** shards, int n, nsk_vec* f, nsk_vec* q, int subtract_q);
int nsk_group_set_baseflow(nsk_ctx** shards, int n, nsk_vec* q);
int nsk_group_set_orbit(nsk_ctx** shards, int n, nsk_vec* q0, double spng_str, nsk_vec* end);
/* Once a process has cut its shard(s): free every device array of the parent that shards do not share (element-major geometry,
 * preconditioner factors, state, work arrays and ALL vectors allocated on the parent -- their handles become invalid).  The 1-D
 * bases and the replicated coarse operator stay.  The parent then only answers nsk_info / nsk_get_stats / nsk_finalize
 * (finalize it after its shards).  Option "shard_graph" (nsk_set_option on a shard): the sharded step runs as one hipGraph per
 * step class; -1 (default) = yes unless an RCCL communicator is attached, 0 = eager, 1 = yes, RCCL calls captured too.
 * Option "shard_hostcheck": eager sharded steps read the device's convergence flags on the host (one 4-128 byte copy and a stream
 * synchronisation, from the iteration where the previous solve of the step class ended) and stop ISSUING iterations -- a launch
 * that finds its solve converged costs 2 us, its halo exchange and all-reduce do not; the flags follow from all-reduced sums, so
 * every rank takes the same decision without talking.  -1 (default) = yes once a transport is attached (RCCL or host-staged),
 * 0 = never (launch budgets, as the captured graphs), 1 = always (also virtual ranks: no graphs then).  Bit-identical maps. */
int nsk_shard_release_parent(nsk_ctx* parent);
/* ---- rank-local set-up (what Nek5000 does by construction: every MPI rank sets up its own elements) ------------
 * Instead of the whole mesh, a rank hands over ITS sub-mesh: the elements it owns plus two rings of node-sharing
 * neighbours, in ascending global element id, with the global node ids (gid), the global vertex ids (vert / nvert) and
 * nglob of the whole mesh; own[e] = 1 for the owned elements.  On the owned elements everything element-local is then
 * exactly what the whole-mesh set-up computes (assembled mass, Jacobi diagonals, Schwarz factors: their stencils end
 * inside the rings).  Global facts travel through the caller (MPI / torch.distributed / a loop over virtual ranks):
 *   nsk_local_info    vol_own (sum), ctarg (max), fd_lmax (max), npr_own (sum), nrows = entries of this rank's coarse rows
 *   nsk_local_rows    the rows of the vertex coarse operator A_c of the vertices this rank owns, as (u, v, a) triplets
 *   nsk_local_finish  the reduced scalars and the triplets of ALL ranks (concatenated): sets dt / nsteps, the Jacobi
 *                     diagonals for that dt and builds the (replicated, vertex-level) coarse solve
 *   nsk_shard_create_local   part_sub[e] = owning rank of sub-mesh element e, elem_glob[e] = its global id
 * The shard behaves exactly like one cut from a whole-mesh parent by nsk_shard_create (tests/test_local_setup_gpu.py holds
 * the two against each other); set-up time and memory scale with the sub-mesh. */
int nsk_init_local(const nsk_case* sub, const int* own, nsk_ctx** out);
int nsk_local_info(nsk_ctx* ctx, double* vol_own, double* ctarg, double* fd_lmax, long long* npr_own, long long* nrows);
int nsk_local_rows(nsk_ctx* ctx, int* u, int* v, double* a);
int nsk_local_finish(nsk_ctx* ctx, double vol, double ctarg, double fd_lmax, long long npr_glob, long long nrows, const int* u,
                     const int* v, const double* a);
int nsk_shard_create_local(nsk_ctx* local_parent, const int* part_sub, const long long* elem_glob, int rank, int nranks, nsk_ctx** out);
/* virtual ranks made from separate rank-local parents: move `shard` onto `leader`'s stream (one process, lock-step) */
int nsk_shard_share_stream(nsk_ctx* shard, nsk_ctx* leader);
/* RCCL transport, one process per GPU: rank 0 creates the id, everybody calls init on its shard;
 * afterwards nsk_group_matvec(&shard, 1, ...) exchanges halos / all-reduces over xGMI. */
/* Host-staged transport for ranks in separate processes WITHOUT RCCL (any host message layer: MPI, torch.distributed gloo;
 * what lets several ranks share one GPU in the tests): the library packs on the device, copies to pinned host buffers and
 * calls back.  exchange: send[k] / recv[k] hold counts[k] doubles for peer peers[k] (both directions have the same count);
 * allreduce: sum buf[0..n) over all ranks in place.  Return 0 on success. */
typedef int (*nsk_exchange_fn)(void* user, int npeers, const int* peers, const int* counts, const double* const* send, double* const* recv);
typedef int (*nsk_allreduce_fn)(void* user, double* buf, int n);
int nsk_comm_init_host(nsk_ctx* shard, nsk_exchange_fn exchange, nsk_allreduce_fn allreduce, void* user);
int nsk_comm_unique_id(unsigned char* out128);
int nsk_comm_init_rccl(nsk_ctx* shard, const unsigned char* id128);
int nsk_allreduce_host(nsk_ctx* shard, double* buf, int n);
int nsk_group_test(nsk_ctx** shards, int n, int which, const double* const* in, double* const* out);  /* 0: dssum, 1: E apply */
/* rank-local pieces of update_hessenberg_matrix: partial bm1s dots and the projection update;
 * the host (or an all-reduce) sums the partial dots over the ranks */
int nsk_local_dots(nsk_ctx* ctx, nsk_vec f, const nsk_vec* Q, int nq, double* out);
int nsk_project_out(nsk_ctx* ctx, nsk_vec f, const nsk_vec* Q, int nq, const double* h);

/* diagnostics: CG (slower component) and GMRES iteration counts of every time step of the last map of this context (what the
 * per-step launch budgets follow); *nsteps = steps recorded (0 before the first map and on shard contexts) */
int nsk_get_step_iters(nsk_ctx* ctx, int n, int* helm, int* pres, int* nsteps);

/* measurement hook for bench.py: average duration (us) of `reps` back-to-back launches of a hot
 * kernel ("helm"), HIP events on the library's stream, full work in every launch */
int nsk_bench_kernel(nsk_ctx* ctx, const char* name, int reps, double* avg_us);

/* ---- kernel-level test hooks (parity against oracle/, tests/test_kernels_gpu.py) ---- */
int nsk_test_axhelm(nsk_ctx* ctx, const double* u, double h1, double h2, double* out); /* local */
int nsk_test_dssum(nsk_ctx* ctx, const double* u, double* out);
int nsk_test_opdiv(nsk_ctx* ctx, const double* u, const double* v, double* out);
int nsk_test_opgradt(nsk_ctx* ctx, const double* p, double* ox, double* oy);
int nsk_test_convect(nsk_ctx* ctx, int adjoint, const double* u, const double* v,
                     double* ox, double* oy);      /* mass-weighted forcing bf (sponge+convection) */
int nsk_test_eapply(nsk_ctx* ctx, const double* p, double* out);   /* D B^-1 D^T p */
int nsk_test_helm_solve(nsk_ctx* ctx, const double* rx, const double* ry, int order,
                        double* ox, double* oy, int* iters);
int nsk_test_pres_solve(nsk_ctx* ctx, const double* g, double* out, int* iters);
/* 3-D element operators, packed arrays: which = 1 weak divergence, 2 D^T p, 3 convection (a = mode), 5 Helmholtz solve (a = order),
 * 8 convection on the matrix cores (lx1 = 8) */
int nsk_test_op3(nsk_ctx* ctx, int which, const double* in, double* out, int a, int* iters);

#ifdef __cplusplus
}
#endif
#endif
