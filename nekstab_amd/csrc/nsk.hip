// libnekstab_hip.so: host driver + C-ABI (include/nekstab_hip.h).
//
// Mirrors, behind nekStab's own operator interface, the hot path
//   matvec(f,q)                      core/matvec.f:64-154
//   forward/adjoint_linearized_map   core/matvec.f:163-326
//   krylov_* vector algebra          core/krylov_subspace.f:24-258
//   update_hessenberg_matrix         core/krylov_decomposition.f:116-202
// and the Nek5000 perturbation step those call.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <utility>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/nekstab_hip.h"
#include "nsk_basis.hpp"
#include "nsk_kernels.hpp"
#include "nsk_persist.hpp"
#include "nsk3_kernels.hpp"
#include "nsk3_mfma.hpp"

using namespace nsk;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess)                                                                \
      return fail(NSK_EHIP, std::string(#x) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
  } while (0)

// kernel set = (dimension, lx1): key = lx1 for quadrilaterals (namespace k2), 100 + lx1 for hexahedra (k3)
#define DISPATCH_N(n, ...)                                                                      \
  switch (n) {                                                                                  \
    case 6: { using namespace nsk::k2; constexpr int N = 6; __VA_ARGS__; } break;               \
    case 8: { using namespace nsk::k2; constexpr int N = 8; __VA_ARGS__; } break;               \
    case 10: { using namespace nsk::k2; constexpr int N = 10; __VA_ARGS__; } break;             \
    case 12: { using namespace nsk::k2; constexpr int N = 12; __VA_ARGS__; } break;             \
    case 106: { using namespace nsk::k3; constexpr int N = 6; __VA_ARGS__; } break;             \
    case 108: { using namespace nsk::k3; constexpr int N = 8; __VA_ARGS__; } break;             \
    case 110: { using namespace nsk::k3; constexpr int N = 10; __VA_ARGS__; } break;           \
    default: return fail(NSK_EINVAL, "unsupported lx1");                                        \
  }

struct nsk_ctx {
  int N = 0, NN = 0, M = 0, MM = 0, ND = 0, NDD = 0, EPB = 0, NT = 0, NTD = 0;
  int ndim = 2, key = 0;                // key selects the kernel set (DISPATCH_N)
  int hrows = 8, hstride = 8;           // rows of Helmholtz partials per parity / stride of their totals (3-D: 12 / 16)
  // large coarse spaces: sparse operator + Chebyshev-Jacobi polynomial instead of the dense inverse
  int coarse_iter = 0, cheb_deg = 0, cA_width = 0;      // cA_*: sparse coarse operator, row-major with padded rows of cA_width entries
  double cheb_theta = 0, cheb_delta = 0;
  const int *cA_rp = nullptr, *cA_ci = nullptr;
  const double *cA_va = nullptr, *cA_dinv = nullptr;
  double *cw_d0 = nullptr, *cw_d1 = nullptr, *cw_r = nullptr;
  // block-circulant coarse solve (meshes of uniform periodic layers): nsk3_setup.inc
  int circ_nz = 0, circ_nv2 = 0, circ_nmat = 0, circ_lda = 0;
  const float* circ_Binv = nullptr; const double* circ_F = nullptr; const int2* circ_cols = nullptr;
  double *circ_rh = nullptr, *circ_xh = nullptr;
  int nel = 0, nblk = 0, nvert = 0;
  long long nloc = 0, npr = 0, nstate = 0;
  int nscal = 0;                        // scalar fields theta(:, 1..ldimt) carried behind the pressure (option "nscal")
  double dt = 0, re = 0, endtime = 0;
  int nsteps = 0;
  int max_helm = 60, max_pres = 40, min_pres = 0, pres_cap = 0, layers = 1;
  int cur_helm[NCLS] = {}, cur_pres[NCLS] = {};       // adaptive launch budgets per BDF order
  int bh_helm[NCLS][8] = {}, bh_pres[NCLS][8] = {}, bh_n = 0;                         // iteration maxima of the last maps (budgets_update)
  int use_graph = 1;
  int gmres_cycle = MAXMR;              // pressure GMRES restarts after this many iterations (option "gmres_cycle": tests use short cycles)
  int mfma_convect = 1;                 // hexahedra, lx1 = 8 and 10: convection kernel with the contractions on the fp64 matrix cores (nsk3_mfma.hpp: k_convect_mfma8, k_convect_mfma<10>, and the full equations' k_convect_mfma_nl<10>)
  int fused = 0;                        // persistent velocity solve (k_helm_fused): one launch per time step instead of one per CG iteration
  unsigned* sync = nullptr;             // grid-barrier counters of the persistent kernels
  int in_test = 0;
  int helm_guess = 1;
  int budget_freeze = 0;
  int helm_fdm = -1;                    // hexahedra: element-block fast-diagonalisation preconditioner of the velocity solves (NSK_HELM_FDM=1 builds it; measured slower than Jacobi-CG, off)
  int eapply_pipe = 4;                  // hexahedra, the Schwarz + D^T kernel: 0 = one workgroup per element, 1 = resident workgroups with the next element's loads in flight (k_schwarz_p), 2 = one wavefront per element (k_schwarz_w), 3 = 2 + k_divgs_w, 4 = the default form: one wavefront per element, sixteen per CU (k_schwarz_w16) at lx1 <= 8, four wavefronts per element (k_schwarz_q, = 5) at lx1 = 10
  int zero_metrics = 1;                 // hexahedra: arrays of the mapping / base-flow constants that are zero on every node are cleaned at set-up and not loaded by the kernels (Dev::zmask, Dev::bfmask); NSK_ZERO_METRICS=0: set-up as rounds 1-4 (rounding noise kept); option zero_metrics = 0: cleaned arrays, every one loaded
  unsigned zmask_built = 0, bfmask_built = 0;
  int helm_pf = 1, helm_pf_grid = 0;    // hexahedra, lx1 = 10: the CG iteration as resident workgroups with LDS-DMA prefetch of the next element (k_helm_p; option "helm_pf", NSK_HELM_PF; 0 = k_helm<10>)
  int divgs_c3 = -1;                    // hexahedra: k_divgs with the three components' pass chains side by side (k_divgs_c3: 5 barriers instead of 12).  -1 (default) = at lx1 = 10 only: there k_divgs<10> (80 registers x 1024 threads) is alone on its CU and k_divgs_c3<10> (56) is not -- 284 against 543 us at 13 824 elements, 1.96 against 3.98 ms at config 5's size; at lx1 = 8 it measured 6 % SLOWER (718 against 678 us at config 4's size: four workgroups per CU either way, and the gather, not the passes, is what a workgroup waits for)
  int eapply_grid[2] = {0, 0};          // their grid sizes (workgroups that fit the device at once; 0 = not yet asked)
  int flat_proj = -1;                   // hexahedra: once-per-step sums over the GMRES / projection bases as streaming kernels (k_pres_comb, k_proj_dots); -1 = yes on single-rank contexts
  int gs_lag = -1;                      // hexahedra: lagged second Gram-Schmidt correction (k_gs_lag: two basis reads per GMRES column instead of four); -1 = yes on single-rank contexts
  int gs2_from = MAXMR;                 // quadrilaterals: GMRES columns from this iteration (of a cycle) on get a second Gram-Schmidt pass (default: never)
  int dbg_max_order = 3, dbg_ab2 = 0, dbg_pext = 1;      // time-scheme sensitivity switches (options of the same names)
  double early_pres_mul = 1e-2;         // pressure tolerance factor of time steps 1-3 of every map
  long long recaptures = 0, retries = 0;
  double recapture_s = 0.0;             // host time spent capturing + instantiating step graphs (since init)
  // since init (nsk_stats: the per-matvec fields are reset by every nsk_matvec, these are not)
  long long tot_capped = 0, tot_helm_iters = 0, tot_pres_iters = 0, tot_steps = 0, tot_pres_jsum = 0;
  double coarse_bytes = 0.0;            // bytes one coarse solve reads (operator storage): nsk_stats::coarse_bytes_per_solve
  double tot_worst_cap = 0.0;
  int debug = 0;
  // graphs[.][NCLS]: the last step class (time steps >= 17: 167 of the 183 steps of a config-2 map) also as ONE graph of
  // `graph_steps` consecutive steps: fewer graph launches and graph-to-graph hand-overs per map (option "graph_steps", 1 = off)
  struct StepGraph { hipGraphExec_t exec = nullptr; int nh = -1, np = -1; } graphs[3][NCLS + 1];
  int graph_steps = 1;                  // (measured on config 2: 11.79 / 11.80 / 11.84 matvecs/s at 1 / 8 / 16 steps per graph: within noise, off)
  // ---- launch budgets PER TIME STEP (round 5; option "step_budgets", default on for graph-replayed single-rank contexts).
  // The iteration counts of a map decay along its time steps (config 2: 28 CG iterations at step 1, 9 at step 40, 7 at step
  // 183) and the pressure counts wander between 2 and 10 with the restarts of the projection space: one budget per step CLASS
  // must cover the largest count of its class (the class of steps >= 17 holds 167 steps), and every budgeted launch beyond a
  // solve's own count still costs its dispatch (2-4 us; 39 % of the kernel time of a config-2 map in round 4).  With the
  // per-step record of the last SBW maps (Dev::step_iters) the budget of step s follows the largest count seen at steps
  // s-2..s+2 in those maps + head-room: measured on 56 maps of config 2 (scripts/step_iters_dump.py, profiles/r05_step_budgets.txt)
  // 14.4 + 9.9 launches per step instead of 20.5 + 13.7, no overflow.  One captured graph per (step class, budget pair), kept
  // in `gcache` (a few dozen pairs occur); a map that overflows is redone with the class budgets.
  static constexpr int SBW = 8, SBN = 2;
  int step_budgets = 1;
  struct StepBudgets {                   // one history per kind of map (0 direct, 1 adjoint): their solves differ
    std::vector<int> hist_h[SBW], hist_p[SBW];
    int n = 0;                           // maps in the history (ring slot = n % SBW)
    std::vector<int> bh, bp;             // budgets of the next map of this kind (empty: class budgets)
  } sb[2];
  int last_map_kind = 0;
  bool sb_force_class = false;           // the redo of an overflowed map runs on the (doubled) class budgets
  // ---- persistent tails (nsk_persist.hpp: k_helm_tail, k_pres_tail; option "tail").  Behind the launches of a solve ONE persistent
  // launch runs whatever iterations are left, to the solver's caps: no budget can overflow, no map is redone.
  //   0: never.   1: HEADS = the median count of the step over the last maps, the tail does the rest (0.25 + 0.52 tail iterations per
  //   step on config 2).   2: SAFETY NET: the per-time-step budgets (largest count + head-room) and the tail behind them: it runs only
  //   when a solve outruns its budget, which lets the head-room shrink from 3 / 2 to 1 / 0.   -1 (default): 2 where every
  //   workgroup of the grid is resident at once (quadrilateral single-rank contexts of at most ~3 workgroups per CU: configs 1, 2),
  //   else 0.
  // Measured on config 2, 118 timed Arnoldi steps (scripts/ab_safety.sh, ab_safety2.sh; run-to-run +-0.1): budgets alone 13.18
  // matvecs/s with 3 redone maps; mode 2 at head-room 3/2: 13.18, 2/1: 13.26-13.47, 1/0: 13.51, no redone map; mode 1: 13.29.
  // All modes return the same bits (tests/test_persistent_gpu.py).  (The kernel trace had promised more: a launch that finds its
  // solve finished shows as 4.4 us there; in an un-profiled graph replay it costs ~1.5 us, and a tail iteration ~2x a launched one:
  // three grid barriers with agent-scope release / acquire, two workgroups per CU.)
  int tail = -1;
  int tail_ok = -1;                      // residency verdict (-1: not asked yet)
  int tail_off_h = 0, tail_off_p = 0;    // heads = median + these (options "tail_off_h" / "tail_off_p": tests push work into the tails with negative values)
  long long tail_maps = 0;
  long long sb_maps = 0, sb_steps = 0, sb_launch_h = 0, sb_launch_p = 0;     // maps / time steps run on per-step budgets and their budgeted launches (diagnostics)
  bool last_map_per_step = false;        // the map just run took the per-step budgets
  std::map<std::array<int, 4>, hipGraphExec_t> gcache;         // (adjoint, class, nh, np) -> captured step
  std::vector<nsk_ctx*> graph_members;  // sharded step graphs (held by the first rank of a process): the ranks they were captured for
  int merged_iters = 24;                // ... for the first merged_iters iterations of a solve (see pres_solve_launch; 12 until round 4: 24 covers the tightened solves of time steps 1-3 too, +2 % on config 2 at identical iteration counts)
  int merged_update = 1;                // GMRES column bookkeeping inside the coarse-solve kernel (k_update_coarse)
  int tc32 = 0;                         // k_divgs_t: the fp32 copy of the coarse image (0 = never (default: it moves a map by 2e-8 for 0.6 us per iteration), 1 = always, -1 = where the solve's relative tolerance is >= 1e-5; option "tc32", NSK_TC32)
  int skip_close = 1;                   // no closing launch behind a pressure tail that covers the merged range (option "skip_close", NSK_SKIP_CLOSE; measured +0.3 %)
  int sb_pct = 90;                      // per-step budgets behind the safety-net tail: percentile of the window's counts (100 = the largest, rounds 5 to mid-6; option "sb_pct", NSK_SB_PCT; 90 measured +0.6 %, with skip_close +1.3 %: profiles/r06_ab_fuse2.txt)
  int fuse2_start = 1;                  // ... and the solve's start inside its first launch (k_proj_apply_e; no k_gmres_update(-1) launch); option "fuse2_start", NSK_FUSE2_START
  int fuse2 = 1;                        // round 6: the merged iteration in TWO launches (k_schwarz_uc, k_divgs_t; option "fuse2", NSK_FUSE2); 0 = the three launches of rounds 3-5
  double* kacc = nullptr;               // nsk_orth: coefficients accumulated over the two passes + the squared norm (device)
  bool released = false;                // nsk_shard_release_parent: only the arrays shards share are left on the device
  static constexpr int ORTH_CHUNKS = 4;
  int orth_overlap = 1;                 // nsk_orth on RCCL ranks: chunked all-reduces on comm_stream overlapped with the next chunk's dots
  hipStream_t comm_stream = nullptr;
  hipEvent_t orth_ev[2 * ORTH_CHUNKS] = {};
  int shard_graph = -1;                 // sharded step in a hipGraph: -1 = yes unless a communicator is attached, 0 = no, 1 = yes (also with RCCL)
  // Eager sharded steps with a real transport: every launched iteration costs its halo exchange and all-reduce whether the solve
  // has converged or not (the host enqueues them; a launch that finds its solve converged is 2 us, a collective is not).  With
  // "shard_hostcheck" the host reads the device's convergence flags (identical on every rank: they come from all-reduced
  // sums) and stops issuing iterations: -1 = yes with a communicator / host transport, 0 = never, 1 = always.
  int shard_hostcheck = -1;
  // RCCL ranks: the all-reduce of an iteration's dot products travels in the SAME ncclGroupStart / ncclGroupEnd as that iteration's
  // halo send / recv pairs (one group per CG iteration instead of two, three per GMRES iteration instead of four): option
  // "rccl_fuse" (NSK_RCCL_FUSE=0 restores the separate calls; bench.py's eager retry attempt does)
  int rccl_fuse = 1;
  int arv_left = 24;                    // all-reduces of this context still to be verified bit for bit across the ranks (option "allred_verify"; k_ar_chunks / k_ar_check)
  double* arv_buf = nullptr;            // 64 doubles
  // The same on a full-mesh context (option "hostcheck"): eager steps, no launch budgets, no redone maps.  -1 = yes for large
  // hexahedral meshes (>= 8192 elements: a launch that only finds its solve converged costs 25-140 us there and a map redone
  // with larger budgets tens of seconds) and for quadrilateral meshes of more than 4096 workgroups (config 3), 0 = never, 1 = always.
  int hostcheck = -1;
  // Halo / interior overlap of the velocity solve on shards (quadrilaterals): a shard keeps its BOUNDARY elements (those with a
  // node another rank shares) first; k_helm is launched for the boundary workgroups, their halo travels on a second stream
  // while the interior workgroups run, and the reduction that follows waits for both.  Option "halo_overlap" (0 / 1).
  int halo_overlap = 0, nbb = 0;                        // nbb: workgroups that hold boundary elements
  hipEvent_t ov_ev[2] = {nullptr, nullptr};
  std::vector<long long> elems_glob;                    // global ids of the owned elements in LOCAL order (nsk_shard_elems)
  int hc_helm[NCLS] = {}, hc_pres[NCLS] = {};           // iterations the last solve of the class used: where the next one is first checked
  long long hc_checks = 0;                              // flag reads since init (diagnostics)
  double* scratch = nullptr;            // one state vector
  const double* xyz = nullptr;          // GLL coordinates [ndim][nloc] (nsk_seed_noise)
  double* rc_big = nullptr;             // coarse restriction for nvert > 3072
  Dev d{};
  Stats hstats{};
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  std::map<void*, size_t> alloc_bytes;  // size of every device allocation of this context (reset_solver_state)
  // per-time-step iteration record of the last map (Dev::step_iters, copied behind every map): [2 * nsteps] ints
  static constexpr int STEP_CAP = 8192;
  int* h_step_iters = nullptr;          // pinned
  int step_rec_n = 0;                   // time steps of the map the record belongs to
  bool state_dirty = false;             // the mutable solver state is not what a map left (nsk_bench_kernel ran on it): the next map starts from a reset state
  // krylov scratch
  double* kpart = nullptr; double* kout = nullptr; double** kptr = nullptr; int kblk = 0;
  double* hpin = nullptr;   // pinned host scratch
  Stats* hstat_pin = nullptr;           // device counters of the last map attempt (pinned)
  nsk_ctx* clone_of = nullptr;          // a second lane of another context (nsk_clone): shares every immutable device array
  // work vectors for tests / setup
  double *wv1 = nullptr, *wv2 = nullptr, *wp1 = nullptr, *wp2 = nullptr;
  std::vector<double> bm1s_host;
  // ---- host copies kept for element sharding (nsk_shard_create)
  std::vector<long long> h_gid;
  std::vector<double> h_cflg, h_dAs, h_bs, h_mask;      // for nsk_set_baseflow (new dt => new Jacobi diagonals)
  double cfl_target = 0.5;
  std::vector<int> h_pidx, h_evert;
  int PS = 0, coarse_lda = 0;
  // ---- shard state (rank-local context cut out of a full-mesh parent)
  nsk_ctx* parent = nullptr;
  int rank = 0, nranks = 1;
  std::vector<int> elems;                               // owned global elements
  int nvh = 0;                                          // velocity halo entries = ghost slots
  const int *vh_off = nullptr, *vh_idx = nullptr;       // pack CSR
  std::vector<int> peers, vh_poff, vh_pcnt;             // per peer: slice of [0, nvh)
  double *vsend = nullptr, *vrecv = nullptr;            // [peer][component][entry], 4 * nvh doubles
  const int2* vh_seg = nullptr;                         // halo slot -> {first slot, slot count} of its peer
  int nps = 0, npg = 0;                                 // pressure halo: send entries / ghost entries
  const int* ph_sidx = nullptr;
  std::vector<int> ph_soff, ph_scnt, ph_goff, ph_gcnt;  // per peer
  double *psend = nullptr, *precv = nullptr;
  double* rc_part = nullptr;                            // coarse restriction (all vertices), summed over ranks
  void* comm = nullptr;                                 // ncclComm_t when ranks are real processes
  // host-staged transport (nsk_comm_init_host): ranks in separate processes without RCCL, e.g. several ranks on one GPU
  nsk_exchange_fn host_xchg = nullptr; nsk_allreduce_fn host_allred = nullptr; void* host_user = nullptr;
  double *hs_send = nullptr, *hs_recv = nullptr; size_t hs_cap = 0;
  // ---- rank-local set-up (nsk_init_local): this context covers a SUB-MESH = the elements a rank owns + two rings of
  // neighbours; everything element-local is exact on the owned elements, the few global facts are exchanged by the caller
  bool local = false;
  std::vector<char> local_own;                          // [nel] 1 = owned by this rank
  std::vector<int> crow_u, crow_v; std::vector<double> crow_a;      // rows of A_c of the vertices this rank owns (triplets)
  double vol_own = 0.0, ctarg = 0.0, fd_lmax = 0.0;
  long long npr_glob = 0;                               // pressure dofs of the WHOLE mesh (0: this context is the whole mesh)
  int has_outflow = 1;
  bool local_done = false;
  // ---- time-periodic base flow (Floquet)
  double* orbit[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const double* steady[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int orbit_steps = 0;
};

// every captured step is stale (Dev, coefficients, kernel choices are baked in): class graphs re-captured on next use, the
// per-step budget cache emptied, the per-step iteration history forgotten (the solves of a new operator behave differently)
static void invalidate_graphs(nsk_ctx* c) {
  for (auto& a : c->graphs) for (auto& g : a) g.nh = -1;
  for (auto& kv : c->gcache) if (kv.second) (void)hipGraphExecDestroy(kv.second);
  c->gcache.clear();
  for (auto& b : c->sb) { b.n = 0; b.bh.clear(); b.bp.clear(); }
}

template <class T>
static int dalloc(nsk_ctx* c, T** p, size_t n) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) return fail(NSK_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  (void)hipMemset(q, 0, std::max<size_t>(n, 1) * sizeof(T));
  c->allocs.push_back(q);
  c->alloc_bytes[q] = std::max<size_t>(n, 1) * sizeof(T);
  *p = (T*)q;
  return 0;
}
template <class T>
static int dupload(nsk_ctx* c, const T** p, const std::vector<T>& h) {
  T* q = nullptr;
  int rc = dalloc(c, &q, h.size());
  if (rc) return rc;
  if (!h.empty()) HIPCHK(hipMemcpy(q, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return 0;
}

static bool lda_ok_for_ecv(const nsk_ctx* c) { return c->ndim == 2 && c->coarse_lda > 0 && c->coarse_lda <= 3072 && !c->local; }      // where k_update_coarse runs

static StepCoef make_coef(const nsk_ctx* c, int istep, int adjoint) {
  static const double BD[3][4] = {{1.0, 1.0, 0.0, 0.0}, {1.5, 2.0, -0.5, 0.0}, {11.0 / 6.0, 3.0, -1.5, 1.0 / 3.0}};
  static const double AB[3][3] = {{1.0, 0.0, 0.0}, {2.0, -1.0, 0.0}, {3.0, -3.0, 1.0}};
  // (the dbg_* members are sensitivity switches of scripts/wake_bisect.py: what the time scheme's details do to the spectrum;
  //  their defaults are the scheme of SURVEY Appendix A)
  const int k = std::min(istep, std::min(3, std::max(1, c->dbg_max_order)));
  StepCoef s;
  for (int q = 0; q < 4; ++q) s.bd[q] = BD[k - 1][q];
  for (int q = 0; q < 3; ++q) s.ab[q] = AB[k - 1][q];
  if (k == 2 && c->dbg_ab2 == 1) { s.ab[0] = 1.5; s.ab[1] = -0.5; }          // Adams-Bashforth 2 instead of the BDF2-consistent extrapolation
  s.pxt = (c->dbg_pext == 0) ? 0.0 : ((k == 3 || (c->dbg_pext == 2 && k == 2 && istep > 1)) ? 1.0 : 0.0);
  s.h2 = s.bd[0] / c->dt;
  s.invdt = 1.0 / c->dt;
  s.k = k;
  s.adjoint = adjoint;
  const bool g = c->helm_guess != 0;
  // du0: nothing / previous / linear / quadratic extrapolation of the previous increments
  static const double XG[4][3] = {{0, 0, 0}, {1, 0, 0}, {2, -1, 0}, {3, -3, 1}};
  const int gi = !g ? 0 : std::min(istep, 4) - 1;
  for (int q = 0; q < 3; ++q) s.xg[q] = XG[gi][q];
  s.cls = step_class(istep);
  return s;
}

// ---------------------------------------------------------------------------
// E-apply helper used by setup, tests and the projection space
// ---------------------------------------------------------------------------
static int eapply(nsk_ctx* c, const double* pin, double* wout) {
  DISPATCH_N(c->key, {
    hipLaunchKernelGGL(k_gradt<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, c->d, pin, c->d.yl);
    hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, c->d, (const double*)c->d.yl, wout, -1, 0);
  });
  return 0;
}

// dense in-place inverse (Gauss-Jordan, partial pivoting), n small
static bool invert_dense(std::vector<double>& A, int n) {
  std::vector<int> piv(n);
  for (int k = 0; k < n; ++k) {
    int p = k; double best = std::fabs(A[(size_t)k * n + k]);
    for (int i = k + 1; i < n; ++i) { double v = std::fabs(A[(size_t)i * n + k]); if (v > best) { best = v; p = i; } }
    if (best == 0.0) return false;
    piv[k] = p;
    if (p != k) for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
    const double d = 1.0 / A[(size_t)k * n + k];
    A[(size_t)k * n + k] = 1.0;
    for (int j = 0; j < n; ++j) A[(size_t)k * n + j] *= d;
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const double f = A[(size_t)i * n + k];
      if (f == 0.0) continue;
      A[(size_t)i * n + k] = 0.0;
      double* ai = &A[(size_t)i * n];
      const double* ak = &A[(size_t)k * n];
      for (int j = 0; j < n; ++j) ai[j] -= f * ak[j];
    }
  }
  for (int k = n - 1; k >= 0; --k)
    if (piv[k] != k) for (int i = 0; i < n; ++i) std::swap(A[(size_t)i * n + k], A[(size_t)i * n + piv[k]]);
  return true;
}

__global__ void k_gj_extract(const double* __restrict__ A, int n, int k, double* __restrict__ rowk, double* __restrict__ colk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double p = 1.0 / A[(size_t)k * n + k];
  rowk[t] = (t == k) ? p : A[(size_t)k * n + t] * p;
  colk[t] = (t == k) ? 0.0 : A[(size_t)t * n + k];
}
__global__ void k_gj_update(double* __restrict__ A, int n, int k, const double* __restrict__ rowk, const double* __restrict__ colk) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= n) return;
  const double p = rowk[k];
  double v;
  if (i == k) v = rowk[j];
  else if (j == k) v = -colk[i] * p;
  else v = A[(size_t)i * n + j] - colk[i] * rowk[j];
  A[(size_t)i * n + j] = v;
}

// ---------------------------------------------------------------------------
// nsk_init
// ---------------------------------------------------------------------------
static int fused_possible(nsk_ctx* c);

// (gid, local index) pairs into ascending (gid, index) order.  The pairs arrive in ascending index, so a counting sort on the
// gid is stable and O(n): 1 s instead of the 6 s std::sort takes on the 1e8 nodes of config 5.  Sparse numberings (a rank's
// sub-mesh keeps the global ids) fall back to the comparison sort when the id range is more than 8x the node count.
static void sort_gid_pairs(std::vector<std::pair<long long, int>>& srt) {
  const size_t n = srt.size();
  long long mx = -1;
  bool neg = false;
  for (const auto& p : srt) { mx = std::max(mx, p.first); neg = neg || p.first < 0; }
  if (neg || mx + 1 > 8 * (long long)n + 1024) { std::sort(srt.begin(), srt.end()); return; }
  std::vector<unsigned> start((size_t)mx + 2, 0u);
  for (const auto& p : srt) start[(size_t)p.first + 1]++;
  for (size_t g = 0; g + 1 < start.size(); ++g) start[g + 1] += start[g];
  std::vector<std::pair<long long, int>> out(n);
  for (const auto& p : srt) out[start[(size_t)p.first]++] = p;
  srt.swap(out);
}

// dense coarse solve: A_c (nvert x nvert, host) -> its inverse on the device (Gauss-Jordan), fp64 (Dev::Aci) and fp32 with
// padded rows (Dev::Acif).  A singular pressure operator has the constant in A_c's null space: shifted out first.
static int coarse_dense_inverse(nsk_ctx* c, std::vector<double>& Ac, bool has_outflow) {
  Dev& d = c->d;
  const int nvert = c->nvert, lda = c->coarse_lda;
  int rc;
  if (!has_outflow) {               // constant null space: shift it out (coarse constant = sum of hats)
    double tr = 0; for (int v = 0; v < nvert; ++v) tr += Ac[(size_t)v * nvert + v];
    const double sh = tr / nvert / nvert;
    for (size_t k = 0; k < Ac.size(); ++k) Ac[k] += sh;
  }
  double* dA = nullptr; double *drow = nullptr, *dcol = nullptr;
  if ((rc = dalloc(c, &dA, Ac.size())) || (rc = dalloc(c, &drow, nvert)) || (rc = dalloc(c, &dcol, nvert))) return rc;
  HIPCHK(hipMemcpy(dA, Ac.data(), Ac.size() * sizeof(double), hipMemcpyHostToDevice));
  for (int k = 0; k < nvert; ++k) {
    hipLaunchKernelGGL(k_gj_extract, dim3((nvert + 255) / 256), dim3(256), 0, c->stream, (const double*)dA, nvert, k, drow, dcol);
    hipLaunchKernelGGL(k_gj_update, dim3((nvert + 255) / 256, nvert), dim3(256), 0, c->stream, dA, nvert, k, (const double*)drow, (const double*)dcol);
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  d.Aci = dA;
  std::vector<double> Ah(Ac.size());
  HIPCHK(hipMemcpy(Ah.data(), dA, Ah.size() * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<float> Af((size_t)nvert * lda, 0.0f);
  for (int r = 0; r < nvert; ++r) for (int q = 0; q < nvert; ++q) Af[(size_t)r * lda + q] = (float)Ah[(size_t)r * nvert + q];
  c->coarse_bytes = 4.0 * nvert * lda + 16.0 * nvert;
  return dupload(c, &d.Acif, Af);
}

static int build(nsk_ctx* c, const nsk_case& cs) {
  const int N = cs.lx1, NN = N * N, M = N - 2, MM = M * M, ND = cs.lxd > 0 ? cs.lxd : 3 * N / 2, NDD = ND * ND;
  if (cs.ndim != 2) return fail(NSK_EINVAL, "build(): ndim must be 2");
  const bool timing = std::getenv("NSK_SETUP_TIMING") != nullptr;       // seconds per set-up section on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto tick = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "set-up %-44s %7.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  c->ndim = 2; c->key = cs.lx1; c->d.ndim = 2;
  if (!(N == 6 || N == 8 || N == 10 || N == 12)) return fail(NSK_EINVAL, "lx1 must be 6, 8, 10 or 12");
  if (ND != 3 * N / 2) return fail(NSK_EINVAL, "lxd must be 3*lx1/2");
  const int nel = cs.nel;
  c->N = N; c->NN = NN; c->M = M; c->MM = MM; c->ND = ND; c->NDD = NDD;
  c->EPB = (256 / NN) > 0 ? 256 / NN : 1;
  c->NT = ((c->EPB * NN + 63) / 64) * 64;
  c->NTD = ((NDD + 63) / 64) * 64;
  c->nel = nel; c->nblk = (nel + c->EPB - 1) / c->EPB;
  c->nloc = (long long)nel * NN; c->npr = (long long)nel * MM; c->nstate = 2 * c->nloc + c->npr;
  c->re = cs.re; c->endtime = cs.endtime;
  c->layers = std::max(1, std::min(cs.schwarz_layers > 0 ? cs.schwarz_layers : 1, std::min(4, M)));
  if (cs.max_helm_iter > 0) c->max_helm = cs.max_helm_iter;
  if (cs.max_pres_iter > 0) c->max_pres = std::min(cs.max_pres_iter, 4 * MAXMR);     // beyond MAXMR: restarted GMRES cycles
  const long long nloc = c->nloc, npr = c->npr;
  HIPCHK(hipStreamCreate(&c->stream));
  Dev& d = c->d;
  d.nel = nel; d.nblk = c->nblk; d.nloc = nloc; d.npr = npr; d.nu = 1.0 / cs.re;
  d.cs = nloc; d.ps = npr; d.npr_glob = npr;          // one rank: no ghost slots
  d.nranks = 1; d.rank = 0;
  c->h_gid.assign(cs.gid, cs.gid + nloc);
  d.tol_helm = cs.tol_helm > 0 ? cs.tol_helm : 1e-9;
  d.tol_pres = cs.tol_pres > 0 ? cs.tol_pres : 1e-7;
  d.tol_relative = cs.tol_relative; d.max_mr = std::min(c->max_pres, MAXMR); d.has_outflow = cs.has_outflow;
  // projection space also with the pressure null space (adjoint runs, closed domains): with the restart policy of k_proj_update it
  // cuts config 4's pressure iterations from 26 to 15 per step (rounds 1-2 switched it off there: their merge policy made such solves slower)
  d.nproj_max = (cs.has_outflow || !std::getenv("NSK_NO_PROJ_SINGULAR")) ? std::min(cs.nproj, MAXPROJ) : 0;
  d.proj_restart = 1;
  if (const char* g = std::getenv("NSK_PROJ_RESTART")) d.proj_restart = std::atoi(g);

  // ---- bases
  std::vector<double> z1, w1, z2, w2, zd, wd;
  zwgll(N, z1, w1); zwgl(M, z2, w2); zwgl(ND, zd, wd);
  std::vector<double> D = deriv_matrix(z1), J12 = interp_matrix(z1, z2), Jd = interp_matrix(z1, zd), Dd = deriv_matrix(zd);
  std::vector<double> D12 = matmul(J12, D, M, N, N);
  std::vector<double> hat(4 * MM);
  for (int b = 0; b < M; ++b)
    for (int a = 0; a < M; ++a) {
      const double rm = 0.5 * (1 - z2[a]), rp = 0.5 * (1 + z2[a]), sm = 0.5 * (1 - z2[b]), sp = 0.5 * (1 + z2[b]);
      hat[0 * MM + b * M + a] = sm * rm; hat[1 * MM + b * M + a] = sm * rp;
      hat[2 * MM + b * M + a] = sp * rm; hat[3 * MM + b * M + a] = sp * rp;
    }
  int rc;
  if ((rc = dupload(c, &d.D, D)) || (rc = dupload(c, &d.J12, J12)) || (rc = dupload(c, &d.D12, D12)) ||
      (rc = dupload(c, &d.Jd, Jd)) || (rc = dupload(c, &d.Dd, Dd)) || (rc = dupload(c, &d.hat, hat))) return rc;

  // ---- geometry  [UPSTREAM coef.f geom1 / geom2 / set_dealias_rx]
  std::vector<double> g1(nloc), g2(nloc), g4(nloc), bm1(nloc), rx(nloc), ry(nloc), sx(nloc), sy(nloc), jac(nloc);
  for (int e = 0; e < nel; ++e) {
    const double* X = cs.x + (size_t)e * NN; const double* Y = cs.y + (size_t)e * NN;
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) {
        double xr = 0, xs = 0, yr = 0, ys = 0;
        for (int k = 0; k < N; ++k) {
          xr += D[i * N + k] * X[j * N + k]; yr += D[i * N + k] * Y[j * N + k];
          xs += D[j * N + k] * X[k * N + i]; ys += D[j * N + k] * Y[k * N + i];
        }
        const size_t l = (size_t)e * NN + j * N + i;
        const double J = xr * ys - xs * yr;
        if (!(J > 0.0)) return fail(NSK_EINVAL, "non-positive Jacobian in element " + std::to_string(e));
        jac[l] = J; rx[l] = ys; ry[l] = -xs; sx[l] = -yr; sy[l] = xr;
        const double W = w1[i] * w1[j];
        bm1[l] = J * W;
        g1[l] = (ys * ys + xs * xs) * W / J;
        g2[l] = (yr * yr + xr * xr) * W / J;
        g4[l] = (ys * (-yr) + (-xs) * xr) * W / J;
      }
  }
  auto interp2 = [&](const std::vector<double>& Jm, int nt, const double* f, double* out) {  // out[b][a] = sum J[b][j] J[a][i] f[j][i]
    std::vector<double> t((size_t)N * nt);
    for (int j = 0; j < N; ++j)
      for (int a = 0; a < nt; ++a) { double s = 0; for (int i = 0; i < N; ++i) s += Jm[a * N + i] * f[j * N + i]; t[j * nt + a] = s; }
    for (int b = 0; b < nt; ++b)
      for (int a = 0; a < nt; ++a) { double s = 0; for (int j = 0; j < N; ++j) s += Jm[b * N + j] * t[j * nt + a]; out[b * nt + a] = s; }
  };
  std::vector<double> w2rx(npr), w2ry(npr), w2sx(npr), w2sy(npr);
  const size_t nfine = (size_t)nel * NDD;
  std::vector<double> cUr(nfine), cUs(nfine), GUx(nfine), GUy(nfine), GVx(nfine), GVy(nfine);
  std::vector<double> rxdA(nfine), rydA(nfine), sxdA(nfine), sydA(nfine);
  {
    std::vector<double> t(std::max(MM, NDD)), rxd(NDD), ryd(NDD), sxd(NDD), syd(NDD), Uf(NDD), Vf(NDD);
    for (int e = 0; e < nel; ++e) {
      const size_t o1 = (size_t)e * NN, o2 = (size_t)e * MM, od = (size_t)e * NDD;
      const std::vector<double>* src[4] = {&rx, &ry, &sx, &sy};
      double* dst2[4] = {&w2rx[o2], &w2ry[o2], &w2sx[o2], &w2sy[o2]};
      double* dstd[4] = {rxd.data(), ryd.data(), sxd.data(), syd.data()};
      for (int q = 0; q < 4; ++q) {
        interp2(J12, M, src[q]->data() + o1, dst2[q]);
        for (int b = 0; b < M; ++b) for (int a = 0; a < M; ++a) dst2[q][b * M + a] *= w2[a] * w2[b];
        interp2(Jd, ND, src[q]->data() + o1, dstd[q]);
        for (int b = 0; b < ND; ++b) for (int a = 0; a < ND; ++a) dstd[q][b * ND + a] *= wd[a] * wd[b];
      }
      interp2(Jd, ND, cs.ub + o1, Uf.data());
      interp2(Jd, ND, cs.vb + o1, Vf.data());
      for (int b = 0; b < ND; ++b)
        for (int a = 0; a < ND; ++a) {
          double Ur = 0, Us = 0, Vr = 0, Vs = 0;
          for (int k = 0; k < ND; ++k) {
            Ur += Dd[a * ND + k] * Uf[b * ND + k]; Us += Dd[b * ND + k] * Uf[k * ND + a];
            Vr += Dd[a * ND + k] * Vf[b * ND + k]; Vs += Dd[b * ND + k] * Vf[k * ND + a];
          }
          const int q = b * ND + a;
          rxdA[od + q] = rxd[q]; rydA[od + q] = ryd[q]; sxdA[od + q] = sxd[q]; sydA[od + q] = syd[q];
          cUr[od + q] = rxd[q] * Uf[q] + ryd[q] * Vf[q];
          cUs[od + q] = sxd[q] * Uf[q] + syd[q] * Vf[q];
          GUx[od + q] = rxd[q] * Ur + sxd[q] * Us;
          GUy[od + q] = ryd[q] * Ur + syd[q] * Us;
          GVx[od + q] = rxd[q] * Vr + sxd[q] * Vs;
          GVy[od + q] = ryd[q] * Vr + syd[q] * Vs;
        }
    }
  }

  tick("bases + geometry");
  // ---- gather-scatter CSR from the global numbering
  std::vector<int> gs_off(nloc + 1), gs_idx(nloc);
  std::vector<std::pair<long long, int>> srt(nloc);
  for (long long l = 0; l < nloc; ++l) srt[l] = {cs.gid[l], (int)l};
  sort_gid_pairs(srt);
  std::vector<int> grp_start(nloc), grp_cnt(nloc);
  {
    std::vector<int> cnt(nloc, 0);
    long long a = 0;
    while (a < nloc) {
      long long b = a;
      while (b < nloc && srt[b].first == srt[a].first) ++b;
      for (long long k = a; k < b; ++k) { grp_start[srt[k].second] = (int)a; grp_cnt[srt[k].second] = (int)(b - a); }
      a = b;
    }
    gs_off[0] = 0;
    for (long long l = 0; l < nloc; ++l) gs_off[l + 1] = gs_off[l] + grp_cnt[l];
    gs_idx.resize(gs_off[nloc]);
    for (long long l = 0; l < nloc; ++l)
      for (int k = 0; k < grp_cnt[l]; ++k) gs_idx[gs_off[l] + k] = srt[grp_start[l] + k].second;
  }
  {
    std::vector<int4> tab(nloc);
    for (long long l = 0; l < nloc; ++l) {
      const int n = gs_off[l + 1] - gs_off[l];
      int v[4] = {-1, -1, -1, -1};
      if (n <= 4) for (int k = 0; k < n; ++k) v[k] = gs_idx[gs_off[l] + k];
      else v[0] = -2;
      tab[l] = make_int4(v[0], v[1], v[2], v[3]);
    }
    if ((rc = dupload(c, &d.gs_tab, tab))) return rc;
  }
  auto dssum_h = [&](const std::vector<double>& f) {
    std::vector<double> o(nloc);
    for (long long l = 0; l < nloc; ++l) { double s = 0; for (int k = gs_off[l]; k < gs_off[l + 1]; ++k) s += f[gs_idx[k]]; o[l] = s; }
    return o;
  };
  std::vector<double> mask(cs.mask, cs.mask + nloc), minv(nloc), binv(nloc), spng(cs.spng, cs.spng + nloc), bm1s(nloc);
  {
    std::vector<double> bs = dssum_h(bm1);
    double vol = 0;
    for (long long l = 0; l < nloc; ++l) {
      // a node fixed in any copy is fixed in all
      double mk = 1.0;
      for (int k = gs_off[l]; k < gs_off[l + 1]; ++k) mk = std::min(mk, cs.mask[gs_idx[k]]);
      mask[l] = mk;
      minv[l] = 1.0 / (gs_off[l + 1] - gs_off[l]);
      binv[l] = mk / bs[l];
      bm1s[l] = (spng[l] != 0.0) ? 0.0 : bm1[l];      // core/usr_extra.f:116-118
      vol += bm1[l];
      if (c->local && c->local_own[l / NN]) c->vol_own += bm1[l];
    }
    d.vol = vol;
  }
  c->bm1s_host = bm1s;

  tick("gather-scatter tables, masks");
  // ---- dt rule  (core/matvec.f:26-46, [UPSTREAM compute_cfl])
  c->cfl_target = cs.cfl;
  {
    c->h_cflg.resize((size_t)4 * nloc);
    std::vector<double> dri(N);
    dri[0] = 1.0 / (z1[1] - z1[0]); dri[N - 1] = 1.0 / (z1[N - 1] - z1[N - 2]);
    for (int i = 1; i < N - 1; ++i) dri[i] = 1.0 / (0.5 * (z1[i + 1] - z1[i - 1]));
    double ctarg = 0;
    for (int e = 0; e < nel; ++e)
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
          const size_t l = (size_t)e * NN + j * N + i;
          const double ur = (cs.ub[l] * rx[l] + cs.vb[l] * ry[l]) / jac[l];
          const double us = (cs.ub[l] * sx[l] + cs.vb[l] * sy[l]) / jac[l];
          ctarg = std::max(ctarg, std::fabs(ur * dri[i]) + std::fabs(us * dri[j]));
          c->h_cflg[4 * l + 0] = rx[l] / jac[l] * dri[i]; c->h_cflg[4 * l + 1] = ry[l] / jac[l] * dri[i];
          c->h_cflg[4 * l + 2] = sx[l] / jac[l] * dri[j]; c->h_cflg[4 * l + 3] = sy[l] / jac[l] * dri[j];
        }
    c->ctarg = ctarg;
    double dt = cs.cfl / ctarg;
    c->nsteps = (int)std::ceil(cs.endtime / dt);
    c->dt = cs.endtime / c->nsteps;
    d.dt = c->dt;
  }

  // ---- Jacobi preconditioner of H for the three BDF orders
  std::vector<double> dinv(3 * nloc);
  {
    std::vector<double> dA(nloc);
    for (int e = 0; e < nel; ++e)
      for (int j = 0; j < N; ++j)
        for (int i = 0; i < N; ++i) {
          const size_t o = (size_t)e * NN;
          double s = 0;
          for (int k = 0; k < N; ++k) s += D[k * N + i] * D[k * N + i] * g1[o + j * N + k] + D[k * N + j] * D[k * N + j] * g2[o + k * N + i];
          s += 2.0 * D[i * N + i] * D[j * N + j] * g4[o + j * N + i];
          dA[o + j * N + i] = s;
        }
    std::vector<double> dAs = dssum_h(dA), bs = dssum_h(bm1);
    c->h_dAs = dAs; c->h_bs = bs; c->h_mask = mask;
    const double bd0[3] = {1.0, 1.5, 11.0 / 6.0};
    for (int k = 0; k < 3; ++k)
      for (long long l = 0; l < nloc; ++l) dinv[(size_t)k * nloc + l] = mask[l] / (d.nu * dAs[l] + bd0[k] / c->dt * bs[l]);
  }

  if ((rc = dupload(c, &d.g1, g1)) || (rc = dupload(c, &d.g2, g2)) || (rc = dupload(c, &d.g4, g4)) ||
      (rc = dupload(c, &d.bm1, bm1)) || (rc = dupload(c, &d.mask, mask)) || (rc = dupload(c, &d.minv, minv)) ||
      (rc = dupload(c, &d.binv, binv)) || (rc = dupload(c, &d.spng, spng)) || (rc = dupload(c, &d.bm1s, bm1s)) ||
      (rc = dupload(c, &d.dinv, dinv)) || (rc = dupload(c, &d.w2rx, w2rx)) || (rc = dupload(c, &d.w2ry, w2ry)) ||
      (rc = dupload(c, &d.w2sx, w2sx)) || (rc = dupload(c, &d.w2sy, w2sy)) || (rc = dupload(c, &d.cUr, cUr)) ||
      (rc = dupload(c, &d.cUs, cUs)) || (rc = dupload(c, &d.GUx, GUx)) || (rc = dupload(c, &d.GUy, GUy)) ||
      (rc = dupload(c, &d.GVx, GVx)) || (rc = dupload(c, &d.GVy, GVy)) || (rc = dupload(c, &d.gs_off, gs_off)) ||
      (rc = dupload(c, &d.gs_idx, gs_idx)) || (rc = dupload(c, &d.rxd, rxdA)) || (rc = dupload(c, &d.ryd, rydA)) ||
      (rc = dupload(c, &d.sxd, sxdA)) || (rc = dupload(c, &d.syd, sydA))) return rc;
  d.nl_spng_str = 0.0; d.spng_vr = nullptr; d.bstep = nullptr; d.bf_stride = 0;
  {
    std::vector<double> xy((size_t)2 * nloc);
    std::copy(cs.x, cs.x + nloc, xy.begin()); std::copy(cs.y, cs.y + nloc, xy.begin() + nloc);
    if ((rc = dupload(c, &c->xyz, xy))) return rc;
  }

  tick("dt rule, Jacobi diagonals, uploads");
  // ---- state + solver work arrays
  if ((rc = dalloc(c, &d.u, 2 * d.cs)) || (rc = dalloc(c, &d.p, npr)) || (rc = dalloc(c, &d.plag, npr)) ||
      (rc = dalloc(c, &d.pext, npr)) || (rc = dalloc(c, &d.ulag, 4 * d.cs)) || (rc = dalloc(c, &d.exlag, 4 * d.cs)) ||
      (rc = dalloc(c, &d.bf, 2 * d.cs)) || (rc = dalloc(c, &d.rloc, 2 * d.cs)) || (rc = dalloc(c, &d.bloc, 2 * d.cs)) || (rc = dalloc(c, &d.dulag, 6 * d.cs)) || (rc = dalloc(c, &d.hx, 2 * d.cs)) ||
      (rc = dalloc(c, &d.hr, 2 * d.cs)) || (rc = dalloc(c, &d.hp, 2 * d.cs)) || (rc = dalloc(c, &d.hs, 2 * d.cs)) ||
      (rc = dalloc(c, &d.hwl, 4 * d.cs)) || (rc = dalloc(c, &d.hpart, (size_t)16 * c->nblk)) || (rc = dalloc(c, &d.hscal, 32)) ||
      (rc = dalloc(c, &d.V, (size_t)(MAXMR + 1) * d.ps)) || (rc = dalloc(c, &d.Z, (size_t)MAXMR * npr)) ||
      (rc = dalloc(c, &d.yl, 2 * d.cs)) || (rc = dalloc(c, &d.ec, (size_t)nel * 4)) ||
      (rc = dalloc(c, &d.gpart, (size_t)(MAXMR + 2) * c->nblk)) || (rc = dalloc(c, &d.gpart2, (size_t)(MAXMR + 2) * c->nblk)) || (rc = dalloc(c, &d.gsc, 1)) ||
      (rc = dalloc(c, &d.stats, 1)) || (rc = dalloc(c, &c->wv1, 2 * d.cs)) || (rc = dalloc(c, &c->wv2, 2 * d.cs)) ||
      (rc = dalloc(c, &c->wp1, npr)) || (rc = dalloc(c, &c->wp2, npr)) || (rc = dalloc(c, &c->scratch, (size_t)c->nstate))) return rc;
  if (d.nproj_max > 0)
    if ((rc = dalloc(c, &d.PX, (size_t)d.nproj_max * npr)) || (rc = dalloc(c, &d.PEX, (size_t)d.nproj_max * npr)) ||
        (rc = dalloc(c, &d.PD, npr)) || (rc = dalloc(c, &d.PED, npr)) || (rc = dalloc(c, &d.ppart, (size_t)(MAXPROJ + 2) * c->nblk))) return rc;
  if ((rc = dalloc(c, &d.xacc, npr))) return rc;
  c->kblk = 256;
  if ((rc = dalloc(c, &c->kpart, (size_t)c->kblk * 1024)) || (rc = dalloc(c, &c->kout, 1024)) || (rc = dalloc(c, &c->kptr, 1024))) return rc;
  HIPCHK(hipHostMalloc((void**)&c->hpin, 4096 * sizeof(double)));
  HIPCHK(hipHostMalloc((void**)&c->hstat_pin, sizeof(Stats)));

  tick("work arrays");
  // ---- element adjacency (shared GLL nodes)
  std::vector<std::vector<int>> nb(nel);
  for (int e = 0; e < nel; ++e) {
    std::vector<int>& v = nb[e];
    v.push_back(e);
    for (int nd = 0; nd < NN; ++nd) {
      const long long l = (long long)e * NN + nd;
      if (gs_off[l + 1] - gs_off[l] == 1) continue;
      for (int k = gs_off[l]; k < gs_off[l + 1]; ++k) {
        const int f = gs_idx[k] / NN;
        if (f != e && std::find(v.begin(), v.end(), f) == v.end()) v.push_back(f);
      }
    }
    std::sort(v.begin() + 1, v.end());
  }
  std::vector<int> nb_off(nel + 1, 0), nb_idx;
  std::vector<long long> blk_off(nel + 1, 0);
  for (int e = 0; e < nel; ++e) {
    nb_off[e + 1] = nb_off[e] + (int)nb[e].size();
    blk_off[e + 1] = blk_off[e] + (long long)nb[e].size() * MM * MM;
    nb_idx.insert(nb_idx.end(), nb[e].begin(), nb[e].end());
  }
  // distance-2 colouring for simultaneous probing
  std::vector<int> colour(nel, -1);
  int ncolour = 0;
  {
    std::vector<int> used;
    for (int e = 0; e < nel; ++e) {
      used.assign(ncolour + 1, 0);
      for (int f : nb[e])
        for (int g : nb[f])
          if (colour[g] >= 0) used[colour[g]] = 1;
      int cc = 0;
      while (cc < ncolour && used[cc]) ++cc;
      colour[e] = cc;
      if (cc == ncolour) ++ncolour;
    }
  }
  tick("adjacency + colouring");
  // ---- probe E = D B^-1 D^T block-sparsely on the device
  std::vector<double> Eblk((size_t)blk_off[nel]);
  {
    double* dE = nullptr; const int *dnb_off = nullptr, *dnb_idx = nullptr; const long long* dblk = nullptr;
    if ((rc = dalloc(c, &dE, Eblk.size())) || (rc = dupload(c, &dnb_off, nb_off)) || (rc = dupload(c, &dnb_idx, nb_idx)) ||
        (rc = dupload(c, &dblk, blk_off))) return rc;
    for (int cc = 0; cc < ncolour; ++cc) {
      std::vector<int> els;
      for (int e = 0; e < nel; ++e) if (colour[e] == cc) els.push_back(e);
      const int* dels = nullptr;
      if ((rc = dupload(c, &dels, els))) return rc;
      const int ne = (int)els.size();
      HIPCHK(hipMemsetAsync(c->wp1, 0, npr * sizeof(double), c->stream));
      for (int k = 0; k < MM; ++k) {
        hipLaunchKernelGGL(k_set_probe, dim3((ne + 255) / 256), dim3(256), 0, c->stream, c->wp1, dels, ne, MM, k, 1.0);
        if ((rc = eapply(c, c->wp1, c->wp2))) return rc;
        hipLaunchKernelGGL(k_collect_probe, dim3(ne), dim3(256), 0, c->stream, (const double*)c->wp2, dels, ne, dnb_off, dnb_idx, dE, dblk, MM, k);
        hipLaunchKernelGGL(k_set_probe, dim3((ne + 255) / 256), dim3(256), 0, c->stream, c->wp1, dels, ne, MM, k, 0.0);
      }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(Eblk.data(), dE, Eblk.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipFree(dE));
    c->allocs.erase(std::find(c->allocs.begin(), c->allocs.end(), (void*)dE));
    c->alloc_bytes.erase((void*)dE);
  }
  auto Eentry = [&](int a, int r, int b, int k) -> double {   // E[(a,r),(b,k)]
    const std::vector<int>& v = nb[b];
    for (size_t s = 0; s < v.size(); ++s)
      if (v[s] == a) return Eblk[blk_off[b] + ((size_t)s * MM + k) * MM + r];
    return 0.0;
  };

  tick("E blocks probed on the device");
  // ---- coarse space: bilinear vertex functions sampled at the Gauss nodes
  const int nvert = (int)cs.nvert;
  c->nvert = nvert; d.nvert = nvert;
  std::vector<int> evert((size_t)nel * 4);
  for (size_t k = 0; k < evert.size(); ++k) evert[k] = (int)cs.vert[k];
  std::vector<int> v_off(nvert + 1, 0), v_ent((size_t)nel * 4);
  for (size_t k = 0; k < evert.size(); ++k) v_off[evert[k] + 1]++;
  for (int v = 0; v < nvert; ++v) v_off[v + 1] += v_off[v];
  { std::vector<int> pos(v_off.begin(), v_off.end() - 1); for (size_t k = 0; k < evert.size(); ++k) v_ent[pos[evert[k]]++] = (int)k; }
  {
    std::vector<double> Ac((size_t)nvert * nvert, 0.0), T(4 * MM);
    for (int b = 0; b < nel; ++b)
      for (size_t s = 0; s < nb[b].size(); ++s) {
        const int a = nb[b][s];
        const double* blk = &Eblk[blk_off[b] + (size_t)s * MM * MM];     // [k][r]
        for (int cc = 0; cc < 4; ++cc)
          for (int k = 0; k < MM; ++k) { double t = 0; for (int r = 0; r < MM; ++r) t += hat[cc * MM + r] * blk[(size_t)k * MM + r]; T[cc * MM + k] = t; }
        for (int cc = 0; cc < 4; ++cc)
          for (int c2 = 0; c2 < 4; ++c2) {
            double t = 0;
            for (int k = 0; k < MM; ++k) t += T[cc * MM + k] * hat[c2 * MM + k];
            Ac[(size_t)evert[a * 4 + cc] * nvert + evert[b * 4 + c2]] += t;
          }
      }
    // leading dimension of the dense coarse inverse: a multiple of 768 (= three 256-column blocks) where the in-LDS coarse solve
    // runs (lda <= 3072), so that the number of column blocks is one of the COMPILE-TIME sizes 3 / 6 / 9 / 12 of k_schwarz_uc's
    // coarse role (run-time loop bounds cost that kernel 70 registers and its load batching: round 6); zero padded
    int lda = ((nvert + 255) / 256) * 256;
    if (((nvert + 767) / 768) * 768 <= 3072) lda = ((nvert + 767) / 768) * 768;
    d.coarse_lda = lda; c->coarse_lda = lda;
    if (lda > 3072 && (rc = dalloc(c, &c->rc_big, lda))) return rc;
    if (lda <= 3072 && (rc = dalloc(c, &d.rch, (size_t)MAXMR * lda))) return rc;     // restriction history (k_update_coarse)
    if (c->local) {
      // rank-local set-up: the rows of the vertices this rank owns (owner = the rank of the lowest element at the vertex; all
      // elements at it, their neighbours and the mass of their nodes are inside the two rings, so these rows are complete);
      // the inverse is built from ALL ranks' rows in nsk_local_finish
      for (int u = 0; u < nvert; ++u) {
        if (v_off[u + 1] == v_off[u] || !c->local_own[v_ent[v_off[u]] / 4]) continue;
        for (int v = 0; v < nvert; ++v) {
          const double a = Ac[(size_t)u * nvert + v];
          if (a != 0.0) { c->crow_u.push_back(u); c->crow_v.push_back(v); c->crow_a.push_back(a); }
        }
      }
    } else if ((rc = coarse_dense_inverse(c, Ac, cs.has_outflow != 0))) return rc;
  }
  {
    std::vector<int> vtab((size_t)nvert * CVT, -1);
    for (int v = 0; v < nvert; ++v) {
      const int n = v_off[v + 1] - v_off[v];
      if (n == 0 && c->local) continue;               // (a vertex outside this rank's sub-mesh)
      if (n < 1 || n > CVT) return fail(NSK_EINVAL, "vertex valence outside 1.." + std::to_string(CVT));
      for (int k = 0; k < n; ++k) vtab[(size_t)v * CVT + k] = v_ent[v_off[v] + k];
    }
    if ((rc = dupload(c, &d.vtab, vtab))) return rc;
    // vertex-major slots of the element-corner restrictions (Dev::ecv): the inverse of vtab
    if (CVT == 8 && lda_ok_for_ecv(c)) {
      std::vector<int> ecslot((size_t)nel * 4, 0);
      for (int v = 0; v < nvert; ++v)
        for (int k = 0; k < v_off[v + 1] - v_off[v]; ++k) ecslot[v_ent[v_off[v] + k]] = v * 8 + k;
      if ((rc = dupload(c, &d.ecslot, ecslot)) || (rc = dalloc(c, &d.ecv, (size_t)8 * d.coarse_lda))) return rc;
    }
  }
  // ---- round 6 (k_schwarz_uc / k_divgs_t): the image of the coarse prolongation under E, block sparse, from the probed E blocks:
  //   Tc[a][s][r] = sum_{b in nb(a)} sum_{c: evert[b][c] = evl[a][s]} sum_k E[(a,r),(b,k)] hat[c][k]
  if (lda_ok_for_ecv(c) && d.ecv) {
    std::vector<std::vector<int>> vl(nel);
    int nvl = 0;
    for (int a = 0; a < nel; ++a) {
      std::vector<int>& v = vl[a];
      for (int b : nb[a]) for (int cc = 0; cc < 4; ++cc) v.push_back(evert[(size_t)b * 4 + cc]);
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
      nvl = std::max(nvl, (int)v.size());
    }
    if (nvl <= nsk::k2::UC_NVL && nsk::k2::UC_NVL <= NN) {          // (k_divgs_t holds exactly nsk::k2::UC_NVL = 20 columns in registers: vertices of valence > 5 take the three-launch form)
      nvl = nsk::k2::UC_NVL;
      std::vector<int> evl((size_t)nel * nvl, 0);
      std::vector<double> Tc((size_t)nel * nvl * MM, 0.0);
      const int nth = std::max(1, std::min<int>(16, (int)std::thread::hardware_concurrency()));
      // deterministic (fixed summation order): one thread per TARGET element a, contributions in the order of nb[a] and corner index
      auto rows_a = [&](int t0) {
        for (int a = t0; a < nel; a += nth)
          for (int b : nb[a]) {
            size_t sb = 0;
            while (sb < nb[b].size() && nb[b][sb] != a) ++sb;
            if (sb == nb[b].size()) continue;                                      // (adjacency is symmetric: not reached)
            const double* blk = &Eblk[blk_off[b] + sb * MM * MM];
            for (int cc = 0; cc < 4; ++cc) {
              const int v = evert[(size_t)b * 4 + cc];
              const int slot = (int)(std::lower_bound(vl[a].begin(), vl[a].end(), v) - vl[a].begin());
              double* dst = &Tc[((size_t)a * nvl + slot) * MM];
              for (int r = 0; r < MM; ++r) { double t = 0; for (int k = 0; k < MM; ++k) t += blk[(size_t)k * MM + r] * hat[cc * MM + k]; dst[r] += t; }
            }
          }
      };
      {
        std::vector<std::thread> pool;
        for (int t = 1; t < nth; ++t) pool.emplace_back(rows_a, t);
        rows_a(0);
        for (auto& th : pool) th.join();
      }
      for (int a = 0; a < nel; ++a) for (size_t k = 0; k < vl[a].size(); ++k) evl[(size_t)a * nvl + k] = vl[a][k];
      d.nvl = nvl;
      std::vector<float> Tc32(Tc.begin(), Tc.end());
      if ((rc = dupload(c, &d.evl, evl)) || (rc = dupload(c, &d.Tc, Tc)) || (rc = dupload(c, &d.Tc32, Tc32)) || (rc = dalloc(c, &d.Wr, (size_t)d.ps))) return rc;
    }
  }
  tick("coarse image under E (Tc)");
  c->h_evert = evert;
  if ((rc = dupload(c, &d.v_off, v_off)) || (rc = dupload(c, &d.v_ent, v_ent)) || (rc = dupload(c, &d.evert, evert)) ||
      (rc = dalloc(c, &d.xc, nvert))) return rc;

  tick("coarse operator + its inverse");
  // ---- restricted additive Schwarz patches: own Gauss nodes + `layers` rows of every neighbour
  {
    const int L = c->layers;
    const int PS = (((M + 2 * L) * (M + 2 * L) + 3) / 4) * 4;
    if (PS > 2 * NN) return fail(NSK_EINVAL, "schwarz_layers too large for this lx1");
    d.p_stride = PS;
    std::vector<int> p_idx((size_t)nel * PS, -1);
    std::vector<float> p_inv((size_t)nel * PS * MM, 0.0f);
    // one dense inverse of ~(lx2 + 2 layers)^2 unknowns per element: 15 s on one host thread at config 3's size -> host threads
    const int nth = std::max(1, std::min<int>(16, (int)std::thread::hardware_concurrency()));
    std::atomic<int> singular{0};
    auto patch_range = [&](int t0) {
    std::vector<int> pe, pr;            // patch dof -> (element, local Gauss index)
    std::vector<double> A;
    for (int e = t0; e < nel; e += nth) {
      pe.clear(); pr.clear();
      for (int k = 0; k < MM; ++k) { pe.push_back(e); pr.push_back(k); }
      for (size_t s = 1; s < nb[e].size(); ++s) {
        const int f = nb[e][s];
        int jmin = N, jmax = -1, imin = N, imax = -1;
        for (int nd = 0; nd < NN; ++nd) {
          const long long l = (long long)f * NN + nd;
          bool sh = false;
          for (int k = gs_off[l]; k < gs_off[l + 1] && !sh; ++k) sh = (gs_idx[k] / NN == e);
          if (sh) { jmin = std::min(jmin, nd / N); jmax = std::max(jmax, nd / N); imin = std::min(imin, nd % N); imax = std::max(imax, nd % N); }
        }
        if (jmax < 0) continue;
        int b0 = 0, b1 = M, a0 = 0, a1 = M;
        if (jmin == jmax) { if (jmin == 0) b1 = L; else if (jmin == N - 1) b0 = M - L; }
        if (imin == imax) { if (imin == 0) a1 = L; else if (imin == N - 1) a0 = M - L; }
        if (b0 == 0 && b1 == M && a0 == 0 && a1 == M) continue;    // degenerate adjacency (wraps both ways)
        for (int b = b0; b < b1; ++b) for (int a = a0; a < a1; ++a) { pe.push_back(f); pr.push_back(b * M + a); }
      }
      int np = (int)pe.size();
      if (np > PS) { np = PS; pe.resize(PS); pr.resize(PS); }       // irregular vertex: drop the excess overlap
      A.assign((size_t)np * np, 0.0);
      for (int q = 0; q < np; ++q)
        for (int r = 0; r < np; ++r) A[(size_t)r * np + q] = Eentry(pe[r], pr[r], pe[q], pr[q]);
      if (!cs.has_outflow) { double tr = 0; for (int q = 0; q < np; ++q) tr += A[(size_t)q * np + q]; for (int q = 0; q < np; ++q) A[(size_t)q * np + q] += 1e-10 * tr / np; }
      if (!invert_dense(A, np)) { singular = 1; return; }
      for (int q = 0; q < np; ++q) {
        p_idx[(size_t)e * PS + q] = pe[q] * MM + pr[q];
        for (int r = 0; r < MM; ++r) p_inv[(size_t)e * PS * MM + ((size_t)(q / 4) * MM + r) * 4 + (q % 4)] = (float)A[(size_t)r * np + q];   // [q/4][own row][q%4]
      }
    }
    };
    {
      std::vector<std::thread> pool;
      for (int t = 1; t < nth; ++t) pool.emplace_back(patch_range, t);
      patch_range(0);
      for (auto& th : pool) th.join();
    }
    if (singular) return fail(NSK_EINVAL, "singular Schwarz patch");
    c->h_pidx = p_idx; c->PS = PS;
    if ((rc = dupload(c, &d.p_idx, p_idx)) || (rc = dupload(c, &d.p_inv, p_inv))) return rc;
  }
  tick("Schwarz patches (host inversions)");
  // many workgroups (lx1 = 12 has one element per workgroup: config 3 has 7984): sum every row of partials once
  // (k_tot2) instead of in every consumer workgroup, which is O(nblk^2)
  if (c->nblk > 1024 || std::getenv("NSK_USE_TOT")) {
    d.use_tot = 1;
    if ((rc = dalloc(c, &d.htot, 32)) || (rc = dalloc(c, &d.gtot, MAXMR + 8)) || (rc = dalloc(c, &d.gtot2, MAXMR + 8)) || (rc = dalloc(c, &d.ptot, MAXPROJ + 2))) return rc;
  }
  for (int k = 0; k < NCLS; ++k) { c->cur_helm[k] = c->max_helm; c->cur_pres[k] = c->max_pres; }
  c->bh_n = 0;
  if ((rc = dalloc(c, &c->sync, 2 * SYNC_WORDS))) return rc;      // two sets: the tails alternate (nsk_persist.hpp)
  // persistent velocity solve: available where the grid is resident, OFF by default -- measured on config 2 it is not faster
  // than the launch-per-iteration form (13.6 vs 13.5 us per CG iteration: DESIGN.md section 5); option "fused" / NSK_FUSED=1
  c->fused = 0;
  if (const char* g = std::getenv("NSK_FUSED")) c->fused = std::atoi(g) && fused_possible(c);
  if (const char* g = std::getenv("NSK_USE_GRAPH")) c->use_graph = std::atoi(g);
  if (const char* g = std::getenv("NSK_MERGED_UPDATE")) c->merged_update = std::atoi(g);
  if (const char* g = std::getenv("NSK_FUSE2")) c->fuse2 = std::atoi(g);
  if (const char* g = std::getenv("NSK_TC32")) c->tc32 = std::atoi(g);
  if (const char* g = std::getenv("NSK_FUSE2_START")) c->fuse2_start = std::atoi(g);
  if (const char* g = std::getenv("NSK_SB_PCT")) c->sb_pct = std::atoi(g);
  if (const char* g = std::getenv("NSK_SKIP_CLOSE")) c->skip_close = std::atoi(g);
  if (const char* g = std::getenv("NSK_HOSTCHECK")) c->hostcheck = std::atoi(g);
  if (const char* g = std::getenv("NSK_GRAPH_STEPS")) c->graph_steps = std::max(1, std::min(std::atoi(g), 64));
  if (const char* g = std::getenv("NSK_MERGED_ITERS")) c->merged_iters = std::max(0, std::min(std::atoi(g), MAXMR));
  if (const char* g = std::getenv("NSK_DEBUG")) c->debug = std::atoi(g);
  if (const char* g = std::getenv("NSK_STEP_BUDGETS")) c->step_budgets = std::atoi(g);
  if (const char* g = std::getenv("NSK_TAIL")) c->tail = std::atoi(g);
  if (const char* g = std::getenv("NSK_TAIL_OFF_H")) c->tail_off_h = std::atoi(g);
  if (const char* g = std::getenv("NSK_TAIL_OFF_P")) c->tail_off_p = std::atoi(g);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

static int scan_bfmask(nsk_ctx* c);
#include "nsk3_setup.inc"

// ---------------------------------------------------------------------------
// one nek_advance() in perturbation mode
// ---------------------------------------------------------------------------
// hexahedral contexts: one workgroup sums each row of per-workgroup partials (Dev::use_tot)
static inline void tot_rows(nsk_ctx* c, const double* part, int rows, double* tot, const int* gate = nullptr) {
  if (c->d.use_tot) hipLaunchKernelGGL(k_tot2, dim3(rows), dim3(256), 0, c->stream, part, c->nblk, tot, gate);
}

// tolerance of the early time steps of a map: tightened, but never below 1e-4 (what 48 GMRES iterations deliver on every mesh)
static inline double early_tol(const Dev& d, double mul) {
  return d.tol_relative ? std::max(d.tol_pres * mul, std::min(d.tol_pres, 1e-4)) : d.tol_pres;
}

// k_update_coarse<MAXIT>: the smallest table size that holds coarse_lda / 256 column blocks
static void launch_update_coarse(nsk_ctx* c, const Dev& d, int j, double scale, int min_iter, int ord) {
  const unsigned cgrid = (d.nvert + 4 * UC_ROWS - 1) / (4 * UC_ROWS);
  const size_t sh = d.coarse_lda * sizeof(double);
  const int nit = d.coarse_lda / 256;
  if (nit <= 3) hipLaunchKernelGGL(k_update_coarse<3>, dim3(cgrid), dim3(256), sh, c->stream, d, j, scale, min_iter, ord);
  else if (nit <= 6) hipLaunchKernelGGL(k_update_coarse<6>, dim3(cgrid), dim3(256), sh, c->stream, d, j, scale, min_iter, ord);
  else if (nit <= 9) hipLaunchKernelGGL(k_update_coarse<9>, dim3(cgrid), dim3(256), sh, c->stream, d, j, scale, min_iter, ord);
  else hipLaunchKernelGGL(k_update_coarse<12>, dim3(cgrid), dim3(256), sh, c->stream, d, j, scale, min_iter, ord);
}

// round 6, the merged iteration in two launches: A_j (Schwarz workgroups + coarse workgroups) and B_j
// (needs: the coarse image Tc with nsk::k2::UC_NVL columns, coarse_lda a multiple of 768, two overlap layers of the Schwarz patches)
static bool fuse2_on(const nsk_ctx* c) {
  if (!(c->fuse2 && c->d.Tc && c->d.Wr && c->d.ecv && c->d.rch && c->ndim == 2)) return false;
  if (c->d.nvl != nsk::k2::UC_NVL || c->d.coarse_lda % 768 != 0 || c->d.coarse_lda > 3072) return false;
  const int M = c->N - 2;
  return c->d.p_stride == (((M + 4) * (M + 4) + 3) / 4) * 4;
}
template <int N>
static void launch_schwarz_uc(nsk_ctx* c, const Dev& d, int j, double scale, int min_iter, int ord) {
  if (c->ndim != 2) return;
  const unsigned cgrid = (d.nvert + 4 * UC_ROWS - 1) / (4 * UC_ROWS), nsw = (unsigned)c->nblk;
  const size_t sh = d.coarse_lda * sizeof(double);
  const int nit = d.coarse_lda / 256;
  const dim3 grid(nsw + cgrid), blk(256);
  // (fuse2_on: nit is exactly 3, 6, 9 or 12)
  if (nit == 3) hipLaunchKernelGGL((nsk::k2::k_schwarz_uc<N, 3>), grid, blk, sh, c->stream, d, j, scale, min_iter, ord, nsw, cgrid);
  else if (nit == 6) hipLaunchKernelGGL((nsk::k2::k_schwarz_uc<N, 6>), grid, blk, sh, c->stream, d, j, scale, min_iter, ord, nsw, cgrid);
  else if (nit == 9) hipLaunchKernelGGL((nsk::k2::k_schwarz_uc<N, 9>), grid, blk, sh, c->stream, d, j, scale, min_iter, ord, nsw, cgrid);
  else if (nit == 12) hipLaunchKernelGGL((nsk::k2::k_schwarz_uc<N, 12>), grid, blk, sh, c->stream, d, j, scale, min_iter, ord, nsw, cgrid);
}
template <int N>
static void launch_proj_apply_e(nsk_ctx* c, const Dev& d) {
  if (c->ndim != 2) return;
  hipLaunchKernelGGL(nsk::k2::k_proj_apply_e<N>, dim3(c->nblk), dim3(256), 0, c->stream, d);
}
template <int N>
static void launch_divgs_t(nsk_ctx* c, const Dev& d, int j) {
  if (c->ndim != 2) return;
  if (d.tc32) hipLaunchKernelGGL((nsk::k2::k_divgs_t<N, true>), dim3(c->nblk), dim3(256), 0, c->stream, d, j);
  else hipLaunchKernelGGL((nsk::k2::k_divgs_t<N, false>), dim3(c->nblk), dim3(256), 0, c->stream, d, j);
}

static bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}
// host-checked convergence (options "hostcheck" / "shard_hostcheck"): the device's convergence flags, read by the host --
// which = 0: every component of the velocity solve after launch `it`; which = 1: the pressure GMRES
static int flags_done(nsk_ctx* c, int which, int it, bool* done) {
  c->hc_checks++;
  if (which == 0) {
    HIPCHK(hipMemcpyAsync(c->hpin, c->d.hscal + (size_t)(it & 1) * c->hstride, c->hstride * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    bool all = true;
    for (int cc = 0; cc < c->ndim; ++cc) all = all && c->hpin[cc * 4 + 2] != 0.0;
    *done = all;
  } else {
    int* flag = (int*)c->hpin;
    HIPCHK(hipMemcpyAsync(flag, &c->d.gsc->done, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *done = *flag != 0;
  }
  return 0;
}
static bool hostcheck_on(const nsk_ctx* c) {
  if (c->parent || c->fused || c->in_test) return false;
  if (c->hostcheck >= 0) return c->hostcheck != 0;
  // quadrilateral meshes of thousands of workgroups (config 3: 7984 at lx1 = 12, kernels of ~65 us): a flag read per solve costs
  // less than the launches a budget spends on converged solves, and budgets that cannot overflow mean no redone maps
  // (measured: 0.262 -> 0.334 Arnoldi steps per second, scripts/cfg3_hostcheck_ab.sh)
  if (c->ndim == 2) return c->nblk > 4096;
  return c->nel >= 8192;
}

// hexahedral GMRES column j in the lagged form: streaming Gram-Schmidt pass (first-pass subtraction, second-pass dots, the
// pending correction of v_j, corner restriction of w'), totals, Hessenberg column.  JB = compile-time bound of j, RB = rows
// of 64 nodes in flight per wavefront.
static bool gs_lag_on(const nsk_ctx* c) {
  if (c->ndim != 3 || c->N > 10) return false;
  if (c->gs_lag >= 0) return c->gs_lag != 0;
  return c->d.nranks <= 1 && !c->parent;
}
// E-apply kernels of the hexahedral pressure iteration as resident workgroups (nsk3_kernels.hpp: k_schwarz_p): as many
// workgroups as the device holds at once, a multiple of 8 so that a workgroup's stride stays inside its XCD's run of elements.
template <int N>
static unsigned eapply_grid(nsk_ctx* c, int which, int count) {
  if constexpr (N <= 10) {
    if (!c->eapply_grid[which]) {
      int per_cu = 0, ncu = 0, dev = 0;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
      if (which == 0) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nsk::k3::k_schwarz_p<N>, nsk::k3::Cfg<N>::NT, 0);
      if (const char* g = std::getenv("NSK_EAPPLY_WGS")) per_cu = std::atoi(g);
      c->eapply_grid[which] = std::max(8, (std::max(1, per_cu) * std::max(1, ncu)) / 8 * 8);
    }
  }
  return (unsigned)std::min(count, c->eapply_grid[which]);
}
#ifndef NSK_SCHW10_MODE
#define NSK_SCHW10_MODE 5             // what "the default form" means at lx1 = 10: four wavefronts per element (k_schwarz_q<10>: 263 us at 13 824 elements; k_schwarz_p<10> 596, k_schwarz<10> 679)
#endif
template <int N>
static void launch_schwarz3(nsk_ctx* c, const Dev& d, int count, const double* vin, double* zout, int use_coarse, int check_done, int mode = -1) {
  if constexpr (N <= 10) {
    if (mode < 0) mode = c->eapply_pipe;
    if (mode == 3) mode = 2;                             // (3 = 2 + the divergence kernel in wavefront form: launch_divgs3)
    if (mode == 2 && N > 8 && !std::getenv("NSK_WAVE_LX10")) mode = 0;      // lx1 = 10: 16 nodes per lane, 234 registers: not measured faster
    if (mode == 5 && N != 10) mode = 4;                  // (5 = four wavefronts per element: lx1 = 10 only)
    if (mode == 4 && N > 8) mode = NSK_SCHW10_MODE;      // lx1 = 10 has no sixteen-per-CU wavefront form (sixteen nodes per lane: 224 registers)
    if (mode == 5) {                                     // four wavefronts per element (lx1 = 10; elsewhere the default form)
      if constexpr (N == 10) {
        hipLaunchKernelGGL(nsk::k3::k_schwarz_q<N>, dim3(count), dim3(256), 0, c->stream, d, vin, zout, use_coarse, check_done);
        return;
      }
    }
    if (mode == 4) {          // one wavefront per element, sixteen per CU (in-place solve, metrics per component)
      if constexpr (N <= 8) hipLaunchKernelGGL(nsk::k3::k_schwarz_w16<N>, dim3(count), dim3(64), 0, c->stream, d, vin, zout, use_coarse, check_done);
    } else if (mode == 2)     // one wavefront per element
      hipLaunchKernelGGL(nsk::k3::k_schwarz_w<N>, dim3(count), dim3(64), 0, c->stream, d, vin, zout, use_coarse, check_done);
    else if (mode == 1)       // resident workgroups, next element's loads in flight
      hipLaunchKernelGGL(nsk::k3::k_schwarz_p<N>, dim3(eapply_grid<N>(c, 0, count)), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, vin, zout,
                         use_coarse, check_done, count);
    else
      hipLaunchKernelGGL(nsk::k3::k_schwarz<N>, dim3(count), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, vin, zout, use_coarse, check_done);
  }
}
// E apply without the Gram-Schmidt dots (j < 0): one wavefront per element where that form exists (lx1 <= 8), else k_divgs
template <int N>
static void launch_divgs3(nsk_ctx* c, const Dev& d, int count, const double* yl, double* wout, int j, int check_done, int mode = -1, int c3 = -1) {
  if constexpr (N <= 10) {
    if (mode < 0) mode = c->eapply_pipe;
    if constexpr (N <= 8) {
      // (measured slower than k_divgs at config 4's size, 997 against 682 us: eight nodes per lane make the gather a chain of seven
      //  dependent round trips where the 512-thread form has two; kept for the record, reachable with mode 3 / NSK_DIVGS_WAVE=1)
      static const bool dv_wave = std::getenv("NSK_DIVGS_WAVE") && std::atoi(std::getenv("NSK_DIVGS_WAVE")) != 0;
      if ((mode == 3 || (mode == 2 && dv_wave)) && j < 0) {
        hipLaunchKernelGGL(nsk::k3::k_divgs_w<N>, dim3(count), dim3(64), 0, c->stream, d, yl, wout, check_done);
        return;
      }
    }
    if (c3 < 0) c3 = c->divgs_c3 >= 0 ? c->divgs_c3 : (N > 8 ? 1 : 0);
    if (c3) hipLaunchKernelGGL(nsk::k3::k_divgs_c3<N>, dim3(count), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, yl, wout, j, check_done);
    else hipLaunchKernelGGL(nsk::k3::k_divgs<N>, dim3(count), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, yl, wout, j, check_done);
  }
}
template <int N>
static void launch_gs_dots3(nsk_ctx* c, const Dev& d, int j) {         // first-pass dots as their own streaming pass (gs_lag = 2)
  if constexpr (N <= 10) {
    constexpr int ROWS = (nsk::k3::Cfg<N>::MM + 63) / 64;
    constexpr int R4 = ROWS < 4 ? ROWS : 4, R2 = ROWS < 2 ? ROWS : 2;
    const dim3 grid(std::min<unsigned>((unsigned)((c->nel + 3) / 4), 2048u)), blk(256);
    if (j < 8) hipLaunchKernelGGL((nsk::k3::k_gs_dots<N, 8, R4>), grid, blk, 0, c->stream, d, j);
    else if (j < 16) hipLaunchKernelGGL((nsk::k3::k_gs_dots<N, 16, R2>), grid, blk, 0, c->stream, d, j);
    else if (j < 32) hipLaunchKernelGGL((nsk::k3::k_gs_dots<N, 32, 1>), grid, blk, 0, c->stream, d, j);
    else hipLaunchKernelGGL((nsk::k3::k_gs_dots<N, MAXMR, 1>), grid, blk, 0, c->stream, d, j);
  }
}
template <int N>
static void launch_gs_lag3(nsk_ctx* c, const Dev& d, int j, double scale, int ord) {
  if constexpr (N <= 10) {
    constexpr int ROWS = (nsk::k3::Cfg<N>::MM + 63) / 64;
    constexpr int R4 = ROWS < 4 ? ROWS : 4, R2 = ROWS < 2 ? ROWS : 2;
    const dim3 grid(std::min<unsigned>((unsigned)((c->nel + 3) / 4), 2048u)), blk(256);
    if (j <= 8) hipLaunchKernelGGL((nsk::k3::k_gs_lag<N, 8, R4>), grid, blk, 0, c->stream, d, j);
    else if (j <= 16) hipLaunchKernelGGL((nsk::k3::k_gs_lag<N, 16, R2>), grid, blk, 0, c->stream, d, j);
    else if (j <= 32) hipLaunchKernelGGL((nsk::k3::k_gs_lag<N, 32, 1>), grid, blk, 0, c->stream, d, j);
    else hipLaunchKernelGGL((nsk::k3::k_gs_lag<N, MAXMR, 1>), grid, blk, 0, c->stream, d, j);
    tot_rows(c, d.gpart2, j + 2, d.gtot2, &d.gsc->done);
    hipLaunchKernelGGL(nsk::k3::k_gmres_col, dim3(1), dim3(64), 0, c->stream, d, j, scale, c->min_pres, ord);
  }
}

// resident workgroups of k_helm_p<10>: what the device holds at once, a multiple of 8 (a workgroup's stride stays inside its XCD's run)
static int helm_pf_grid(nsk_ctx* c) {
  if (!c->helm_pf_grid) {
    int per_cu = 0, ncu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nsk::k3::k_helm_p<10>, nsk::k3::Cfg<10>::NT, 0);
    if (const char* g = std::getenv("NSK_HELM_PF_WGS")) per_cu = std::atoi(g);
    c->helm_pf_grid = std::max(8, (std::max(1, per_cu) * std::max(1, ncu)) / 8 * 8);
  }
  return c->helm_pf_grid;
}
// one CG iteration of the velocity solve: k_helm, or (hexahedra with Dev::helm_fdm) the two launches of the block-preconditioned form
template <int N>
static void launch_helm_iter(nsk_ctx* c, const Dev& d, const StepCoef& sc, int it, const double* rhs) {
  if constexpr (N <= 10) {
    if (c->ndim == 3 && d.helm_fdm) {
      hipLaunchKernelGGL(nsk::k3::k_helm_fa<N>, dim3(c->nblk), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, sc, it, rhs);
      hipLaunchKernelGGL(nsk::k3::k_helm_fb<N>, dim3(c->nblk), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, sc, it);
      return;
    }
  }
  if (c->ndim == 3) {
    if constexpr (N == 10) {
      // lx1 = 10: resident workgroups with the next element's r, p, s, x arriving in LDS under the current element's A z (k_helm_p;
      // it = 0 stays k_helm's helm_first)
      if (it >= 1 && c->helm_pf != 0 && d.boff == 0) {
        hipLaunchKernelGGL(nsk::k3::k_helm_p<N>, dim3(std::min(c->nblk, helm_pf_grid(c))), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, sc, it, c->nblk);
        return;
      }
    }
    if constexpr (N <= 10) hipLaunchKernelGGL(nsk::k3::k_helm<N>, dim3(c->nblk), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, sc, it, rhs);
  } else {
    hipLaunchKernelGGL(nsk::k2::k_helm<N>, dim3(c->nblk), dim3(nsk::k2::Cfg<N>::NT), 0, c->stream, d, sc, it, rhs);
  }
}

static bool flat_proj_on(const nsk_ctx* c) {
  if (c->ndim != 3 || c->N > 10 || !c->d.dpw) return false;
  if (c->flat_proj >= 0) return c->flat_proj != 0;
  return c->d.nranks <= 1 && !c->parent;
}
template <int N>
static void launch_pres_comb3(nsk_ctx* c, const Dev& d, const StepCoef& sc) {        // dp, p and PD in one streaming pass, then yl = D^T dp
  if constexpr (N <= 10) {
    const dim3 grid(std::min<unsigned>((unsigned)((c->nel + 3) / 4), 2048u)), blk(256);
    hipLaunchKernelGGL(nsk::k3::k_pres_comb<N>, grid, blk, 0, c->stream, d, sc);
    hipLaunchKernelGGL(nsk::k3::k_gradt<N>, dim3(c->nblk), dim3(nsk::k3::Cfg<N>::NT), 0, c->stream, d, (const double*)d.dpw, d.yl);
  }
}
template <int N>
static void launch_proj_dots3(nsk_ctx* c, const Dev& d) {
  if constexpr (N <= 10) {
    const dim3 grid(std::min<unsigned>((unsigned)((c->nel + 3) / 4), 2048u)), blk(256);
    hipLaunchKernelGGL(nsk::k3::k_proj_dots<N>, grid, blk, 0, c->stream, d);
  }
}

template <int N> static void launch_pres_tail(nsk_ctx* c, const Dev& d, int j0, int j1, double scale, int min_iter, int ord, bool start = false);
static int pres_solve_launch(nsk_ctx* c, double h2, int ord, int np, double tol_mul = 1.0, bool allow_cap = false, bool hc = false, bool tail = false) {
  Dev d = c->d;                                            // by value: the early steps of a map run with a tighter tolerance
  d.tol_pres = early_tol(d, tol_mul);
  // time steps >= 4: optionally a bounded solve (min_pres_iter .. pres_cap iterations): nothing is launched beyond the cap
  if (c->pres_cap > 0 && ord >= 3 && allow_cap && c->ndim == 2) {   // validated on quadrilateral linearised maps only
    d.pres_cap = std::max(c->pres_cap, c->min_pres); np = std::min(np, d.pres_cap);
  }
  const double scale = 1.0 / (h2 * std::sqrt(d.vol));
  const bool lag = gs_lag_on(c);
  d.gs_lag = lag ? 1 : 0;
  DISPATCH_N(c->key, {
    constexpr int NT = Cfg<N>::NT;
    tot_rows(c, d.gpart, d.has_outflow ? 1 : 2, d.gtot);
    if (d.nproj_max > 0 && !c->in_test) tot_rows(c, d.ppart, MAXPROJ + 1, d.ptot);
    if (!d.has_outflow && !c->in_test) { hipLaunchKernelGGL(k_ortho, dim3(c->nblk), dim3(256), 0, c->stream, d); tot_rows(c, d.gpart, 1, d.gtot); }
    // round 6: with the two-launch iteration the solve starts INSIDE A_0 (element-aligned projection kernel -> raw g' in Wr
    // + its corner restrictions; no k_gmres_update(-1) launch) where the projection kernel is the last producer of g'
    const bool merged0 = c->merged_update && c->ndim == 2 && d.coarse_lda <= 3072 && !d.use_tot && d.nranks <= 1 && d.rch && d.ecv && !hc;
    // (the choice must not depend on the launch budgets or on the tail mode: the two projection kernels sum their partials in
    //  different groups, so a step that switched form with its budget would change bits between otherwise identical runs)
    const bool f2start = merged0 && fuse2_on(c) && c->fuse2_start && d.nproj_max > 0 && d.nproj_max <= MAXPROJ && !c->in_test && np > 0 &&
                         std::min(c->merged_iters, c->gmres_cycle) > 0;
    if (f2start) launch_proj_apply_e<N>(c, d);
    else if (d.nproj_max > 0 && !c->in_test) { hipLaunchKernelGGL(k_proj_apply, dim3(c->nblk), dim3(256), 0, c->stream, d); tot_rows(c, d.gpart, 1, d.gtot); }
    if (!f2start) hipLaunchKernelGGL(k_gmres_update<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, -1, scale, c->min_pres, ord);
    // merged bookkeeping (k_update_coarse): quadrilateral single-rank contexts with the dense in-LDS coarse solve.
    // Only the first `merged_iters` (24) iterations of a solve: x_c(v_j) by linearity is a recurrence, its rounding error grows by
    // ~|h_jj / h_{j+1,j}| per iteration (harmless in a preconditioner for a dozen iterations, a stalled solve after forty:
    // measured on the 1e-8 solves of test_newton_gpu); later iterations take the classic four kernels.
    const bool merged = c->merged_update && c->ndim == 2 && d.coarse_lda <= 3072 && !d.use_tot && d.nranks <= 1 && d.rch && d.ecv && !hc;
    // with a persistent tail `np` is the HEAD (merged iterations launched one by one); the tail covers the rest of the merged
    // range in one launch, and iterations beyond it (classic form) keep the class budget
    const bool tl = tail && merged && d.pres_cap == 0;
    const int nhead = tl ? std::max(0, std::min(np, std::min(c->merged_iters, c->gmres_cycle) - 1)) : 0;
    if (tl) np = std::max(c->cur_pres[ord], std::min(c->max_pres, std::min(c->merged_iters, c->gmres_cycle)));
    const int nm = merged ? std::min(np, std::min(c->merged_iters, c->gmres_cycle)) : 0;
    if (hc && c->hc_pres[ord] <= 1) {                     // the projection space may have solved this right-hand side alone
      bool done = false;
      int rc2 = flags_done(c, 1, 0, &done);
      if (rc2) return rc2;
      if (done) { c->hc_pres[ord] = 0; np = 0; }
    }
    const bool f2 = merged && fuse2_on(c);
    // the fp32 copy of the coarse image where the tolerance of THIS solve is loose (relative, >= 1e-5): option "tc32" = -1 (default) / 0 / 1
    d.tc32 = (c->tc32 < 0) ? ((d.tol_relative && d.tol_pres >= 1e-5 && d.Tc32) ? 1 : 0) : ((c->tc32 && d.Tc32) ? 1 : 0);
    for (int j = 0; j < (tl ? nhead : nm); ++j) {
      if (f2) {                                            // two launches per iteration (round 6)
        Dev d0 = d; d0.uc_start = (f2start && j == 0) ? 1 : 0;
        launch_schwarz_uc<N>(c, d0, j, scale, c->min_pres, ord);
        launch_divgs_t<N>(c, d, j);
        continue;
      }
      launch_update_coarse(c, d, j, scale, c->min_pres, ord);
      hipLaunchKernelGGL(k_schwarz<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)(d.V + (size_t)j * d.ps), d.Z + (size_t)j * d.npr, 1, 1);
      hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)d.yl, d.V + (size_t)(j + 1) * d.ps, j, 2);
    }
    if (tl && nhead < nm) launch_pres_tail<N>(c, d, nhead, nm, scale, c->min_pres, ord, f2start && nhead == 0);
    // a velocity tail ran in this step but no pressure tail will: the velocity tail's barrier words (set 0) are re-zeroed by
    // the pressure tail only, so zero them here (a dirty set lets the next velocity tail's first barriers fall through)
    if (tail && !(tl && nhead < nm) && c->sync) (void)hipMemsetAsync(c->sync, 0, SYNC_WORDS * sizeof(unsigned), c->stream);
    // closes the last merged column (normalises v_nm and writes its corner restriction: what the classic iteration nm reads)
    // (behind a persistent tail that covers the whole merged range, with no classic iteration budgeted, nothing is left to close: a solve
    //  that is not finished after the tail's last iteration is unconverged whatever the closing launch records -- option "skip_close")
    if (nm > 0 && !(c->skip_close && tl && f2 && np <= nm)) {
      if (f2 && nm < std::min(c->merged_iters, c->gmres_cycle)) {
        // two-launch form, budget below the merged range: the last column is closed by A_nm -- the SAME code (uc_rotate) that closes
        // every other column and that the persistent tail runs, so that a solve of exactly nm iterations ends with the same bits
        // whatever budget or tail mode ended it (k_gmres_update's own copy of the rotation differs from it by rounding)
        launch_schwarz_uc<N>(c, d, nm, scale, c->min_pres, ord);
      } else {
        Dev dcl = d;
        if (f2) dcl.wraw = d.Wr;                           // the raw w of B_{nm-1}; classic iterations follow: they need v_nm normalised and its corner restrictions
        hipLaunchKernelGGL(k_gmres_update<N>, dim3(c->nblk), dim3(NT), 0, c->stream, dcl, nm - 1, scale, c->min_pres, ord);
      }
    }
    for (int jt = nm; jt < np; ++jt) {
      const int j = jt % c->gmres_cycle;                 // index inside the current GMRES cycle
      if (jt > 0 && j == 0) {                            // cycle full and not converged: restart on the residual
        hipLaunchKernelGGL(k_gmres_restart, dim3(c->nblk), dim3(256), 0, c->stream, d, c->gmres_cycle);
        tot_rows(c, d.gpart, 1, d.gtot, &d.gsc->done);
        hipLaunchKernelGGL(k_gmres_update<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, -2, scale, c->min_pres, ord);
      }
      if (c->ndim == 3) {
        hipLaunchKernelGGL(k_coarse_restrict_csr, dim3(d.coarse_lda / 256), dim3(256), 0, c->stream, d, c->rc_big);
        if (c->coarse_iter) { int rc2 = coarse_iterative(c, c->rc_big); if (rc2) return rc2; }
        else hipLaunchKernelGGL(k_coarse_big, dim3((d.nvert + 3) / 4), dim3(256), 0, c->stream, d, (const double*)c->rc_big);
      } else if (d.coarse_lda <= 3072) {
        hipLaunchKernelGGL(k_coarse, dim3((d.nvert + 4 * CROWS_W - 1) / (4 * CROWS_W)), dim3(256), d.coarse_lda * sizeof(double), c->stream, d);
      } else {
        hipLaunchKernelGGL(k_coarse_restrict, dim3(d.coarse_lda / 256), dim3(256), 0, c->stream, d, c->rc_big);
        hipLaunchKernelGGL(k_coarse_big, dim3((d.nvert + 3) / 4), dim3(256), 0, c->stream, d, (const double*)c->rc_big);
      }
      if (c->ndim == 3) launch_schwarz3<N>(c, d, c->nblk, (const double*)(d.V + (size_t)j * d.ps), d.Z + (size_t)j * d.npr, 1, 1);
      else hipLaunchKernelGGL(k_schwarz<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)(d.V + (size_t)j * d.ps), d.Z + (size_t)j * d.npr, 1, 1);
      if (lag && c->gs_lag != 1) {                        // (default, and gs_lag = 2: the first-pass dots as their own streaming pass)
        launch_divgs3<N>(c, d, c->nblk, (const double*)d.yl, d.V + (size_t)(j + 1) * d.ps, -1, 1);
        launch_gs_dots3<N>(c, d, j);
      } else {
        if (c->ndim == 3) launch_divgs3<N>(c, d, c->nblk, (const double*)d.yl, d.V + (size_t)(j + 1) * d.ps, j, 1);
        else hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)d.yl, d.V + (size_t)(j + 1) * d.ps, j, 1);
      }
      tot_rows(c, d.gpart, j + 2, d.gtot, &d.gsc->done);
      if (lag) {
        launch_gs_lag3<N>(c, d, j, scale, ord);
      } else {
        // second Gram-Schmidt pass: always on hexahedra, on quadrilaterals from iteration gs2_from of a cycle on
        const bool two = c->ndim == 3 || j >= c->gs2_from;
        if (two) {
          hipLaunchKernelGGL(k_gmres_reorth<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, j);
          tot_rows(c, d.gpart2, j + 2, d.gtot2, &d.gsc->done);
        }
        Dev d2 = d; d2.gs2 = two ? 1 : 0;
        hipLaunchKernelGGL(k_gmres_update<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d2, j, scale, c->min_pres, ord);
      }
      // large meshes: an iteration takes milliseconds, a flag read microseconds -> after every iteration; otherwise from three
      // iterations before the previous solve's count on
      if (hc && jt + 1 < np && (c->nel >= 8192 || jt + 4 >= c->hc_pres[ord])) {
        bool done = false;
        int rc2 = flags_done(c, 1, 0, &done);
        if (rc2) return rc2;
        if (done) { c->hc_pres[ord] = jt + 1; break; }
      }
    }
  });
  return 0;
}

// ---- persistent velocity solve (nsk_persist.hpp): only the quadrilateral kernel set has it
namespace nsk { namespace k3 { template <int N> __global__ void k_helm_fused(Dev, StepCoef, int, unsigned*) {} } }
template <int N>
static void launch_fused(nsk_ctx* c, const StepCoef& sc, int max_it = -1, const Dev* dd = nullptr) {
  if (c->ndim != 2) return;
  hipLaunchKernelGGL(nsk::k2::k_helm_fused<N>, dim3(c->nblk), dim3(nsk::k2::Cfg<N>::NT), 0, c->stream, dd ? *dd : c->d, sc,
                     max_it < 0 ? c->max_helm : max_it, c->sync);
}
// Can every workgroup of the grid be resident at once?  (grid barrier => all or nothing.)  The occupancy query can be
// one block per CU high at 81..112 SGPRs (MI355X_MICROARCH.md, residency), so one block per CU of margin is kept.
static int fused_possible(nsk_ctx* c) {
  if (c->ndim != 2 || c->parent || c->d.use_tot || c->d.nranks > 1) return 0;
  int ncu = 0, dev = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  hipError_t e = hipErrorUnknown;
  switch (c->N) {
    case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, nsk::k2::k_helm_fused<6>, nsk::k2::Cfg<6>::NT, 0); break;
    case 8: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, nsk::k2::k_helm_fused<8>, nsk::k2::Cfg<8>::NT, 0); break;
    case 10: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, nsk::k2::k_helm_fused<10>, nsk::k2::Cfg<10>::NT, 0); break;
    case 12: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, nsk::k2::k_helm_fused<12>, nsk::k2::Cfg<12>::NT, 0); break;
    default: return 0;
  }
  if (e != hipSuccess || per < 2 || ncu < 1) return 0;
  const int need = (c->nblk + ncu - 1) / ncu;
  return need <= std::min(per, 5);      // 5: what ~110 SGPRs admit (floor(800 / 128) = 6) minus one
}

// persistent tails: both kernels resident with one workgroup per CU of margin (the occupancy query can be one high)
template <int N>
static int tails_resident(nsk_ctx* c, int ncu) {
  const size_t sh = (size_t)c->d.coarse_lda * sizeof(double);
  const int nit = c->d.coarse_lda / 256;
  int ph = 0, pp = 0;
  hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&ph, nsk::k2::k_helm_tail<N>, nsk::k2::Cfg<N>::NT, 0), e2 = hipErrorUnknown;
  if (nit <= 3) e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pp, (nsk::k2::k_pres_tail<N, 3>), nsk::k2::Cfg<N>::NT, sh);
  else if (nit <= 6) e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pp, (nsk::k2::k_pres_tail<N, 6>), nsk::k2::Cfg<N>::NT, sh);
  else if (nit <= 9) e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pp, (nsk::k2::k_pres_tail<N, 9>), nsk::k2::Cfg<N>::NT, sh);
  else e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pp, (nsk::k2::k_pres_tail<N, 12>), nsk::k2::Cfg<N>::NT, sh);
  if (e1 != hipSuccess || e2 != hipSuccess) return 0;
  if (fuse2_on(c)) {                             // the two-launch form's tail (256 threads whatever lx1)
    int p2 = 0;
    hipError_t e3;
    if (nit <= 3) e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&p2, (nsk::k2::k_pres_tail2<N, 3>), 256, sh);
    else if (nit <= 6) e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&p2, (nsk::k2::k_pres_tail2<N, 6>), 256, sh);
    else if (nit <= 9) e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&p2, (nsk::k2::k_pres_tail2<N, 9>), 256, sh);
    else e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&p2, (nsk::k2::k_pres_tail2<N, 12>), 256, sh);
    if (e3 != hipSuccess) return 0;
    pp = p2;
  } else {
    // k_pres_tail runs update_coarse_body, written for 256-thread workgroups, on its first cgrid workgroups
    const unsigned cgrid = (c->d.nvert + 4 * UC_ROWS - 1) / (4 * UC_ROWS);
    if (nsk::k2::Cfg<N>::NT != 256 || cgrid > (unsigned)c->nblk) return 0;
  }
  const int need = (c->nblk + ncu - 1) / ncu;
  if (c->debug) fprintf(stderr, "persistent tails: %d workgroups per CU needed, occupancy %d (velocity) / %d (pressure)\n", need, ph, pp);
  // (the occupancy query is one workgroup per CU high only where it is bound by scalar registers, 7-8 per CU: MI355X_MICROARCH.md,
  //  residency; grids of up to 5 per CU take it as it is, as fused_possible does)
  return need <= std::min(std::min(ph, pp), 5);
}
static bool tails_on(nsk_ctx* c) {
  if (c->tail == 0 || c->fused || c->ndim != 2 || c->parent || c->clone_of || c->d.use_tot || c->d.nranks > 1 || !c->d.ecv || !c->d.rch || !c->merged_update) return false;
  if (hostcheck_on(c) || !c->use_graph) return false;
  if (c->tail_ok < 0) {
    int ncu = 0, dev = 0;
    c->tail_ok = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) {
      switch (c->N) {
        case 6: c->tail_ok = tails_resident<6>(c, ncu); break;
        case 8: c->tail_ok = tails_resident<8>(c, ncu); break;
        case 10: c->tail_ok = tails_resident<10>(c, ncu); break;
        case 12: c->tail_ok = tails_resident<12>(c, ncu); break;
        default: break;
      }
    }
  }
  return c->tail_ok > 0;
}
template <int N>
static void launch_helm_tail(nsk_ctx* c, const Dev& d, const StepCoef& sc, int it0, int it_end) {
  if (c->ndim != 2) return;
  hipLaunchKernelGGL(nsk::k2::k_helm_tail<N>, dim3(c->nblk), dim3(nsk::k2::Cfg<N>::NT), 0, c->stream, d, sc, it0, it_end, (const double*)d.rloc, c->sync, c->sync + SYNC_WORDS);
}
template <int N>
static void launch_pres_tail(nsk_ctx* c, const Dev& d, int j0, int j1, double scale, int min_iter, int ord, bool start) {
  if (c->ndim != 2) return;
  const unsigned cgrid = (d.nvert + 4 * UC_ROWS - 1) / (4 * UC_ROWS);
  const size_t sh = d.coarse_lda * sizeof(double);
  const int nit = d.coarse_lda / 256;
  unsigned* sy = c->sync + SYNC_WORDS;          // (set 1; zeroes set 0 for the next velocity tail)
  if (fuse2_on(c)) {
    Dev d0 = d; d0.uc_start = (start && j0 == 0) ? 1 : 0;      // (a head of zero launches: the solve starts in this one)
    launch_schwarz_uc<N>(c, d0, j0, scale, min_iter, ord);   // A_{j0} as a launch: closes column j0-1 (a solve of exactly j0 iterations ends here: skip_a)
    const dim3 grid(c->nblk), blk(256);
    if (nit <= 3) hipLaunchKernelGGL((nsk::k2::k_pres_tail2<N, 3>), grid, blk, sh, c->stream, d, j0, j1, scale, min_iter, ord, cgrid, 1, sy, c->sync);
    else if (nit <= 6) hipLaunchKernelGGL((nsk::k2::k_pres_tail2<N, 6>), grid, blk, sh, c->stream, d, j0, j1, scale, min_iter, ord, cgrid, 1, sy, c->sync);
    else if (nit <= 9) hipLaunchKernelGGL((nsk::k2::k_pres_tail2<N, 9>), grid, blk, sh, c->stream, d, j0, j1, scale, min_iter, ord, cgrid, 1, sy, c->sync);
    else hipLaunchKernelGGL((nsk::k2::k_pres_tail2<N, 12>), grid, blk, sh, c->stream, d, j0, j1, scale, min_iter, ord, cgrid, 1, sy, c->sync);
    return;
  }
  launch_update_coarse(c, d, j0, scale, min_iter, ord);      // closes column j0-1 as a launch: a solve of exactly j0 iterations ends here (k_pres_tail: skip_a)
  const dim3 grid(c->nblk), blk(nsk::k2::Cfg<N>::NT);
  if (nit <= 3) hipLaunchKernelGGL((nsk::k2::k_pres_tail<N, 3>), grid, blk, sh, c->stream, d, j0, j1, scale, min_iter, ord, cgrid, 1, sy, c->sync);
  else if (nit <= 6) hipLaunchKernelGGL((nsk::k2::k_pres_tail<N, 6>), grid, blk, sh, c->stream, d, j0, j1, scale, min_iter, ord, cgrid, 1, sy, c->sync);
  else if (nit <= 9) hipLaunchKernelGGL((nsk::k2::k_pres_tail<N, 9>), grid, blk, sh, c->stream, d, j0, j1, scale, min_iter, ord, cgrid, 1, sy, c->sync);
  else hipLaunchKernelGGL((nsk::k2::k_pres_tail<N, 12>), grid, blk, sh, c->stream, d, j0, j1, scale, min_iter, ord, cgrid, 1, sy, c->sync);
}

static int step(nsk_ctx* c, int istep, int adjoint, int nh_over = -1, int np_over = -1, bool tail = false) {
  Dev& d = c->d;
  const StepCoef sc = make_coef(c, istep, adjoint);
  const bool hc = hostcheck_on(c) && !stream_capturing(c->stream);
  int nh = hc ? c->max_helm : (nh_over > 0 ? nh_over : c->cur_helm[sc.cls]);
  DISPATCH_N(c->key, {
    constexpr int NT = Cfg<N>::NT;
    if (c->key == 108 && adjoint != 2 && c->mfma_convect)
      hipLaunchKernelGGL(nsk::k3::k_convect_mfma8, dim3(c->nel), dim3(512), 0, c->stream, d, (const double*)d.u, d.bf, adjoint);
    else if (c->key == 110 && adjoint != 2 && c->mfma_convect)
      hipLaunchKernelGGL(nsk::k3::k_convect_mfma<10>, dim3(c->nel), dim3(512), 0, c->stream, d, (const double*)d.u, d.bf, adjoint);
    else if (c->key == 110 && adjoint == 2 && c->mfma_convect)      // the full equations (Newton-Krylov)
      hipLaunchKernelGGL(nsk::k3::k_convect_mfma_nl<10>, dim3(c->nel), dim3(1024), 0, c->stream, d, (const double*)d.u, d.bf);
    else
      hipLaunchKernelGGL(k_convect<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, (const double*)d.u, d.bf, adjoint);
    if (c->fused) {
      // persistent velocity solve: rhs + every CG iteration + pressure right-hand side in one launch
      HIPCHK(hipMemsetAsync(c->sync, 0, SYNC_WORDS * sizeof(unsigned), c->stream));
      launch_fused<N>(c, sc);
    } else {
      hipLaunchKernelGGL(k_rhs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc);
      if (tail) nh = std::max(1, std::min(nh, c->max_helm - 1));      // HEAD launches; the persistent tail runs launches nh .. max_helm-1
      for (int it = 0; it < nh; ++it) {
        launch_helm_iter<N>(c, d, sc, it, (const double*)d.rloc);
        tot_rows(c, d.hpart + (size_t)(it & 1) * c->hrows * c->nblk, c->hrows, d.htot + (it & 1) * c->hstride);
        if (hc && it >= 1 && it >= c->hc_helm[sc.cls] && ((it - c->hc_helm[sc.cls]) % 2 == 0)) {
          bool done = false;
          int rc2 = flags_done(c, 0, it, &done);
          if (rc2) return rc2;
          if (done) { c->hc_helm[sc.cls] = std::max(1, it - 1); nh = it + 1; break; }
        }
      }
      if (tail) { launch_helm_tail<N>(c, d, sc, nh, c->max_helm); nh = c->max_helm; }
      hipLaunchKernelGGL(k_pres_rhs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc, (nh - 1) & 1, nh - 1);
    }
  });
  // The first steps of a map project out whatever divergence the input vector has (a noise seed is far from
  // solenoidal): an error there survives to the end of the map, so those solves are converged further.
  int rc = pres_solve_launch(c, sc.h2, sc.cls, hc ? c->max_pres : (np_over > 0 ? np_over : c->cur_pres[sc.cls]), istep <= 3 ? c->early_pres_mul : 1.0, adjoint != 2, hc, tail);
  if (rc) return rc;
  const bool flat = flat_proj_on(c);
  Dev df = d; df.flat_proj = flat ? 1 : 0;
  DISPATCH_N(c->key, {
    if (flat) launch_pres_comb3<N>(c, df, sc);
    else hipLaunchKernelGGL(k_pres_update<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, sc);
  });
  if (d.nproj_max > 0) {
    DISPATCH_N(c->key, {
      hipLaunchKernelGGL(k_vel_update_proj<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, df, sc);
      if (flat) launch_proj_dots3<N>(c, df);
    });
    tot_rows(c, d.ppart, MAXPROJ + 1, d.ptot);
    hipLaunchKernelGGL(k_proj_update, dim3(c->nblk), dim3(256), 0, c->stream, d);
  } else {
    hipLaunchKernelGGL(k_vel_update, dim3((unsigned)((d.nloc + 255) / 256)), dim3(256), 0, c->stream, d, sc);
  }
  return 0;
}

static const int CLS_ISTEP[NCLS] = {1, 2, 3, 4, 7, 17};    // first step of every class

static int ensure_graph(nsk_ctx* c, int cls, int adjoint) {
  nsk_ctx::StepGraph& g = c->graphs[adjoint][cls];
  if (g.exec && g.nh == c->cur_helm[cls] && g.np == c->cur_pres[cls]) return 0;
  const auto t_cap0 = std::chrono::steady_clock::now();
  if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  hipGraph_t graph = nullptr;
  HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = step(c, CLS_ISTEP[cls], adjoint);
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc) return rc;
  if (e != hipSuccess) return fail(NSK_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  HIPCHK(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
  c->recaptures++;
  HIPCHK(hipGraphDestroy(graph));
  g.nh = c->cur_helm[cls]; g.np = c->cur_pres[cls];
  c->recapture_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_cap0).count();
  return 0;
}

static int ensure_graph_multi(nsk_ctx* c, int adjoint) {
  nsk_ctx::StepGraph& g = c->graphs[adjoint][NCLS];
  const int cls = NCLS - 1;
  if (g.exec && g.nh == c->cur_helm[cls] && g.np == c->cur_pres[cls]) return 0;
  const auto t_cap0 = std::chrono::steady_clock::now();
  if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  hipGraph_t graph = nullptr;
  HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int r = 0; r < c->graph_steps && !rc; ++r) rc = step(c, CLS_ISTEP[cls], adjoint);
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc) return rc;
  if (e != hipSuccess) return fail(NSK_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  HIPCHK(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
  c->recaptures++;
  HIPCHK(hipGraphDestroy(graph));
  g.nh = c->cur_helm[cls]; g.np = c->cur_pres[cls];
  c->recapture_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_cap0).count();
  return 0;
}

// the captured step of class `cls` with the launch budgets (nh, np): per-step budgets (nsk_ctx::gcache)
static int graph_for(nsk_ctx* c, int adjoint, int cls, int nh, int np, hipGraphExec_t* out) {
  const bool tail = tails_on(c);
  const std::array<int, 4> key{adjoint, cls + (tail ? 16 : 0), nh, np};
  auto it = c->gcache.find(key);
  if (it != c->gcache.end()) { *out = it->second; return 0; }
  // (the cache is bounded by run_map, BEFORE it builds a plan: evicting here would leave handles of the plan under construction dangling)
  const auto t_cap0 = std::chrono::steady_clock::now();
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = step(c, CLS_ISTEP[cls], adjoint, nh, np, tail);
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) return fail(NSK_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIPCHK(hipGraphDestroy(graph));
  c->recaptures++;
  c->recapture_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_cap0).count();
  c->gcache[key] = exec;
  *out = exec;
  return 0;
}
static bool step_budgets_ready(const nsk_ctx* c, int kind) {
  if (kind < 0 || kind > 1) return false;
  const nsk_ctx::StepBudgets& b = c->sb[kind];
  return c->step_budgets && !c->sb_force_class && !c->fused && !c->budget_freeze && c->d.step_iters && (int)b.bh.size() == c->nsteps && (int)b.bp.size() == c->nsteps;
}

static int run_map(nsk_ctx* c, int adjoint, double* f, const double* q) {
  Dev& d = c->d;
  // one- and two-step maps (newton.py: time derivative of the orbit) run eagerly: capturing six step-class graphs for them
  // costs more than the steps, and their iteration counts say nothing about the budgets of the real maps
  const bool use_graph = c->use_graph && c->nsteps > 2 && !hostcheck_on(c);
  const bool per_step = use_graph && step_budgets_ready(c, adjoint);
  c->last_map_per_step = per_step; c->last_map_kind = adjoint;
  std::vector<hipGraphExec_t> plan;
  if (per_step) {                                           // every graph of the plan exists before the first launch (captures end the stream's queue)
    // bound the cache of captured steps here, where no plan holds a handle of it (never seen: a few dozen budget pairs occur);
    // a plan adds at most nsteps entries
    if (c->gcache.size() + (size_t)c->nsteps > (size_t)std::max(4096, 2 * c->nsteps)) {
      HIPCHK(hipStreamSynchronize(c->stream));              // (no replay of an evicted graph may still be queued)
      for (auto& kv : c->gcache) if (kv.second) (void)hipGraphExecDestroy(kv.second);
      c->gcache.clear();
    }
    plan.resize(c->nsteps);
    for (int istep = 1; istep <= c->nsteps; ++istep) {
      const nsk_ctx::StepBudgets& b = c->sb[adjoint];
      int rc = graph_for(c, adjoint, step_class(istep), b.bh[istep - 1], b.bp[istep - 1], &plan[istep - 1]);
      if (rc) return rc;
      c->sb_launch_h += b.bh[istep - 1]; c->sb_launch_p += b.bp[istep - 1];
    }
    c->sb_maps++; c->sb_steps += c->nsteps;
    if (tails_on(c)) c->tail_maps++;
  } else if (use_graph)
    for (int k = 0; k < NCLS; ++k) { int rc = ensure_graph(c, k, adjoint); if (rc) return rc; }
  const int gsteps = (use_graph && !per_step && c->graph_steps > 1 && c->nsteps >= CLS_ISTEP[NCLS - 1] + 2 * c->graph_steps) ? c->graph_steps : 1;
  if (gsteps > 1) { int rc = ensure_graph_multi(c, adjoint); if (rc) return rc; }
  for (int cc = 0; cc < c->ndim; ++cc)
    HIPCHK(hipMemcpyAsync(d.u + cc * d.cs, q + cc * d.nloc, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d.p, q + c->ndim * d.nloc, d.npr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  if (d.bf_stride) {
    if (adjoint != 2 && c->nsteps > c->orbit_steps) return fail(NSK_EINVAL, "map longer than the stored base-flow orbit");
    HIPCHK(hipMemsetAsync(d.bstep, 0, sizeof(int), c->stream));
  }
  for (int istep = 1; istep <= c->nsteps; ++istep) {
    if (per_step) {
      HIPCHK(hipGraphLaunch(plan[istep - 1], c->stream));
    } else if (use_graph) {
      if (gsteps > 1 && istep >= CLS_ISTEP[NCLS - 1] && istep + gsteps - 1 <= c->nsteps) {
        HIPCHK(hipGraphLaunch(c->graphs[adjoint][NCLS].exec, c->stream));
        istep += gsteps - 1;
        continue;
      }
      HIPCHK(hipGraphLaunch(c->graphs[adjoint][step_class(istep)].exec, c->stream));
    } else {
      int rc = step(c, istep, adjoint);
      if (rc) return rc;
      if (c->debug) {
        GmresScal G; double hs[32];
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(&G, d.gsc, sizeof(G), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hs, d.hscal, sizeof(hs), hipMemcpyDeviceToHost));
        Stats hs2; HIPCHK(hipMemcpy(&hs2, d.stats, sizeof(hs2), hipMemcpyDeviceToHost));
        static long long ph = 0, pp = 0;
        fprintf(stderr, "step %3d: helm_it=%lld pres_it=%lld ", istep, hs2.helm_iters - (istep == 1 ? 0 : ph), hs2.pres_iters - (istep == 1 ? 0 : pp));
        ph = hs2.helm_iters; pp = hs2.pres_iters;
        fprintf(stderr, "step %3d: |g|=%.3e |g'|=%.3e nit=%d resid=%.3e nproj=%d pcnt=%d st_n=%.3e  helm ref %.3e %.3e\n", istep, G.gnorm0, G.beta0, G.nit, G.resid, G.nproj, G.pcnt, G.st_n, hs[16], hs[17]);
      }
    }
  }
  for (int cc = 0; cc < c->ndim; ++cc)
    HIPCHK(hipMemcpyAsync(f + cc * d.nloc, d.u + cc * d.cs, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(f + c->ndim * d.nloc, d.p, d.npr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  if (c->nscal > 0) {                         // ifheat = .false.: the stepper leaves the scalars alone (core/matvec.f nopcopy in / out)
    const long long toff = (long long)c->ndim * d.nloc + d.npr;
    HIPCHK(hipMemcpyAsync(f + toff, q + toff, (size_t)c->nscal * d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  return 0;
}

// Launch budgets of the step classes from the iteration maxima of the maps run so far.  The maxima of one
// class move by +-3 between consecutive Krylov vectors; budgets that follow the last map alone are cut after a
// cheap map, the next map runs out of launches and is redone as a whole (measured on cfg 2: 29 redone maps and
// 449 graph re-captures in 130 maps).  So: a window of the last BW maps; grow at once; shrink to the window
// maximum + head-room only when the window is full, or when the budget is more than twice what the window asks
// for (start-up budgets, doubled budgets after a redone map).
constexpr int BW = 8;
static int bhead() { static const int v = std::getenv("NSK_BHEAD") ? std::max(0, std::atoi(std::getenv("NSK_BHEAD"))) : 3; return v; }      // head-room above the window maximum (A/B switch)
static void budgets_update(nsk_ctx* c, const Stats& h) {
  if (c->budget_freeze) return;                         // measurement switch (scripts/noop_cost.py): budgets stay where the caller put them
  const int slot = c->bh_n % BW;
  for (int k = 0; k < NCLS; ++k) { c->bh_helm[k][slot] = (int)h.max_helm_k[k]; c->bh_pres[k][slot] = (int)h.max_pres_k[k]; }
  c->bh_n++;
  const int nv = std::min(c->bh_n, BW);
  for (int k = 0; k < NCLS; ++k) {
    if (CLS_ISTEP[k] > c->nsteps) break;
    int mh = 0, mp = 0;
    for (int i = 0; i < nv; ++i) { mh = std::max(mh, c->bh_helm[k][i]); mp = std::max(mp, c->bh_pres[k][i]); }
    // the classes of time steps 1..6 hold one to three steps per map and their counts vary most (pressure
    // solves of steps 2-3: 4 to 23 iterations): spare launches there cost microseconds, a redone map 0.1 s
    const int xh = k <= 3 ? std::max(4, mh / 2) : (k == 4 ? 1 : 0), xp = k <= 3 ? std::max(4, mp) : (k == 4 ? 1 : 0);
    // the first maps of a run differ most from one another (noise seed, empty projection space): wider margin
    const int sh = nv < 4 ? std::max(2, mh / 4) : 0, sp = nv < 4 ? std::max(4, mp) : 0;
    const int th = std::min(c->max_helm, mh + bhead() + xh + sh), tp = std::min(c->max_pres, mp + bhead() + xp + sp);
    // before the window is full only the long classes are cut (their spare launches are what costs); the classes of
    // steps 1-6 keep what they have: the second Krylov vector of a run can need 34 pressure iterations where the
    // noise seed needed 4
    const bool early = nv < BW && k >= 4;
    if (!c->fused && (th > c->cur_helm[k] || (nv == BW ? th < c->cur_helm[k] - 1 : (early && c->cur_helm[k] > 2 * th)))) c->cur_helm[k] = th;   // fused: the solve ends on the device
    if (tp > c->cur_pres[k] || (nv == BW ? tp < c->cur_pres[k] - 1 : (early && c->cur_pres[k] > 2 * tp))) c->cur_pres[k] = tp;
  }
}

// Per-step budgets of the NEXT map from the per-step iteration counts of the last SBW maps (see nsk_ctx::step_budgets).
// Helmholtz: launches = largest count at steps s-SBN..s+SBN + 3 (a solve that took I iterations needs I + 1 launches: 2 spare);
// pressure: + 2.  Steps 1-6 and the first maps of a run vary most (the class budgets' rule): wider margins there.
static void step_budgets_update(nsk_ctx* c) {
  const int ns = c->step_rec_n, kind = c->last_map_kind;
  if (kind < 0 || kind > 1) return;
  nsk_ctx::StepBudgets& b = c->sb[kind];
  if (!c->step_budgets || !c->h_step_iters || ns != c->nsteps || ns < 3) { b.bh.clear(); b.bp.clear(); return; }
  if (b.n > 0 && (int)b.hist_h[0].size() != ns) b.n = 0;                     // (nsteps changed under us: start again)
  const int slot = b.n % nsk_ctx::SBW;
  b.hist_h[slot].resize(ns); b.hist_p[slot].resize(ns);
  for (int s = 0; s < ns; ++s) { b.hist_h[slot][s] = c->h_step_iters[2 * s]; b.hist_p[slot][s] = c->h_step_iters[2 * s + 1]; }
  b.n++;
  const int nv = std::min(b.n, nsk_ctx::SBW);
  b.bh.assign(ns, 0); b.bp.assign(ns, 0);
  // (tail = 2: the budgets below + a persistent tail behind them as a SAFETY NET: a solve that outruns its budget continues
  //  in the tail instead of costing a redone map)
  const bool net = tails_on(c);
  static const int env_h = std::getenv("NSK_SB_HEAD_H") ? std::atoi(std::getenv("NSK_SB_HEAD_H")) : -1;
  static const int env_p = std::getenv("NSK_SB_HEAD_P") ? std::atoi(std::getenv("NSK_SB_HEAD_P")) : -1;
  // (options "tail_off_h" / "tail_off_p" > 0 add to the head-room of the plain budgets too: tests use it for a reference run that cannot overflow)
  const int head_h = (env_h >= 0 ? env_h : (net ? 1 : 3)) + std::max(0, c->tail_off_h), head_p = (env_p >= 0 ? env_p : (net ? 0 : 2)) + std::max(0, c->tail_off_p);
  if (tails_on(c) && c->tail == 1) {
    // HEADS for the persistent tails: the MEDIAN count of this step over the window (offline on 56 maps of config 2: the cheapest
    // predictor, 0.44 launches that find nothing to do and 0.56 tail iterations per pressure solve; profiles/r05_step_budgets.txt).
    // Velocity: a solve of I iterations is found finished by launch I (0-based), i.e. I + 1 launches.
    const int off_h = c->tail_off_h, off_p = c->tail_off_p;
    int vh[nsk_ctx::SBW], vp[nsk_ctx::SBW];
    for (int s = 0; s < ns; ++s) {
      for (int i = 0; i < nv; ++i) { vh[i] = b.hist_h[i][s]; vp[i] = b.hist_p[i][s]; }
      std::sort(vh, vh + nv); std::sort(vp, vp + nv);
      const int mh = vh[nv / 2], mp = vp[nv / 2];
      b.bh[s] = std::max(1, std::min(c->max_helm - 1, mh + 1 + off_h));
      b.bp[s] = std::max(0, std::min(c->max_pres, mp + off_p));
    }
    return;
  }
  // (option "sb_pct" < 100, with the safety-net tail only: the budget is that PERCENTILE of the counts at steps s-1..s+1 of the window
  //  instead of the largest at s-2..s+2 -- fewer launches that find nothing to do, more solves that end in the tail; offline on the
  //  recorded counts of 64 maps the 85th-90th percentile is the cheapest: DESIGN.md section 7)
  const int pct = (net && nv >= 4) ? std::max(50, std::min(100, c->sb_pct)) : 100;
  const int win = pct < 100 ? 1 : nsk_ctx::SBN;
  for (int s = 0; s < ns; ++s) {
    int mh = 0, mp = 0;
    if (pct < 100) {
      int vh[nsk_ctx::SBW * 3], vp[nsk_ctx::SBW * 3], n = 0;
      for (int i = 0; i < nv; ++i)
        for (int t = std::max(0, s - win); t <= std::min(ns - 1, s + win); ++t) { vh[n] = b.hist_h[i][t]; vp[n] = b.hist_p[i][t]; ++n; }
      std::sort(vh, vh + n); std::sort(vp, vp + n);
      const int k = std::min(n - 1, (int)std::ceil(0.01 * pct * (n - 1)));
      mh = vh[k]; mp = vp[k];
    } else
    for (int i = 0; i < nv; ++i)
      for (int t = std::max(0, s - nsk_ctx::SBN); t <= std::min(ns - 1, s + nsk_ctx::SBN); ++t) { mh = std::max(mh, b.hist_h[i][t]); mp = std::max(mp, b.hist_p[i][t]); }
    int xh = 0, xp = 0;
    if (s < 6) { xh = std::max(4, mh / 2); xp = std::max(4, mp); }                    // time steps 1-6: counts vary most (noise left by the input vector)
    if (nv < 4 && !net) { xh += std::max(2, mh / 4); xp += std::max(3, mp / 2); }    // first maps of a run: noise seed, empty projection space (with the safety-net tail an early overflow costs a tail iteration, not a map)
    b.bh[s] = std::min(c->max_helm, mh + head_h + xh);
    b.bp[s] = std::min(c->max_pres, std::max(c->min_pres, mp) + head_p + xp);
  }
}

// run one map with the adaptive launch budgets: every inner solve early-exits on its
// own convergence flag; a solve that runs out of launched iterations is counted on the
// device and the whole map is redone with larger budgets.
// One attempt of a map, asynchronous: everything is queued on the lane's stream, the device-side counters follow into pinned
// host memory; map_finish waits, books the counters and says whether the attempt stands (0), must be redone with larger
// launch budgets (1) or failed (< 0).  Split so that several lanes (nsk_matvec_batch) can have their maps in flight at once.
// Everything a map may leave behind for the next one -- time-stepper lags, CG / GMRES work arrays, the pressure projection
// space and its counters, partial sums -- back to what nsk_init left: zeros.  The immutable arrays (const members of Dev) and
// the launch budgets stay.  Called (on the stream) before the next map when nsk_bench_kernel has run on the state, and by
// map_finish after a map whose state went non-finite: a NaN / Inf in a lag array or in the projection space would otherwise
// survive every later map (step 1 multiplies the lags by ZERO coefficients: 0 x Inf = NaN; the projection space is kept from
// map to map by design).  This is what made BENCH_r04 fail: scripts/repro_r04_nan.py.
static int reset_solver_state(nsk_ctx* c) {
  Dev& d = c->d;
  void* mut[] = {d.u, d.p, d.plag, d.pext, d.ulag, d.exlag, d.bf, d.rloc, d.bloc, d.dulag, d.hx, d.hr, d.hp, d.hs, d.hwl, d.hpart, d.hscal,
                 d.V, d.Z, d.yl, d.ec, d.ecv, d.Wr, d.xc, d.gpart, d.xacc, d.dpw, d.hz, d.hy, d.rch, d.gsc, d.PX, d.PEX, d.PD, d.PED, d.ppart, d.stats,
                 d.gpart2, d.gtot2, d.ptot, d.htot, d.gtot, c->rc_big, c->cw_d0, c->cw_d1, c->cw_r, c->circ_rh, c->circ_xh, c->rc_part, c->sync};
  for (void* p : mut) {
    if (!p) continue;
    auto it = c->alloc_bytes.find(p);
    if (it == c->alloc_bytes.end()) continue;            // (an array this context shares with its parent: the parent resets it)
    HIPCHK(hipMemsetAsync(p, 0, it->second, c->stream));
  }
  c->state_dirty = false;
  return 0;
}

// per-step iteration record: device arrays + pinned host copy, made at the first map of a context (before any graph is captured:
// the kernels take the pointers by value inside Dev)
static int ensure_step_record(nsk_ctx* c) {
  if (c->d.step_iters) return 0;
  int rc;
  int *ctr = nullptr, *rec = nullptr;
  if ((rc = dalloc(c, &ctr, 4)) || (rc = dalloc(c, &rec, (size_t)2 * nsk_ctx::STEP_CAP))) return rc;
  HIPCHK(hipHostMalloc((void**)&c->h_step_iters, (size_t)2 * nsk_ctx::STEP_CAP * sizeof(int)));
  std::memset(c->h_step_iters, 0, (size_t)2 * nsk_ctx::STEP_CAP * sizeof(int));
  c->d.stepctr = ctr; c->d.step_iters = rec; c->d.step_cap = nsk_ctx::STEP_CAP;
  invalidate_graphs(c);
  return 0;
}

static int map_launch(nsk_ctx* c, int adjoint, double* f, const double* src) {
  if (c->state_dirty) { int rc0 = reset_solver_state(c); if (rc0) return rc0; }
  { int rc0 = ensure_step_record(c); if (rc0) return rc0; }
  const int nrec = std::min(c->nsteps, nsk_ctx::STEP_CAP);
  HIPCHK(hipMemsetAsync(c->d.stats, 0, sizeof(Stats), c->stream));
  HIPCHK(hipMemsetAsync(c->d.stepctr, 0, sizeof(int), c->stream));
  HIPCHK(hipMemsetAsync(c->d.step_iters, 0, (size_t)2 * nrec * sizeof(int), c->stream));
  int rc = run_map(c, adjoint, f, src);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(c->hstat_pin, c->d.stats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_step_iters, c->d.step_iters, (size_t)2 * nrec * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->step_rec_n = nrec;
  return 0;
}
static int map_finish(nsk_ctx* c) {
  HIPCHK(hipStreamSynchronize(c->stream));
  const Stats h = *c->hstat_pin;
  c->hstats.helm_iters += h.helm_iters; c->hstats.pres_iters += h.pres_iters; c->hstats.steps += c->nsteps;
  c->hstats.max_helm = std::max(c->hstats.max_helm, h.max_helm); c->hstats.max_pres = std::max(c->hstats.max_pres, h.max_pres);
  c->hstats.last_helm_res = h.last_helm_res; c->hstats.last_pres_res = h.last_pres_res;
  c->hstats.capped_solves += h.capped_solves; c->hstats.worst_cap_ratio = std::max(c->hstats.worst_cap_ratio, h.worst_cap_ratio);
  c->tot_capped += h.capped_solves; c->tot_worst_cap = std::max(c->tot_worst_cap, h.worst_cap_ratio);
  c->tot_helm_iters += h.helm_iters; c->tot_pres_iters += h.pres_iters; c->tot_steps += c->nsteps; c->tot_pres_jsum += h.pres_jsum;
  for (int k = 0; k < NCLS; ++k) {
    c->hstats.max_helm_k[k] = std::max(c->hstats.max_helm_k[k], h.max_helm_k[k]);
    c->hstats.max_pres_k[k] = std::max(c->hstats.max_pres_k[k], h.max_pres_k[k]);
  }
  if (c->debug >= 2) {
    fprintf(stderr, "map: unconverged %llu | helm max/budget", (unsigned long long)h.unconverged);
    for (int k = 0; k < NCLS; ++k) fprintf(stderr, " %llu/%d", (unsigned long long)h.max_helm_k[k], c->cur_helm[k]);
    fprintf(stderr, " | pres max/budget");
    for (int k = 0; k < NCLS; ++k) fprintf(stderr, " %llu/%d", (unsigned long long)h.max_pres_k[k], c->cur_pres[k]);
    fprintf(stderr, "\n");
  }
  if (h.sync_timeouts) {
    if (!c->fused && c->tail != 0 && c->last_map_per_step) {
      // a grid barrier of a persistent tail gave up (its workgroups were not all resident: another process on the device, fewer
      // CUs than the occupancy query promised): this context goes back to launch budgets for good and the map is redone
      fprintf(stderr, "libnekstab_hip: a grid barrier of the persistent tails timed out (%lld): tails off for this context, map redone on launch budgets\n", (long long)h.sync_timeouts);
      c->tail = 0;
      invalidate_graphs(c);
      (void)reset_solver_state(c);
      c->retries++;
      return 1;
    }
    return fail(NSK_EHIP, "grid barrier of the persistent velocity solve timed out (workgroups not co-resident?): set option fused = 0");
  }
  if (h.nonfinite) {                    // the MAP says so (the reference's only guard is the NaN check of the next inner product, core/krylov_subspace.f:52-55)
    (void)reset_solver_state(c);        // ... and the next map of this context starts clean
    return fail(NSK_ENAN, "non-finite state inside the map: " + std::to_string((long long)h.nonfinite) + " time steps with a NaN / Inf pressure right-hand side (input vector not finite?)");
  }
  const bool hcm = hostcheck_on(c);     // host-checked eager steps iterate to the caps whatever the budgets say
  if (h.unconverged == 0) {
    if (c->nsteps > 2 && !hcm) { budgets_update(c, h); step_budgets_update(c); }
    c->sb_force_class = false;
    return 0;
  }
  if (c->last_map_per_step) {           // a per-step budget overflowed: the redo runs on the class budgets as they stand (they cover the class maxima of the last maps)
    c->sb_force_class = true;
    c->retries++;
    return 1;
  }
  bool capped = true;
  for (int k = 0; k < NCLS; ++k) capped = capped && c->cur_helm[k] >= c->max_helm && c->cur_pres[k] >= c->max_pres;
  if (hcm) capped = true;               // ... so an unconverged solve there IS capped: redoing the map with doubled budgets the mode ignores would repeat it (as group_run_map)
  if (capped) {
    c->hstats.unconverged += h.unconverged;
    return fail(NSK_ENOCONV, "inner solve hit its iteration cap (" + std::to_string(h.unconverged) + " solves)");
  }
  c->retries++;
  for (int k = 0; k < NCLS; ++k) {
    c->cur_helm[k] = std::min(c->max_helm, 2 * c->cur_helm[k] + 4);
    c->cur_pres[k] = std::min(c->max_pres, 2 * c->cur_pres[k] + 4);
  }
  return 1;
}

// run one map with the adaptive launch budgets: every inner solve early-exits on its
// own convergence flag; a solve that runs out of launched iterations is counted on the
// device and the whole map is redone with larger budgets.
static int run_map_adaptive(nsk_ctx* c, int adjoint, double* f, const double* q) {
  const double* src = q;
  if (f == q) {
    HIPCHK(hipMemcpyAsync(c->scratch, q, c->nstate * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    src = c->scratch;
  }
  for (;;) {
    int rc = map_launch(c, adjoint, f, src);
    if (rc) return rc;
    rc = map_finish(c);
    if (rc <= 0) return rc;
  }
}

#include "nsk_shard.inc"

// ===========================================================================
// C-ABI
// ===========================================================================
extern "C" {

// Cut the shard of `rank` (part[e] = owner of global element e) out of a full-mesh context.
int nsk_shard_create(nsk_ctx* parent, const int* part, int rank, int nranks, nsk_ctx** out) {
  if (!parent || !part || !out) return fail(NSK_EINVAL, "bad argument");
  if (parent->released) return fail(NSK_EINVAL, "parent context was released (nsk_shard_release_parent)");
  if (parent->nscal > 0) return fail(NSK_EINVAL, "sharded vectors do not carry scalar fields yet");
  return shard_create(parent, part, rank, nranks, out);
}

// Free everything of a full-mesh context that its shards do not share: element-major geometry, preconditioner factors, state,
// work arrays, Krylov vectors.  What stays on the device: the 1-D bases and the (replicated) coarse operator.  A rank of a
// multi-process run calls this once its shard is cut, so that its GPU holds one shard, not the whole mesh.
int nsk_shard_release_parent(nsk_ctx* P) {
  if (!P) return fail(NSK_EINVAL, "bad argument");
  if (P->parent) return fail(NSK_EINVAL, "not a full-mesh context");
  if (P->released) return 0;
  HIPCHK(hipStreamSynchronize(P->stream));
  const Dev& d = P->d;
  const void* keep[] = {d.D, d.J12, d.D12, d.Jd, d.Dd, d.hat, d.Aci, d.Acif, d.p_off, d.p_invoff,
                        P->cA_rp, P->cA_ci, P->cA_va, P->cA_dinv, P->cw_d0, P->cw_d1, P->cw_r,
                        P->circ_Binv, P->circ_F, P->circ_cols, P->circ_rh, P->circ_xh};
  std::vector<void*> left;
  for (void* p : P->allocs) {
    bool k = false;
    for (const void* q : keep) k = k || (q && q == p);
    if (k) left.push_back(p); else { (void)hipFree(p); P->alloc_bytes.erase(p); }
  }
  P->allocs.swap(left);
  for (auto& a : P->graphs) for (auto& g : a) if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; g.nh = -1; }
  P->released = true;
  return 0;
}

// f_r = map(q_r) for the ranks living in this process, in lock-step (virtual ranks: all of them).
int nsk_group_matvec(nsk_ctx** shards, int n, int mode, nsk_vec* f, nsk_vec* q) {
  if (!shards || n < 1 || !f || !q) return fail(NSK_EINVAL, "bad argument");
  std::vector<nsk_ctx*> G(shards, shards + n);
  int rc;
  switch (mode) {
    case NSK_DIRECT: return group_run_map(G, 0, (double* const*)f, (const double* const*)q);
    case NSK_ADJOINT: return group_run_map(G, 1, (double* const*)f, (const double* const*)q);
    case NSK_DIRECT_ADJOINT:                                     // core/matvec.f:343-346 (f may not alias q on shards)
      for (int r = 0; r < n; ++r) if (f[r] == q[r]) return fail(NSK_EINVAL, "sharded composed maps need f != q");
      if ((rc = group_run_map(G, 0, (double* const*)f, (const double* const*)q))) return rc;
      for (int r = 0; r < n; ++r) HIPCHK(hipMemcpyAsync(G[r]->scratch, f[r], G[r]->nstate * sizeof(double), hipMemcpyDeviceToDevice, G[r]->stream));
      {
        std::vector<const double*> src(n);
        for (int r = 0; r < n; ++r) src[r] = G[r]->scratch;
        return group_run_map(G, 1, (double* const*)f, src.data());
      }
    case NSK_NEWTON:                                             // core/matvec.f:398-401
      for (int r = 0; r < n; ++r) if (f[r] == q[r]) return fail(NSK_EINVAL, "newton map needs f != q");
      if ((rc = group_run_map(G, 0, (double* const*)f, (const double* const*)q))) return rc;
      for (int r = 0; r < n; ++r)
        hipLaunchKernelGGL(k_axpby, dim3((unsigned)((G[r]->nstate + 255) / 256)), dim3(256), 0, G[r]->stream, (double*)f[r], -1.0, (const double*)q[r], 1.0, G[r]->nstate);
      return 0;
    case NSK_FORCE_SENSITIVITY:                                  // core/matvec.f:357-374
      for (int r = 0; r < n; ++r) if (f[r] == q[r]) return fail(NSK_EINVAL, "force-sensitivity map needs f != q");
      if ((rc = group_run_map(G, 1, (double* const*)f, (const double* const*)q))) return rc;
      for (int r = 0; r < n; ++r)
        hipLaunchKernelGGL(k_axpby, dim3((unsigned)((G[r]->nstate + 255) / 256)), dim3(256), 0, G[r]->stream, (double*)f[r], 1.0, (const double*)q[r], -1.0, G[r]->nstate);
      return 0;
    default: return fail(NSK_EINVAL, "unknown mode");
  }
}

// Phi_T(q) of the full equations on the ranks of this process (nonlinear_forward_map, core/newton_krylov.f:336-378, which the
// reference runs under MPI like every other map); subtract_q != 0 returns Phi_T(q) - q.
int nsk_group_nonlinear_map(nsk_ctx** shards, int n, nsk_vec* f, nsk_vec* q, int subtract_q) {
  if (!shards || n < 1 || !f || !q) return fail(NSK_EINVAL, "bad argument");
  for (int r = 0; r < n; ++r) if (!shards[r] || !f[r] || !q[r] || f[r] == q[r]) return fail(NSK_EINVAL, "needs f != q on every rank");
  std::vector<nsk_ctx*> G(shards, shards + n);
  int rc = group_run_map(G, 2, (double* const*)f, (const double* const*)q);
  if (rc) return rc;
  if (subtract_q)
    for (int r = 0; r < n; ++r) {
      nsk_ctx* c = G[r];
      hipLaunchKernelGGL(k_axpby, dim3((unsigned)((c->nstate + 255) / 256)), dim3(256), 0, c->stream, (double*)f[r], -1.0, (const double*)q[r], 1.0, c->nstate);
    }
  return 0;
}

// max over all ranks of a per-rank scalar through the sum transports: every rank fills its own slot of a zeroed vector
static int group_allmax(std::vector<nsk_ctx*>& G, double* vals) {
  nsk_ctx* c0 = G[0];
  const int nr = c0->nranks;
  std::vector<double> slot(nr, 0.0);
  for (size_t r = 0; r < G.size(); ++r) slot[G[r]->rank] = vals[r];
  if (G.size() == 1 && (c0->comm || c0->host_allred)) { int rc = nsk_allreduce_host(c0, slot.data(), nr); if (rc) return rc; }
  double m = 0.0;
  for (double v : slot) m = std::max(m, v);
  for (size_t r = 0; r < G.size(); ++r) vals[r] = m;
  return 0;
}

// New linearisation point on element shards (nsk_set_baseflow for the ranks of this process): every rank recomputes the
// base-flow constants of its own elements; dt / nsteps follow from the CFL maximum over ALL ranks (compute_cfl is a glmax).
static int group_set_baseflow(std::vector<nsk_ctx*>& G, const double* const* q) {
  std::vector<double> ctarg(G.size(), 0.0);
  for (size_t r = 0; r < G.size(); ++r) {
    nsk_ctx* c = G[r]; nsk_ctx* P = c->parent; Dev& d = c->d;
    if (!P) return fail(NSK_EINVAL, "needs shard contexts");
    const int nd = c->ndim, NN = c->NN;
    d.bfmask = 0;                                          // (the new constants are not scanned for zero arrays on shards)
    DISPATCH_N(c->key, {                                   // (hexahedra: Dev::cUr aliases the 12-constant array bfc)
      hipLaunchKernelGGL(k_baseflow<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, q[r], (double*)d.cUr, (double*)d.cUs,
                         (double*)d.GUx, (double*)d.GUy, (double*)d.GVx, (double*)d.GVy);
    });
    std::vector<double> u((size_t)nd * c->nloc);
    HIPCHK(hipMemcpyAsync(u.data(), q[r], (size_t)nd * c->nloc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double ct = 0.0;
    for (int le = 0; le < c->nel; ++le)
      for (int k = 0; k < NN; ++k) {
        const long long l = (long long)le * NN + k, lg = (long long)c->elems[le] * NN + k;
        double sm = 0.0;
        for (int a = 0; a < nd; ++a) {
          double t = 0.0;
          for (int cc = 0; cc < nd; ++cc) t += u[(size_t)cc * c->nloc + l] * P->h_cflg[(size_t)nd * nd * lg + a * nd + cc];
          sm += std::fabs(t);
        }
        ct = std::max(ct, sm);
      }
    ctarg[r] = ct;
  }
  int rc = group_allmax(G, ctarg.data());
  if (rc) return rc;
  if (!(ctarg[0] > 0.0)) return fail(NSK_EINVAL, "base flow is zero");
  for (size_t r = 0; r < G.size(); ++r) {
    nsk_ctx* c = G[r]; nsk_ctx* P = c->parent; Dev& d = c->d;
    const int NN = c->NN;
    const double dt0 = P->cfl_target / ctarg[r];
    c->nsteps = (int)std::ceil(c->endtime / dt0);
    c->dt = c->endtime / c->nsteps;
    d.dt = c->dt;
    std::vector<double> dinv((size_t)3 * c->nloc);
    const double bd0[3] = {1.0, 1.5, 11.0 / 6.0};
    for (int k = 0; k < 3; ++k)
      for (int le = 0; le < c->nel; ++le)
        for (int i = 0; i < NN; ++i) {
          const long long lg = (long long)c->elems[le] * NN + i;
          dinv[(size_t)k * c->nloc + (size_t)le * NN + i] = P->h_mask[lg] / (d.nu * P->h_dAs[lg] + bd0[k] / c->dt * P->h_bs[lg]);
        }
    HIPCHK(hipMemcpy((double*)d.dinv, dinv.data(), dinv.size() * sizeof(double), hipMemcpyHostToDevice));
    invalidate_graphs(c);
    for (int k = 0; k < NCLS; ++k) { c->cur_helm[k] = c->max_helm; c->cur_pres[k] = c->max_pres; }
    c->bh_n = 0;
  }
  return 0;
}
int nsk_group_set_baseflow(nsk_ctx** shards, int n, nsk_vec* q) {
  if (!shards || n < 1 || !q) return fail(NSK_EINVAL, "bad argument");
  std::vector<nsk_ctx*> G(shards, shards + n);
  for (int r = 0; r < n; ++r) if (!G[r] || !q[r]) return fail(NSK_EINVAL, "bad argument");
  if (G[0]->d.bf_stride) return fail(NSK_EINVAL, "a stored orbit is active: call nsk_group_set_orbit again instead");
  return group_set_baseflow(G, (const double* const*)q);
}

// Time-periodic base flow on element shards (nsk_set_orbit for the ranks of this process): the full equations are integrated
// over one period by the sharded stepper and every rank stores the base-flow constants of its own elements per step.
int nsk_group_set_orbit(nsk_ctx** shards, int n, nsk_vec* q0, double spng_str, nsk_vec* end) {
  if (!shards || n < 1 || !q0) return fail(NSK_EINVAL, "bad argument");
  std::vector<nsk_ctx*> G(shards, shards + n);
  for (int r = 0; r < n; ++r)
    if (!G[r] || !q0[r] || !G[r]->parent) return fail(NSK_EINVAL, "needs shard contexts");
  const int nd = G[0]->ndim;
  // quadrilaterals: six arrays of nel*lxd^2 per step; hexahedra: one array of 12*nel*lxd^3 per step behind Dev::bfc (= cUr)
  const int narr = nd == 3 ? 1 : 6;
  auto steady_back = [&](nsk_ctx* c) {
    Dev& d = c->d;
    if (!d.bf_stride) return;
    if (nd == 3) { d.bfc = c->steady[0]; d.cUr = c->steady[0]; }
    else { d.cUr = c->steady[0]; d.cUs = c->steady[1]; d.GUx = c->steady[2]; d.GUy = c->steady[3]; d.GVx = c->steady[4]; d.GVy = c->steady[5]; }
    d.bf_stride = 0;
  };
  int rc;
  for (nsk_ctx* c : G) steady_back(c);
  if ((rc = group_set_baseflow(G, (const double* const*)q0))) return rc;
  for (size_t r = 0; r < G.size(); ++r) {
    nsk_ctx* c = G[r]; Dev& d = c->d;
    const long long per_step = nd == 3 ? 12 * d.nfine : (long long)c->nel * c->NDD;
    for (int k = 0; k < narr; ++k) {
      if (c->orbit[k]) { nsk_vec v = c->orbit[k]; nsk_vec_free(c, 1, &v); c->orbit[k] = nullptr; }
      if ((rc = dalloc(c, &c->orbit[k], (size_t)c->nsteps * per_step))) return rc;
    }
    double* vr = const_cast<double*>(d.spng_vr);
    if (!vr && (rc = dalloc(c, &vr, nd * d.cs))) return rc;
    const double* q = (const double*)q0[r];
    for (int cc = 0; cc < nd; ++cc) HIPCHK(hipMemcpyAsync(vr + cc * d.cs, q + cc * d.nloc, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    d.spng_vr = vr; d.nl_spng_str = spng_str;
    if (!d.bstep && (rc = dalloc(c, &d.bstep, 4))) return rc;
    HIPCHK(hipMemsetAsync(d.stats, 0, sizeof(Stats), c->stream));
    for (int cc = 0; cc < nd; ++cc) HIPCHK(hipMemcpyAsync(d.u + cc * d.cs, q + cc * d.nloc, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d.p, q + nd * d.nloc, d.npr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  const int nsteps = G[0]->nsteps;
  for (int istep = 1; istep <= nsteps; ++istep) {
    for (nsk_ctx* c : G) {
      Dev& d = c->d;
      const long long off = (long long)(istep - 1) * (nd == 3 ? 12 * d.nfine : (long long)c->nel * c->NDD);
      // k_baseflow reads a state vector (component stride nloc); the stepper's field has stride cs = nloc + ghost slots
      for (int cc = 0; cc < nd; ++cc)
        HIPCHK(hipMemcpyAsync(c->scratch + cc * d.nloc, d.u + cc * d.cs, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      if (nd == 3) {
        DISPATCH_N(c->key, {
          hipLaunchKernelGGL(k_baseflow<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, (const double*)c->scratch, c->orbit[0] + off,
                             (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr);
        });
      } else {
        DISPATCH_N(c->key, {
          hipLaunchKernelGGL(k_baseflow<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, (const double*)c->scratch, c->orbit[0] + off, c->orbit[1] + off,
                             c->orbit[2] + off, c->orbit[3] + off, c->orbit[4] + off, c->orbit[5] + off);
        });
      }
    }
    if ((rc = group_step(G, istep, 2))) return rc;
  }
  Stats h;
  HIPCHK(hipMemcpyAsync(&h, G[0]->d.stats, sizeof(Stats), hipMemcpyDeviceToHost, G[0]->stream));
  for (size_t r = 0; r < G.size(); ++r) {
    nsk_ctx* c = G[r]; Dev& d = c->d;
    if (end && end[r]) {
      double* f = (double*)end[r];
      for (int cc = 0; cc < nd; ++cc) HIPCHK(hipMemcpyAsync(f + cc * d.nloc, d.u + cc * d.cs, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(f + nd * d.nloc, d.p, d.npr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
  }
  for (nsk_ctx* c : G) HIPCHK(hipStreamSynchronize(c->stream));
  if (h.unconverged > 0) return fail(NSK_ENOCONV, "inner solve hit its iteration cap while integrating the base-flow orbit (sharded)");
  for (nsk_ctx* c : G) {
    Dev& d = c->d;
    if (nd == 3) {
      c->steady[0] = d.bfc;
      d.bfc = c->orbit[0]; d.cUr = c->orbit[0]; d.bfmask = 0;
      d.bf_stride = 12 * d.nfine;
    } else {
      c->steady[0] = d.cUr; c->steady[1] = d.cUs; c->steady[2] = d.GUx; c->steady[3] = d.GUy; c->steady[4] = d.GVx; c->steady[5] = d.GVy;
      d.cUr = c->orbit[0]; d.cUs = c->orbit[1]; d.GUx = c->orbit[2]; d.GUy = c->orbit[3]; d.GVx = c->orbit[4]; d.GVy = c->orbit[5];
      d.bf_stride = (long long)c->nel * c->NDD;
    }
    c->orbit_steps = c->nsteps;
    invalidate_graphs(c);
  }
  return 0;
}

// ---- RCCL transport for ranks in separate processes (one process per GPU) ----
int nsk_comm_unique_id(unsigned char* out128) {
  if (!out128) return fail(NSK_EINVAL, "bad argument");
  if (!rccl_rt::load()) return fail(NSK_EHIP, "librccl not found");
  rccl_rt::UniqueId id;
  if (rccl_rt::GetUniqueId(&id) != 0) return fail(NSK_EHIP, "ncclGetUniqueId failed");
  std::memcpy(out128, id.internal, 128);
  return 0;
}
int nsk_comm_init_host(nsk_ctx* shard, nsk_exchange_fn xchg, nsk_allreduce_fn allred, void* user) {
  if (!shard || !xchg || !allred || !shard->parent) return fail(NSK_EINVAL, "needs a shard context and both callbacks");
  shard->host_xchg = xchg; shard->host_allred = allred; shard->host_user = user;
  return 0;
}
int nsk_comm_init_rccl(nsk_ctx* shard, const unsigned char* id128) {
  if (!shard || !id128 || !shard->parent) return fail(NSK_EINVAL, "needs a shard context");
  if (!rccl_rt::load()) return fail(NSK_EHIP, "librccl not found");
  rccl_rt::UniqueId id;
  std::memcpy(id.internal, id128, 128);
  rccl_rt::Comm comm = nullptr;
  if (rccl_rt::CommInitRank(&comm, shard->nranks, id, shard->rank) != 0) return fail(NSK_EHIP, "ncclCommInitRank failed");
  shard->comm = comm;
  return 0;
}
// sum a small host array over the ranks (Krylov inner products), through the device
int nsk_allreduce_host(nsk_ctx* shard, double* buf, int n) {
  if (!shard || !buf || n < 1 || n > 1024) return fail(NSK_EINVAL, "bad argument");
  if (shard->host_allred) {
    if (shard->host_allred(shard->host_user, buf, n) != 0) return fail(NSK_EHIP, "host all-reduce callback failed");
    return 0;
  }
  if (!shard->comm) return 0;                                   // single process: nothing to do
  HIPCHK(hipMemcpyAsync(shard->kout, buf, n * sizeof(double), hipMemcpyHostToDevice, shard->stream));
  if (rccl_rt::AllReduce(shard->kout, shard->kout, n, rccl_rt::kDouble, rccl_rt::kSum, shard->comm, shard->stream) != 0)
    return fail(NSK_EHIP, "ncclAllReduce failed");
  HIPCHK(hipMemcpyAsync(buf, shard->kout, n * sizeof(double), hipMemcpyDeviceToHost, shard->stream));
  HIPCHK(hipStreamSynchronize(shard->stream));
  return 0;
}

// test hook: which = 0: dssum of a velocity-mesh field across shards; 1: E = D B^-1 D^T apply
int nsk_group_test(nsk_ctx** shards, int n, int which, const double* const* in, double* const* out) {
  std::vector<nsk_ctx*> G(shards, shards + n);
  int rc;
  if (which == 0) {
    for (int r = 0; r < n; ++r) HIPCHK(hipMemcpyAsync(G[r]->d.rloc, in[r], G[r]->nloc * sizeof(double), hipMemcpyHostToDevice, G[r]->stream));
    if ((rc = xchg_vel(G, &Dev::rloc, 0, 1))) return rc;
    for (int r = 0; r < n; ++r) {
      nsk_ctx* c = G[r];
      hipLaunchKernelGGL(k_dssum_test, dim3((unsigned)((c->nloc + 255) / 256)), dim3(256), 0, c->stream, c->d, (const double*)c->d.rloc, c->wv2);
      HIPCHK(hipMemcpyAsync(out[r], c->wv2, c->nloc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
  } else if (which == 1) {
    for (int r = 0; r < n; ++r) HIPCHK(hipMemcpyAsync(G[r]->wp1, in[r], G[r]->npr * sizeof(double), hipMemcpyHostToDevice, G[r]->stream));
    DISPATCH_N(G[0]->key, {
      for (nsk_ctx* c : G) hipLaunchKernelGGL(k_gradt<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, c->d, (const double*)c->wp1, c->d.yl);
      if ((rc = xchg_vel(G, &Dev::yl, 0, G[0]->ndim))) return rc;
      for (nsk_ctx* c : G) hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, c->d, (const double*)c->d.yl, c->wp2, -1, 0);
    });
    for (int r = 0; r < n; ++r) HIPCHK(hipMemcpyAsync(out[r], G[r]->wp2, G[r]->npr * sizeof(double), hipMemcpyDeviceToHost, G[r]->stream));
  } else return fail(NSK_EINVAL, "unknown test");
  HIPCHK(hipStreamSynchronize(G[0]->stream));
  return 0;
}

// rank-local part of the bm1s inner products (f, Q_k): the caller sums over ranks
int nsk_local_dots(nsk_ctx* c, nsk_vec f, const nsk_vec* Q, int nq, double* out) {
  if (!c || !f || !Q || !out || nq < 1) return fail(NSK_EINVAL, "bad argument");
  HIPCHK(hipStreamSynchronize(c->stream));
  if (nq > 1024) return fail(NSK_EINVAL, "too many vectors");
  for (int k = 0; k < nq; ++k) ((double**)c->hpin)[k] = (double*)Q[k];
  HIPCHK(hipMemcpyAsync(c->kptr, c->hpin, nq * sizeof(double*), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_dots, dim3(c->kblk), dim3(256), 0, c->stream, (const double*)f, (const double* const*)c->kptr, nq, c->d.bm1s, c->nloc, c->kpart, c->kblk, c->ndim, (long long)c->ndim * c->nloc + c->npr, c->nscal);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, (const double*)c->kpart, nq, c->kblk, c->kout);
  HIPCHK(hipMemcpyAsync(out, c->kout, nq * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// f -= sum_k h_k Q_k over the rank-local state (coefficients from the host)
int nsk_project_out(nsk_ctx* c, nsk_vec f, const nsk_vec* Q, int nq, const double* h) {
  if (!c || !f || !Q || !h || nq < 1 || nq > 1024) return fail(NSK_EINVAL, "bad argument");
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < nq; ++k) ((double**)c->hpin)[k] = (double*)Q[k];
  HIPCHK(hipMemcpyAsync(c->kptr, c->hpin, nq * sizeof(double*), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->kout, h, nq * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_project_out, dim3((unsigned)((c->nstate + 255) / 256)), dim3(256), 0, c->stream, (double*)f, (const double* const*)c->kptr, nq, (const double*)c->kout, c->nstate);
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}


const char* nsk_last_error(void) { return g_err.c_str(); }

int nsk_init(const nsk_case* cs, nsk_ctx** out) {
  if (!cs || !out) return fail(NSK_EINVAL, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NSK_EHIP, "no HIP device: libnekstab_hip has no CPU fallback");
  nsk_ctx* c = new nsk_ctx();
  int rc = (cs->ndim == 3) ? build3(c, *cs) : build(c, *cs);
  if (!rc) rc = ensure_step_record(c);       // (here, not at the first map: an allocation memsets on the null stream, which would end another lane's stream capture)
  if (rc) { std::string keep = g_err; nsk_finalize(c); g_err = keep; return rc; }
  *out = c;
  return 0;
}

// ---- rank-local set-up --------------------------------------------------------------------------------------------------
// A rank of an element-sharded run builds a context over ITS sub-mesh (own elements + two rings of node-sharing neighbours,
// ascending global element id, global node / vertex ids kept), not over the whole mesh:
//   nsk_init_local   geometry, gather tables, Jacobi diagonals, Schwarz factors: exact on the owned elements;
//                    the coarse operator: the rows of the vertices this rank owns
//   nsk_local_info / nsk_local_rows      what the other ranks need: volume of the owned elements, CFL maximum, the largest
//                    fast-diagonalisation eigenvalue, the coarse rows
//   nsk_local_finish the same, reduced / gathered over all ranks by the caller: dt and nsteps, Jacobi diagonals for that dt,
//                    the replicated coarse solve
//   nsk_shard_create_local   the rank's shard, halos keyed by global ids
// Set-up time and memory then scale with the sub-mesh; only the vertex-level coarse operator is replicated.
int nsk_init_local(const nsk_case* cs, const int* own, nsk_ctx** out) {
  if (!cs || !own || !out) return fail(NSK_EINVAL, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NSK_EHIP, "no HIP device: libnekstab_hip has no CPU fallback");
  nsk_ctx* c = new nsk_ctx();
  c->local = true;
  c->local_own.assign(own, own + cs->nel);
  for (char& f : c->local_own) f = f ? 1 : 0;
  c->has_outflow = cs->has_outflow;
  int rc = (cs->ndim == 3) ? build3(c, *cs) : build(c, *cs);
  if (rc) { std::string keep = g_err; nsk_finalize(c); g_err = keep; return rc; }
  *out = c;
  return 0;
}

int nsk_local_info(nsk_ctx* c, double* vol_own, double* ctarg, double* fd_lmax, long long* npr_own, long long* nrows) {
  if (!c || !c->local) return fail(NSK_EINVAL, "needs a context made by nsk_init_local");
  long long nown = 0;
  for (char f : c->local_own) nown += f;
  if (vol_own) *vol_own = c->vol_own;
  if (ctarg) *ctarg = c->ctarg;
  if (fd_lmax) *fd_lmax = c->fd_lmax;
  if (npr_own) *npr_own = nown * c->MM;
  if (nrows) *nrows = (long long)c->crow_u.size();
  return 0;
}

int nsk_local_rows(nsk_ctx* c, int* u, int* v, double* a) {
  if (!c || !c->local || !u || !v || !a) return fail(NSK_EINVAL, "needs a context made by nsk_init_local");
  std::copy(c->crow_u.begin(), c->crow_u.end(), u);
  std::copy(c->crow_v.begin(), c->crow_v.end(), v);
  std::copy(c->crow_a.begin(), c->crow_a.end(), a);
  return 0;
}

int nsk_local_finish(nsk_ctx* c, double vol, double ctarg, double fd_lmax, long long npr_glob, long long nrows, const int* u, const int* v,
                     const double* a) {
  if (!c || !c->local) return fail(NSK_EINVAL, "needs a context made by nsk_init_local");
  if (c->local_done) return fail(NSK_EINVAL, "nsk_local_finish was already called");
  if (!(vol > 0.0) || !(ctarg > 0.0) || npr_glob < 1 || nrows < 1 || !u || !v || !a) return fail(NSK_EINVAL, "bad argument");
  Dev& d = c->d;
  int rc;
  d.vol = vol; c->npr_glob = npr_glob; d.npr_glob = npr_glob;
  c->ctarg = ctarg;
  const double dt0 = c->cfl_target / ctarg;
  c->nsteps = (int)std::ceil(c->endtime / dt0);
  c->dt = c->endtime / c->nsteps;
  d.dt = c->dt;
  {
    std::vector<double> dinv((size_t)3 * c->nloc);
    const double bd0[3] = {1.0, 1.5, 11.0 / 6.0};
    for (int k = 0; k < 3; ++k)
      for (long long l = 0; l < c->nloc; ++l) dinv[(size_t)k * c->nloc + l] = c->h_mask[l] / (d.nu * c->h_dAs[l] + bd0[k] / c->dt * c->h_bs[l]);
    HIPCHK(hipMemcpy((double*)d.dinv, dinv.data(), dinv.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  const int nvert = c->nvert;
  for (long long k = 0; k < nrows; ++k)
    if (u[k] < 0 || u[k] >= nvert || v[k] < 0 || v[k] >= nvert) return fail(NSK_EINVAL, "coarse row entry outside 0..nvert-1");
  if (c->ndim == 3) {
    c->fd_lmax = fd_lmax; d.fd_eps = 1e-11 * fd_lmax;
    std::vector<std::vector<std::pair<int, double>>> rows(nvert);
    for (long long k = 0; k < nrows; ++k) rows[u[k]].push_back({v[k], a[k]});
    for (int w = 0; w < nvert; ++w) if (rows[w].empty()) return fail(NSK_EINVAL, "coarse operator: no row for vertex " + std::to_string(w) + " (rows of all ranks are needed)");
    if ((rc = coarse_build3(c, rows, c->has_outflow != 0))) return rc;
  } else {
    std::vector<double> Ac((size_t)nvert * nvert, 0.0);
    std::vector<char> seen(nvert, 0);
    for (long long k = 0; k < nrows; ++k) { Ac[(size_t)u[k] * nvert + v[k]] = a[k]; seen[u[k]] = 1; }
    for (int w = 0; w < nvert; ++w) if (!seen[w]) return fail(NSK_EINVAL, "coarse operator: no row for vertex " + std::to_string(w) + " (rows of all ranks are needed)");
    if ((rc = coarse_dense_inverse(c, Ac, c->has_outflow != 0))) return rc;
  }
  std::vector<int>().swap(c->crow_u); std::vector<int>().swap(c->crow_v); std::vector<double>().swap(c->crow_a);
  c->local_done = true;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int nsk_shard_create_local(nsk_ctx* P, const int* part_sub, const long long* elem_glob, int rank, int nranks, nsk_ctx** out) {
  if (!P || !part_sub || !elem_glob || !out) return fail(NSK_EINVAL, "bad argument");
  if (!P->local || !P->local_done) return fail(NSK_EINVAL, "needs a context made by nsk_init_local and completed by nsk_local_finish");
  if (P->released) return fail(NSK_EINVAL, "parent context was released (nsk_shard_release_parent)");
  for (int e = 0; e < P->nel; ++e) {
    if (e && elem_glob[e] <= elem_glob[e - 1]) return fail(NSK_EINVAL, "elem_glob must ascend");
    if ((part_sub[e] == rank) != (P->local_own[e] != 0)) return fail(NSK_EINVAL, "part_sub disagrees with the own flags given to nsk_init_local");
  }
  return shard_create(P, part_sub, rank, nranks, out, elem_glob);
}

// Global ids of a shard's elements in its LOCAL order (boundary elements first, each group ascending): the order of the
// element blocks inside its vectors, i.e. what nsk_vec_upload / nsk_vec_download of a shard expect and return.
int nsk_shard_elems(nsk_ctx* shard, long long* out) {
  if (!shard || !shard->parent || !out) return fail(NSK_EINVAL, "needs a shard context");
  std::copy(shard->elems_glob.begin(), shard->elems_glob.end(), out);
  return 0;
}

int nsk_shard_halo_counts(nsk_ctx* shard, int* vel, int* pres_send, int* pres_recv) {
  if (!shard || !shard->parent || !vel || !pres_send || !pres_recv) return fail(NSK_EINVAL, "needs a shard context");
  for (int p = 0; p < shard->nranks; ++p) {
    vel[p] = 0;
    pres_send[p] = (p < (int)shard->ph_scnt.size()) ? shard->ph_scnt[p] : 0;
    pres_recv[p] = (p < (int)shard->ph_gcnt.size()) ? shard->ph_gcnt[p] : 0;
  }
  for (size_t k = 0; k < shard->peers.size(); ++k) vel[shard->peers[k]] = shard->vh_pcnt[k];
  return 0;
}

// Virtual ranks (several shards in one process) advance in stream order on ONE stream.  Shards cut from one parent share its
// stream; shards of separate rank-local parents are moved onto the first one's stream with this call.
int nsk_shard_share_stream(nsk_ctx* shard, nsk_ctx* leader) {
  if (!shard || !leader || !shard->parent || !leader->parent) return fail(NSK_EINVAL, "needs two shard contexts");
  HIPCHK(hipStreamSynchronize(shard->stream));
  shard->stream = leader->stream;
  return 0;
}

int nsk_finalize(nsk_ctx* c) {
  if (!c) return 0;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& a : c->graphs) for (auto& g : a) if (g.exec) (void)hipGraphExecDestroy(g.exec);
  for (auto& kv : c->gcache) if (kv.second) (void)hipGraphExecDestroy(kv.second);
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->hpin) (void)hipHostFree(c->hpin);
  if (c->hstat_pin) (void)hipHostFree(c->hstat_pin);
  if (c->h_step_iters) (void)hipHostFree(c->h_step_iters);
  if (c->hs_send) (void)hipHostFree(c->hs_send);
  if (c->hs_recv) (void)hipHostFree(c->hs_recv);
  if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); for (auto& e : c->orth_ev) if (e) (void)hipEventDestroy(e); (void)hipStreamDestroy(c->comm_stream); }
  for (auto& e : c->ov_ev) if (e) (void)hipEventDestroy(e);
  if (c->comm && rccl_rt::CommDestroy) (void)rccl_rt::CommDestroy(c->comm);
  if (c->stream && !c->parent) (void)hipStreamDestroy(c->stream);      // shards share the parent's stream (clones have their own)
  delete c;
  return 0;
}

int nsk_get_info(nsk_ctx* c, double* dt, int* nsteps, long long* nstate, long long* nvel, long long* npres) {
  if (!c) return fail(NSK_EINVAL, "null ctx");
  if (dt) *dt = c->dt;
  if (nsteps) *nsteps = c->nsteps;
  if (nstate) *nstate = c->nstate;
  if (nvel) *nvel = c->nloc;
  if (npres) *npres = c->npr;
  return 0;
}

int nsk_set_nsteps(nsk_ctx* c, int nsteps) {
  if (!c || nsteps < 1) return fail(NSK_EINVAL, "bad nsteps");
  c->nsteps = nsteps;
  return 0;
}

int nsk_set_tolerances(nsk_ctx* c, double th, double tp, int relative) {
  if (!c) return fail(NSK_EINVAL, "null ctx");
  c->d.tol_helm = th; c->d.tol_pres = tp; c->d.tol_relative = relative;
  invalidate_graphs(c);   // Dev is captured by value: re-capture
  for (int k = 0; k < NCLS; ++k) { c->cur_helm[k] = c->max_helm; c->cur_pres[k] = c->max_pres; }
  c->bh_n = 0;
  return 0;
}

int nsk_set_option(nsk_ctx* c, const char* name, double value) {
  if (!c || !name) return fail(NSK_EINVAL, "bad argument");
  const std::string n(name);
  if (n == "use_graph") c->use_graph = (int)value;
  else if (n == "shard_graph") c->shard_graph = (int)value;
  else if (n == "shard_hostcheck") c->shard_hostcheck = (int)value;
  else if (n == "hostcheck") c->hostcheck = (int)value;
  else if (n == "halo_overlap") c->halo_overlap = value != 0.0;
  else if (n == "rccl_fuse") c->rccl_fuse = (int)value;
  else if (n == "allred_verify") c->arv_left = (int)value;
  else if (n == "orth_overlap") c->orth_overlap = (int)value;
  else if (n == "nscal") {
    // krylov_vector%theta (core/krylov_subspace.f:13): carried by every vector operation and by the inner product; the time
    // steppers act on it as the reference does with ifheat = .false. -- identity (the scalar equation itself is not built)
    if (value < 0 || value > 8 || c->parent) return fail(NSK_EINVAL, "nscal: 0..8, full-mesh contexts only");
    c->nscal = (int)value;
    c->nstate = (long long)c->ndim * c->nloc + c->npr + (long long)c->nscal * c->nloc;
    int rc = dalloc(c, &c->scratch, (size_t)c->nstate);       // vectors allocated before this call keep the old length: set it first
    if (rc) return rc;
  }
  else if (n == "merged_iters") { c->merged_iters = std::max(0, std::min((int)value, MAXMR)); invalidate_graphs(c); }
  else if (n == "merged_update") { c->merged_update = (int)value; invalidate_graphs(c); }
  else if (n == "fuse2") { c->fuse2 = (int)value; c->tail_ok = -1; invalidate_graphs(c); }
  else if (n == "tc32") { c->tc32 = (int)value; invalidate_graphs(c); }
  else if (n == "fuse2_start") { c->fuse2_start = (int)value; invalidate_graphs(c); }
  else if (n == "sb_pct") { c->sb_pct = (int)value; }
  else if (n == "skip_close") { c->skip_close = (int)value; invalidate_graphs(c); }
  else if (n == "min_pres_iter") c->min_pres = (int)value;
  else if (n == "pres_cap") {
    if (value > 0 && c->ndim != 2) return fail(NSK_EINVAL, "pres_cap is validated on quadrilateral linearised maps only (DESIGN.md section 1)");
    c->pres_cap = (int)value;
  }
  else if (n == "helm_guess") c->helm_guess = (int)value;
  else if (n == "gs2_from") c->gs2_from = std::max(0, (int)value);
  else if (n == "gs_lag") c->gs_lag = (int)value;
  else if (n == "flat_proj") c->flat_proj = (int)value;
  else if (n == "divgs_c3") { c->divgs_c3 = (int)value; invalidate_graphs(c); }
  else if (n == "helm_pf") { c->helm_pf = (int)value; invalidate_graphs(c); }
  else if (n == "helm_pf_grid") { c->helm_pf_grid = std::max(8, (int)value / 8 * 8); invalidate_graphs(c); }      // resident workgroups of k_helm_p (tests: fewer than elements; default: what the device holds)
  else if (n == "zero_metrics") {       // 0: every array is loaded (the cleaned ones included: same bits); 1: back to the masks of the set-up
    c->d.zmask = value != 0.0 ? c->zmask_built : 0u; c->d.bfmask = (value != 0.0 && !c->d.bf_stride) ? c->bfmask_built : 0u;
    invalidate_graphs(c);
  }
  else if (n == "eapply_pipe") { c->eapply_pipe = (int)value; invalidate_graphs(c); }
  else if (n == "helm_fdm") {           // 0: back to Jacobi (the factors stay); 1: only if the set-up built them (NSK_HELM_FDM=1 or an anisotropic mesh)
    if (value != 0.0 && !c->d.hfS) return fail(NSK_EINVAL, "helm_fdm: the fast-diagonalisation factors were not built at set-up (NSK_HELM_FDM=1)");
    c->d.helm_fdm = value != 0.0 ? 1 : 0;
    invalidate_graphs(c);
  }
  else if (n == "graph_steps") { c->graph_steps = std::max(1, std::min((int)value, 64)); for (auto& a : c->graphs) a[NCLS].nh = -1; }
  else if (n == "dbg_max_order") c->dbg_max_order = (int)value;
  else if (n == "dbg_ab2") c->dbg_ab2 = (int)value;
  else if (n == "dbg_pext") c->dbg_pext = (int)value;
  else if (n == "mfma_convect") c->mfma_convect = (int)value;
  else if (n == "endtime") {                     // param(10): the sampling period T (Newton for periodic orbits changes it every iteration)
    if (!(value > 0.0)) return fail(NSK_EINVAL, "endtime must be positive");
    c->endtime = value;                          // takes effect at the next nsk_set_baseflow / nsk_set_orbit (dt, nsteps from the CFL rule)
  }
  else if (n == "gmres_cycle") c->gmres_cycle = std::max(2, std::min((int)value, MAXMR));
  else if (n == "fused") {
    if (value != 0 && !fused_possible(c)) return fail(NSK_EINVAL, "persistent velocity solve not available for this context (needs a quadrilateral single-rank context whose workgroups are all resident)");
    c->fused = value != 0;
    for (int k = 0; k < NCLS; ++k) c->cur_helm[k] = c->max_helm;
  }
  else if (n == "early_pres_mul") c->early_pres_mul = value;
  else if (n == "proj_reset") c->d.proj_reset = (int)value;
  else if (n == "proj_restart") c->d.proj_restart = (int)value;
  else if (n == "pres_floor") c->d.tol_pres_floor = value;
  else if (n == "dbg") { int v = (int)value; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &v, sizeof(int))); }
  else if (n == "budget_helm") { for (int k = 0; k < NCLS; ++k) c->cur_helm[k] = std::min(c->max_helm, std::max(1, (int)value)); }
  else if (n == "budget_pres") { for (int k = 0; k < NCLS; ++k) c->cur_pres[k] = std::min(c->max_pres, std::max(1, (int)value)); }
  else if (n == "budget_add_helm") { for (int k = 0; k < NCLS; ++k) c->cur_helm[k] = std::min(c->max_helm, std::max(1, c->cur_helm[k] + (int)value)); }
  else if (n == "budget_add_pres") { for (int k = 0; k < NCLS; ++k) c->cur_pres[k] = std::min(c->max_pres, std::max(1, c->cur_pres[k] + (int)value)); }
  else if (n == "budget_freeze") c->budget_freeze = (int)value;
  else if (n == "step_budgets") c->step_budgets = (int)value;
  else if (n == "tail") c->tail = (int)value;
  else if (n == "tail_off_h") c->tail_off_h = (int)value;
  else if (n == "tail_off_p") c->tail_off_p = (int)value;
  else return fail(NSK_EINVAL, "unknown option " + n);
  invalidate_graphs(c);
  return 0;
}

int nsk_vec_alloc(nsk_ctx* c, int n, nsk_vec* out) {
  if (!c || !out || n < 0) return fail(NSK_EINVAL, "bad argument");
  if (c->released) return fail(NSK_EINVAL, "context was released (nsk_shard_release_parent): only its shards compute");
  for (int k = 0; k < n; ++k) {
    double* p = nullptr;
    int rc = dalloc(c, &p, (size_t)c->nstate);
    if (rc) return rc;
    out[k] = p;
  }
  return 0;
}

int nsk_vec_free(nsk_ctx* c, int n, nsk_vec* v) {
  if (!c || !v) return fail(NSK_EINVAL, "bad argument");
  for (int k = 0; k < n; ++k) {
    auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)v[k]);
    if (it == c->allocs.end()) return fail(NSK_EINVAL, "unknown vector handle");
    (void)hipFree(v[k]);
    c->allocs.erase(it);
    c->alloc_bytes.erase((void*)v[k]);
    v[k] = nullptr;
  }
  return 0;
}

int nsk_vec_upload(nsk_ctx* c, nsk_vec v, const double* vx, const double* vy, const double* pr) {
  if (!c || !v) return fail(NSK_EINVAL, "bad argument");
  if (c->ndim != 2) return fail(NSK_EINVAL, "3-D context: use nsk_vec_upload3");
  double* p = (double*)v;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (vx) HIPCHK(hipMemcpy(p, vx, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  if (vy) HIPCHK(hipMemcpy(p + c->nloc, vy, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  if (pr) HIPCHK(hipMemcpy(p + 2 * c->nloc, pr, c->npr * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

int nsk_vec_upload_scalar(nsk_ctx* c, nsk_vec v, int m, const double* theta) {
  if (!c || !v || !theta || m < 0 || m >= c->nscal) return fail(NSK_EINVAL, "bad argument (scalar index < nscal)");
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy((double*)v + (long long)c->ndim * c->nloc + c->npr + (long long)m * c->nloc, theta, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}
int nsk_vec_download_scalar(nsk_ctx* c, nsk_vec v, int m, double* theta) {
  if (!c || !v || !theta || m < 0 || m >= c->nscal) return fail(NSK_EINVAL, "bad argument (scalar index < nscal)");
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(theta, (const double*)v + (long long)c->ndim * c->nloc + c->npr + (long long)m * c->nloc, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int nsk_vec_download(nsk_ctx* c, nsk_vec v, double* vx, double* vy, double* pr) {
  if (!c || !v) return fail(NSK_EINVAL, "bad argument");
  if (c->ndim != 2) return fail(NSK_EINVAL, "3-D context: use nsk_vec_download3");
  const double* p = (const double*)v;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (vx) HIPCHK(hipMemcpy(vx, p, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  if (vy) HIPCHK(hipMemcpy(vy, p + c->nloc, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  if (pr) HIPCHK(hipMemcpy(pr, p + 2 * c->nloc, c->npr * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int nsk_vec_upload3(nsk_ctx* c, nsk_vec v, const double* vx, const double* vy, const double* vz, const double* pr) {
  if (!c || !v) return fail(NSK_EINVAL, "bad argument");
  if (c->ndim != 3) return fail(NSK_EINVAL, "nsk_vec_upload3 needs a 3-D context");
  double* p = (double*)v;
  HIPCHK(hipStreamSynchronize(c->stream));
  const double* src[3] = {vx, vy, vz};
  for (int k = 0; k < 3; ++k)
    if (src[k]) HIPCHK(hipMemcpy(p + k * c->nloc, src[k], c->nloc * sizeof(double), hipMemcpyHostToDevice));
  if (pr) HIPCHK(hipMemcpy(p + 3 * c->nloc, pr, c->npr * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

int nsk_vec_download3(nsk_ctx* c, nsk_vec v, double* vx, double* vy, double* vz, double* pr) {
  if (!c || !v) return fail(NSK_EINVAL, "bad argument");
  if (c->ndim != 3) return fail(NSK_EINVAL, "nsk_vec_download3 needs a 3-D context");
  const double* p = (const double*)v;
  HIPCHK(hipStreamSynchronize(c->stream));
  double* dst[3] = {vx, vy, vz};
  for (int k = 0; k < 3; ++k)
    if (dst[k]) HIPCHK(hipMemcpy(dst[k], p + k * c->nloc, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  if (pr) HIPCHK(hipMemcpy(pr, p + 3 * c->nloc, c->npr * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int nsk_get_stats(nsk_ctx* c, nsk_stats* s) {
  if (!c || !s) return fail(NSK_EINVAL, "bad argument");
  const Stats& h = c->hstats;
  s->steps = h.steps; s->helm_iters = h.helm_iters; s->pres_iters = h.pres_iters;
  s->unconverged = h.unconverged; s->last_helm_res = h.last_helm_res; s->last_pres_res = h.last_pres_res;
  s->max_helm_iter = h.max_helm; s->max_pres_iter = h.max_pres;
  s->budget_helm = c->cur_helm[5]; s->budget_pres = c->cur_pres[5];
  s->recaptures = c->recaptures; s->retries = c->retries;
  s->capped_solves = h.capped_solves; s->worst_cap_ratio = h.worst_cap_ratio;
  s->total_capped_solves = c->tot_capped; s->total_worst_cap_ratio = c->tot_worst_cap;
  s->total_helm_iters = c->tot_helm_iters; s->total_pres_iters = c->tot_pres_iters; s->total_steps = c->tot_steps;
  s->recapture_seconds = c->recapture_s;
  s->total_pres_jsum = c->tot_pres_jsum; s->coarse_bytes_per_solve = c->coarse_bytes;
  s->step_budget_maps = c->sb_maps;
  s->tail_maps = c->tail_maps;
  s->zero_arrays = (long long)c->d.zmask | ((long long)c->d.bfmask << 12);
  s->step_budget_helm_mean = c->sb_maps ? (double)c->sb_launch_h / ((double)c->sb_steps) : 0.0;
  s->step_budget_pres_mean = c->sb_maps ? (double)c->sb_launch_p / ((double)c->sb_steps) : 0.0;
  return 0;
}

int nsk_matvec(nsk_ctx* c, int mode, nsk_vec fv, nsk_vec qv) {
  if (!c || !fv || !qv) return fail(NSK_EINVAL, "bad argument");
  if (c->released) return fail(NSK_EINVAL, "context was released (nsk_shard_release_parent): only its shards compute");
  double* f = (double*)fv; const double* q = (const double*)qv;
  c->hstats = Stats{};
  int rc = 0;
  switch (mode) {
    case NSK_DIRECT: rc = run_map_adaptive(c, 0, f, q); break;
    case NSK_ADJOINT: rc = run_map_adaptive(c, 1, f, q); break;
    case NSK_DIRECT_ADJOINT:                                     // core/matvec.f:343-346
      rc = run_map_adaptive(c, 0, f, q);
      if (!rc) rc = run_map_adaptive(c, 1, f, f);
      break;
    case NSK_NEWTON:                                             // core/matvec.f:398-401
      if (f == q) return fail(NSK_EINVAL, "newton map needs f != q");
      rc = run_map_adaptive(c, 0, f, q);
      if (!rc) hipLaunchKernelGGL(k_axpby, dim3((unsigned)((c->nstate + 255) / 256)), dim3(256), 0, c->stream, f, -1.0, q, 1.0, c->nstate);
      break;
    case NSK_FORCE_SENSITIVITY:                                  // core/matvec.f:366-371: f = -(exp(L^+ T) q - q)
      if (f == q) return fail(NSK_EINVAL, "force-sensitivity map needs f != q");
      rc = run_map_adaptive(c, 1, f, q);
      if (!rc) hipLaunchKernelGGL(k_axpby, dim3((unsigned)((c->nstate + 255) / 256)), dim3(256), 0, c->stream, f, 1.0, q, -1.0, c->nstate);
      break;
    default: return fail(NSK_EINVAL, "unknown mode");
  }
  return rc;
}

// Which of the twelve base-flow constants of the hexahedral convection kernel vanish on every dealiasing node (a two-dimensional
// base flow on a spanwise-extruded mesh: the third convecting component and five of the nine gradient terms): Dev::bfmask.
// "Vanish": largest magnitude below 1e-10 of the largest of its group (convecting field / gradient) -- d U / d z of a z-invariant
// field comes out of the derivative matrix as rounding noise, as the mapping's cross derivatives do (nsk3_setup.inc) -- and such an
// array is then SET to zero, so that loading it or not gives the same bits.  Steady sets only; a stored orbit loads everything.
__global__ void k_absmax_scan(const double* __restrict__ a, size_t n, unsigned long long* out) {
  const int q = blockIdx.y;
  const double* p = a + (size_t)q * n;
  double m = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double v = fabs(p[i]);
    if (!(v == v)) v = INFINITY;
    m = fmax(m, v);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out + q, (unsigned long long)__double_as_longlong(m));     // (non-negative doubles order as integers)
}
static int scan_bfmask(nsk_ctx* c) {
  Dev& d = c->d;
  d.bfmask = 0; c->bfmask_built = 0;
  if (c->ndim != 3 || !c->zero_metrics || !d.zmask || d.bf_stride || !d.bfc) return 0;
  unsigned long long* dm = nullptr;
  unsigned long long hm[12];
  HIPCHK(hipMalloc(&dm, sizeof(hm)));
  HIPCHK(hipMemsetAsync(dm, 0, sizeof(hm), c->stream));
  hipLaunchKernelGGL(k_absmax_scan, dim3(1024, 12), dim3(256), 0, c->stream, d.bfc, (size_t)d.nfine, dm);
  HIPCHK(hipMemcpyAsync(hm, dm, sizeof(hm), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(dm));
  double mx[12], gmax[2] = {0.0, 0.0};
  for (int q = 0; q < 12; ++q) { std::memcpy(&mx[q], &hm[q], sizeof(double)); gmax[q >= 3] = std::max(gmax[q >= 3], mx[q]); }
  unsigned mask = 0;
  for (int q = 0; q < 12; ++q)
    if (mx[q] <= 1e-10 * gmax[q >= 3]) {
      mask |= 1u << q;
      if (mx[q] != 0.0) HIPCHK(hipMemsetAsync((double*)d.bfc + (size_t)q * d.nfine, 0, (size_t)d.nfine * sizeof(double), c->stream));
    }
  HIPCHK(hipStreamSynchronize(c->stream));
  c->bfmask_built = mask;
  d.bfmask = mask;
  return 0;
}

// New linearisation point (core/newton_krylov.f:371-372 + prepare_linearized_solver): base-flow
// constants of the convection kernels, dt / nsteps from the CFL rule, Jacobi diagonals for the new dt.
int nsk_set_baseflow(nsk_ctx* c, nsk_vec qv) {
  if (!c || !qv) return fail(NSK_EINVAL, "bad argument");
  if (c->parent || c->nranks > 1) return fail(NSK_EINVAL, "shards: use nsk_group_set_baseflow");
  if (c->clone_of) return fail(NSK_EINVAL, "lanes share the base-flow constants of the context they were cloned from: set the base flow there, before nsk_clone");
  Dev& d = c->d;
  const double* q = (const double*)qv;
  DISPATCH_N(c->key, {
    hipLaunchKernelGGL(k_baseflow<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, q, (double*)d.cUr, (double*)d.cUs,
                       (double*)d.GUx, (double*)d.GUy, (double*)d.GVx, (double*)d.GVy);
  });
  const int nd = c->ndim;
  std::vector<double> u((size_t)nd * c->nloc);
  HIPCHK(hipMemcpyAsync(u.data(), q, (size_t)nd * c->nloc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  double ctarg = 0.0;
  for (long long l = 0; l < c->nloc; ++l) {
    double s = 0.0;
    for (int a = 0; a < nd; ++a) {
      double t = 0.0;
      for (int cc = 0; cc < nd; ++cc) t += u[(size_t)cc * c->nloc + l] * c->h_cflg[(size_t)nd * nd * l + a * nd + cc];
      s += std::fabs(t);
    }
    ctarg = std::max(ctarg, s);
  }
  if (!(ctarg > 0.0)) return fail(NSK_EINVAL, "base flow is zero");
  const double dt0 = c->cfl_target / ctarg;
  c->nsteps = (int)std::ceil(c->endtime / dt0);
  c->dt = c->endtime / c->nsteps;
  d.dt = c->dt;
  std::vector<double> dinv((size_t)3 * c->nloc);
  const double bd0[3] = {1.0, 1.5, 11.0 / 6.0};
  for (int k = 0; k < 3; ++k)
    for (long long l = 0; l < c->nloc; ++l) dinv[(size_t)k * c->nloc + l] = c->h_mask[l] / (d.nu * c->h_dAs[l] + bd0[k] / c->dt * c->h_bs[l]);
  HIPCHK(hipMemcpy((double*)d.dinv, dinv.data(), dinv.size() * sizeof(double), hipMemcpyHostToDevice));
  { const int rc = scan_bfmask(c); if (rc) return rc; }
  invalidate_graphs(c);           // coefficients are baked into the captured graphs
  return 0;
}

// Time-periodic base flow for Floquet analysis (uparam(1)=3.11, core/matvec.f:191-236, core/eigensolvers.f:201-210):
// integrate the full equations from q0 over one period T = endtime (DNS sponge towards the initial field with
// strength spng_str, core/utils.f:165-170) and store the base-flow constants of every step; the linearised
// maps then read slot istep-1 at step istep.  Returns Phi_T(q0) in `end` (periodicity check) if not NULL.
int nsk_set_orbit(nsk_ctx* c, nsk_vec q0v, double spng_str, nsk_vec end) {
  if (!c || !q0v) return fail(NSK_EINVAL, "bad argument");
  if (c->parent) return fail(NSK_EINVAL, "shards: use nsk_group_set_orbit");
  if (c->clone_of) return fail(NSK_EINVAL, "not on a lane (nsk_clone)");
  Dev& d = c->d;
  const double* q0 = (const double*)q0v;
  if (c->ndim == 3) {
    // hexahedra: the twelve dealiasing-mesh constants of every time step of the orbit, [nsteps][12][nfine] behind Dev::bfc
    if (d.bf_stride) { d.bfc = c->steady[0]; d.cUr = c->steady[0]; d.bf_stride = 0; }
    int rc = nsk_set_baseflow(c, q0v);                          // dt, nsteps from the CFL of the initial field
    if (rc) return rc;
    const long long nfine = d.nfine;
    if (c->orbit[0]) { nsk_vec v = c->orbit[0]; nsk_vec_free(c, 1, &v); c->orbit[0] = nullptr; }
    if ((rc = dalloc(c, &c->orbit[0], (size_t)c->nsteps * 12 * nfine))) return rc;
    double* vr = const_cast<double*>(d.spng_vr);
    if (!vr && (rc = dalloc(c, &vr, 3 * d.cs))) return rc;
    for (int cc = 0; cc < 3; ++cc) HIPCHK(hipMemcpyAsync(vr + cc * d.cs, q0 + cc * d.nloc, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    d.spng_vr = vr; d.nl_spng_str = spng_str;
    if (!d.bstep && (rc = dalloc(c, &d.bstep, 4))) return rc;
    invalidate_graphs(c);
    for (int k = 0; k < NCLS; ++k) { c->cur_helm[k] = c->max_helm; c->cur_pres[k] = c->max_pres; }
    c->bh_n = 0;
    HIPCHK(hipMemsetAsync(d.stats, 0, sizeof(Stats), c->stream));
    for (int cc = 0; cc < 3; ++cc) HIPCHK(hipMemcpyAsync(d.u + cc * d.cs, q0 + cc * d.nloc, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d.p, q0 + 3 * d.nloc, d.npr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    for (int istep = 1; istep <= c->nsteps; ++istep) {
      DISPATCH_N(c->key, {
        hipLaunchKernelGGL(k_baseflow<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, (const double*)d.u, c->orbit[0] + (size_t)(istep - 1) * 12 * nfine,
                           (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr);
      });
      if ((rc = step(c, istep, 2))) return rc;
    }
    Stats h;
    HIPCHK(hipMemcpyAsync(&h, d.stats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream));
    if (end) {
      double* f = (double*)end;
      for (int cc = 0; cc < 3; ++cc) HIPCHK(hipMemcpyAsync(f + cc * d.nloc, d.u + cc * d.cs, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(f + 3 * d.nloc, d.p, d.npr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (h.unconverged > 0) return fail(NSK_ENOCONV, "inner solve hit its iteration cap while integrating the base-flow orbit");
    c->steady[0] = d.bfc;
    d.bfc = c->orbit[0]; d.cUr = c->orbit[0]; d.bfmask = 0;
    d.bf_stride = 12 * nfine; c->orbit_steps = c->nsteps;
    invalidate_graphs(c);
    return 0;
  }
  if (d.bf_stride) {                                            // back to the steady arrays first
    d.cUr = c->steady[0]; d.cUs = c->steady[1]; d.GUx = c->steady[2]; d.GUy = c->steady[3]; d.GVx = c->steady[4]; d.GVy = c->steady[5];
    d.bf_stride = 0;
  }
  int rc = nsk_set_baseflow(c, q0v);                            // dt, nsteps from the CFL of the initial field
  if (rc) return rc;
  const long long nfine = (long long)c->nel * c->NDD;
  for (int k = 0; k < 6; ++k) {
    if (c->orbit[k]) { nsk_vec v = c->orbit[k]; nsk_vec_free(c, 1, &v); c->orbit[k] = nullptr; }
    if ((rc = dalloc(c, &c->orbit[k], (size_t)c->nsteps * nfine))) return rc;
  }
  double* vr = const_cast<double*>(d.spng_vr);                  // one buffer per context: Newton calls this every iteration
  if (!vr && (rc = dalloc(c, &vr, 2 * d.cs))) return rc;
  HIPCHK(hipMemcpyAsync(vr, q0, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(vr + d.cs, q0 + d.nloc, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  d.spng_vr = vr; d.nl_spng_str = spng_str;
  if (!d.bstep && (rc = dalloc(c, &d.bstep, 4))) return rc;
  invalidate_graphs(c);
  for (int k = 0; k < NCLS; ++k) { c->cur_helm[k] = c->max_helm; c->cur_pres[k] = c->max_pres; }
  c->bh_n = 0;
  HIPCHK(hipMemsetAsync(d.stats, 0, sizeof(Stats), c->stream));
  HIPCHK(hipMemcpyAsync(d.u, q0, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d.u + d.cs, q0 + d.nloc, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d.p, q0 + 2 * d.nloc, d.npr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  for (int istep = 1; istep <= c->nsteps; ++istep) {
    const long long off = (long long)(istep - 1) * nfine;
    DISPATCH_N(c->key, {
      hipLaunchKernelGGL(k_baseflow<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, (const double*)d.u, c->orbit[0] + off, c->orbit[1] + off,
                         c->orbit[2] + off, c->orbit[3] + off, c->orbit[4] + off, c->orbit[5] + off);
    });
    if ((rc = step(c, istep, 2))) return rc;
  }
  Stats h;
  HIPCHK(hipMemcpyAsync(&h, d.stats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream));
  if (end) {
    double* f = (double*)end;
    HIPCHK(hipMemcpyAsync(f, d.u, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(f + d.nloc, d.u + d.cs, d.nloc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(f + 2 * d.nloc, d.p, d.npr * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (h.unconverged > 0) return fail(NSK_ENOCONV, "inner solve hit its iteration cap while integrating the base-flow orbit");
  c->steady[0] = d.cUr; c->steady[1] = d.cUs; c->steady[2] = d.GUx; c->steady[3] = d.GUy; c->steady[4] = d.GVx; c->steady[5] = d.GVy;
  d.cUr = c->orbit[0]; d.cUs = c->orbit[1]; d.GUx = c->orbit[2]; d.GUy = c->orbit[3]; d.GVx = c->orbit[4]; d.GVy = c->orbit[5];
  d.bf_stride = nfine; c->orbit_steps = c->nsteps;
  invalidate_graphs(c);
  return 0;
}

// Phi_T(q): nsteps of the full Navier-Stokes equations (nonlinear_forward_map, core/newton_krylov.f:336-378);
// subtract_q != 0 returns Phi_T(q) - q, the Newton right-hand side.
int nsk_nonlinear_map(nsk_ctx* c, nsk_vec fv, nsk_vec qv, int subtract_q) {
  if (!c || !fv || !qv) return fail(NSK_EINVAL, "bad argument");
  if (fv == qv) return fail(NSK_EINVAL, "needs f != q");
  if (c->parent) return fail(NSK_EINVAL, "shards: use nsk_group_nonlinear_map");
  c->hstats = Stats{};
  int rc = run_map_adaptive(c, 2, (double*)fv, (const double*)qv);
  if (rc) return rc;
  if (subtract_q)
    hipLaunchKernelGGL(k_axpby, dim3((unsigned)((c->nstate + 255) / 256)), dim3(256), 0, c->stream, (double*)fv, -1.0, (const double*)qv, 1.0, c->nstate);
  return 0;
}

// ---- Krylov vector algebra ------------------------------------------------
static int dots_to_device(nsk_ctx* c, const double* f, const nsk_vec* Q, int nq) {
  if (nq > 1024) return fail(NSK_EINVAL, "too many vectors");
  for (int k = 0; k < nq; ++k) ((double**)c->hpin)[k] = (double*)Q[k];
  HIPCHK(hipMemcpyAsync(c->kptr, c->hpin, nq * sizeof(double*), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_dots, dim3(c->kblk), dim3(256), 0, c->stream, f, (const double* const*)c->kptr, nq, c->d.bm1s, c->nloc, c->kpart, c->kblk, c->ndim, (long long)c->ndim * c->nloc + c->npr, c->nscal);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, (const double*)c->kpart, nq, c->kblk, c->kout);
  return 0;
}

int nsk_dot(nsk_ctx* c, nsk_vec p, nsk_vec q, double* alpha) {
  if (!c || !p || !q || !alpha) return fail(NSK_EINVAL, "bad argument");
  HIPCHK(hipStreamSynchronize(c->stream));   // hpin reuse
  int rc = dots_to_device(c, (const double*)p, &q, 1);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(alpha, c->kout, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (std::isnan(*alpha)) return fail(NSK_ENAN, "NaN inner product");
  return 0;
}

int nsk_norm(nsk_ctx* c, nsk_vec p, double* alpha) {
  int rc = nsk_dot(c, p, p, alpha);
  if (!rc) *alpha = std::sqrt(*alpha);
  return rc;
}

int nsk_scal(nsk_ctx* c, nsk_vec p, double a) {
  if (!c || !p) return fail(NSK_EINVAL, "bad argument");
  hipLaunchKernelGGL(k_axpby, dim3((unsigned)((c->nstate + 255) / 256)), dim3(256), 0, c->stream, (double*)p, 0.0, (const double*)p, a, c->nstate);
  return 0;
}

int nsk_axpy(nsk_ctx* c, nsk_vec p, double a, nsk_vec q) {
  if (!c || !p || !q) return fail(NSK_EINVAL, "bad argument");
  hipLaunchKernelGGL(k_axpby, dim3((unsigned)((c->nstate + 255) / 256)), dim3(256), 0, c->stream, (double*)p, a, (const double*)q, 1.0, c->nstate);
  return 0;
}

int nsk_copy(nsk_ctx* c, nsk_vec dst, nsk_vec src) {
  if (!c || !dst || !src) return fail(NSK_EINVAL, "bad argument");
  HIPCHK(hipMemcpyAsync(dst, src, c->nstate * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int nsk_zero(nsk_ctx* c, nsk_vec p) {
  if (!c || !p) return fail(NSK_EINVAL, "bad argument");
  HIPCHK(hipMemsetAsync(p, 0, c->nstate * sizeof(double), c->stream));
  return 0;
}

// Rank-sharded contexts: the inner products are sums over ranks (`glsc3`, core/krylov_subspace.f:37-43).  The coefficients stay
// on the device: rank-local dots -> all-reduce of the j-vector on the stream (RCCL, or the host-staged transport) -> projection.
static int dots_allreduce(nsk_ctx* c, int n) {
  if (!c->parent || (!c->comm && !c->host_allred)) return 0;
  std::vector<nsk_ctx*> G(1, c);
  return allreduce_ptr(G, &nsk_ctx::kout, n);
}

int nsk_orth(nsk_ctx* c, nsk_vec fv, const nsk_vec* Q, int j, double* h, double* beta) {
  if (!c || !fv || (j > 0 && !Q) || !h || !beta || j < 0) return fail(NSK_EINVAL, "bad argument");
  if (j > 1000) return fail(NSK_EINVAL, "too many vectors");
  double* f = (double*)fv;
  int rc;
  if (!c->kacc && (rc = dalloc(c, &c->kacc, 1024))) return rc;
  // Everything below is queued on the stream without a host round trip: pointer table once, then per pass
  // local dots -> sum over ranks (device all-reduce) -> projection (+ accumulation of the coefficients on the device),
  // then norm -> sum over ranks -> scaling; ONE download of [h(0..j-1), |f|^2] at the end.
  HIPCHK(hipStreamSynchronize(c->stream));   // hpin reuse
  for (int k = 0; k < j; ++k) ((double**)c->hpin)[k] = (double*)Q[k];
  ((double**)c->hpin)[j] = f;
  HIPCHK(hipMemcpyAsync(c->kptr, c->hpin, (j + 1) * sizeof(double*), hipMemcpyHostToDevice, c->stream));
  const unsigned gridn = (unsigned)((c->nstate + 255) / 256);
  // RCCL ranks: the all-reduce of one chunk of coefficients runs on a second stream WHILE the dots of the next chunk are
  // computed (north_star: "inner-product all-reduces overlapped with orthogonalisation"; the reference issues 2 j
  // sequential scalar all-reduces, core/krylov_decomposition.f:165-196).  Chunks of >= 16 vectors, at most ORTH_CHUNKS.
  const bool overlap = c->parent && c->comm && !c->host_allred && c->orth_overlap && j >= 32;
  if (overlap && !c->comm_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (auto& e : c->orth_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  for (int ps = 0; ps < 2 && j > 0; ++ps) {        // two projection passes (re-orthogonalisation)
    if (overlap) {
      const int nch = std::min(nsk_ctx::ORTH_CHUNKS, j / 16);
      for (int ch = 0; ch < nch; ++ch) {
        const int o = (int)((long long)j * ch / nch), m = (int)((long long)j * (ch + 1) / nch) - o;
        hipLaunchKernelGGL(k_dots, dim3(c->kblk), dim3(256), 0, c->stream, (const double*)f, (const double* const*)(c->kptr + o), m, c->d.bm1s, c->nloc, c->kpart + (size_t)o * c->kblk, c->kblk, c->ndim, (long long)c->ndim * c->nloc + c->npr, c->nscal);
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, (const double*)(c->kpart + (size_t)o * c->kblk), m, c->kblk, c->kout + o);
        HIPCHK(hipEventRecord(c->orth_ev[2 * ch], c->stream));
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->orth_ev[2 * ch], 0));
        if (rccl_rt::AllReduce(c->kout + o, c->kout + o, m, rccl_rt::kDouble, rccl_rt::kSum, c->comm, c->comm_stream) != 0) return fail(NSK_EHIP, "ncclAllReduce failed (nsk_orth)");
        HIPCHK(hipEventRecord(c->orth_ev[2 * ch + 1], c->comm_stream));
      }
      for (int ch = 0; ch < nch; ++ch) HIPCHK(hipStreamWaitEvent(c->stream, c->orth_ev[2 * ch + 1], 0));
    } else {
      hipLaunchKernelGGL(k_dots, dim3(c->kblk), dim3(256), 0, c->stream, (const double*)f, (const double* const*)c->kptr, j, c->d.bm1s, c->nloc, c->kpart, c->kblk, c->ndim, (long long)c->ndim * c->nloc + c->npr, c->nscal);
      hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, (const double*)c->kpart, j, c->kblk, c->kout);
      if ((rc = dots_allreduce(c, j))) return rc;
    }
    hipLaunchKernelGGL(k_project_out_acc, dim3(gridn), dim3(256), 0, c->stream, f, (const double* const*)c->kptr, j, (const double*)c->kout, c->nstate, c->kacc, ps);
  }
  hipLaunchKernelGGL(k_dots, dim3(c->kblk), dim3(256), 0, c->stream, (const double*)f, (const double* const*)(c->kptr + j), 1, c->d.bm1s, c->nloc, c->kpart, c->kblk, c->ndim, (long long)c->ndim * c->nloc + c->npr, c->nscal);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, (const double*)c->kpart, 1, c->kblk, c->kout);
  if ((rc = dots_allreduce(c, 1))) return rc;
  hipLaunchKernelGGL(k_scale_rsqrt, dim3(gridn), dim3(256), 0, c->stream, f, (const double*)c->kout, c->nstate);
  HIPCHK(hipMemcpyAsync(c->kacc + j, c->kout, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  double* res = c->hpin + 2048;                    // pinned; the pointer table (first 1001 slots) was consumed in stream order
  HIPCHK(hipMemcpyAsync(res, c->kacc, (j + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < j; ++k) { if (std::isnan(res[k])) return fail(NSK_ENAN, "NaN inner product"); h[k] = res[k]; }
  if (std::isnan(res[j])) return fail(NSK_ENAN, "NaN norm");
  *beta = std::sqrt(res[j]);
  return 0;
}

int nsk_basis_gemm(nsk_ctx* c, nsk_vec* Q, int k, const double* Z, int ldz) {
  if (!c || !Q || !Z || k < 1 || k > 256 || ldz < k) return fail(NSK_EINVAL, "bad argument (k <= 256)");
  double* dZ = nullptr; double** dQ = nullptr; int rc;
  if ((rc = dalloc(c, &dZ, (size_t)k * ldz)) || (rc = dalloc(c, &dQ, k))) return rc;
  HIPCHK(hipMemcpy(dZ, Z, (size_t)k * ldz * sizeof(double), hipMemcpyHostToDevice));   // Z column-major: Z[c*ldz + q]
  HIPCHK(hipMemcpy(dQ, Q, k * sizeof(double*), hipMemcpyHostToDevice));
  const unsigned grid = (unsigned)((c->nstate + 63) / 64);                              // 4 waves x 16 entries
  const int ct = (k + 15) / 16;
#define NSK_GEMM(CT) hipLaunchKernelGGL(k_basis_gemm_mfma<CT>, dim3(grid), dim3(256), 0, c->stream, (double* const*)dQ, k, (const double*)dZ, ldz, c->nstate)
  if (ct <= 1) NSK_GEMM(1); else if (ct <= 2) NSK_GEMM(2); else if (ct <= 4) NSK_GEMM(4); else if (ct <= 8) NSK_GEMM(8); else NSK_GEMM(16);
#undef NSK_GEMM
  HIPCHK(hipStreamSynchronize(c->stream));
  nsk_vec a = dZ, b = dQ;
  nsk_vec_free(c, 1, &a); nsk_vec_free(c, 1, &b);
  return 0;
}

int nsk_basis_gemv(nsk_ctx* c, const nsk_vec* Q, int k, const double* y_re, const double* y_im, nsk_vec re, nsk_vec im) {
  if (!c || !Q || !y_re || !re || k < 1 || k > 1024) return fail(NSK_EINVAL, "bad argument");
  std::vector<double> Z((size_t)2 * k);
  for (int q = 0; q < k; ++q) { Z[q] = y_re[q]; Z[k + q] = y_im ? y_im[q] : 0.0; }
  double* dZ = nullptr; double** dQ = nullptr; double** dO = nullptr; int rc;
  if ((rc = dalloc(c, &dZ, Z.size())) || (rc = dalloc(c, &dQ, k)) || (rc = dalloc(c, &dO, 2))) return rc;
  double* outs[2] = {(double*)re, (double*)(im ? im : re)};
  HIPCHK(hipMemcpy(dZ, Z.data(), Z.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dQ, Q, k * sizeof(double*), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dO, outs, 2 * sizeof(double*), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_basis_comb, dim3((unsigned)((c->nstate + 255) / 256)), dim3(256), 0, c->stream, (const double* const*)dQ, k, (const double*)dZ, k, 0, (im && y_im) ? 2 : 1, (double* const*)dO, c->nstate);
  HIPCHK(hipStreamSynchronize(c->stream));
  nsk_vec a = dZ, b = dQ, e = dO;
  nsk_vec_free(c, 1, &a); nsk_vec_free(c, 1, &b); nsk_vec_free(c, 1, &e);
  return 0;
}

int nsk_seed_noise(nsk_ctx* c, nsk_vec v) {
  if (!c || !v) return fail(NSK_EINVAL, "bad argument");
  if (c->parent) return fail(NSK_EINVAL, "nsk_seed_noise: seed the full-mesh context and scatter (shards do not hold the global element ids)");
  if (!c->xyz) return fail(NSK_EINVAL, "context holds no coordinates");
  double* q = (double*)v;
  const unsigned grid = (unsigned)((c->nloc + 255) / 256);
  HIPCHK(hipMemsetAsync(q, 0, c->nstate * sizeof(double), c->stream));                // pressure (and scalars): zero
  hipLaunchKernelGGL(k_seed_rand, dim3(grid), dim3(256), 0, c->stream, (const double*)c->xyz, c->nloc, c->ndim, c->N, q);
  hipLaunchKernelGGL(k_seed_avg, dim3(grid), dim3(256), 0, c->stream, c->d, (const double*)q, c->scratch, 0);
  hipLaunchKernelGGL(k_seed_avg, dim3(grid), dim3(256), 0, c->stream, c->d, (const double*)c->scratch, q, 1);
  return 0;
}

// diagnostic (NSK_STAMPS build): run `reps` full pressure iterations' k_divgs and dump the stamps
int nsk_debug_stamps(nsk_ctx* c, unsigned long long* out, int nblk_max) {
  if (!c || !out) return fail(NSK_EINVAL, "bad argument");
  Dev& d = c->d;
  const size_t nrow = (size_t)c->nblk + (size_t)(d.nvert + 7) / 8 + 8;           // (k_schwarz_uc: Schwarz + coarse workgroups)
  if (!d.dbg) { int rc = dalloc(c, &d.dbg, (size_t)16 * nrow); if (rc) return rc; }
  HIPCHK(hipMemset(d.dbg, 0, (size_t)16 * nrow * sizeof(unsigned long long)));
  const char* sk = std::getenv("NSK_STAMP_KERNEL");
  if (sk && (std::string(sk) == "schwarz_uc" || std::string(sk) == "divgs_t")) {
    if (!fuse2_on(c)) return fail(NSK_EINVAL, "two-launch GMRES iteration not available in this context");
    Dev dd = d; dd.tol_pres = 0.0; dd.tol_relative = 0; dd.pres_cap = 0;
    dd.tc32 = (c->tc32 != 0 && d.Tc32) ? 1 : 0;
    const StepCoef sc = make_coef(c, 17, 0);
    const double scale = 1.0 / (sc.h2 * std::sqrt(d.vol));
    const int jd = std::getenv("NSK_STAMP_J") ? std::atoi(std::getenv("NSK_STAMP_J")) : 3;
    HIPCHK(hipMemsetAsync((char*)d.gsc + offsetof(GmresScal, done), 0, sizeof(int), c->stream));
    DISPATCH_N(c->key, {
      for (int r = 0; r < 5; ++r) {
        if (std::string(sk) == "schwarz_uc") { launch_divgs_t<N>(c, dd, jd - 1); launch_schwarz_uc<N>(c, dd, jd, scale, 2, 5); }
        else { launch_schwarz_uc<N>(c, dd, jd, scale, 2, 5); launch_divgs_t<N>(c, dd, jd); }
      }
    });
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, d.dbg, (size_t)16 * std::min<size_t>((size_t)nblk_max, nrow) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    d.dbg = nullptr;
    c->state_dirty = true;
    return 0;
  }
  DISPATCH_N(c->key, {
    if (nblk_max < 0) {
      Dev dd = d; dd.tol_helm = 0.0; dd.tol_relative = 0;
      const StepCoef sc = make_coef(c, 3, 0);
      for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(k_helm<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, dd, sc, r, (const double*)d.rloc);
      nblk_max = -nblk_max;
    } else if (std::getenv("NSK_STAMP_KERNEL") && std::string(std::getenv("NSK_STAMP_KERNEL")) == "divgs_w") {
      for (int r = 0; r < 5; ++r) launch_divgs3<N>(c, d, c->nblk, (const double*)d.yl, c->wp2, -1, 0, 3);
    } else if (std::getenv("NSK_STAMP_KERNEL") && std::string(std::getenv("NSK_STAMP_KERNEL")) == "schwarz_w") {
      for (int r = 0; r < 5; ++r) launch_schwarz3<N>(c, d, c->nblk, (const double*)d.V, d.Z, 1, 0, 2);
    } else if (std::getenv("NSK_STAMP_KERNEL") && std::string(std::getenv("NSK_STAMP_KERNEL")) == "schwarz_p") {
      for (int r = 0; r < 5; ++r) launch_schwarz3<N>(c, d, c->nblk, (const double*)d.V, d.Z, 1, 0, 1);
    } else if (std::getenv("NSK_STAMP_KERNEL") && std::string(std::getenv("NSK_STAMP_KERNEL")) == "schwarz") {
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_schwarz<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, (const double*)d.V, d.Z, 1, 0);
    } else {
      const int jd = std::getenv("NSK_STAMP_J") ? std::atoi(std::getenv("NSK_STAMP_J")) : 5;
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, (const double*)d.yl, c->wp2, jd, 0);
    }
  });
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, d.dbg, (size_t)16 * std::min(nblk_max, c->nblk) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long* p = d.dbg; d.dbg = nullptr; (void)p;
  return 0;
}

// ---- a second LANE of a context: its own stream and state, every immutable device array shared -----------------------------
// A single map leaves ~40 % of the launch pipeline idle on config 2 (a time step is a chain of dependent 5-15 us kernels);
// independent maps on separate streams fill it: two lanes reach 1.6x the matvecs/s of one.  Who has two
// independent vectors: a band (block) Arnoldi factorisation (nekstab_amd/krylov.py: band_arnoldi), parameter sweeps.
// Quadrilateral, full-mesh contexts; vectors of any lane are plain device memory and may be used on every lane.
int nsk_clone(nsk_ctx* P, nsk_ctx** out) {
  if (!P || !out) return fail(NSK_EINVAL, "bad argument");
  if (P->parent || P->clone_of || P->ndim != 2 || P->released) return fail(NSK_EINVAL, "nsk_clone: quadrilateral full-mesh contexts only");
  if (P->d.bf_stride) return fail(NSK_EINVAL, "nsk_clone: not with a stored base-flow orbit");
  nsk_ctx* c = new nsk_ctx();
  auto bail = [&](int rc) { std::string keep = g_err; nsk_finalize(c); g_err = keep; return rc; };
  // scalars and options
  c->N = P->N; c->NN = P->NN; c->M = P->M; c->MM = P->MM; c->ND = P->ND; c->NDD = P->NDD; c->EPB = P->EPB; c->NT = P->NT; c->NTD = P->NTD;
  c->ndim = 2; c->key = P->key; c->hrows = P->hrows; c->hstride = P->hstride;
  c->nel = P->nel; c->nblk = P->nblk; c->nvert = P->nvert; c->nloc = P->nloc; c->npr = P->npr; c->nstate = P->nstate; c->nscal = 0;
  if (P->nscal) { delete c; return fail(NSK_EINVAL, "nsk_clone: not with scalar fields"); }
  c->dt = P->dt; c->re = P->re; c->endtime = P->endtime; c->nsteps = P->nsteps;
  c->max_helm = P->max_helm; c->max_pres = P->max_pres; c->min_pres = P->min_pres; c->pres_cap = P->pres_cap; c->layers = P->layers;
  c->use_graph = P->use_graph; c->gmres_cycle = P->gmres_cycle; c->helm_guess = P->helm_guess; c->early_pres_mul = P->early_pres_mul;
  c->merged_iters = P->merged_iters; c->merged_update = P->merged_update; c->gs2_from = P->gs2_from; c->gs_lag = P->gs_lag; c->debug = P->debug;
  c->dbg_max_order = P->dbg_max_order; c->dbg_ab2 = P->dbg_ab2; c->dbg_pext = P->dbg_pext;
  c->PS = P->PS; c->coarse_lda = P->coarse_lda; c->cfl_target = P->cfl_target; c->xyz = P->xyz;
  c->clone_of = P;
  // two persistent grids launched at once can each get a share of the CUs' slots and wait for the rest for ever: contexts with
  // lanes run their maps on launch budgets only
  if (P->tail != 0) { P->tail = 0; invalidate_graphs(P); }
  c->tail = 0;
  for (int k = 0; k < NCLS; ++k) { c->cur_helm[k] = c->max_helm; c->cur_pres[k] = c->max_pres; }
  if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return fail(NSK_EHIP, "hipStreamCreate"); }
  Dev& d = c->d;
  d = P->d;                                               // every immutable pointer: geometry, bases, gather tables, preconditioner
  d.dbg = nullptr;
  d.stepctr = nullptr; d.step_iters = nullptr;            // (the lane's own per-step record is made at its first map)
  const long long npr = c->npr;
  int rc;
  // the mutable set of build(): time-stepper state, CG / GMRES work arrays, projection space, partial sums, counters
  if ((rc = dalloc(c, &d.u, 2 * d.cs)) || (rc = dalloc(c, &d.p, npr)) || (rc = dalloc(c, &d.plag, npr)) ||
      (rc = dalloc(c, &d.pext, npr)) || (rc = dalloc(c, &d.ulag, 4 * d.cs)) || (rc = dalloc(c, &d.exlag, 4 * d.cs)) ||
      (rc = dalloc(c, &d.bf, 2 * d.cs)) || (rc = dalloc(c, &d.rloc, 2 * d.cs)) || (rc = dalloc(c, &d.bloc, 2 * d.cs)) || (rc = dalloc(c, &d.dulag, 6 * d.cs)) || (rc = dalloc(c, &d.hx, 2 * d.cs)) ||
      (rc = dalloc(c, &d.hr, 2 * d.cs)) || (rc = dalloc(c, &d.hp, 2 * d.cs)) || (rc = dalloc(c, &d.hs, 2 * d.cs)) ||
      (rc = dalloc(c, &d.hwl, 4 * d.cs)) || (rc = dalloc(c, &d.hpart, (size_t)16 * c->nblk)) || (rc = dalloc(c, &d.hscal, 32)) ||
      (rc = dalloc(c, &d.V, (size_t)(MAXMR + 1) * d.ps)) || (rc = dalloc(c, &d.Z, (size_t)MAXMR * npr)) ||
      (rc = dalloc(c, &d.yl, 2 * d.cs)) || (rc = dalloc(c, &d.ec, (size_t)c->nel * 4)) ||
      (rc = dalloc(c, &d.gpart, (size_t)(MAXMR + 2) * c->nblk)) || (rc = dalloc(c, &d.gpart2, (size_t)(MAXMR + 2) * c->nblk)) || (rc = dalloc(c, &d.gsc, 1)) ||
      (rc = dalloc(c, &d.stats, 1)) || (rc = dalloc(c, &c->wv1, 2 * d.cs)) || (rc = dalloc(c, &c->wv2, 2 * d.cs)) ||
      (rc = dalloc(c, &c->wp1, npr)) || (rc = dalloc(c, &c->wp2, npr)) || (rc = dalloc(c, &c->scratch, (size_t)c->nstate)) ||
      (rc = dalloc(c, &d.xacc, npr)) || (rc = dalloc(c, &d.xc, c->nvert))) return bail(rc);
  if (d.nproj_max > 0)
    if ((rc = dalloc(c, &d.PX, (size_t)d.nproj_max * npr)) || (rc = dalloc(c, &d.PEX, (size_t)d.nproj_max * npr)) ||
        (rc = dalloc(c, &d.PD, npr)) || (rc = dalloc(c, &d.PED, npr)) || (rc = dalloc(c, &d.ppart, (size_t)(MAXPROJ + 2) * c->nblk))) return bail(rc);
  if (P->rc_big && (rc = dalloc(c, &c->rc_big, c->coarse_lda))) return bail(rc);
  if (P->d.rch && (rc = dalloc(c, &d.rch, (size_t)MAXMR * c->coarse_lda))) return bail(rc);
  if (P->d.ecv && (rc = dalloc(c, &d.ecv, (size_t)8 * c->coarse_lda))) return bail(rc);
  if (P->d.Wr && (rc = dalloc(c, &d.Wr, (size_t)d.ps))) return bail(rc);            // (Tc / evl are immutable: shared with the parent)
  if (d.use_tot && ((rc = dalloc(c, &d.htot, 32)) || (rc = dalloc(c, &d.gtot, MAXMR + 8)) || (rc = dalloc(c, &d.gtot2, MAXMR + 8)) || (rc = dalloc(c, &d.ptot, MAXPROJ + 2)))) return bail(rc);
  if ((rc = dalloc(c, &c->sync, 2 * SYNC_WORDS))) return bail(rc);
  c->kblk = 256;
  if ((rc = dalloc(c, &c->kpart, (size_t)c->kblk * 1024)) || (rc = dalloc(c, &c->kout, 1024)) || (rc = dalloc(c, &c->kptr, 1024))) return bail(rc);
  if (hipHostMalloc((void**)&c->hpin, 4096 * sizeof(double)) != hipSuccess || hipHostMalloc((void**)&c->hstat_pin, sizeof(Stats)) != hipSuccess)
    return bail(fail(NSK_ENOMEM, "hipHostMalloc"));
  c->bm1s_host = P->bm1s_host;
  if ((rc = ensure_step_record(c))) return bail(rc);
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = c;
  return 0;
}

// b independent maps f_k = map(q_k), one per lane, all in flight at once: launch everything, then collect; a lane whose
// launch budgets ran out repeats its map on its own.
int nsk_matvec_batch(nsk_ctx** lanes, int b, int mode, nsk_vec* f, nsk_vec* q) {
  if (!lanes || b < 1 || b > 8 || !f || !q) return fail(NSK_EINVAL, "bad argument (1 <= b <= 8)");
  if (mode != NSK_DIRECT && mode != NSK_ADJOINT) return fail(NSK_EINVAL, "nsk_matvec_batch: direct or adjoint map");
  for (int k = 0; k < b; ++k) {
    if (!lanes[k] || !f[k] || !q[k] || f[k] == q[k]) return fail(NSK_EINVAL, "nsk_matvec_batch: needs f != q on every lane");
    if (lanes[k]->parent || lanes[k]->released) return fail(NSK_EINVAL, "nsk_matvec_batch: full-mesh lanes only");
    for (int j = 0; j < k; ++j) if (lanes[j] == lanes[k]) return fail(NSK_EINVAL, "nsk_matvec_batch: one map per lane");
    lanes[k]->hstats = Stats{};
  }
  const int adj = mode == NSK_ADJOINT ? 1 : 0;
  // inputs were written on other lanes' streams (orthogonalisation runs on lane 0): every lane starts behind all of them
  for (int k = 0; k < b; ++k) HIPCHK(hipStreamSynchronize(lanes[k]->stream));
  // one host thread per lane: a map is ~180 graph launches (or ~10^4 kernel launches in eager mode), and two lanes fed by one
  // thread reach 1.3x the rate of one where two threads reach 1.5-1.6x (scripts/lanes_bench.py)
  std::vector<int> rcs(b, 0);
  std::vector<std::string> errs(b);
  auto run_lane = [&](int k) {
    for (;;) {
      int rc = map_launch(lanes[k], adj, (double*)f[k], (const double*)q[k]);
      if (!rc) rc = map_finish(lanes[k]);
      if (rc <= 0) { rcs[k] = rc; if (rc) errs[k] = g_err; return; }       // 1: redo with the larger budgets map_finish has set
    }
  };
  std::vector<std::thread> th;
  for (int k = 1; k < b; ++k) th.emplace_back(run_lane, k);
  run_lane(0);
  for (auto& t : th) t.join();
  for (int k = 0; k < b; ++k) if (rcs[k]) return fail(rcs[k], "lane " + std::to_string(k) + ": " + errs[k]);
  return 0;
}

// iteration counts of every time step of the LAST map of this context: helm[s] CG iterations (the slower component), pres[s]
// GMRES iterations of step s + 1; returns the number of steps recorded through *nsteps (0: no map yet, or a sharded context)
int nsk_get_step_iters(nsk_ctx* c, int n, int* helm, int* pres, int* nsteps) {
  if (!c || n < 0 || (n > 0 && (!helm || !pres)) || !nsteps) return fail(NSK_EINVAL, "bad argument");
  HIPCHK(hipStreamSynchronize(c->stream));
  const int m = c->h_step_iters ? c->step_rec_n : 0;
  *nsteps = m;
  for (int s = 0; s < std::min(n, m); ++s) { helm[s] = c->h_step_iters[2 * s]; pres[s] = c->h_step_iters[2 * s + 1]; }
  return 0;
}

// average duration of one hot kernel, measured with HIP events on the library's stream
int nsk_bench_kernel(nsk_ctx* c, const char* name, int reps, double* avg_us) {
  if (!c || !name || reps < 1 || !avg_us) return fail(NSK_EINVAL, "bad argument");
  const std::string n(name);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  Dev d = c->d;
  float ms = 0.f;
  if (n == "helm" || n == "helm_wg") {
    d.tol_helm = 0.0; d.tol_relative = 0;                       // never converge: every launch does full work
    const StepCoef sc = make_coef(c, 3, 0);
    const int cyc = 8;
    const int keep_pf = c->helm_pf;
    if (n == "helm_wg") c->helm_pf = 0;                         // one workgroup per element whatever the context runs
    HIPCHK(hipStreamSynchronize(c->stream));
    DISPATCH_N(c->key, {
      for (int r = 0; r < cyc; ++r) {                             // warm
        launch_helm_iter<N>(c, d, sc, r % cyc, (const double*)d.rloc);
        tot_rows(c, d.hpart + (size_t)(r & 1) * c->hrows * c->nblk, c->hrows, d.htot + (r & 1) * c->hstride);
      }
      HIPCHK(hipEventRecord(e0, c->stream));
      for (int r = 0; r < reps; ++r) {
        launch_helm_iter<N>(c, d, sc, r % cyc, (const double*)d.rloc);
        tot_rows(c, d.hpart + (size_t)(r & 1) * c->hrows * c->nblk, c->hrows, d.htot + (r & 1) * c->hstride);   // use_tot contexts only (a few us, included)
      }
      HIPCHK(hipEventRecord(e1, c->stream));
    });
    c->helm_pf = keep_pf;
  } else if (n == "convect_nl" || n == "convect_mfma_nl") {
    // the full equations' convection term (mode 2): thread-per-node kernel / matrix-core kernel (hexahedra, lx1 = 10)
    if (n == "convect_mfma_nl" && c->key != 110) return fail(NSK_EINVAL, "k_convect_mfma_nl<10>: hexahedra with lx1 = 10");
    HIPCHK(hipStreamSynchronize(c->stream));
    DISPATCH_N(c->key, {
      for (int r = 0; r < reps + 3; ++r) {
        if (r == 3) HIPCHK(hipEventRecord(e0, c->stream));
        if (n == "convect_mfma_nl") hipLaunchKernelGGL(nsk::k3::k_convect_mfma_nl<10>, dim3(c->nel), dim3(1024), 0, c->stream, d, (const double*)d.u, d.bf);
        else hipLaunchKernelGGL(k_convect<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, (const double*)d.u, d.bf, 2);
      }
      HIPCHK(hipEventRecord(e1, c->stream));
    });
  } else if (n == "convect" || n == "convect_mfma") {
    if (n == "convect_mfma" && c->key != 108 && c->key != 110) return fail(NSK_EINVAL, "k_convect_mfma8 / k_convect_mfma<10>: hexahedra with lx1 = 8 or 10");
    HIPCHK(hipStreamSynchronize(c->stream));
    DISPATCH_N(c->key, {
      for (int r = 0; r < reps + 3; ++r) {
        if (r == 3) HIPCHK(hipEventRecord(e0, c->stream));
        if (n == "convect_mfma" && c->key == 108) hipLaunchKernelGGL(nsk::k3::k_convect_mfma8, dim3(c->nel), dim3(512), 0, c->stream, d, (const double*)d.u, d.bf, 0);
        else if (n == "convect_mfma") hipLaunchKernelGGL(nsk::k3::k_convect_mfma<10>, dim3(c->nel), dim3(512), 0, c->stream, d, (const double*)d.u, d.bf, 0);
        else hipLaunchKernelGGL(k_convect<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, (const double*)d.u, d.bf, 0);
      }
      HIPCHK(hipEventRecord(e1, c->stream));
    });
  } else if (n == "helm_fused" || n == "helm_fused0") {
    // persistent velocity solve with 8 (or 0) CG iterations per launch, never converging: the difference of the two
    // is the cost of 8 iterations (axhelm + updates + one grid barrier each)
    if (!c->fused) return fail(NSK_EINVAL, "persistent velocity solve not enabled in this context");
    d.tol_helm = 0.0; d.tol_relative = 0;
    const StepCoef sc = make_coef(c, 3, 0);
    const int its = (n == "helm_fused") ? 8 : 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    DISPATCH_N(c->key, {
      for (int r = 0; r < 4; ++r) { HIPCHK(hipMemsetAsync(c->sync, 0, SYNC_WORDS * sizeof(unsigned), c->stream)); launch_fused<N>(c, sc, its, &d); }
      HIPCHK(hipEventRecord(e0, c->stream));
      for (int r = 0; r < reps; ++r) { HIPCHK(hipMemsetAsync(c->sync, 0, SYNC_WORDS * sizeof(unsigned), c->stream)); launch_fused<N>(c, sc, its, &d); }
      HIPCHK(hipEventRecord(e1, c->stream));
    });
  } else if (c->ndim == 3 && (n == "divgs" || n == "divgs_wg" || n == "divgs_w" || n == "divgs_c3" || n == "schwarz" || n == "schwarz_wg" || n == "schwarz_p" || n == "schwarz_w" || n == "schwarz_w16" || n == "schwarz_q" || n == "gradt" || n.rfind("gs_lag", 0) == 0 || n.rfind("gs_dots", 0) == 0 || n == "pres_update" || n == "vel_update_proj" || n == "pres_rhs" || n == "rhs")) {
    // hexahedral pressure kernels back to back on the state the last map left: the E apply without its dots ("divgs"), the
    // Schwarz preconditioner + D^T ("schwarz"), the streaming Gram-Schmidt passes at basis index j ("gs_lag<j>", "gs_dots<j>")
    d.tol_pres = 0.0; d.tol_relative = 0; d.pres_cap = 0;
    const StepCoef sc = make_coef(c, 17, 0);
    const int jj = (n.rfind("gs_lag", 0) == 0) ? std::atoi(n.c_str() + 6) : ((n.rfind("gs_dots", 0) == 0) ? std::atoi(n.c_str() + 7) : 3);
    if (jj < 0 || jj + 1 >= MAXMR) return fail(NSK_EINVAL, "basis index out of range");
    HIPCHK(hipMemsetAsync((char*)d.gsc + offsetof(GmresScal, done), 0, sizeof(int), c->stream));
    HIPCHK(hipMemsetAsync((char*)d.gsc + offsetof(GmresScal, pending), 0, sizeof(int), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    DISPATCH_N(c->key, {
      constexpr int NT = Cfg<N>::NT;
      for (int r = -3; r < reps; ++r) {
        if (r == 0) HIPCHK(hipEventRecord(e0, c->stream));
        if (n == "divgs") launch_divgs3<N>(c, d, c->nblk, (const double*)d.yl, c->wp2, -1, 0, 0);                 // the form the context runs
        else if (n == "divgs_wg") launch_divgs3<N>(c, d, c->nblk, (const double*)d.yl, c->wp2, -1, 0, 0, 0);
        else if (n == "divgs_w") launch_divgs3<N>(c, d, c->nblk, (const double*)d.yl, c->wp2, -1, 0, 3);
        else if (n == "divgs_c3") launch_divgs3<N>(c, d, c->nblk, (const double*)d.yl, c->wp2, -1, 0, 0, 1);
        else if (n == "schwarz") launch_schwarz3<N>(c, d, c->nblk, (const double*)(d.V + (size_t)jj * d.ps), d.Z + (size_t)jj * d.npr, 1, 0);      // the form the context runs
        else if (n == "schwarz_wg") launch_schwarz3<N>(c, d, c->nblk, (const double*)(d.V + (size_t)jj * d.ps), d.Z + (size_t)jj * d.npr, 1, 0, 0);
        else if (n == "schwarz_p") launch_schwarz3<N>(c, d, c->nblk, (const double*)(d.V + (size_t)jj * d.ps), d.Z + (size_t)jj * d.npr, 1, 0, 1);
        else if (n == "schwarz_w") launch_schwarz3<N>(c, d, c->nblk, (const double*)(d.V + (size_t)jj * d.ps), d.Z + (size_t)jj * d.npr, 1, 0, 2);
        else if (n == "schwarz_w16") launch_schwarz3<N>(c, d, c->nblk, (const double*)(d.V + (size_t)jj * d.ps), d.Z + (size_t)jj * d.npr, 1, 0, 4);
        else if (n == "schwarz_q") launch_schwarz3<N>(c, d, c->nblk, (const double*)(d.V + (size_t)jj * d.ps), d.Z + (size_t)jj * d.npr, 1, 0, 5);
        else if (n == "gradt") hipLaunchKernelGGL(k_gradt<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)c->wp1, d.yl);
        else if (n == "pres_update") hipLaunchKernelGGL(k_pres_update<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc);
        else if (n == "vel_update_proj") hipLaunchKernelGGL(k_vel_update_proj<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc);
        else if (n == "pres_rhs") hipLaunchKernelGGL(k_pres_rhs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc, 1, 7);
        else if (n == "rhs") hipLaunchKernelGGL(k_rhs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc);
        else if (n.rfind("gs_dots", 0) == 0) launch_gs_dots3<N>(c, d, jj);
        else {                                               // gs_lag<j>: the streaming pass alone (its totals and the column kernel are microseconds)
          if constexpr (N <= 10) {
            constexpr int ROWS = (nsk::k3::Cfg<N>::MM + 63) / 64;
            constexpr int R4 = ROWS < 4 ? ROWS : 4, R2 = ROWS < 2 ? ROWS : 2;
            const dim3 grid(std::min<unsigned>((unsigned)((c->nel + 3) / 4), 2048u)), blk(256);
            if (jj <= 8) hipLaunchKernelGGL((nsk::k3::k_gs_lag<N, 8, R4>), grid, blk, 0, c->stream, d, jj);
            else if (jj <= 16) hipLaunchKernelGGL((nsk::k3::k_gs_lag<N, 16, R2>), grid, blk, 0, c->stream, d, jj);
            else if (jj <= 32) hipLaunchKernelGGL((nsk::k3::k_gs_lag<N, 32, 1>), grid, blk, 0, c->stream, d, jj);
            else hipLaunchKernelGGL((nsk::k3::k_gs_lag<N, MAXMR, 1>), grid, blk, 0, c->stream, d, jj);
          }
        }
      }
      HIPCHK(hipEventRecord(e1, c->stream));
    });
  } else if (n == "coarse" || n == "schwarz" || n == "divgs" || n == "gmres_update" || n == "pres_chain" || n == "pres_chain3" || n == "pres_chain_merged" || n == "pres_chain_fused" || n.rfind("schwarz_uc", 0) == 0 || n == "divgs_t" || n.rfind("update_coarse", 0) == 0 || n == "divgs2" ||
             n == "proj_apply" || n == "proj_apply_e" || n == "proj_update" || n == "pres_update" || n == "vel_update_proj" || n == "pres_rhs" || n == "rhs") {
    // kernels of the pressure solve, back to back on the state the last map left (run one first).  Tolerance 0 and a cleared
    // `done` flag: every launch does full work.  `pres_chain`: whole GMRES iterations j = 0..7 (coarse, Schwarz, E, update).
    if (c->ndim != 2 || d.coarse_lda > 3072) return fail(NSK_EINVAL, "pressure-kernel timing: quadrilateral contexts with the dense in-LDS coarse solve");
    if ((n.rfind("update_coarse", 0) == 0 || n == "pres_chain_merged") && !d.ecv) return fail(NSK_EINVAL, "merged coarse-solve kernel not available in this context");
    if ((n.rfind("schwarz_uc", 0) == 0 || n == "divgs_t" || n == "pres_chain_fused") && !(d.Tc && d.Wr && d.ecv && d.rch)) return fail(NSK_EINVAL, "two-launch GMRES iteration not available in this context");
    d.tol_pres = 0.0; d.tol_relative = 0; d.pres_cap = 0;
    d.tc32 = (c->tc32 != 0 && d.Tc32) ? 1 : 0;              // (as a production solve)
    const StepCoef sc = make_coef(c, 17, 0);
    const double scale = 1.0 / (sc.h2 * std::sqrt(d.vol));
    const int jj = 3;
    auto clear_done = [&]() { return hipMemsetAsync((char*)d.gsc + offsetof(GmresScal, done), 0, sizeof(int), c->stream); };
    HIPCHK(hipStreamSynchronize(c->stream));
    DISPATCH_N(c->key, {
      constexpr int NT = Cfg<N>::NT;
      for (int r = -3; r < reps; ++r) {
        if (r == 0) HIPCHK(hipEventRecord(e0, c->stream));
        if (r <= 0 || n.rfind("update_coarse", 0) == 0 || n.rfind("schwarz_uc", 0) == 0 || n == "gmres_update") HIPCHK(clear_done());   // (a 4-byte memset node per launch, same for all)
        if (n == "coarse") hipLaunchKernelGGL(k_coarse, dim3((d.nvert + 4 * CROWS_W - 1) / (4 * CROWS_W)), dim3(256), d.coarse_lda * sizeof(double), c->stream, d);
        else if (n.rfind("update_coarse", 0) == 0) launch_update_coarse(c, d, std::atoi(n.c_str() + 13), scale, 2, 5);
        else if (n.rfind("schwarz_uc", 0) == 0) launch_schwarz_uc<N>(c, d, std::atoi(n.c_str() + 10), scale, 2, 5);
        else if (n == "divgs_t") launch_divgs_t<N>(c, d, jj);
        else if (n == "pres_chain_fused") {
          HIPCHK(clear_done());
          for (int j = 0; j < 8; ++j) { launch_schwarz_uc<N>(c, d, j, scale, 2, 5); launch_divgs_t<N>(c, d, j); }
        }
        else if (n == "divgs2") hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)d.yl, d.V + (size_t)(jj + 1) * d.ps, jj, 2);
        else if (n == "schwarz") hipLaunchKernelGGL(k_schwarz<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)(d.V + (size_t)jj * d.ps), d.Z + (size_t)jj * d.npr, 1, 1);
        else if (n == "divgs") hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)d.yl, d.V + (size_t)(jj + 1) * d.ps, jj, 1);
        else if (n == "gmres_update") hipLaunchKernelGGL(k_gmres_update<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, jj, scale, 2, 5);
        else if (n == "proj_apply") hipLaunchKernelGGL(k_proj_apply, dim3(c->nblk), dim3(256), 0, c->stream, d);
        else if (n == "proj_apply_e") launch_proj_apply_e<N>(c, d);
        else if (n == "proj_update") hipLaunchKernelGGL(k_proj_update, dim3(c->nblk), dim3(256), 0, c->stream, d);
        else if (n == "pres_update") hipLaunchKernelGGL(k_pres_update<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc);
        else if (n == "vel_update_proj") hipLaunchKernelGGL(k_vel_update_proj<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc);
        else if (n == "pres_rhs") hipLaunchKernelGGL(k_pres_rhs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc, 1, 7);
        else if (n == "rhs") hipLaunchKernelGGL(k_rhs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, sc);
        else if (n == "pres_chain_merged") {
          HIPCHK(clear_done());
          for (int j = 0; j < 8; ++j) {
            launch_update_coarse(c, d, j, scale, 2, 5);
            hipLaunchKernelGGL(k_schwarz<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)(d.V + (size_t)j * d.ps), d.Z + (size_t)j * d.npr, 1, 1);
            hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)d.yl, d.V + (size_t)(j + 1) * d.ps, j, 2);
          }
        } else {
          HIPCHK(clear_done());
          for (int j = 0; j < 8; ++j) {
            hipLaunchKernelGGL(k_coarse, dim3((d.nvert + 4 * CROWS_W - 1) / (4 * CROWS_W)), dim3(256), d.coarse_lda * sizeof(double), c->stream, d);
            hipLaunchKernelGGL(k_schwarz<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)(d.V + (size_t)j * d.ps), d.Z + (size_t)j * d.npr, 1, 1);
            hipLaunchKernelGGL(k_divgs<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, (const double*)d.yl, d.V + (size_t)(j + 1) * d.ps, j, 1);
            if (n == "pres_chain") hipLaunchKernelGGL(k_gmres_update<N>, dim3(c->nblk), dim3(NT), 0, c->stream, d, j, scale, 2, 5);
          }
        }
      }
      HIPCHK(hipEventRecord(e1, c->stream));
    });
    if (n == "pres_chain" || n == "pres_chain3" || n == "pres_chain_merged" || n == "pres_chain_fused") reps *= 8;                  // per GMRES iteration
  } else {
    return fail(NSK_EINVAL, "unknown kernel " + n);
  }
  HIPCHK(hipEventSynchronize(e1));
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *avg_us = 1e3 * ms / reps;
  HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1));
  HIPCHK(hipMemsetAsync(c->d.stats, 0, sizeof(Stats), c->stream));
  // the kernels above ran hundreds of times on the live solver state (lags shifted, the projection space fed with itself):
  // not a state any map left.  The next map of this context starts from the state nsk_init left (reset_solver_state).
  c->state_dirty = true;
  return 0;
}

// ---- test hooks -------------------------------------------------------------
int nsk_test_axhelm(nsk_ctx* c, const double* u, double h1, double h2, double* out) {
  HIPCHK(hipMemcpy(c->wv1, u, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  DISPATCH_N(c->key, { hipLaunchKernelGGL(k_axhelm_test<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, c->d, (const double*)c->wv1, h1, h2, c->wv2); });
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, c->wv2, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int nsk_test_dssum(nsk_ctx* c, const double* u, double* out) {
  HIPCHK(hipMemcpy(c->wv1, u, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_dssum_test, dim3((unsigned)((c->nloc + 255) / 256)), dim3(256), 0, c->stream, c->d, (const double*)c->wv1, c->wv2);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, c->wv2, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int nsk_test_opdiv(nsk_ctx* c, const double* u, const double* v, double* out) {
  if (c->ndim != 2) return fail(NSK_EINVAL, "2-D hook: use nsk_test_op3");
  HIPCHK(hipMemcpy(c->wv1, u, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->wv1 + c->d.cs, v, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  DISPATCH_N(c->key, { hipLaunchKernelGGL(k_opdiv_test<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, c->d, (const double*)c->wv1, c->wp1); });
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, c->wp1, c->npr * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int nsk_test_opgradt(nsk_ctx* c, const double* p, double* ox, double* oy) {
  if (c->ndim != 2) return fail(NSK_EINVAL, "2-D hook: use nsk_test_op3");
  HIPCHK(hipMemcpy(c->wp1, p, c->npr * sizeof(double), hipMemcpyHostToDevice));
  DISPATCH_N(c->key, { hipLaunchKernelGGL(k_gradt<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, c->d, (const double*)c->wp1, c->wv1); });
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(ox, c->wv1, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(oy, c->wv1 + c->d.cs, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int nsk_test_convect(nsk_ctx* c, int adjoint, const double* u, const double* v, double* ox, double* oy) {
  if (c->ndim != 2) return fail(NSK_EINVAL, "2-D hook: use nsk_test_op3");
  HIPCHK(hipMemcpy(c->wv1, u, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->wv1 + c->d.cs, v, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  DISPATCH_N(c->key, { hipLaunchKernelGGL(k_convect<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, c->d, (const double*)c->wv1, c->wv2, adjoint); });
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(ox, c->wv2, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(oy, c->wv2 + c->d.cs, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

// 3-D element operators with packed arguments: which = 1: weak divergence ([3*P] -> [P2]); 2: D^T p ([P2] -> [3*P]);
// 3: convection term, mode a = 0 direct / 1 adjoint / 2 full equations ([3*P] -> [3*P]);
// 5: Helmholtz solve of BDF order a ([3*P] unassembled rhs -> [3*P]), iteration count in *iters
int nsk_test_op3(nsk_ctx* c, int which, const double* in, double* out, int a, int* iters) {
  if (!c || !in || !out) return fail(NSK_EINVAL, "bad argument");
  if (c->ndim != 3) return fail(NSK_EINVAL, "nsk_test_op3 needs a 3-D context");
  Dev& d = c->d;
  const size_t nv = 3 * (size_t)c->nloc;
  if (which == 1) {
    HIPCHK(hipMemcpy(c->wv1, in, nv * sizeof(double), hipMemcpyHostToDevice));
    DISPATCH_N(c->key, { hipLaunchKernelGGL(k_opdiv_test<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, (const double*)c->wv1, c->wp1); });
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->wp1, c->npr * sizeof(double), hipMemcpyDeviceToHost));
  } else if (which == 2) {
    HIPCHK(hipMemcpy(c->wp1, in, c->npr * sizeof(double), hipMemcpyHostToDevice));
    DISPATCH_N(c->key, { hipLaunchKernelGGL(k_gradt<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, (const double*)c->wp1, c->wv1); });
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->wv1, nv * sizeof(double), hipMemcpyDeviceToHost));
  } else if (which == 3) {
    HIPCHK(hipMemcpy(c->wv1, in, nv * sizeof(double), hipMemcpyHostToDevice));
    DISPATCH_N(c->key, { hipLaunchKernelGGL(k_convect<N>, dim3(c->nel), dim3(Cfg<N>::NTD), 0, c->stream, d, (const double*)c->wv1, c->wv2, a); });
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->wv2, nv * sizeof(double), hipMemcpyDeviceToHost));
  } else if (which == 8) {                             // convection term on the matrix cores (lx1 = 8; a = 0 direct / 1 adjoint)
    if ((c->key != 108 && c->key != 110) || (a == 2 && c->key != 110)) return fail(NSK_EINVAL, "k_convect_mfma8 / k_convect_mfma<10>: hexahedra with lx1 = 8 or 10, modes 0 and 1; k_convect_mfma_nl<10>: mode 2 at lx1 = 10");
    HIPCHK(hipMemcpy(c->wv1, in, nv * sizeof(double), hipMemcpyHostToDevice));
    if (c->key == 108) hipLaunchKernelGGL(nsk::k3::k_convect_mfma8, dim3(c->nel), dim3(512), 0, c->stream, d, (const double*)c->wv1, c->wv2, a);
    else if (a == 2) hipLaunchKernelGGL(nsk::k3::k_convect_mfma_nl<10>, dim3(c->nel), dim3(1024), 0, c->stream, d, (const double*)c->wv1, c->wv2);
    else hipLaunchKernelGGL(nsk::k3::k_convect_mfma<10>, dim3(c->nel), dim3(512), 0, c->stream, d, (const double*)c->wv1, c->wv2, a);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->wv2, nv * sizeof(double), hipMemcpyDeviceToHost));
  } else if (which == 5) {
    HIPCHK(hipMemcpy(d.rloc, in, nv * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.bloc, d.rloc, nv * sizeof(double), hipMemcpyDeviceToDevice));
    HIPCHK(hipMemset(d.stats, 0, sizeof(Stats)));
    const StepCoef sc = make_coef(c, a, 0);
    DISPATCH_N(c->key, {
      for (int it = 0; it < c->max_helm; ++it) {
        launch_helm_iter<N>(c, d, sc, it, (const double*)d.rloc);
        tot_rows(c, d.hpart + (size_t)(it & 1) * c->hrows * c->nblk, c->hrows, d.htot + (it & 1) * c->hstride);
      }
    });
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, d.hx, nv * sizeof(double), hipMemcpyDeviceToHost));
    Stats h; HIPCHK(hipMemcpy(&h, d.stats, sizeof(Stats), hipMemcpyDeviceToHost));
    if (iters) *iters = (int)h.helm_iters;
  } else if (which == 6 || which == 7) {               // pressure preconditioner: 6 = Schwarz only, 7 = Schwarz + coarse
    HIPCHK(hipMemcpy(d.V, in, c->npr * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d.gsc, 0, sizeof(GmresScal)));
    std::vector<double> part(c->nblk, 0.0);
    for (long long q = 0; q < c->npr; ++q) part[q / c->MM] += in[q] * in[q];
    HIPCHK(hipMemcpy(d.gpart, part.data(), c->nblk * sizeof(double), hipMemcpyHostToDevice));
    DISPATCH_N(c->key, {
      tot_rows(c, d.gpart, 1, d.gtot);
      hipLaunchKernelGGL(k_gmres_update<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, -1, 1.0, 1, 0);   // normalises V[0], fills ec
      hipLaunchKernelGGL(k_coarse_restrict_csr, dim3(d.coarse_lda / 256), dim3(256), 0, c->stream, d, c->rc_big);
      if (c->coarse_iter) { int rc2 = coarse_iterative(c, c->rc_big); if (rc2) return rc2; }
      else hipLaunchKernelGGL(k_coarse_big, dim3((d.nvert + 3) / 4), dim3(256), 0, c->stream, d, (const double*)c->rc_big);
      hipLaunchKernelGGL(k_schwarz<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, (const double*)d.V, d.Z, which == 7 ? 1 : 0, 0);
    });
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, d.Z, c->npr * sizeof(double), hipMemcpyDeviceToHost));
    double nrm = 0; for (long long q = 0; q < c->npr; ++q) nrm += in[q] * in[q];
    nrm = std::sqrt(nrm);
    for (long long q = 0; q < c->npr; ++q) out[q] *= nrm;     // undo the normalisation of V[0]
  } else return fail(NSK_EINVAL, "unknown operator");
  return 0;
}

int nsk_test_eapply(nsk_ctx* c, const double* p, double* out) {
  HIPCHK(hipMemcpy(c->wp1, p, c->npr * sizeof(double), hipMemcpyHostToDevice));
  int rc = eapply(c, c->wp1, c->wp2);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, c->wp2, c->npr * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int nsk_test_helm_solve(nsk_ctx* c, const double* rx, const double* ry, int order, double* ox, double* oy, int* iters) {
  if (c->ndim != 2) return fail(NSK_EINVAL, "2-D hook: use nsk_test_op3");
  Dev& d = c->d;
  HIPCHK(hipMemcpy(d.rloc, rx, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.rloc + d.cs, ry, c->nloc * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.bloc, d.rloc, 2 * d.cs * sizeof(double), hipMemcpyDeviceToDevice));
  HIPCHK(hipMemset(d.stats, 0, sizeof(Stats)));
  const StepCoef sc = make_coef(c, order, 0);
  const int nh = c->max_helm;
  DISPATCH_N(c->key, {
    for (int it = 0; it < nh; ++it) {
      hipLaunchKernelGGL(k_helm<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, sc, it, (const double*)d.rloc);
      tot_rows(c, d.hpart + (size_t)(it & 1) * c->hrows * c->nblk, c->hrows, d.htot + (it & 1) * c->hstride);
    }
  });
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(ox, d.hx, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(oy, d.hx + d.cs, c->nloc * sizeof(double), hipMemcpyDeviceToHost));
  Stats h; HIPCHK(hipMemcpy(&h, d.stats, sizeof(Stats), hipMemcpyDeviceToHost));
  if (iters) *iters = (int)h.helm_iters;
  return 0;
}

int nsk_test_pres_solve(nsk_ctx* c, const double* g, double* out, int* iters) {
  Dev& d = c->d;
  // V[0] = g, partial norms, then GMRES; result y with E0 y = g  (returned without the h2 factor)
  HIPCHK(hipMemcpy(d.V, g, c->npr * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d.stats, 0, sizeof(Stats)));
  HIPCHK(hipMemset(d.pext, 0, c->npr * sizeof(double)));
  std::vector<double> part(c->nblk, 0.0);
  for (long long q = 0; q < c->npr; ++q) part[(q / c->MM) / c->EPB] += g[q] * g[q];
  HIPCHK(hipMemcpy(d.gpart, part.data(), c->nblk * sizeof(double), hipMemcpyHostToDevice));
  StepCoef sc = make_coef(c, 3, 0);
  const int savep = d.nproj_max; d.nproj_max = 0; c->in_test = 1;
  int rc = pres_solve_launch(c, 1.0, 3, c->max_pres);
  c->in_test = 0;
  if (rc) return rc;
  sc.h2 = 1.0;
  DISPATCH_N(c->key, { hipLaunchKernelGGL(k_pres_update<N>, dim3(c->nblk), dim3(Cfg<N>::NT), 0, c->stream, d, sc); });
  HIPCHK(hipStreamSynchronize(c->stream));
  d.nproj_max = savep;
  HIPCHK(hipMemcpy(out, d.p, c->npr * sizeof(double), hipMemcpyDeviceToHost));
  Stats h; HIPCHK(hipMemcpy(&h, d.stats, sizeof(Stats), hipMemcpyDeviceToHost));
  if (iters) *iters = (int)h.pres_iters;
  return 0;
}

}  // extern "C"
