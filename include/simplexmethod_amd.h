/*
 * simplexmethod_amd.h — C ABI of the MI355X-native dense-LP hot path.
 *
 * The reference (haskell-md2/SimplexMethod) has no FFI/plugin interface: its
 * boundary is the C++ class API `Solver(const Canonical&)` + `solve()`
 * (/root/reference/src/SimplexSolover.h:285-328) and, by README intent
 * (README.md:27,40-42), the same shape for `EnumerationSolver`
 * (src/EnumerationSolver.h:3-10, an empty stub).  The functions below are what a
 * maintainer binds instead of the Eigen arithmetic inside those two classes; the
 * C++ wrappers in simplexmethod_amd/host/ (Solver, EnumerationSolver) are that
 * binding and keep the reference's names, arguments and exception behaviour.
 * See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - plain pointers and sizes only; caller allocates every output; no
 *     ownership transfer; no exceptions cross the ABI.
 *   - matrices are COLUMN-MAJOR fp64 (Eigen's default order, so the reference's
 *     `A.data()` can be passed as is); indices are 32-bit int; combination ranks
 *     are 64-bit unsigned.
 *   - m = rows of the canonical A, n = ALL canonical columns (slacks included),
 *     n_orig = Canonical::GetOriginalVariablesCount() (Canonical.cpp:151-154).
 *   - every entry point returns one of the LP_* codes below; negative values are
 *     HIP runtime failures (-(int)hipError_t); lp_last_error() has the text.
 *   - there is NO CPU fallback: without a usable gfx950 device
 *     lp_context_create fails and every other call needs a context.
 */
#ifndef SIMPLEXMETHOD_AMD_H
#define SIMPLEXMETHOD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LP_ABI_VERSION 4 /* 2: lp_simplex_stats grew algo_used / fell_back; lp_enum_shard_abstain, lp_batched_shard_bounds; 3: LP_SIMPLEX_ALGO_OVERLAP, lp_enum_exact_division, lp_debug_reciprocal; 4: lp_debug_division */

/* Status codes; the C++ wrappers map them back to the
 * reference's exception types and messages.                                     */
enum {
    LP_OPTIMAL = 0,     /* Solver::solve returned                  SimplexSolover.h:432-440 */
    LP_UNBOUNDED = 1,   /* std::runtime_error, objective unbounded SimplexSolover.h:442-444 */
    LP_ITER_LIMIT = 2,  /* std::runtime_error, iteration limit     SimplexSolover.h:450     */
    LP_SINGULAR = 3,    /* std::runtime_error "Singular basis matrix"        :125-126       */
    LP_INFEASIBLE = 4,  /* enumeration found no feasible basis                              */
    LP_BAD_ARG = 5      /* std::invalid_argument from Canonical's ctor, Canonical.cpp:27-46;
                           "basis index out of range", SimplexSolover.h:101                 */
};

/* Enumeration verdict for one basis subset. */
enum { LP_SUBSET_FEASIBLE = 0, LP_SUBSET_INFEASIBLE = 1, LP_SUBSET_SINGULAR = 2 };

typedef struct lp_context lp_context; /* one HIP device + stream + scratch */

int lp_abi_version(void);
int lp_device_count(void);
/* Binds HIP device `device` (must be gfx950-class; fails loudly otherwise).
 * `stream` may be NULL (the library creates its own non-blocking stream) or an
 * existing hipStream_t owned by the caller.                                      */
int lp_context_create(int device, void* stream, lp_context** ctx_out);
void lp_context_destroy(lp_context* ctx);
const char* lp_last_error(const lp_context* ctx);
const char* lp_status_string(int status);
int lp_context_sync(lp_context* ctx);

/* =========================================================================
 * Simplex  —  replaces Solver::solveWithBasis / simplexIter / computeBFS
 *             (SimplexSolover.h:408-451, :135-209, :117-133)
 * ========================================================================= */

/* One-shot: upload, solve on the GPU, download.
 * basis_in : Canonical::GetBasisIndices() (m entries, by basis position).
 * eps      : Solver::EPS (1e-9, SimplexSolover.h:13).
 * max_iter : MAX_ITER (10000, SimplexSolover.h:426).
 * x_out    : n_orig doubles = solve()'s return value (:435-439).
 * basis_out: m ints, final basis BY POSITION (the reference's local N, :419).
 * obj_out  : c . x (Canonical::Evaluate, Canonical.cpp:86).
 * iters_out: pivots executed.   Any of basis_out/obj_out/iters_out may be NULL. */
int lp_simplex_solve(lp_context* ctx, const double* A, int m, int n, const double* b,
                     const double* c, const int* basis_in, int maximize, int n_orig, double eps,
                     int max_iter, double* x_out, int* basis_out, double* obj_out, int* iters_out);

/* Device-resident form (used by bench.py so that timed regions start with the
 * tableau already in HBM).                                                       */
typedef struct lp_simplex_problem lp_simplex_problem;

enum {
    LP_SIMPLEX_ALGO_AUTO = 0,
    LP_SIMPLEX_ALGO_LAUNCH = 1,    /* one select + one rank-1-update launch per pivot   */
    LP_SIMPLEX_ALGO_LOOKAHEAD = 2, /* J pivots staged by a one-workgroup selector, then
                                      one rank-J update pass over the tableau            */
    LP_SIMPLEX_ALGO_RESIDENT = 3,  /* one launch per solve: the tableau stays in the registers of
                                      co-resident workgroups, one per CU (32 columns each for
                                      m <= 512, 16 columns each for 512 < m <= 960; at most 256
                                      workgroups, i.e. n <= 8192 resp. n <= 4096), one all-to-all
                                      hand-off per pivot; AUTO's choice when the shape fits.  A
                                      hand-off that times out (the workgroups never became
                                      co-resident) re-runs the solve on another algorithm:
                                      lp_simplex_stats::fell_back                          */
    LP_SIMPLEX_ALGO_OVERLAP = 4    /* one launch per pivot: the rank-1 update of pivot k streams the
                                      tableau out of place while one more workgroup of the same launch
                                      selects pivot k+1 from the old tableau and pivot k's eta; shapes
                                      beyond the chip-resident ones (a second tableau buffer is
                                      allocated on first use)                                */
};

typedef struct lp_simplex_stats {
    int status;
    int pivots;             /* pivots executed                                            */
    int launches;           /* kernel launches issued                                      */
    float solve_ms;         /* HIP-event time of the whole solve on the library's stream  */
    float update_ms;        /* HIP-event time spent in tableau-update launches: the rank-1 / rank-J
                               update launches when lp_simplex_profile is on; for the chip-
                               resident algorithm the one kernel that runs every pivot (always) */
    int update_launches;    /* launches counted in update_ms                               */
    double bytes_per_pivot; /* algorithmic bytes of one rank-1 update: 16*m*(n+1)          */
    int algo_used;          /* LP_SIMPLEX_ALGO_* that produced the answer (never AUTO)             */
    int fell_back;          /* 1: the chip-resident algorithm was asked for (or chosen by AUTO), a
                               hand-off timed out and the solve was re-run on algo_used; the answer is
                               the same, the solve took >= 200 ms longer.  0 otherwise.              */
} lp_simplex_stats;

int lp_simplex_upload(lp_context* ctx, const double* A, int m, int n, const double* b,
                      const double* c, const int* basis_in, int maximize, int n_orig,
                      lp_simplex_problem** problem_out);
/* Restores the initial tableau (device-to-device) so a solve can be repeated.   */
int lp_simplex_reset(lp_simplex_problem* p);
int lp_simplex_run(lp_simplex_problem* p, double eps, int max_iter, int algo,
                   lp_simplex_stats* stats_out);
/* on != 0: the next runs bracket every tableau-update launch with HIP events so that
 * lp_simplex_stats::update_ms / update_launches are filled (costs ~1-2 us per launch; off by
 * default, in which case those two fields are 0).                                         */
int lp_simplex_profile(lp_simplex_problem* p, int on);
/* trace_*: first trace_cap pivots (entering column, leaving POSITION); tableau_out:
 * (m+1) x (n+1) row-major, rows by basis position, row m = reduced costs,
 * column n = xB.  Every pointer may be NULL.                                      */
int lp_simplex_download(lp_simplex_problem* p, double* x_out, int* basis_out, double* obj_out,
                        int* trace_enter, int* trace_leave, int trace_cap, double* tableau_out);
void lp_simplex_free(lp_simplex_problem* p);

/* ---- Two-phase simplex ---------------------------------------
 * For canonical problems without a usable starting basis (Symmetrical min problems,
 * negative b).  Replaces the flow the reference sketches in code its public API cannot reach
 * (SimplexSolover.h:61-68 make_b_nonneg, :70-95 createAuxiliaryProblem, :331-381
 * replaceArtificialColumns, :383-406 twoPhaseSimplex), made consistent: rows with b < -eps
 * change sign; phase I minimises the sum of m artificials [A' | I] from their identity basis;
 * LP_INFEASIBLE iff that sum > eps; an artificial still basic leaves for the first non-basic
 * original column with |T[pos][cand]| > eps (none: LP_SINGULAR, dependent constraints); phase II
 * CONTINUES ON THE PHASE-I TABLEAU: the reduced-cost row is re-priced from the original costs
 * over the current basis, the artificial columns stay but are barred from entering (no second
 * upload, no re-inversion; the vertex carries phase I's rounding, ~1e-16 relative).  Every pivot
 * runs on the GPU.
 * iters_out (optional): 3 ints = pivots of phase I, drive-out pivots, pivots of phase II.       */
int lp_simplex_two_phase(lp_context* ctx, const double* A, int m, int n, const double* b,
                         const double* c, int maximize, int n_orig, double eps, int max_iter,
                         double* x_out, int* basis_out, double* obj_out, int* iters_out);
/* Row `row` (0..m-1 by basis position, m = reduced costs) of the current tableau: n+1 doubles. */
int lp_simplex_row(lp_simplex_problem* p, int row, double* out);
/* One Gauss-Jordan pivot at (row, col) of the current tableau, chosen by the caller
 * (replaceArtificialColumns, :357-366: N(pos) = cand; Binv = F * Binv).                        */
int lp_simplex_force_pivot(lp_simplex_problem* p, int row, int col);

/* Times `iters` launches of the rank-1 update kernel alone on the problem's
 * current tableau (pivot element (row, col) must be non-zero; the tableau is
 * restored afterwards).  ms_per_launch = HIP-event time / iters.                  */
int lp_bench_rank1_update(lp_simplex_problem* p, int row, int col, int iters,
                          float* ms_per_launch_out);

/* The same for the look-ahead path's rank-J update: the selector stages one batch of J pivots
 * on the problem's initial tableau (call after lp_simplex_reset), then the update launch is
 * replayed `iters` times between two HIP events.  *pivots_per_launch_out = J actually staged;
 * algorithmic bytes per launch = J * 16*m*(n+1).  State is restored afterwards.            */
int lp_bench_rankj_update(lp_simplex_problem* p, int iters, float* ms_per_launch_out,
                          int* pivots_per_launch_out);

/* Diagnostic (not part of the drop-in surface): first call with cap_pivots > 0 turns the
 * selectors' per-phase cycle stamps on; a later call copies the stamps (s_memtime ticks) of the
 * last run into `out`: after a look-ahead run 8 per pivot for the first cap_pivots pivots; after a
 * chip-resident run 16 per-phase cycle sums over the solve for each of the first cap_pivots (<= 256)
 * workgroups.                                                                              */
int lp_debug_simplex_stamps(lp_simplex_problem* p, int cap_pivots, unsigned long long* out);

/* Batched simplex (BASELINE.json configs[4]): `batch` independent LPs of one
 * shape, one LP per workgroup.  Arrays are concatenated per LP: A batch*m*n
 * (each column-major), b batch*m, c batch*n, basis_in batch*m; outputs x_out
 * batch*n_orig, basis_out batch*m, obj_out/iters_out/status_out batch.           */
int lp_simplex_solve_batched(lp_context* ctx, int batch, const double* A, int m, int n,
                             const double* b, const double* c, const int* basis_in, int maximize,
                             int n_orig, double eps, int max_iter, double* x_out, int* basis_out,
                             double* obj_out, int* iters_out, int* status_out);

typedef struct lp_batched_problem lp_batched_problem;
int lp_batched_upload(lp_context* ctx, int batch, const double* A, int m, int n, const double* b,
                      const double* c, const int* basis_in, int maximize, int n_orig,
                      lp_batched_problem** problem_out);
int lp_batched_run(lp_batched_problem* p, double eps, int max_iter, float* ms_out);
int lp_batched_download(lp_batched_problem* p, double* x_out, int* basis_out, double* obj_out,
                        int* iters_out, int* status_out);
void lp_batched_free(lp_batched_problem* p);
/* BASELINE.json configs[4] "1 -> 8 GPUs": the LPs of a batch are independent, so participant `shard`
 * of `shards` (one process or host thread per GPU) uploads and solves the LPs [*lo, *hi) of the batch
 * and nobody exchanges anything (replicas of the code, no collective; the caller concatenates the
 * outputs).  The convention every binding uses: *lo = batch*shard/shards, *hi = batch*(shard+1)/shards
 * (64-bit arithmetic) - contiguous, disjoint, covering, sizes differing by at most one.             */
int lp_batched_shard_bounds(int batch, int shard, int shards, int* lo, int* hi);

/* =========================================================================
 * Enumeration — EnumerationSolver (src/EnumerationSolver.h:3-10 is a stub; spec
 * README.md:27,40-42; per-basis kernel = Canonical::GetBasicSolution /
 * IsFeasibleBasis / Evaluate, Canonical.cpp:165-197, :79-87).
 * Semantics: rank k in [0, C(n,m)) = k-th sorted
 * m-subset in lexicographic order; per subset a Gauss-Jordan solve with partial
 * pivoting; singular / infeasible (some xB < -1e-9) / feasible; winner = best
 * objective, ties within 1e-9 broken towards the smallest rank.
 * ========================================================================= */

uint64_t lp_binom(int n, int k); /* C(n,k), 0 on overflow */
/* Shard `shard` of `shards` contiguous rank ranges of EQUAL ESTIMATED COST for the shared-prefix
 * enumeration (the rank space shards across the GPUs of a node): cost(x) = x +
 * 170 * (depth m-7 tree nodes before subset x) — late prefixes have few subsets per node, and
 * equal-size shards differ by 1.5x in run time.  Pure host arithmetic; the solver's answer does
 * not depend on where the cuts are (tie rule).  Small problems get equal-size ranges.         */
int lp_enum_shard_bounds(int n, int m, int shard, int shards, uint64_t* begin_out,
                         uint64_t* end_out);

/* One-shot single-GPU solve.  counts_out[3] = {feasible, infeasible, singular}. */
int lp_enum_solve(lp_context* ctx, const double* A, int m, int n, const double* b,
                  const double* c, int maximize, int n_orig, double* x_out, int* basis_out,
                  uint64_t* rank_out, double* obj_out, uint64_t* counts_out);

typedef struct lp_enum_problem lp_enum_problem;

enum {
    LP_ENUM_ALGO_AUTO = 0,
    LP_ENUM_ALGO_DIRECT = 1, /* one independent m x m solve per subset                   */
    LP_ENUM_ALGO_PREFIX = 2  /* shared-prefix elimination over the combination tree
                                (bit-identical results, far fewer flops): 6 <= m <= 16 with
                                2 <= n-m <= 16 on the tuned kernels; 7 <= m <= 16 with any
                                n <= 64, or 17 <= m <= 32 with n-m <= 32, on the general one;
                                AUTO takes it for ranges of 2^15 subsets or more (2^8 for m > 16)         */
};

typedef struct lp_enum_stats {
    float kernel_ms;    /* HIP-event time of the enumeration kernel(s)                    */
    uint64_t subsets;   /* subsets processed                                              */
    int launches;
} lp_enum_stats;

int lp_enum_upload(lp_context* ctx, const double* A, int m, int n, const double* b,
                   const double* c, int maximize, lp_enum_problem** problem_out);
/* Pass 1 over the shard [rank_begin, rank_end): best objective over feasible
 * subsets (-inf/+inf if none) and the three counts.  This is what each GPU runs
 * on its slice of the rank space; the caller then reduces zbest over GPUs
 * (one RCCL all-reduce).                                                          */
int lp_enum_range(lp_enum_problem* p, uint64_t rank_begin, uint64_t rank_end, int algo,
                  double* zbest_out, uint64_t* counts_out, lp_enum_stats* stats_out);
/* Pass 2: smallest feasible rank in [rank_begin, rank_end) whose objective is
 * within tol of zstar on the better-or-equal side; UINT64_MAX if none.           */
int lp_enum_first_within(lp_enum_problem* p, uint64_t rank_begin, uint64_t rank_end, double zstar,
                         double tol, uint64_t* rank_out);
/* Vertex of one rank: x (n_orig), sorted basis (m), objective, verdict.          */
int lp_enum_vertex(lp_enum_problem* p, uint64_t rank, int n_orig, double* x_out, int* basis_out,
                   double* obj_out, int* verdict_out);
void lp_enum_free(lp_enum_problem* p);
/* 1 if the problem's leaf kernels divide plainly: a pass of the default kernels (fast reciprocal,
 * the same bits for a pivot magnitude within [2^-500, 2^500]) met a pivot outside that range on a
 * subset that was not singular anyway, and was repeated; later passes start there.  0 otherwise.  */
int lp_enum_exact_division(const lp_enum_problem* p);
/* Diagnostic (not part of the drop-in surface): the leaf kernels' fast reciprocal and the plain
 * division 1.0 / x[i], both computed on the device, for the parity test of the two.              */
int lp_debug_reciprocal(lp_context* ctx, const double* x, int n, double* fast_out, double* plain_out);
/* Diagnostic (not part of the drop-in surface): the chip-resident simplex's quotient by the pivot element (the
 * division's own instruction sequence without its range scaling, applied only to operands inside [2^-500, 2^501) or a
 * zero numerator) and the plain division num[i] / den[i], both computed on the device, for the parity test.      */
int lp_debug_division(lp_context* ctx, const double* num, const double* den, int n, double* fast_out, double* plain_out);

/* ---- Enumeration sharded over the GPUs of a node (README.md:27,40-42) ------
 * One participant per GPU — a process, or a host thread of one process — each with its own
 * lp_context and its own lp_enum_problem (the tiny problem is replicated).  Participant `rank`
 * of `world` enumerates shard lp_enum_shard_bounds(n, m, rank, world) with no data-path
 * collective; the only exchange is ONE all-gather of a 48-byte record per participant (best
 * score, smallest rank within 1e-9 of it, the three counts, status) — over RCCL/xGMI for
 * lp_comm_create_rccl communicators.  A second all-gather happens only when two shards hold
 * different vertices within 1e-9 of the optimum.  Results are identical on every participant and
 * for every world size (tie rule).                                                            */
typedef struct lp_comm lp_comm;
/* 128 bytes (ncclUniqueId): one participant calls this and hands the bytes to the others (file,
 * environment, MPI, a TCP store ...).  RCCL is loaded on first use (dlopen).                   */
int lp_comm_unique_id(void* id_out_128_bytes);
/* Collective: every participant calls it with the same id; blocks until all `world` have joined.
 * The communicator is bound to ctx's device and stream.                                        */
int lp_comm_create_rccl(lp_context* ctx, int rank, int world, const void* unique_id, lp_comm** comm_out);
/* `world` communicators for host threads of ONE process, exchanging through host memory;
 * participants may share a device (more shards than GPUs: what the one-GPU tests use).         */
int lp_comm_create_local(int world, lp_comm** comms_out /* world entries */);
int lp_comm_rank(const lp_comm* c);
int lp_comm_world(const lp_comm* c);
void lp_comm_destroy(lp_comm* c);
/* comm == NULL: a single participant (no exchange).  Outputs as lp_enum_solve (x_out and basis_out
 * may both be NULL: the winning rank, objective and counts then come straight from the exchange and
 * no vertex is evaluated); counts_out[3] are the node-wide totals.  Every participant must call it (a failing one still takes part in the
 * exchange, and every participant returns its status).                                          */
int lp_enum_solve_sharded(lp_comm* comm, lp_enum_problem* p, int n_orig, double* x_out, int* basis_out,
                          uint64_t* rank_out, double* obj_out, uint64_t* counts_out);
/* A participant that cannot enumerate (its context, upload or communicator set-up failed) calls this
 * INSTEAD of lp_enum_solve_sharded, so that the others are not left waiting in the exchange: it
 * contributes a failed record and returns `status` (LP_BAD_ARG if LP_OPTIMAL was passed); every other
 * participant's lp_enum_solve_sharded returns that status.  lp_enum_solve_sharded(comm, NULL, ...)
 * does the same with LP_BAD_ARG.                                                                 */
int lp_enum_shard_abstain(lp_comm* comm, int status);

#ifdef __cplusplus
}
#endif
#endif
