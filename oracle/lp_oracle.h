/*
 * lp_oracle.h — CPU restatement (plain C) of the dense-LP hot path of
 * haskell-md2/SimplexMethod.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link, load or call anything in oracle/.  The product path
 * (simplexmethod_amd/csrc, simplexmethod_amd/host) never does.
 *
 * PARITY STATUS.  The reference cannot be built here: every translation unit
 * includes <Eigen/Dense> (src/ProblemTypes/IProblem.h:3) and Eigen 3.4.0 is
 * fetched from the network by CMakeLists.txt:12-17, so there is no oracle/_ref.
 *   - per-basis solve + feasibility (Canonical.cpp:165-197): PINNED by the
 *     reference's own fixture tests/test_canonical.cpp:41-66.
 *   - Symmetrical::ToCanonical max branch: PINNED by tests/test_symmetrical.cpp:55-72.
 *   - Solver::solve (SimplexSolover.h:288-451): the reference holds no test, no
 *     golden vector and no recorded output for it => "parity unpinned"; the
 *     restatement follows the source line by line and is cross-checked against
 *     scipy.optimize.linprog (objective only) when tests/golden is generated.
 *   - EnumerationSolver: no reference implementation exists
 *     (src/EnumerationSolver.h:3-10 is an empty stub; spec = README.md:27,40-42)
 *     => semantics are build-defined and "parity unpinned".
 *
 * All matrices cross this API column-major (Eigen's default storage order, so
 * A.col(j) is contiguous exactly as in the reference).
 */
#ifndef LP_ORACLE_H
#define LP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes, shared with include/simplexmethod_amd.h. */
enum {
    ORC_OPTIMAL = 0,     /* solve() returned (SimplexSolover.h:432-440)                     */
    ORC_UNBOUNDED = 1,   /* runtime_error "Целевая функция неограничена"  (:442-444)        */
    ORC_ITER_LIMIT = 2,  /* runtime_error "Достигнут лимит итераций"      (:450)            */
    ORC_SINGULAR = 3,    /* runtime_error "Singular basis matrix"         (:125-126)        */
    ORC_INFEASIBLE = 4,  /* enumeration: no feasible basis                                   */
    ORC_BAD_ARG = 5      /* invalid_argument from Canonical's ctor (Canonical.cpp:27-46),
                            runtime_error "basis index out of range" (SimplexSolover.h:101) */
};

/* ---- Simplex, reference-shaped (SimplexSolover.h:408-451) -------------------
 * Revised simplex with an explicit dense Binv that is recomputed from scratch by
 * a full-pivot LU after every pivot (computeBFS, :117-133, called at :423,:433,:446).
 * Pivot rules: simplexIter :135-209.  If dense_eta_product != 0 the discarded
 * Binv = F*Binv product (:198-206) is also executed as a dense m x m GEMM so
 * that the cost shape equals the reference's (this is the timed CPU baseline).
 * trace_enter/trace_leave (optional, capacity trace_cap) receive the pivot
 * sequence: entering column, leaving basis POSITION (:196).
 */
int orc_simplex_reference(const double* A, int m, int n, const double* b, const double* c,
                          const int* basis_in, int maximize, int n_orig, double eps, int max_iter,
                          int dense_eta_product,
                          double* x_out /* n_orig */, int* basis_out /* m, by position */,
                          double* obj_out, int* iters_out,
                          int* trace_enter, int* trace_leave, int trace_cap);

/* ---- Simplex, tableau form (what the GPU executes) ---------------------------
 * Same pivot rules (:152-196) on the full tableau T = Binv*[A | b] plus a reduced-
 * cost row, updated by the rank-1 Gauss-Jordan step that F*Binv (:198-206) is:
 * rows i != r: T_i += (-u_i/u_r) * T_r ; row r: T_r *= (1/u_r).  Every element is
 * updated by one fma with fixed operands, so a GPU that does the same is
 * bit-identical to this function.  tableau_out (optional) receives the final
 * (m+1) x (n+1) tableau row-major: rows 0..m-1 = constraint rows by basis
 * position, row m = reduced costs d_j = c_j - z_j (entry [m][n] = -objective);
 * column n = xB.
 */
int orc_simplex_tableau(const double* A, int m, int n, const double* b, const double* c,
                        const int* basis_in, int maximize, int n_orig, double eps, int max_iter,
                        double* x_out, int* basis_out, double* obj_out, int* iters_out,
                        int* trace_enter, int* trace_leave, int trace_cap,
                        double* tableau_out);

/* ---- Two-phase simplex (build-defined, parity unpinned) -----------
 * For canonical problems without a usable starting basis (Symmetrical min problems, negative
 * b).  Flow of the reference's unreachable sketch (SimplexSolover.h:61-95, :331-406), made
 * consistent: rows with b < -eps change sign; phase I minimises the sum of m artificials
 * [A' | I] from their identity basis with the tableau simplex above; infeasible iff that sum
 * > eps; an artificial still basic is pivoted out on the first non-basic original column with
 * |T[pos][cand]| > eps (none: ORC_SINGULAR, dependent rows); phase II = orc_simplex_tableau on
 * (A', b', c) from the clean basis.  iters_out[3] = pivots of phase I, drive-out, phase II.
 */
int orc_two_phase(const double* A, int m, int n, const double* b, const double* c, int maximize,
                  int n_orig, double eps, int max_iter, double* x_out, int* basis_out,
                  double* obj_out, int* iters_out);

/* One sequential "chain" selection as written at SimplexSolover.h:153-161
 * (maximize: take j if d > best + eps) / :164-172 (minimize) / :181-192 (ratio
 * test = minimize flavour).  mask[j] != 0 marks eligible entries.  Returns the
 * selected index or -1; *best_out receives the chain's final value.
 */
int orc_chain_select(const double* v, const unsigned char* mask, int len, int want_max, double eps,
                     double* best_out);

/* ---- Per-basis solve (Canonical.cpp:165-197, :79-87) ------------------------ */
/* GetBasicSolution: B = A[:,basis]; column-pivoted Householder QR solve of
 * B*xB = b; scatter into a length-n vector (zeros elsewhere).                      */
int orc_basic_solution(const double* A, int m, int n, const double* b, const int* basis,
                       double* x_out /* n */);
/* IsFeasibleBasis: all entries of the above >= -1e-9.                              */
int orc_is_feasible_basis(const double* A, int m, int n, const double* b, const int* basis);
/* Evaluate: c . x                                                                   */
double orc_evaluate(const double* c, const double* x, int n);

/* ---- Enumeration (build-defined) ----------------------------
 * Rank k in [0, C(n,m)) indexes the sorted m-subsets of {0..n-1} in lexicographic
 * order.  Per subset: Gauss-Jordan on [A[:,S] | b] with partial (row) pivoting,
 * columns in ascending order, the last two columns solved as a 2x2 block with
 * back-substitution (see lp_oracle.c for the exact operation order);
 * singular iff a pivot is exactly 0 or min|pivot| <= DBL_EPSILON*m*max|pivot|
 * (Eigen FullPivLU::isInvertible's default threshold, the test the reference
 * applies at SimplexSolover.h:124-126); feasible iff all xB >= -1e-9
 * (Canonical.cpp:169-175); objective z = sum_j c_j x_j by sequential fma in
 * ascending j (Canonical.cpp:86).
 */
enum { ORC_SUBSET_FEASIBLE = 0, ORC_SUBSET_INFEASIBLE = 1, ORC_SUBSET_SINGULAR = 2 };

uint64_t orc_binom(int n, int k);                       /* C(n,k); 0 if it overflows u64 */
void orc_unrank(int n, int m, uint64_t rank, int* subset /* m */);
uint64_t orc_rank(int n, int m, const int* subset);
int orc_next_subset(int n, int m, int* subset);        /* lexicographic successor; 0 at end */

/* Solve one subset.  xB_out[t] is the value of variable subset[t].                 */
int orc_enum_subset(const double* A, int m, int n, const double* b, const double* c,
                    const int* subset, double* xB_out /* m */, double* z_out);

/* Pass 1 over ranks [begin,end): best objective over feasible subsets (max if
 * maximize else min), counts[3] = {feasible, infeasible, singular}.
 * Returns ORC_OPTIMAL or ORC_INFEASIBLE (no feasible subset in range).             */
int orc_enum_range(const double* A, int m, int n, const double* b, const double* c, int maximize,
                   uint64_t begin, uint64_t end, double* zbest_out, uint64_t counts[3]);
/* Pass 2: smallest rank in [begin,end) that is feasible with |z - zstar| <= tol on
 * the better-or-equal side, i.e. z >= zstar - tol (max) / z <= zstar + tol (min).
 * UINT64_MAX if none.                                                               */
uint64_t orc_enum_first_within(const double* A, int m, int n, const double* b, const double* c,
                               int maximize, uint64_t begin, uint64_t end, double zstar, double tol);
/* Whole solver: pass 1, pass 2 with tol = 1e-9, then the winner's vertex.          */
int orc_enum_solve(const double* A, int m, int n, const double* b, const double* c, int maximize,
                   int n_orig, double* x_out /* n_orig */, int* basis_out /* m sorted */,
                   uint64_t* rank_out, double* obj_out, uint64_t counts[3]);

/* ---- Synthetic dense LPs ------------------------------------
 * Canonical [A_orig | I], m rows, n = total columns: A_orig ~ U(0,1), b ~
 * U(1,2)*(n-m)/2, c ~ U(0,1) (0 on slacks), basis = slack columns, maximise.
 * Generator: splitmix64 counter stream keyed by seed (identical in
 * simplexmethod_amd/capi.py: gen_lp).                                              */
void orc_gen_lp(uint64_t seed, int m, int n, double* A /* m*n col-major */, double* b, double* c,
                int* basis);

#ifdef __cplusplus
}
#endif
#endif
