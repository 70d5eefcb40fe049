// simplex_problem.hpp — device-side state of one single-LP tableau solve.
#pragma once

#include "lp_internal.hpp"

struct SimplexState {
    int status;       // kRunning (-100) while pivoting, then an LP_* code
    int iters;        // pivots executed (the reference's `iteration`, SimplexSolover.h:427)
    int max_iter;     // MAX_ITER, :426
    int enter;        // entering column of the pending pivot
    int leave;        // leaving basis POSITION (= tableau row) of the pending pivot
    int pivot_valid;  // 1 if select staged a pivot for the update kernel
    int pad0, pad1;
    double eps;       // Solver::EPS, :13
    double minpiv, maxpiv;  // crash: smallest / largest |pivot|
};

// Everything the kernels need, passed by value.
struct SimplexDev {
    int m, n, ld;     // ld = padded row length of T (multiple of 8 doubles)
    int maximize;
    int trace_cap;
    double* T;        // (m+1) x ld row-major tableau
    double* lcol;     // m+1: eta column of F (:198-204); entry r holds 1/u_r
    double* prow;     // ld: copy of the pivot row before the update
    int* basis;       // m: N, basis by position (:419)
    unsigned char* nonbasic;  // n: complement(n, N) as flags (:97-108)
    unsigned char* rowused;   // m: crash bookkeeping
    int* rowpos;      // m: crash — tableau row chosen for basis position t
    int* trace_enter; // trace_cap
    int* trace_leave;
    SimplexState* state;
};

// Look-ahead batch buffers (simplex_lookahead.hip).
struct LookDev {
    int J;          // pivots staged per batch
    int rows_pad;   // row stride of etaL (m+1 rounded up to 8)
    double* etaL;   // J x rows_pad: eta columns of F (:198-204); entry r = 1/u_r, entry m = cost row
    double* etaP;   // J x ld: pivot rows before scaling (entry n = xB_r)
    double* dvec;   // ld: current reduced-cost row (entry n = -objective)
    double* rhs;    // rows_pad: current xB
    int* piv;       // 2 x J: (entering column, leaving position) of each staged pivot
    int* count;     // pivots staged by the last selector launch
    unsigned long long* stamps;  // diagnostic: 8 s_memtime stamps per pivot (nullptr = off)
};

// Chip-resident solve (simplex_resident.hip): the tableau lives in the registers of G co-resident
// workgroups (cpt columns each) for the whole solve; per pivot they exchange one 16-byte pricing record, 8-15
// slice records of the ratio test and one candidate column each through these buffers (8-byte {epoch tag, value}
// granules).
struct ResidentDev {
    int G;          // participating workgroups = ceil(n / columns per workgroup)
    int stride;     // participants are the blocks b with b % stride == 0 (8: one XCD under round-robin dispatch)
    int mpad;       // row threads per workgroup = m rounded up to 64 (one tableau row per thread)
    int cpt;        // tableau columns per workgroup: 32 (m <= 512) or 16 (m <= 960)
    int flags;      // bit 0: write-through stores even when all participants share an XCD; bit 1: injected failure (tests)
    char* comm;     // one allocation, zeroed before every launch; carved below (byte offsets)
    unsigned prec_off;           // [2][G] pricing records (16 bytes)
    unsigned srec_off;           // [2][G][2][16] slice records of the ratio test (A granules, B granules)
    unsigned col_off;            // [3][G][mpad] candidate columns
    unsigned dpub_off, recS_off, colS_off, census_off, abort_off, comm_bytes;
};

struct lp_simplex_problem {
    lp_context* ctx = nullptr;
    SimplexDev dev{};
    LookDev look{};
    ResidentDev res{};            // res.G == 0: shape outside the chip-resident path
    int n_orig = 0;
    size_t tableau_bytes = 0;
    void* arena = nullptr;        // ONE device allocation (from the context's pool) behind every pointer below
    size_t arena_bytes = 0;
    double* dT0 = nullptr;        // pristine initial tableau (after crash) for lp_simplex_reset; also the
                                  // staging area of the upload (column-major A) and of the crash's row permutation
    double* dscratchT = nullptr;  // scratch copy used by the update micro-benchmarks (allocated on first use)
    double* ov_T = nullptr;       // simplex_overlap.hip: the second tableau buffer, and the second eta slot
    double* ov_vec = nullptr;     // (pivot row, eta column, 8 ints); both allocated on first use
    bool ov_attr = false;         // the overlapped kernel's dynamic-LDS opt-in has been made
    int* dbasis0 = nullptr;
    unsigned char* dnonbasic0 = nullptr;
    double* dx = nullptr;         // n: extracted vertex
    SimplexState* h_state = nullptr;  // pinned
    std::vector<double> h_c;      // objective coefficients (Canonical::Evaluate on the host)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t res_ev0 = nullptr, res_ev1 = nullptr;   // around the chip-resident kernel launch
    std::vector<hipEvent_t> upd_events;  // 2 per timed rank-J update launch (look-ahead path)
    bool profile_updates = false;        // lp_simplex_profile: bracket update launches with events
    int init_status = LP_OPTIMAL; // LP_SINGULAR if the initial basis was singular
    int last_status = -100;
    int last_iters = 0;
    int last_algo = 0;            // LP_SIMPLEX_ALGO_* of the last run (which stamp buffer is current)
};

// simplex_launch.hip
void lp_simplex_launch_update(lp_simplex_problem* p);
int lp_simplex_crash(lp_simplex_problem* p);
int lp_simplex_price_out_identity(lp_simplex_problem* p);  // unit-vector basis with non-zero costs
int lp_simplex_force(lp_simplex_problem* p, int row, int col);  // host-chosen pivot on the current tableau
int lp_simplex_driveout(lp_simplex_problem* p, const int* positions, int count, int n_limit, double eps, int* applied);
int lp_simplex_phase2_costs(lp_simplex_problem* p, const double* cost, int n_real, int maximize, int n_orig);
int lp_simplex_run_launch(lp_simplex_problem* p, double eps, int max_iter, lp_simplex_stats* stats);
int lp_simplex_extract_x(lp_simplex_problem* p, double* dx);
int lp_simplex_bench_update(lp_simplex_problem* p, int row, int col, int iters, float* ms_out);

// simplex_overlap.hip
bool lp_overlap_fits(int m);
bool lp_overlap_auto(int m);    // what LP_SIMPLEX_ALGO_AUTO requires of the shape
int lp_overlap_prepare(lp_simplex_problem* p);   // allocates the second tableau buffer (first use)
int lp_simplex_run_overlap(lp_simplex_problem* p, double eps, int max_iter, lp_simplex_stats* stats);

// simplex_resident.hip
int lp_resident_plan(int m, int n, ResidentDev* out);   // fills G/stride/mpad/offsets; 0 if the shape does not fit
int lp_simplex_run_resident(lp_simplex_problem* p, double eps, int max_iter, lp_simplex_stats* stats);
int lp_simplex_debug_division(lp_context* ctx, const double* num, const double* den, int n, double* fast_out, double* plain_out);

// simplex_lookahead.hip
int lp_lookahead_pick_j(int m, int n);
int lp_lookahead_prepare(lp_simplex_problem* p);
int lp_lookahead_init_vectors(lp_simplex_problem* p);
int lp_simplex_run_lookahead(lp_simplex_problem* p, double eps, int max_iter, lp_simplex_stats* stats);
int lp_lookahead_bench_update(lp_simplex_problem* p, int iters, float* ms_per_launch, int* pivots_out);
