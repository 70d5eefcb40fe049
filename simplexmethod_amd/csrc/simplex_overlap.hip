// simplex_overlap.hip — single LP beyond the chip-resident shapes: ONE launch per pivot, in which the
// rank-1 update of pivot k streams the tableau (HBM-bound, every CU) while one more workgroup of the same
// launch already selects pivot k+1.
//
// The two-launch form (simplex_launch.hip) is a chain  select -> update -> select -> ...: its update runs
// at 5.9-6.9 TB/s on tableaus of 64-270 MB, but the one-workgroup selector between two updates (pricing
// over n, a strided column gather, ratio test, pivot-row copy: 10-25 us) and the second kernel boundary
// add 30-65 % to every pivot.  Nothing in the selection of pivot k+1 needs the UPDATED tableau as a whole:
// it needs row m, column n, one column and one row of it, and each such entry is one fma away from the old
// tableau — T_k[i][j] = fma(l_k[i], prow_k[j], T_{k-1}[i][j]) (pivot row: prow_k[j] * l_k[r]; entering
// column: the unit vector), the very operations the update performs, so the bits are the same.
// So the tableau is updated OUT OF PLACE between two buffers (T_{k-1} -> T_k), and launch k consists of
//   workgroup 0        : selects pivot k+1 from T_{k-1} and pivot k's eta (l_k, prow_k, r_k, e_k): pricing
//                        (SimplexSolover.h:152-174), unbounded test (:179), ratio test (:181-194), basis
//                        bookkeeping (:196), eta column of F (:198-204) and the pivot row -> slot (k+1)&1;
//   workgroups 1 ...   : T_k = update of T_{k-1} with pivot k's eta (slot k&1), 128 columns x 64 rows each.
// Per pivot: max(update, select) + one kernel boundary instead of update + select + two boundaries.
// Same pivots, same tableau bits as the other algorithms (tests/test_gpu_simplex.py).
#include "device_select.hpp"
#include "simplex_problem.hpp"

namespace {

constexpr int kRunning = -100;  // SimplexState::status while pivoting

constexpr int OV_TX = 64;    // column pairs per workgroup (128 columns: 1 KiB per row segment)
constexpr int OV_TY = 16;    // row groups per workgroup
constexpr int OV_RPT = 4;    // rows per thread
constexpr int OV_LIN = 8;    // column pairs per thread in flight (linear update)

struct OverlapDev {
    double* T[2];      // T[0] = SimplexDev::T; T_k lives in T[k & 1]
    double* lcol[2];   // eta column of pivot k in lcol[k & 1] (entry r = 1/u_r, entry m = the cost row's multiplier)
    double* prow[2];   // pivot row of pivot k before the update
    int* slot;         // [2][4]: entering column, leaving position, valid, -
};

__global__ void k_overlap_init(SimplexDev d, OverlapDev ov, double eps, int max_iter) {
    SimplexState* st = d.state;
    st->status = kRunning;
    st->iters = 0;
    st->max_iter = max_iter;
    st->enter = st->leave = -1;
    st->pivot_valid = 0;
    st->eps = eps;
    for (int k = 0; k < 8; ++k) ov.slot[k] = 0;
}

// workgroup 0 of launch k: pivot k+1 from the tableau T_{k-1} (T_0 for k = 0) seen through pivot k's eta
__device__ __forceinline__ void overlap_select(const SimplexDev& d, const OverlapDev& ov, int k, double* s_dyn) {
    SimplexState* st = d.state;
    const int m = d.m, n = d.n, ld = d.ld;
    double* s_u = s_dyn;
    double* s_ratio = s_dyn + (m + 2);
    int* s_int = reinterpret_cast<int*>(s_dyn + 2 * (m + 2));
    int& s_enter = s_int[0];
    int& s_leave = s_int[1];
    int& s_flag = s_int[2];
    lpdev::BlockChainScratch* s_sc = reinterpret_cast<lpdev::BlockChainScratch*>(s_dyn + 2 * (m + 2) + 2);
    const int tid = threadIdx.x;
    const int cur = k & 1, nxt = cur ^ 1;
    int* out = ov.slot + 4 * nxt;
    if (st->status != kRunning) {
        if (tid == 0) out[2] = 0;
        return;
    }
    const double eps = st->eps;
    if (st->iters >= st->max_iter) {  // while (iteration < MAX_ITER) ... throw, :429,:450
        if (tid == 0) {
            st->status = LP_ITER_LIMIT;
            st->pivot_valid = 0;
            out[2] = 0;
        }
        return;
    }
    const double* Ts = ov.T[k == 0 ? 0 : nxt];   // T_{k-1}
    const int* in = ov.slot + 4 * cur;
    const bool prev = k > 0 && in[2] != 0;       // pivot k is pending on T_{k-1} (applied by this launch's other workgroups)
    const int ep = in[0], rp = in[1];
    const double* lp = ov.lcol[cur];
    const double* pp = Ts + (size_t)(prev ? rp : 0) * ld;   // pivot k's row: row r_k of T_{k-1} itself
    // entry (i, j) of T_k: what the update writes there (simplex_launch.hip: k_simplex_update, operand for operand)
    auto val = [&](int i, int j) {
        double v = Ts[(size_t)i * ld + j];
        if (prev) {
            const double l = lp[i], pr = pp[j];
            v = (i == rp) ? pr * l : fma(l, pr, v);
            if (j == ep) v = (i == rp) ? 1.0 : 0.0;
        }
        return v;
    };
    // pricing: the updated cost row, staged in the (still unused) next pivot-row slot, every wave scanning
    double* stage = ov.prow[nxt];
    {
        double best;
        int e;
        auto price = [&](int j) { return d.nonbasic[j] != 0 ? val(m, j) : (d.maximize ? -INFINITY : INFINITY); };   // complement(), :97-108
        if (d.maximize)
            e = lpdev::block_chain_select<true, true>(n, eps, best, price, stage, s_sc);
        else
            e = lpdev::block_chain_select<false, true>(n, eps, best, price, stage, s_sc);
        const bool optimal = d.maximize ? (best <= eps) : (best >= -eps);  // :162 / :173
        if (tid == 0) {
            s_enter = optimal ? -1 : e;
            s_flag = 0;
        }
    }
    __syncthreads();
    const int e = s_enter;
    if (e < 0) {
        if (tid == 0) {
            st->status = LP_OPTIMAL;
            st->pivot_valid = 0;
            out[2] = 0;
        }
        return;
    }
    int any_pos = 0;
    for (int i = tid; i <= m; i += blockDim.x) {   // u = Binv * A.col(enter), :176, incl. the reduced-cost row
        const double ui = val(i, e);
        s_u[i] = ui;
        if (i < m) {
            s_ratio[i] = (ui > eps) ? val(i, n) / ui : INFINITY;  // :185-186
            if (!(ui <= eps)) any_pos = 1;  // (u.array() <= EPS).all(), :179
        }
    }
    if (any_pos) s_flag = 1;
    __syncthreads();
    if (!s_flag) {
        if (tid == 0) {
            st->status = LP_UNBOUNDED;
            st->pivot_valid = 0;
            out[2] = 0;
        }
        return;
    }
    {
        double theta;
        auto ratio = [&](int i) { return s_ratio[i]; };   // ineligible rows hold +inf, which the < scan never takes
        const int r = lpdev::block_chain_select<false, false>(m, eps, theta, ratio, s_ratio, s_sc);  // :187-190
        if (tid == 0) s_leave = r;
    }
    __syncthreads();
    const int r = s_leave;
    if (r < 0) {  // :194
        if (tid == 0) {
            st->status = LP_UNBOUNDED;
            st->pivot_valid = 0;
            out[2] = 0;
        }
        return;
    }
    const double ur = s_u[r];
    double* lout = ov.lcol[nxt];
    for (int i = tid; i <= m; i += blockDim.x)  // F(i,r) = -u_i/u_r, F(r,r) = 1/u_r, :198-204
        lout[i] = (i == r) ? 1.0 / ur : -s_u[i] / ur;
    // (the pivot row of pivot k+1 is row r of T_k: the next launch reads it there)
    if (tid == 0) {
        const int old = d.basis[r];
        d.basis[r] = e;  // N(leave_pos) = enter, :196
        d.nonbasic[e] = 0;
        d.nonbasic[old] = 1;
        const int it = st->iters;
        if (it < d.trace_cap) {
            d.trace_enter[it] = e;
            d.trace_leave[it] = r;
        }
        st->iters = it + 1;
        st->enter = e;
        st->leave = r;
        st->pivot_valid = 1;
        out[0] = e;
        out[1] = r;
        out[2] = 1;
    }
}

// The same selection with TWO dependent global round trips instead of six (round 4; the selector shares the chip
// with an update that saturates the memory system, so every dependent access costs microseconds: 2048 x 4096 ran
// at 27.8 us per pivot for a 22 us update).  For m + 1 <= OVF_ROWS * 1024 rows and an LDS budget that still lets two
// update workgroups share a CU:
//   trip 1  everything that does not depend on the entering column, all in flight together: the cost row, the
//           non-basic flags and pivot k's row (pricing), pivot k's eta column and the xB column (ratio test) —
//           the priced values go to LDS (they were staged in global memory and read back), xB stays in registers;
//   trip 2  the entering column (strided gather) -> u, ratios, the ratio chain in LDS; the eta column comes out of the
//           registers that hold u.
// The pivot row is not copied any more: pivot k's row is row r_k of T_{k-1}, where both this selection and the update
// read it.
constexpr int OVF_ROWS = 4;    // rows per thread held in registers
constexpr int OVF_COLS = 4;    // columns per thread and chunk of 4096

__device__ __forceinline__ void overlap_select_fast(const SimplexDev& d, const OverlapDev& ov, int k, double* s_dyn) {
    SimplexState* st = d.state;
    const int m = d.m, n = d.n, ld = d.ld;
    double* s_u = s_dyn;
    double* s_ratio = s_dyn + (m + 2);
    int* s_int = reinterpret_cast<int*>(s_dyn + 2 * (m + 2));
    int& s_enter = s_int[0];
    int& s_leave = s_int[1];
    int& s_flag = s_int[2];
    lpdev::BlockChainScratch* s_sc = reinterpret_cast<lpdev::BlockChainScratch*>(s_dyn + 2 * (m + 2) + 2);
    double* s_price = reinterpret_cast<double*>(s_sc + 1);   // n doubles
    const int tid = threadIdx.x, T = (int)blockDim.x;
    const int cur = k & 1, nxt = cur ^ 1;
    int* out = ov.slot + 4 * nxt;
    const double* Ts = ov.T[k == 0 ? 0 : nxt];   // T_{k-1}
    const int* in = ov.slot + 4 * cur;
    const double* lp = ov.lcol[cur];
    // ---- trip 1: issued before anything is looked at (reads only; an early exit simply drops them)
    const int status = st->status, iters = st->iters, max_iter = st->max_iter;
    const double eps = st->eps;
    const int in_valid = in[2], ep = in[0], rp = in[1];
    const double* pp = Ts + (size_t)((k > 0 && in_valid != 0) ? rp : 0) * ld;   // pivot k's row: row r_k of T_{k-1} itself
    const double lpm = lp[m], ppn = pp[n];
    double xb_raw[OVF_ROWS], lrow[OVF_ROWS];
#pragma unroll
    for (int q = 0; q < OVF_ROWS; ++q) {
        const int i = tid + q * T;
        xb_raw[q] = (i <= m) ? Ts[(size_t)i * ld + n] : 0.0;
        lrow[q] = (i <= m) ? lp[i] : 0.0;
    }
    if (status != kRunning) {
        if (tid == 0) out[2] = 0;
        return;
    }
    if (iters >= max_iter) {  // while (iteration < MAX_ITER) ... throw, :429,:450
        if (tid == 0) {
            st->status = LP_ITER_LIMIT;
            st->pivot_valid = 0;
            out[2] = 0;
        }
        return;
    }
    const bool prev = k > 0 && in_valid != 0;    // pivot k is pending on T_{k-1} (applied by this launch's other workgroups)
    // entry (i, j) of T_k from its three operands: what the update writes there (k_simplex_update, operand for operand)
    auto upd = [&](double v, double l, double pr, int i, int j) {
        if (prev) {
            v = (i == rp) ? pr * l : fma(l, pr, v);
            if (j == ep) v = (i == rp) ? 1.0 : 0.0;
        }
        return v;
    };
    // pricing: the updated cost row -> LDS
    const double ineligible = d.maximize ? -INFINITY : INFINITY;   // complement(), :97-108
    for (int base = 0; base < n; base += OVF_COLS * T) {
        double cr[OVF_COLS], pr[OVF_COLS];
        unsigned char nb[OVF_COLS];
#pragma unroll
        for (int q = 0; q < OVF_COLS; ++q) {
            const int j = base + tid + q * T;
            const int jc = j < n ? j : n - 1;
            cr[q] = Ts[(size_t)m * ld + jc];
            pr[q] = pp[jc];
            nb[q] = d.nonbasic[jc];
        }
#pragma unroll
        for (int q = 0; q < OVF_COLS; ++q) {
            const int j = base + tid + q * T;
            if (j < n) {
                double v = nb[q] != 0 ? upd(cr[q], lpm, pr[q], m, j) : ineligible;
                s_price[j] = (v == v) ? v : ineligible;
            }
        }
    }
    // xB of T_k (column n is never the entering column)
    double xbv[OVF_ROWS];
#pragma unroll
    for (int q = 0; q < OVF_ROWS; ++q) xbv[q] = upd(xb_raw[q], lrow[q], ppn, tid + q * T, n);
    __syncthreads();
    {
        double best;
        int e;
        auto price = [&](int j) { return s_price[j]; };
        if (d.maximize)
            e = lpdev::block_chain_select<true, false>(n, eps, best, price, s_price, s_sc);
        else
            e = lpdev::block_chain_select<false, false>(n, eps, best, price, s_price, s_sc);
        const bool optimal = d.maximize ? (best <= eps) : (best >= -eps);  // :162 / :173
        if (tid == 0) {
            s_enter = optimal ? -1 : e;
            s_flag = 0;
        }
    }
    __syncthreads();
    const int e = s_enter;
    if (e < 0) {
        if (tid == 0) {
            st->status = LP_OPTIMAL;
            st->pivot_valid = 0;
            out[2] = 0;
        }
        return;
    }
    // ---- trip 2: u = Binv * A.col(enter), :176, incl. the reduced-cost row
    const double ppe = pp[e];
    double uu[OVF_ROWS];
#pragma unroll
    for (int q = 0; q < OVF_ROWS; ++q) {
        const int i = tid + q * T;
        uu[q] = (i <= m) ? Ts[(size_t)i * ld + e] : 0.0;
    }
    int any_pos = 0;
#pragma unroll
    for (int q = 0; q < OVF_ROWS; ++q) {
        const int i = tid + q * T;
        if (i <= m) {
            const double ui = upd(uu[q], lrow[q], ppe, i, e);
            uu[q] = ui;
            s_u[i] = ui;
            if (i < m) {
                s_ratio[i] = (ui > eps) ? xbv[q] / ui : INFINITY;  // :185-186
                if (!(ui <= eps)) any_pos = 1;  // (u.array() <= EPS).all(), :179
            }
        }
    }
    if (any_pos) s_flag = 1;
    __syncthreads();
    if (!s_flag) {
        if (tid == 0) {
            st->status = LP_UNBOUNDED;
            st->pivot_valid = 0;
            out[2] = 0;
        }
        return;
    }
    {
        double theta;
        auto ratio = [&](int i) { return s_ratio[i]; };   // ineligible rows hold +inf, which the < scan never takes
        const int r = lpdev::block_chain_select<false, false>(m, eps, theta, ratio, s_ratio, s_sc);  // :187-190
        if (tid == 0) s_leave = r;
    }
    __syncthreads();
    const int r = s_leave;
    if (r < 0) {  // :194
        if (tid == 0) {
            st->status = LP_UNBOUNDED;
            st->pivot_valid = 0;
            out[2] = 0;
        }
        return;
    }
    // (no third trip: the pivot row of pivot k+1 is row r of T_k, which the next launch reads where this launch's
    // update workgroups are writing it)
    const double ur = s_u[r];
    double* lout = ov.lcol[nxt];
#pragma unroll
    for (int q = 0; q < OVF_ROWS; ++q) {  // F(i,r) = -u_i/u_r, F(r,r) = 1/u_r, :198-204
        const int i = tid + q * T;
        if (i <= m) lout[i] = (i == r) ? 1.0 / ur : -uu[q] / ur;
    }
    if (tid == 0) {
        const int old = d.basis[r];
        d.basis[r] = e;  // N(leave_pos) = enter, :196
        d.nonbasic[e] = 0;
        d.nonbasic[old] = 1;
        if (iters < d.trace_cap) {
            d.trace_enter[iters] = e;
            d.trace_leave[iters] = r;
        }
        st->iters = iters + 1;
        st->enter = e;
        st->leave = r;
        st->pivot_valid = 1;
        out[0] = e;
        out[1] = r;
        out[2] = 1;
    }
}

// Launch k.  nbx = column tiles of the update (its workgroups are 1 ... nbx * nby).
template <bool LINEAR>
__global__ __launch_bounds__(OV_TX* OV_TY) void k_simplex_overlap(SimplexDev d, OverlapDev ov, int k, int nbx, int fast) {
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    if (blockIdx.x == 0) {
        __builtin_amdgcn_s_setprio(3);
#ifdef LP_OVERLAP_UPDATE_ONLY   // diagnostic builds: after the first real selection every launch repeats pivot 1's eta
        if (k >= 2) {           // (numerically meaningless: the out-of-place update's own time per launch)
            if (threadIdx.x == 0) {
                SimplexState* st = d.state;
                int* o = ov.slot + 4 * ((k & 1) ^ 1);
                const int* in = ov.slot + 4 * (k & 1);
                if (st->status != kRunning) { o[2] = 0; return; }
                if (st->iters >= st->max_iter) { st->status = LP_ITER_LIMIT; o[2] = 0; return; }
                st->iters = st->iters + 1;
                o[0] = in[0]; o[1] = in[1]; o[2] = in[2];
            }
            return;
        }
#endif
        if (fast)
            overlap_select_fast(d, ov, k, s_dyn);
        else
            overlap_select(d, ov, k, s_dyn);
        return;
    }
    if (k == 0) return;
#ifdef LP_OVERLAP_SELECT_ONLY   // diagnostic builds (scripts/ab_overlap_variants): the selection's own time per launch
    return;
#endif
    const int cur = k & 1;
    const int* sl = ov.slot + 4 * cur;
    if (!sl[2]) return;
    const int e = sl[0], r = sl[1];
    const int ld2 = d.ld >> 1;
    const int rows = d.m + 1;
    const double2* S2 = reinterpret_cast<const double2*>(ov.T[cur ^ 1]);
    double2* D2 = reinterpret_cast<double2*>(ov.T[cur]);
    const double* lcol = ov.lcol[cur];
    const int je = e >> 1;
    if constexpr (LINEAR) {
        // Few tiles for the chip (fewer than six per CU: at 2048 x 4096 the 1089 tiles of one workgroup per CU — the
        // kernel's registers allow no second one — are 4.25 rounds, 26.6 us per launch against the 22.7 of the
        // stand-alone kernel's 4257 small workgroups): the update workgroups are PERSISTENT, one per CU (the launch has
        // 256 workgroups, the selector included), and share the tableau evenly and linearly — workgroup w streams the
        // contiguous piece [G w / W, G (w+1) / W) of the row-major tableau (G 16-byte column pairs) through a ring of
        // eight pairs per thread; the pivot-row pair and the eta entry of each come from L1/L2 (row r and the eta column
        // are 50 KB, hot): 23.2 us.  On larger tableaus the tiles win (2048 x 8192: 48 us in tiles, 55 linear; 3072 x
        // 6144 57 against 69): the other instantiation of this kernel.
        const int W = (int)gridDim.x - 1, w = (int)blockIdx.x - 1;
        const int T = OV_TX * OV_TY;
        const double2* P2 = S2 + (size_t)r * ld2;   // the pivot row: row r of T_{k-1}
        const long long G = (long long)rows * ld2;
        const long long g0 = G * w / W, g1 = G * (w + 1) / W;
        // A thread's elements are T apart; their row and column pair are carried (one division per thread: eight
        // 64-bit divisions per round had been ~1 k VALU instructions per thread in a loop that otherwise has ~100).
        // The OV_LIN slots are a RING: as soon as a slot's element is stored, the element OV_LIN further on is
        // requested into it, and a slot is waited for with a COUNTED s_waitcnt — vmcnt returns in issue order, so
        // "at most 28 younger operations outstanding" (seven slots x (one store + three loads)) says exactly that
        // this slot's three loads have landed.  The compiler cannot count across the loop's branches (it waits for
        // vmcnt(0) at the loop head: whole rounds of OV_LIN, and a share of 16.08 elements per thread — 2049 x 2049
        // pairs on 255 workgroups — then costs THREE round trips, the third for two of the sixteen waves with the CU
        // idle behind them), so loads, stores and waits are inline asm and every one of them is issued by every wave
        // in every step: loads with clamped indices, stores under an exec mask rather than a branch.
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        const dbl2* Sv = reinterpret_cast<const dbl2*>(S2);
        const dbl2* Pv = reinterpret_cast<const dbl2*>(P2);
        dbl2* Dv = reinterpret_cast<dbl2*>(D2);
        const int dT = T / ld2, rT = T - dT * ld2;
        auto advance = [&](int& i, int& j) {
            i += dT;
            j += rT;
            if (j >= ld2) {
                j -= ld2;
                ++i;
            }
        };
        // (element indices fit 32 bits: the host takes this form only below 6 x 256 tiles of 8192 pairs)
        const int gend = (int)g1, gbeg = (int)g0;
        int g_i = gbeg + (int)threadIdx.x;   // next element to request
        int i_i = g_i / ld2, j_i = g_i - i_i * ld2;
        int g_c = g_i;                       // next element to finish
        int i_c = i_i, j_c = j_i;
        dbl2 t0, t1, t2, t3, t4, t5, t6, t7, p0, p1, p2, p3, p4, p5, p6, p7;
        double l0, l1, l2, l3, l4, l5, l6, l7;
        int pad_ = 0;
        static_assert(OV_LIN == 8, "the ring below is written out for eight slots");
#define OV_REQUEST(TQ, PQ, LQ)                                                                                        \
    do {                                                                                                              \
        const unsigned ot_ = (unsigned)(g_i < gend ? g_i : gbeg) << 4;   /* (past the share: one hot line) */           \
        const unsigned op_ = (unsigned)j_i << 4;                                                                      \
        const unsigned ol_ = (unsigned)(i_i < rows ? i_i : rows - 1) << 3;                                            \
        asm volatile("global_load_dwordx4 %0, %3, %6\n\tglobal_load_dwordx4 %1, %4, %7\n\tglobal_load_dwordx2 %2, %5, %8" \
                     : "=&v"(TQ), "=&v"(PQ), "=&v"(LQ)                                                                \
                     : "v"(ot_), "v"(op_), "v"(ol_), "s"(Sv), "s"(Pv), "s"(lcol)                                      \
                     : "memory");                                                                                     \
        g_i += T;                                                                                                     \
        advance(i_i, j_i);                                                                                            \
    } while (0)
        // (the prologue's requests carry a fourth operation each — a load of one hot word — so that the count of
        // younger operations is 28 from the first turn of the ring on: 4 per slot, as in the loop, where it is the store)
// (its destination stays ONE live register up to the end of the ring: a dead one would be reused while the load is in flight)
#define OV_PAD() asm volatile("global_load_dword %0, %1, %2" : "+v"(pad_) : "v"(0u), "s"(lcol) : "memory")
#define OV_FINISH(TQ, PQ, LQ)                                                                                         \
    do {                                                                                                              \
        asm volatile("s_waitcnt vmcnt(28)" : "+v"(TQ), "+v"(PQ), "+v"(LQ));                                           \
        dbl2 v_ = TQ;                                                                                                 \
        if (i_c == r) {                                                                                               \
            v_.x = PQ.x * LQ;                                                                                         \
            v_.y = PQ.y * LQ;                                                                                         \
        } else {                                                                                                      \
            v_.x = fma(LQ, PQ.x, v_.x);                                                                               \
            v_.y = fma(LQ, PQ.y, v_.y);                                                                               \
        }                                                                                                             \
        if (j_c == je) {                                                                                              \
            const double unit_ = (i_c == r) ? 1.0 : 0.0;                                                              \
            if (e & 1) v_.y = unit_; else v_.x = unit_;                                                               \
        }                                                                                                             \
        const unsigned long long live_ = __ballot(g_c < gend);                                                        \
        const unsigned od_ = (unsigned)(g_c < gend ? g_c : gbeg) << 4;                                                \
        unsigned long long sv_;                                                                                       \
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tglobal_store_dwordx4 %2, %3, %4\n\ts_mov_b64 exec, %0" \
                     : "=&s"(sv_)                                                                                     \
                     : "s"(live_), "v"(od_), "v"(v_), "s"(Dv)                                                         \
                     : "memory");                                                                                     \
        g_c += T;                                                                                                     \
        advance(i_c, j_c);                                                                                            \
    } while (0)
        OV_REQUEST(t0, p0, l0); OV_PAD(); OV_REQUEST(t1, p1, l1); OV_PAD(); OV_REQUEST(t2, p2, l2); OV_PAD();
        OV_REQUEST(t3, p3, l3); OV_PAD(); OV_REQUEST(t4, p4, l4); OV_PAD(); OV_REQUEST(t5, p5, l5); OV_PAD();
        OV_REQUEST(t6, p6, l6); OV_PAD(); OV_REQUEST(t7, p7, l7); OV_PAD();
        while (g_c < gend) {
            OV_FINISH(t0, p0, l0); OV_REQUEST(t0, p0, l0);
            OV_FINISH(t1, p1, l1); OV_REQUEST(t1, p1, l1);
            OV_FINISH(t2, p2, l2); OV_REQUEST(t2, p2, l2);
            OV_FINISH(t3, p3, l3); OV_REQUEST(t3, p3, l3);
            OV_FINISH(t4, p4, l4); OV_REQUEST(t4, p4, l4);
            OV_FINISH(t5, p5, l5); OV_REQUEST(t5, p5, l5);
            OV_FINISH(t6, p6, l6); OV_REQUEST(t6, p6, l6);
            OV_FINISH(t7, p7, l7); OV_REQUEST(t7, p7, l7);
        }
        asm volatile("" ::"v"(pad_));
#undef OV_PAD
#undef OV_REQUEST
#undef OV_FINISH
        return;
    }
    // ---- one tile of 128 columns x 64 rows per workgroup
    const int b = (int)blockIdx.x - 1;
    const int by = b / nbx, bx = b - by * nbx;
    const int tx = threadIdx.x & (OV_TX - 1), ty = threadIdx.x / OV_TX;
    const int jp = bx * OV_TX + tx;  // column pair
    if (jp >= ld2) return;
    const double2 pr = S2[(size_t)r * ld2 + jp];   // the pivot row: row r of T_{k-1}
    const int i0 = (by * OV_TY + ty) * OV_RPT;
    double2 t[OV_RPT];
    double l[OV_RPT];
#pragma unroll
    for (int q = 0; q < OV_RPT; ++q) {
        const int i = i0 + q;
        if (i < rows) {
            t[q] = S2[(size_t)i * ld2 + jp];
            l[q] = lcol[i];
        }
    }
#pragma unroll
    for (int q = 0; q < OV_RPT; ++q) {
        const int i = i0 + q;
        if (i < rows) {
            double2 v = t[q];
            if (i == r) {
                v.x = pr.x * l[q];
                v.y = pr.y * l[q];
            } else {
                v.x = fma(l[q], pr.x, v.x);
                v.y = fma(l[q], pr.y, v.y);
            }
            if (jp == je) {
                const double unit = (i == r) ? 1.0 : 0.0;
                if (e & 1) v.y = unit; else v.x = unit;
            }
            D2[(size_t)i * ld2 + jp] = v;
        }
    }
}

}  // namespace

size_t lp_overlap_lds_bytes(int m) { return 2 * sizeof(double) * (size_t)(m + 2) + 16 + sizeof(lpdev::BlockChainScratch); }
// the three-round-trip selector: the priced cost row in LDS too, rows in registers — if that still leaves room for
// two update workgroups per CU (every workgroup of the launch is given the selector's dynamic LDS)
static bool overlap_fast_fits(int m, int n) {
    return m + 1 <= OVF_ROWS * OV_TX * OV_TY && lp_overlap_lds_bytes(m) + sizeof(double) * (size_t)n <= 78 * 1024;
}

// The shape runs on this path if the selector's two m-vectors fit one CU's LDS.
bool lp_overlap_fits(int m) { return lp_overlap_lds_bytes(m) <= 156 * 1024; }
// What AUTO takes: every workgroup of the launch is given the selector's dynamic LDS, so beyond 80 KB (m ~ 5000) the
// update would run ONE workgroup per CU — never measured; the launch pair per pivot has no such coupling.
bool lp_overlap_auto(int m) { return lp_overlap_lds_bytes(m) <= 80 * 1024; }

// Second tableau and second eta slot, on first use (each checked on its own: a failed second allocation must not
// leave a later call with the first one only).  AUTO calls this before it commits to the algorithm: without the
// memory for a second tableau it takes the launch pair per pivot instead.
int lp_overlap_prepare(lp_simplex_problem* p) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    if (!lp_overlap_fits(d.m)) LP_FAIL(ctx, LP_BAD_ARG, "overlapped simplex: m too large for the selector's LDS");
    if (!p->ov_T) LP_HIP(ctx, hipMalloc(&p->ov_T, p->tableau_bytes));
    if (!p->ov_vec) LP_HIP(ctx, hipMalloc(&p->ov_vec, sizeof(double) * ((size_t)d.m + 2 + (size_t)d.ld) + 64));
    return LP_OPTIMAL;
}

int lp_simplex_run_overlap(lp_simplex_problem* p, double eps, int max_iter, lp_simplex_stats* stats) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    hipStream_t s = ctx->stream;
    {
        const int rc = lp_overlap_prepare(p);
        if (rc) return rc;
    }
    OverlapDev ov;
    ov.T[0] = d.T;
    ov.T[1] = p->ov_T;
    ov.lcol[0] = d.lcol;
    ov.prow[0] = d.prow;
    ov.prow[1] = p->ov_vec;                         // ld doubles (16-byte aligned: read as double2)
    ov.lcol[1] = p->ov_vec + d.ld;                  // m + 2 doubles
    ov.slot = reinterpret_cast<int*>(p->ov_vec + d.ld + d.m + 2);   // 8 ints
    const int fast = overlap_fast_fits(d.m, d.n) ? 1 : 0;
    const size_t shm = lp_overlap_lds_bytes(d.m) + (fast ? sizeof(double) * (size_t)d.n : 0);
    if (shm > 48 * 1024 && !p->ov_attr) {
        LP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_simplex_overlap<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        LP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_simplex_overlap<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        p->ov_attr = true;
    }
    const int nbx = lp_ceil_div(d.ld / 2, OV_TX);
    const int nby = lp_ceil_div(d.m + 1, OV_TY * OV_RPT);
    // the update: a tile per workgroup, or — fewer than six tiles per CU — one persistent workgroup per CU with even
    // linear shares (see the kernel; two per CU, as the thread count would allow, do not fit its registers: the second
    // generation of 255 short shares ran behind the first, 29 us against 23)
#ifndef LP_OVERLAP_LINEAR_ROUNDS   // diagnostic builds: tile rounds below which the linear form is taken / its workgroups per CU
#define LP_OVERLAP_LINEAR_ROUNDS 6
#endif
#ifndef LP_OVERLAP_LINEAR_WGS
#define LP_OVERLAP_LINEAR_WGS 1
#endif
    const long long pairs = (long long)(d.m + 1) * (d.ld / 2);
    const int linear = (nbx * nby < LP_OVERLAP_LINEAR_ROUNDS * ctx->num_cus && pairs < (1LL << 27)) ? 1 : 0;   // (32-bit byte offsets in the ring)
    const unsigned grid = 1u + (unsigned)(linear ? std::max(1, std::min(nbx * nby, LP_OVERLAP_LINEAR_WGS * ctx->num_cus - 1)) : nbx * nby);
    int launches = 0;
    LP_HIP(ctx, hipEventRecord(p->ev0, s));
    hipLaunchKernelGGL(k_overlap_init, 1, 1, 0, s, d, ov, eps, max_iter);
    ++launches;
    int batch = 16, k = 0;
    int status = kRunning;
    // Launches turn into no-ops once the state leaves kRunning, so they are queued in growing batches and
    // the status word is polled once per batch.
    for (;;) {
        for (int q = 0; q < batch; ++q, ++k)
            if (linear)
                hipLaunchKernelGGL(k_simplex_overlap<true>, grid, OV_TX * OV_TY, shm, s, d, ov, k, nbx, fast);
            else
                hipLaunchKernelGGL(k_simplex_overlap<false>, grid, OV_TX * OV_TY, shm, s, d, ov, k, nbx, fast);
        launches += batch;
        LP_HIP(ctx, hipMemcpyAsync(p->h_state, d.state, sizeof(SimplexState), hipMemcpyDeviceToHost, s));
        LP_HIP(ctx, hipStreamSynchronize(s));
        status = p->h_state->status;
        if (status != kRunning) break;
        if (batch < 256) batch *= 2;
    }
    // T_N lives in buffer N & 1 (launch N applied the last pivot; the launches behind it did nothing)
    if (p->h_state->iters & 1) LP_HIP(ctx, hipMemcpyAsync(d.T, p->ov_T, p->tableau_bytes, hipMemcpyDeviceToDevice, s));
    LP_HIP(ctx, hipEventRecord(p->ev1, s));
    LP_HIP(ctx, hipEventSynchronize(p->ev1));
    LP_HIP(ctx, hipGetLastError());
    float ms = 0.f;
    LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->last_status = status;
    p->last_algo = LP_SIMPLEX_ALGO_OVERLAP;
    p->last_iters = p->h_state->iters;
    if (stats) {
        stats->status = status;
        stats->pivots = p->h_state->iters;
        stats->launches = launches;
        stats->solve_ms = ms;
        stats->update_ms = 0.f;
        stats->update_launches = 0;
        stats->bytes_per_pivot = 16.0 * (double)d.m * (double)(d.n + 1);
    }
    return status;
}
