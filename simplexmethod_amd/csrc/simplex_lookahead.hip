// simplex_lookahead.hip — single-LP tableau simplex with J-pivot look-ahead
// (LP_SIMPLEX_ALGO_LOOKAHEAD; what LP_SIMPLEX_ALGO_AUTO selects whenever the selector's
// working set fits one CU's LDS).
//
// Same pivot rules as simplex_launch.hip (/root/reference/src/SimplexSolover.h:152-196)
// and bit-identical tableau values, but the 4.2 MB tableau is read and written once per
// J pivots instead of once per pivot:
//
//   selector (ONE workgroup):  chooses J consecutive pivots.  Everything a pivot rule
//     needs — the reduced-cost row, xB, the entering column, the pivot row — is O(m+n)
//     data; it is obtained from the STALE tableau in HBM plus the etas staged so far in
//     this batch (kept in LDS), applying to each needed element exactly the fma sequence
//     the update would have applied.  The reduced-cost row and xB live in LDS across the
//     batch.  Output: J eta columns (F of SimplexSolover.h:198-204) and J pivot rows.
//   rank-J update (whole chip): every tableau element takes its J fused multiply-adds
//     in pivot order in registers — one HBM read + one write per J pivots, and the same
//     bits as J successive rank-1 updates (each element's operation sequence is
//     unchanged).
//
// Algorithmic bytes stay 16*m*(n+1) per pivot; HBM traffic per pivot
// drops by J.
#include "device_select.hpp"
#include "lp_internal.hpp"
#include "simplex_problem.hpp"

namespace {

constexpr int kRunning = -100;
#ifndef LP_SEL_THREADS
#define LP_SEL_THREADS 1024
#endif
constexpr int SEL_THREADS = LP_SEL_THREADS;

// LDS carve of the selector (all in the dynamic region, 16-B aligned pieces)
struct QInfo {   // one staged pivot, read with a single 16-B LDS load
    double inv;  // 1/u_r
    int rq;      // leaving position
    int eq;      // entering column
};

struct SelScratch {  // what the scanning wave hands to the rest of the workgroup
    int sel;
    int pad[3];
};
constexpr int kScratchDoubles = (int)(sizeof(SelScratch) / 8);

struct SelLds {
    SelScratch* sc;  // selected index of the last scan (16-B aligned, first)
    QInfo* qi;      // J   : staged pivots (16-B aligned)
    double* d;      // n+1 : reduced-cost row (entry n = -objective)
    double* rhs;    // m   : xB
    double* u;      // m   : entering column of the current tableau
    double* ratio;  // m   : xB_i/u_i where u_i > eps, +inf elsewhere (:185-186)
    double* lcH;    // J x m     : staged eta columns (entry r = 1/u_r)
    double* prH;    // J x (n+1) : staged pivot rows (before scaling)
    double* f64;    // 4 scalars: ur, inv, lm, rhs_r
    int* basis;     // m : N by position
    int* i32;       // 4 scalars: enter, leave
    unsigned char* nb;  // n : non-basic flags
};

__host__ __device__ inline size_t sel_lds_doubles(int m, int n, int J) {
    return (size_t)kScratchDoubles + 2 * (size_t)J + (size_t)(n + 1) + 3 * (size_t)m +
           (size_t)J * ((size_t)m + n + 1) + 4;
}

__host__ __device__ inline size_t sel_lds_bytes(int m, int n, int J) {
    size_t bytes = sel_lds_doubles(m, n, J) * 8 + (size_t)(m + 4) * 4;
    bytes = (bytes + 15) & ~(size_t)15;
    return bytes + (size_t)n + 16;
}

__device__ inline SelLds carve(double* base, int m, int n, int J) {
    SelLds s;
    s.sc = reinterpret_cast<SelScratch*>(base);
    s.qi = reinterpret_cast<QInfo*>(base + kScratchDoubles);
    s.d = base + kScratchDoubles + 2 * (size_t)J;
    s.rhs = s.d + (n + 1);
    s.u = s.rhs + m;
    s.ratio = s.u + m;
    s.lcH = s.ratio + m;
    s.prH = s.lcH + (size_t)J * m;
    s.f64 = s.prH + (size_t)J * (n + 1);
    s.basis = reinterpret_cast<int*>(base + sel_lds_doubles(m, n, J));
    s.i32 = s.basis + m;
    size_t bytes = sel_lds_doubles(m, n, J) * 8 + (size_t)(m + 4) * 4;
    bytes = (bytes + 15) & ~(size_t)15;
    s.nb = reinterpret_cast<unsigned char*>(base) + bytes;
    return s;
}

__global__ __launch_bounds__(SEL_THREADS) void k_look_select(SimplexDev d, LookDev la) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    SimplexState* st = d.state;
    const int tid = threadIdx.x;
    if (st->status != kRunning) {
        if (tid == 0) *la.count = 0;
        return;
    }
    const int m = d.m, n = d.n, ld = d.ld, J = la.J;
    SelLds s = carve(smem, m, n, J);
    const double eps = st->eps;
    const int max_iter = st->max_iter;
    int iters = st->iters;
    // The LDS copy of the reduced-cost row holds the pricing scan's sentinel (-inf / +inf) in the
    // basic columns instead of their exact zeros: the scan then needs no mask (one LDS read per
    // entry instead of two and a select).  A basic column's entry is only ever read again when its
    // variable leaves the basis, where its reduced cost is known to be 0.
    const double d_sentinel = d.maximize ? -INFINITY : INFINITY;
    for (int j = tid; j < n; j += SEL_THREADS) {
        const unsigned char nbj = d.nonbasic[j];
        s.nb[j] = nbj;
        s.d[j] = nbj ? la.dvec[j] : d_sentinel;
    }
    if (tid == 0) s.d[n] = la.dvec[n];
    for (int i = tid; i < m; i += SEL_THREADS) {
        s.rhs[i] = la.rhs[i];
        s.basis[i] = d.basis[i];
    }
    __syncthreads();

    int cnt = 0;
    int status = kRunning;
    const double* T = d.T;
    unsigned long long* stamps = la.stamps ? la.stamps + (size_t)iters * 8 : nullptr;
#define LP_STAMP(k)                                                                       \
    do {                                                                                  \
        if (stamps && tid == 0) stamps[(size_t)cnt * 8 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
    for (int q0 = 0; q0 < J; ++q0) {
        const int sidx = q0;  // index of the pivot being staged
        if (iters >= max_iter) {  // SimplexSolover.h:429,:450
            status = LP_ITER_LIMIT;
            break;
        }
        // ---- pricing, :152-174, on the LDS-resident reduced-cost row
        LP_STAMP(0);
        // (one wave scans the whole row — 16 entries per lane at n = 1024 — and publishes the result:
        // two wave reductions and one barrier instead of a two-stage block reduction with two)
        if (tid < 64) {
            double best;
            auto load = [&](int j, bool& ok) {
                ok = true;          // basic columns hold the sentinel (complement(), :97-108)
                return s.d[j];
            };
            int e0 = d.maximize ? lpdev::wave_chain_select<true>(n, eps, best, load)
                                : lpdev::wave_chain_select<false>(n, eps, best, load);
            const bool optimal = d.maximize ? (best <= eps) : (best >= -eps);
            if (optimal) e0 = -1;
            if (tid == 0) s.sc->sel = e0;
        }
        __syncthreads();
        const int e = s.sc->sel;
        LP_STAMP(1);
        if (e < 0) {
            status = LP_OPTIMAL;
            break;
        }
        // ---- entering column of the CURRENT tableau: stale column + staged etas (:176)
        int any_pos = 0;
        for (int i = tid; i < m; i += SEL_THREADS) {
            double t = T[(size_t)i * ld + e];
#pragma unroll 4
            for (int q = 0; q < sidx; ++q) {
                const QInfo qi = s.qi[q];
                const double l = s.lcH[(size_t)q * m + i];
                const double pe = s.prH[(size_t)q * (n + 1) + e];
                t = (i == qi.rq) ? t * qi.inv : fma(l, pe, t);
                if (e == qi.eq) t = (i == qi.rq) ? 1.0 : 0.0;
            }
            s.u[i] = t;
            s.ratio[i] = (t > eps) ? s.rhs[i] / t : INFINITY;  // :185-186
            if (!(t <= eps)) any_pos = 1;  // :179
        }
        if (!__syncthreads_or(any_pos)) {
            status = LP_UNBOUNDED;
            break;
        }
        LP_STAMP(2);
        // ---- ratio test, :181-194
        // ineligible rows hold +inf, which the < scan never takes
        if (tid < 64) {
            double theta;
            auto load = [&](int i, bool& ok) {
                ok = true;
                return s.ratio[i];
            };
            // entries per lane sized to m: a 16-entry tile would scan clamped duplicates for m <= 512
            const int r0 = (m <= 256)   ? lpdev::wave_chain_select<false, 4>(m, eps, theta, load)
                           : (m <= 512) ? lpdev::wave_chain_select<false, 8>(m, eps, theta, load)
                                        : lpdev::wave_chain_select<false, 16>(m, eps, theta, load);
            if (tid == 0) s.sc->sel = r0;
        }
        __syncthreads();
        const int r = s.sc->sel;
        LP_STAMP(3);
        if (r < 0) {
            status = LP_UNBOUNDED;
            break;
        }
        const double ur = s.u[r];
        const double inv = 1.0 / ur;       // F(r,r), :204
        const double lm = -s.d[e] / ur;    // F row of the reduced costs
        const double rhs_r = s.rhs[r];
        const int oldb = s.basis[r];       // leaves the basis with this pivot (read before thread 0 rewrites it)
        __syncthreads();                   // everyone has read d[e], rhs[r], basis[r] before they change
        // ---- pivot row of the CURRENT tableau (before scaling) + reduced-cost update.  The stale
        // row's HBM read is issued first and the eta column + xB update (LDS only) run under it.
        double* prS = s.prH + (size_t)sidx * (n + 1);
        double* etaP = la.etaP + (size_t)sidx * ld;
        const double t_first = (tid < n) ? T[(size_t)r * ld + tid] : 0.0;
        // ---- eta column (:198-204) + xB update
        double* lcS = s.lcH + (size_t)sidx * m;
        double* etaL = la.etaL + (size_t)sidx * la.rows_pad;
        for (int i = tid; i < m; i += SEL_THREADS) {
            const double l = (i == r) ? inv : -s.u[i] / ur;
            lcS[i] = l;
            etaL[i] = l;
            s.rhs[i] = (i == r) ? rhs_r * inv : fma(l, rhs_r, s.rhs[i]);
        }
        LP_STAMP(4);
        for (int j = tid; j < n; j += SEL_THREADS) {
            double t = (j == tid) ? t_first : T[(size_t)r * ld + j];
#pragma unroll 4
            for (int q = 0; q < sidx; ++q) {
                const QInfo qi = s.qi[q];
                const double l = s.lcH[(size_t)q * m + r];
                const double pj = s.prH[(size_t)q * (n + 1) + j];
                t = (r == qi.rq) ? t * qi.inv : fma(l, pj, t);
                if (j == qi.eq) t = (r == qi.rq) ? 1.0 : 0.0;
            }
            prS[j] = t;
            etaP[j] = t;
            const double dj = fma(lm, t, (j == oldb) ? 0.0 : s.d[j]);   // (the leaving variable's was 0)
            s.d[j] = (j == e) ? d_sentinel : dj;                          // (e becomes basic)
        }
        LP_STAMP(5);
        if (tid == 0) {
            prS[n] = rhs_r;
            etaP[n] = rhs_r;
            s.d[n] = fma(lm, rhs_r, s.d[n]);
            etaL[m] = lm;
            QInfo qi;
            qi.inv = inv;
            qi.rq = r;
            qi.eq = e;
            s.qi[sidx] = qi;
            la.piv[2 * sidx] = e;
            la.piv[2 * sidx + 1] = r;
            s.basis[r] = e;  // :196
            s.nb[e] = 0;
            s.nb[oldb] = 1;
            if (iters < d.trace_cap) {
                d.trace_enter[iters] = e;
                d.trace_leave[iters] = r;
            }
        }
        LP_STAMP(6);
        __syncthreads();
        LP_STAMP(7);
        ++iters;
        ++cnt;
    }
#undef LP_STAMP
    __syncthreads();
    for (int j = tid; j <= n; j += SEL_THREADS) la.dvec[j] = s.d[j];
    for (int j = tid; j < n; j += SEL_THREADS) d.nonbasic[j] = s.nb[j];
    for (int i = tid; i < m; i += SEL_THREADS) {
        la.rhs[i] = s.rhs[i];
        d.basis[i] = s.basis[i];
    }
    if (tid == 0) {
        *la.count = cnt;
        st->iters = iters;
        st->status = status;
        st->pivot_valid = 0;
    }
}

// rank-J update, in place: t <- eta_{count-1}( ... eta_0(t) ... ) per element.
// A block owns a LU_ROWS x (2*LU_TX) tile of the tableau.  The eta operands of the whole batch
// that touch the tile — J eta-column entries per tile row, J pivot-row entries per tile column —
// are staged once in LDS (18 KB at J = 16), so per launch the tableau is read and written once
// and the eta traffic is a fraction of that; each thread then applies the J fused multiply-adds
// to its LU_RPT rows x one 16-B column pair in registers.
constexpr int LU_TX = 64;    // column pairs per block (128 columns)
constexpr int LU_TY = 4;
#ifndef LP_LU_RPT
#define LP_LU_RPT 4
#endif
constexpr int LU_RPT = LP_LU_RPT;    // rows per thread
constexpr int LU_ROWS = LU_TY * LU_RPT;  // 16 rows per block
constexpr int LU_JMAX = 16;  // upper bound of LookDev::J

__global__ __launch_bounds__(LU_TX* LU_TY) void k_look_update(SimplexDev d, LookDev la) {
    __shared__ __attribute__((aligned(16))) double s_pr[LU_JMAX][2 * LU_TX];
    __shared__ double s_l[LU_JMAX][LU_ROWS];
    __shared__ int s_piv[2 * LU_JMAX];
    const int ld2 = d.ld >> 1;
    const int rows = d.m + 1;
    const int tid = threadIdx.y * LU_TX + threadIdx.x;
    const int col0 = blockIdx.x * 2 * LU_TX;   // first column of the tile
    const int row0 = blockIdx.y * LU_ROWS;
    const int jp = blockIdx.x * LU_TX + threadIdx.x;
    const bool live = jp < ld2;
    const int i0 = row0 + threadIdx.y * LU_RPT;
    double2* T2 = reinterpret_cast<double2*>(d.T);
    // Everything below is issued before the staged-pivot count is known, so that the count, the
    // tile and the eta operands are all in flight together (one memory round trip, not three);
    // the whole J-slot eta buffer is staged, entries beyond `count` are simply not used.
    double2 t[LU_RPT];
#pragma unroll
    for (int k = 0; k < LU_RPT; ++k)
        t[k] = (live && i0 + k < rows) ? T2[(size_t)(i0 + k) * ld2 + jp] : make_double2(0.0, 0.0);
    const int J = la.J;
    for (int k = tid; k < J * 2 * LU_TX; k += LU_TX * LU_TY) {
        const int q = k / (2 * LU_TX), j = k - q * 2 * LU_TX;
        s_pr[q][j] = (col0 + j < d.ld) ? la.etaP[(size_t)q * d.ld + col0 + j] : 0.0;
    }
    for (int k = tid; k < J * LU_ROWS; k += LU_TX * LU_TY) {
        const int q = k / LU_ROWS, i = k - q * LU_ROWS;
        s_l[q][i] = (row0 + i < rows) ? la.etaL[(size_t)q * la.rows_pad + row0 + i] : 0.0;
    }
    if (tid < 2 * J) s_piv[tid] = la.piv[tid];
    const int count = *la.count;
    if (count == 0) return;  // block-uniform
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < count; ++q) {
        const int e = s_piv[2 * q], r = s_piv[2 * q + 1];
        const int je = e >> 1;
        const double2 pr = *reinterpret_cast<const double2*>(&s_pr[q][2 * threadIdx.x]);
#pragma unroll
        for (int k = 0; k < LU_RPT; ++k) {
            const int i = i0 + k;
            const double lk = s_l[q][threadIdx.y * LU_RPT + k];
            if (i == r) {
                t[k].x = t[k].x * lk;
                t[k].y = t[k].y * lk;
            } else {
                t[k].x = fma(lk, pr.x, t[k].x);
                t[k].y = fma(lk, pr.y, t[k].y);
            }
            if (jp == je) {
                const double unit = (i == r) ? 1.0 : 0.0;
                if (e & 1) t[k].y = unit; else t[k].x = unit;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < LU_RPT; ++k)
        if (live && i0 + k < rows) T2[(size_t)(i0 + k) * ld2 + jp] = t[k];
}

// dvec <- row m of T, rhs <- column n of T (after upload / crash / reset)
__global__ void k_look_init(SimplexDev d, LookDev la) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= d.n) la.dvec[k] = d.T[(size_t)d.m * d.ld + k];
    if (k < d.m) la.rhs[k] = d.T[(size_t)k * d.ld + d.n];
    if (k == 0) *la.count = 0;
}

__global__ void k_look_state_init(SimplexDev d, double eps, int max_iter) {
    SimplexState* st = d.state;
    st->status = kRunning;
    st->iters = 0;
    st->max_iter = max_iter;
    st->enter = st->leave = -1;
    st->pivot_valid = 0;
    st->eps = eps;
}

}  // namespace

// Largest J (<= 16) whose selector fits the 160 KiB LDS of one CU; 0 = does not fit.
int lp_lookahead_pick_j(int m, int n) {
    const size_t cap = 156 * 1024;
    int J = 0;
    for (int j = 1; j <= 16; ++j)
        if (sel_lds_bytes(m, n, j) <= cap) J = j;
    return J;
}

int lp_lookahead_prepare(lp_simplex_problem* p) {
    // one-time opt-in to > 64 KiB of dynamic LDS for the selector
    const size_t bytes = sel_lds_bytes(p->dev.m, p->dev.n, p->look.J);
    LP_HIP(p->ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_look_select),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return LP_OPTIMAL;
}

int lp_lookahead_init_vectors(lp_simplex_problem* p) {
    const SimplexDev& d = p->dev;
    const int span = (d.n + 1 > d.m) ? d.n + 1 : d.m;
    hipLaunchKernelGGL(k_look_init, lp_ceil_div(span, 256), 256, 0, p->ctx->stream, d, p->look);
    return LP_OPTIMAL;
}

int lp_simplex_run_lookahead(lp_simplex_problem* p, double eps, int max_iter,
                             lp_simplex_stats* stats) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    const LookDev& la = p->look;
    hipStream_t s = ctx->stream;
    const size_t shm = sel_lds_bytes(d.m, d.n, la.J);
    const dim3 ugrid(lp_ceil_div(d.ld / 2, LU_TX), lp_ceil_div(d.m + 1, LU_ROWS));
    int launches = 0;
    LP_HIP(ctx, hipEventRecord(p->ev0, s));
    hipLaunchKernelGGL(k_look_state_init, 1, 1, 0, s, d, eps, max_iter);
    lp_lookahead_init_vectors(p);
    launches += 2;
    int batches = 4;
    int status = kRunning;
    // HIP events around the first kMaxTimed rank-J update launches (the kernel the HBM roofline
    // is quoted on); launches after termination are no-ops and are not counted.
    constexpr int kMaxTimed = 512;
    if (p->upd_events.empty()) {
        p->upd_events.resize(2 * kMaxTimed);
        for (auto& e : p->upd_events) LP_HIP(ctx, hipEventCreate(&e));
    }
    int timed = 0;
    for (;;) {
        for (int k = 0; k < batches; ++k) {
            hipLaunchKernelGGL(k_look_select, 1, SEL_THREADS, shm, s, d, la);
            if (p->profile_updates && timed < kMaxTimed) LP_HIP(ctx, hipEventRecord(p->upd_events[2 * timed], s));
            hipLaunchKernelGGL(k_look_update, ugrid, dim3(LU_TX, LU_TY), 0, s, d, la);
            if (p->profile_updates && timed < kMaxTimed) {
                LP_HIP(ctx, hipEventRecord(p->upd_events[2 * timed + 1], s));
                ++timed;
            }
        }
        launches += 2 * batches;
        LP_HIP(ctx, hipMemcpyAsync(p->h_state, d.state, sizeof(SimplexState), hipMemcpyDeviceToHost, s));
        LP_HIP(ctx, hipStreamSynchronize(s));
        status = p->h_state->status;
        if (status != kRunning) break;
        if (batches < 64) batches *= 2;
    }
    LP_HIP(ctx, hipEventRecord(p->ev1, s));
    LP_HIP(ctx, hipEventSynchronize(p->ev1));
    LP_HIP(ctx, hipGetLastError());
    float ms = 0.f;
    LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->last_status = status;
    p->last_algo = LP_SIMPLEX_ALGO_LOOKAHEAD;
    p->last_iters = p->h_state->iters;
    if (stats) {
        stats->status = status;
        stats->pivots = p->h_state->iters;
        stats->launches = launches;
        stats->solve_ms = ms;
        // update launches that did real work: one per started batch of J pivots
        int real = (p->h_state->iters + la.J - 1) / la.J;
        if (real > timed) real = timed;
        float upd = 0.f;
        for (int k = 0; k < real; ++k) {
            float t = 0.f;
            LP_HIP(ctx, hipEventElapsedTime(&t, p->upd_events[2 * k], p->upd_events[2 * k + 1]));
            upd += t;
        }
        stats->update_ms = upd;
        stats->update_launches = real;
        stats->bytes_per_pivot = 16.0 * (double)d.m * (double)(d.n + 1);
    }
    return status;
}

// Micro-benchmark of the rank-J update alone: stage one batch of J pivots on the problem's
// current tableau with the selector, then replay the update launch `iters` times between two
// HIP events (the same etas are re-applied, so values drift — irrelevant for timing; the
// tableau and solver state are restored afterwards).
int lp_lookahead_bench_update(lp_simplex_problem* p, int iters, float* ms_per_launch, int* pivots_out) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    const LookDev& la = p->look;
    hipStream_t s = ctx->stream;
    if (la.J < 1 || iters <= 0) LP_FAIL(ctx, LP_BAD_ARG, "look-ahead path unavailable for this problem");
    int rc = lp_lookahead_prepare(p);
    if (rc) return rc;
    const size_t shm = sel_lds_bytes(d.m, d.n, la.J);
    const dim3 ugrid(lp_ceil_div(d.ld / 2, LU_TX), lp_ceil_div(d.m + 1, LU_ROWS));
    if (!p->dscratchT) LP_HIP(ctx, hipMalloc(&p->dscratchT, p->tableau_bytes));   // (micro-benchmarks only)
    LP_HIP(ctx, hipMemcpyAsync(p->dscratchT, d.T, p->tableau_bytes, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_look_state_init, 1, 1, 0, s, d, 1e-9, 1 << 30);
    lp_lookahead_init_vectors(p);
    hipLaunchKernelGGL(k_look_select, 1, SEL_THREADS, shm, s, d, la);
    int count = 0;
    LP_HIP(ctx, hipMemcpyAsync(&count, la.count, sizeof(int), hipMemcpyDeviceToHost, s));
    LP_HIP(ctx, hipStreamSynchronize(s));
    if (count <= 0) LP_FAIL(ctx, LP_BAD_ARG, "no pivot could be staged on the current tableau");
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_look_update, ugrid, dim3(LU_TX, LU_TY), 0, s, d, la);
    LP_HIP(ctx, hipEventRecord(p->ev0, s));
    for (int k = 0; k < iters; ++k) hipLaunchKernelGGL(k_look_update, ugrid, dim3(LU_TX, LU_TY), 0, s, d, la);
    LP_HIP(ctx, hipEventRecord(p->ev1, s));
    LP_HIP(ctx, hipEventSynchronize(p->ev1));
    float ms = 0.f;
    LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
    if (ms_per_launch) *ms_per_launch = ms / (float)iters;
    if (pivots_out) *pivots_out = count;
    // restore: tableau, basis bookkeeping (the selector moved it), staged-count
    LP_HIP(ctx, hipMemcpyAsync(d.T, p->dscratchT, p->tableau_bytes, hipMemcpyDeviceToDevice, s));
    LP_HIP(ctx, hipMemcpyAsync(d.basis, p->dbasis0, sizeof(int) * (size_t)d.m, hipMemcpyDeviceToDevice, s));
    LP_HIP(ctx, hipMemcpyAsync(d.nonbasic, p->dnonbasic0, (size_t)d.n, hipMemcpyDeviceToDevice, s));
    LP_HIP(ctx, hipMemsetAsync(la.count, 0, sizeof(int), s));
    LP_HIP(ctx, hipStreamSynchronize(s));
    LP_HIP(ctx, hipGetLastError());
    return LP_OPTIMAL;
}
