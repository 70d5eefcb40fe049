// simplex_launch.hip — single-LP tableau simplex, one select + one rank-1-update
// launch per pivot (LP_SIMPLEX_ALGO_LAUNCH), plus the device-side computeBFS
// ("crash") that brings an arbitrary initial basis to tableau form.
//
// Replaces Solver::solveWithBasis / simplexIter / computeBFS
// (/root/reference/src/SimplexSolover.h:408-451, :135-209, :117-133).  The
// reference keeps a dense Binv and recomputes it by FullPivLU after every pivot;
// this path keeps the full tableau T = Binv*[A | b] plus the reduced-cost row in
// HBM and applies Binv = F*Binv (:198-206) as the equivalent elementwise rank-1
// Gauss-Jordan update.  Pivot RULES (:152-196) are reproduced exactly; see
// device_select.hpp for how the sequential EPS-hysteresis scans are replayed.
//
// HBM layout: T is (m+1) x ld row-major fp64, ld = round_up(n+1, 8) so every row
// starts on a 64-B boundary and rows are read/written as 16-B double2 lanes.
// Row i < m is the constraint row of basis position i, row m the reduced costs
// d_j = c_j - z_j; column n is xB (entry [m][n] = -objective); pad columns are 0.
#include "device_select.hpp"
#include "lp_internal.hpp"
#include "simplex_problem.hpp"

namespace {

constexpr int kRunning = -100;  // SimplexState::status while pivoting

// ---------------------------------------------------------------------------
// select: pricing (:152-174), unbounded test (:179), ratio test (:181-194),
// basis bookkeeping (:196) and the eta column of F (:198-204).  One workgroup.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_simplex_select(SimplexDev d) {
    SimplexState* st = d.state;
    // all LDS in the dynamic region (keeps its base 16-B aligned): u[m+1], ratio[m], 3 ints
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    double* s_u = s_dyn;
    double* s_ratio = s_dyn + (d.m + 2);
    int* s_int = reinterpret_cast<int*>(s_dyn + 2 * (d.m + 2));
    int& s_enter = s_int[0];
    int& s_leave = s_int[1];
    int& s_flag = s_int[2];

    const int tid = threadIdx.x;
    if (st->status != kRunning) {
        if (tid == 0) st->pivot_valid = 0;
        return;
    }
    const int m = d.m, n = d.n, ld = d.ld;
    const double eps = st->eps;
    if (st->iters >= st->max_iter) {  // while (iteration < MAX_ITER) ... throw, :429,:450
        if (tid == 0) {
            st->status = LP_ITER_LIMIT;
            st->pivot_valid = 0;
        }
        return;
    }
    const double* drow = d.T + (size_t)m * ld;
    if (tid < 64) {
        double best;
        int e;
        auto load = [&](int j, bool& ok) {
            ok = d.nonbasic[j] != 0;  // complement(), :97-108
            return drow[j];
        };
        if (d.maximize)
            e = lpdev::wave_chain_select<true>(n, eps, best, load);
        else
            e = lpdev::wave_chain_select<false>(n, eps, best, load);
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
        }
        return;
    }
    // u = column e of the tableau (Binv*A.col(enter), :176), incl. the reduced-cost row
    int any_pos = 0;
    for (int i = tid; i <= m; i += blockDim.x) {
        const double ui = d.T[(size_t)i * ld + e];
        s_u[i] = ui;
        if (i < m) {
            s_ratio[i] = (ui > eps) ? d.T[(size_t)i * ld + n] / ui : INFINITY;  // :185-186
            if (!(ui <= eps)) any_pos = 1;  // (u.array() <= EPS).all(), :179
        }
    }
    if (any_pos) s_flag = 1;
    __syncthreads();
    if (!s_flag) {
        if (tid == 0) {
            st->status = LP_UNBOUNDED;
            st->pivot_valid = 0;
        }
        return;
    }
    if (tid < 64) {
        double theta;
        auto load = [&](int i, bool& ok) {
            ok = true;  // ineligible rows hold +inf, which the < scan never takes
            return s_ratio[i];
        };
        const int r = lpdev::wave_chain_select<false>(m, eps, theta, load);  // :187-190
        if (tid == 0) s_leave = r;
    }
    __syncthreads();
    const int r = s_leave;
    if (r < 0) {  // :194
        if (tid == 0) {
            st->status = LP_UNBOUNDED;
            st->pivot_valid = 0;
        }
        return;
    }
    const double ur = s_u[r];
    for (int i = tid; i <= m; i += blockDim.x)  // F(i,r) = -u_i/u_r, F(r,r) = 1/u_r, :198-204
        d.lcol[i] = (i == r) ? 1.0 / ur : -s_u[i] / ur;
    const double* trow = d.T + (size_t)r * ld;
    for (int j = tid; j < ld; j += blockDim.x) d.prow[j] = trow[j];
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
    }
}

// ---------------------------------------------------------------------------
// rank-1 Gauss-Jordan update: T_i += l_i * T_r (i != r), T_r *= 1/u_r; column e
// becomes the exact unit vector.  HBM-bound: every element read once, written
// once (16*m*(n+1) algorithmic bytes per pivot).  Each thread owns one 16-B
// column pair and RPT rows; the pivot-row pair is read once and kept in
// registers; multipliers come from the contiguous lcol vector.
// ---------------------------------------------------------------------------
constexpr int UPD_TX = 64;   // column pairs per block (128 columns, 1 KiB per row segment)
constexpr int UPD_TY = 4;    // row groups per block
constexpr int UPD_RPT = 4;   // rows per thread

__global__ __launch_bounds__(UPD_TX* UPD_TY) void k_simplex_update(SimplexDev d) {
    const SimplexState* st = d.state;
    if (!st->pivot_valid) return;
    const int r = st->leave, e = st->enter;
    const int ld2 = d.ld >> 1;
    const int jp = blockIdx.x * UPD_TX + threadIdx.x;  // column pair
    if (jp >= ld2) return;
    const int rows = d.m + 1;
    const double2 pr = reinterpret_cast<const double2*>(d.prow)[jp];
    double2* T2 = reinterpret_cast<double2*>(d.T);
    const int i0 = (blockIdx.y * UPD_TY + threadIdx.y) * UPD_RPT;
    const int je = e >> 1;
    double2 t[UPD_RPT];
    double l[UPD_RPT];
#pragma unroll
    for (int k = 0; k < UPD_RPT; ++k) {
        const int i = i0 + k;
        if (i < rows) {
            t[k] = T2[(size_t)i * ld2 + jp];
            l[k] = d.lcol[i];
        }
    }
#pragma unroll
    for (int k = 0; k < UPD_RPT; ++k) {
        const int i = i0 + k;
        if (i < rows) {
            double2 v = t[k];
            if (i == r) {
                v.x = pr.x * l[k];
                v.y = pr.y * l[k];
            } else {
                v.x = fma(l[k], pr.x, v.x);
                v.y = fma(l[k], pr.y, v.y);
            }
            if (jp == je) {
                const double unit = (i == r) ? 1.0 : 0.0;
                if (e & 1) v.y = unit; else v.x = unit;
            }
            T2[(size_t)i * ld2 + jp] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// crash (computeBFS at :423 for a basis that is not the slack identity):
// step t brings column N(t) to a unit vector by a Gauss-Jordan pivot on the row
// of largest |entry| among rows not used yet (first maximum).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_crash_select(SimplexDev d, int t) {
    SimplexState* st = d.state;
    __shared__ double s_val[16];
    __shared__ int s_idx[16];
    __shared__ int s_p;
    const int tid = threadIdx.x;
    if (st->status != kRunning) {
        if (tid == 0) st->pivot_valid = 0;
        return;
    }
    const int m = d.m, ld = d.ld;
    const int q = d.basis[t];
    double big = -1.0;
    int p = INT_MAX;
    for (int i = tid; i < m; i += blockDim.x) {
        if (d.rowused[i]) continue;
        const double a = fabs(d.T[(size_t)i * ld + q]);
        if (a > big) {  // i ascending per thread: strict > keeps the first maximum
            big = a;
            p = i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(big, off, 64);
        const int op = __shfl_xor(p, off, 64);
        if (ob > big || (ob == big && op < p)) {
            big = ob;
            p = op;
        }
    }
    if ((tid & 63) == 0) {
        s_val[tid >> 6] = big;
        s_idx[tid >> 6] = p;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (s_val[w] > big || (s_val[w] == big && s_idx[w] < p)) {
                big = s_val[w];
                p = s_idx[w];
            }
        if (!(big > 0.0)) {
            st->status = LP_SINGULAR;
            st->pivot_valid = 0;
            p = -1;
        } else {
            if (big < st->minpiv) st->minpiv = big;
            if (big > st->maxpiv) st->maxpiv = big;
            d.rowused[p] = 1;
            d.rowpos[t] = p;
            st->enter = q;
            st->leave = p;
            st->pivot_valid = 1;
        }
        s_p = p;
    }
    __syncthreads();
    const int pr = s_p;
    if (pr < 0) return;
    const double ur = d.T[(size_t)pr * ld + q];
    for (int i = tid; i <= m; i += blockDim.x)
        d.lcol[i] = (i == pr) ? 1.0 / ur : -d.T[(size_t)i * ld + q] / ur;
    const double* trow = d.T + (size_t)pr * ld;
    for (int j = tid; j < ld; j += blockDim.x) d.prow[j] = trow[j];
}

// A pivot the HOST chose (two-phase drive-out, SimplexSolover.h:331-381 replaceArtificialColumns):
// stage the eta column and the pivot-row copy for (row, col) of the current tableau with the
// selectors' arithmetic, update the basis bookkeeping; k_simplex_update then applies it.
__global__ __launch_bounds__(1024) void k_force_select(SimplexDev d, int row, int col) {
    SimplexState* st = d.state;
    const int tid = threadIdx.x, m = d.m, ld = d.ld;
    const double ur = d.T[(size_t)row * ld + col];
    for (int i = tid; i <= m; i += blockDim.x)
        d.lcol[i] = (i == row) ? 1.0 / ur : -d.T[(size_t)i * ld + col] / ur;
    const double* trow = d.T + (size_t)row * ld;
    for (int j = tid; j < ld; j += blockDim.x) d.prow[j] = trow[j];
    if (tid == 0) {
        const int old = d.basis[row];
        d.basis[row] = col;
        d.nonbasic[col] = 0;
        d.nonbasic[old] = 1;
        st->enter = col;
        st->leave = row;
        st->pivot_valid = 1;
    }
}

// Drive-out step of the two-phase flow (replaceArtificialColumns, SimplexSolover.h:331-381), chosen
// ON THE DEVICE: the artificial basic at position `pos` leaves for the first non-basic column
// j < n_limit with |T[pos][j]| > eps; none = linearly dependent constraints (sticky flag pad0).
// Stages the pivot exactly as k_force_select does; k_simplex_update then applies it.
__global__ __launch_bounds__(1024) void k_driveout_select(SimplexDev d, int pos, int n_limit, double eps) {
    __shared__ int s_cand;
    SimplexState* st = d.state;
    const int tid = threadIdx.x, m = d.m, ld = d.ld;
    if (tid == 0) {
        s_cand = INT_MAX;
        st->pivot_valid = 0;
    }
    __syncthreads();
    if (st->pad0 != 0) return;   // an earlier position already failed
    const double* trow = d.T + (size_t)pos * ld;
    int mine = INT_MAX;
    for (int j = tid; j < n_limit; j += blockDim.x)
        if (d.nonbasic[j] && fabs(trow[j]) > eps) {
            mine = j;
            break;   // (ascending within the thread; the minimum over threads is the first overall)
        }
    if (mine != INT_MAX) atomicMin(&s_cand, mine);
    __syncthreads();
    const int col = s_cand;
    if (col == INT_MAX) {
        if (tid == 0) st->pad0 = 1;
        return;
    }
    const double ur = trow[col];
    for (int i = tid; i <= m; i += blockDim.x)
        d.lcol[i] = (i == pos) ? 1.0 / ur : -d.T[(size_t)i * ld + col] / ur;
    for (int j = tid; j < ld; j += blockDim.x) d.prow[j] = trow[j];
    if (tid == 0) {
        const int old = d.basis[pos];
        d.basis[pos] = col;
        d.nonbasic[col] = 0;
        d.nonbasic[old] = 1;
        st->enter = col;
        st->leave = pos;
        st->pivot_valid = 1;
        st->pad1 += 1;   // drive-out pivots applied
    }
}
__global__ void k_driveout_begin(SimplexDev d) {
    d.state->pad0 = 0;
    d.state->pad1 = 0;
    d.state->pivot_valid = 0;
}

// Phase II on the phase-I tableau: row m <- the original costs (0 for the artificial columns and the
// right-hand side); k_price_out_identity then prices it out over the current basis.  The artificial
// columns are barred from entering by clearing their non-basic flag (none of them is basic any more).
__global__ __launch_bounds__(256) void k_phase2_costs(SimplexDev d, const double* cost, int n_real) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > d.n) return;
    d.T[(size_t)d.m * d.ld + j] = (j < n_real) ? cost[j] : 0.0;
    if (j >= n_real && j < d.n) d.nonbasic[j] = 0;
}

// computeBFS for a basis whose columns are the unit vectors e_t in order but whose costs are not
// zero (the artificial basis of a phase-I problem): the m crash pivots would have pivot element 1
// and multipliers 0 on every constraint row, so all they do is eliminate the basic costs from the
// reduced-cost row, d_j <- fma(-c_N(t), T[t][j], d_j) for t = 0 .. m-1 in order, and set
// d_N(t) = 0.  One thread per column replays exactly that chain (same values as
// oracle/lp_oracle.c's crash loop; signs of zero entries of the constraint rows aside).
__global__ __launch_bounds__(256) void k_price_out_identity(SimplexDev d) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > d.n) return;
    const int m = d.m, ld = d.ld;
    const double* crow = d.T + (size_t)m * ld;   // still the original costs c
    double dj = crow[j];
    int own = -1;   // basis position whose column this is
    for (int t = 0; t < m; ++t) {
        const int e = d.basis[t];
        const double l = -crow[e] / 1.0;
        dj = (own >= 0) ? dj : fma(l, d.T[(size_t)t * ld + j], dj);
        if (e == j) {
            dj = 0.0;
            own = t;
        }
    }
    // other threads still need the original costs of the basic columns: the result goes to scratch
    // (prow, ld doubles) and is copied back by k_price_out_commit
    d.prow[j] = dj;
}
__global__ __launch_bounds__(256) void k_price_out_commit(SimplexDev d) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j <= d.n) d.T[(size_t)d.m * d.ld + j] = d.prow[j];
}

// Singularity verdict after the m crash pivots (min|piv| <= eps_mach*m*max|piv|, the
// FullPivLU::isInvertible threshold the reference relies on at :124-126) and the
// row permutation to basis-position order: dst row t = src row rowpos[t].
__global__ void k_crash_finish(SimplexDev d) {
    SimplexState* st = d.state;
    if (st->status == kRunning && st->minpiv <= 2.220446049250313e-16 * (double)d.m * st->maxpiv)
        st->status = LP_SINGULAR;
    st->pivot_valid = 0;
}

__global__ __launch_bounds__(256) void k_permute_rows(SimplexDev d, const double* src, double* dst) {
    const int t = blockIdx.x;  // destination row, 0..m
    // rowpos[] is complete only if all m crash pivots succeeded; after a singular verdict the
    // tableau is never used again and rows are copied in place (rowpos[t..] was never written)
    const int s = (t < d.m && d.state->status == kRunning) ? d.rowpos[t] : t;
    const double2* a = reinterpret_cast<const double2*>(src + (size_t)s * d.ld);
    double2* b = reinterpret_cast<double2*>(dst + (size_t)t * d.ld);
    for (int j = threadIdx.x; j < (d.ld >> 1); j += blockDim.x) b[j] = a[j];
}

__global__ void k_state_init(SimplexDev d, double eps, int max_iter) {
    SimplexState* st = d.state;
    st->status = kRunning;
    st->iters = 0;
    st->max_iter = max_iter;
    st->enter = st->leave = -1;
    st->pivot_valid = 0;
    st->eps = eps;
    st->minpiv = INFINITY;
    st->maxpiv = 0.0;
}

// x(N(t)) = xB(t), zeros elsewhere (:131-132)
__global__ void k_extract_x(SimplexDev d, double* x) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < d.n) x[j] = 0.0;
}
__global__ void k_scatter_x(SimplexDev d, double* x) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < d.m) x[d.basis[t]] = d.T[(size_t)t * d.ld + d.n];
}

}  // namespace

// ---------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------

static dim3 update_grid(const SimplexDev& d) {
    return dim3(lp_ceil_div(d.ld / 2, UPD_TX), lp_ceil_div(d.m + 1, UPD_TY * UPD_RPT));
}

int lp_simplex_force(lp_simplex_problem* p, int row, int col) {
    lp_context* ctx = p->ctx;
    hipLaunchKernelGGL(k_force_select, 1, 1024, 0, ctx->stream, p->dev, row, col);
    lp_simplex_launch_update(p);
    LP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LP_HIP(ctx, hipGetLastError());
    return LP_OPTIMAL;
}

void lp_simplex_launch_update(lp_simplex_problem* p) {
    hipLaunchKernelGGL(k_simplex_update, update_grid(p->dev), dim3(UPD_TX, UPD_TY), 0,
                       p->ctx->stream, p->dev);
}

// Queues the drive-out of the artificial basics at the given positions; one host sync at the end.
// Returns LP_OPTIMAL / LP_SINGULAR; *applied = pivots performed.
int lp_simplex_driveout(lp_simplex_problem* p, const int* positions, int count, int n_limit, double eps,
                        int* applied) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_driveout_begin, 1, 1, 0, s, d);
    for (int k = 0; k < count; ++k) {
        hipLaunchKernelGGL(k_driveout_select, 1, 1024, 0, s, d, positions[k], n_limit, eps);
        lp_simplex_launch_update(p);
    }
    LP_HIP(ctx, hipMemcpyAsync(p->h_state, d.state, sizeof(SimplexState), hipMemcpyDeviceToHost, s));
    LP_HIP(ctx, hipStreamSynchronize(s));
    LP_HIP(ctx, hipGetLastError());
    if (applied) *applied = p->h_state->pad1;
    return p->h_state->pad0 ? LP_SINGULAR : LP_OPTIMAL;
}

// Phase II continues on the current tableau with new costs (cost: n_real doubles on the host).
int lp_simplex_phase2_costs(lp_simplex_problem* p, const double* cost, int n_real, int maximize, int n_orig) {
    lp_context* ctx = p->ctx;
    SimplexDev& d = p->dev;
    hipStream_t s = ctx->stream;
    LP_HIP(ctx, hipMemcpyAsync(p->dx, cost, sizeof(double) * (size_t)n_real, hipMemcpyHostToDevice, s));   // dx: n doubles of scratch
    hipLaunchKernelGGL(k_phase2_costs, lp_ceil_div(d.n + 1, 256), 256, 0, s, d, (const double*)p->dx, n_real);
    hipLaunchKernelGGL(k_price_out_identity, lp_ceil_div(d.n + 1, 256), 256, 0, s, d);
    hipLaunchKernelGGL(k_price_out_commit, lp_ceil_div(d.n + 1, 256), 256, 0, s, d);
    LP_HIP(ctx, hipStreamSynchronize(s));
    LP_HIP(ctx, hipGetLastError());
    d.maximize = maximize ? 1 : 0;
    p->n_orig = n_orig;
    p->h_c.assign((size_t)d.n, 0.0);
    for (int j = 0; j < n_real; ++j) p->h_c[(size_t)j] = cost[j];
    return LP_OPTIMAL;
}

int lp_simplex_price_out_identity(lp_simplex_problem* p) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    hipLaunchKernelGGL(k_price_out_identity, lp_ceil_div(d.n + 1, 256), 256, 0, ctx->stream, d);
    hipLaunchKernelGGL(k_price_out_commit, lp_ceil_div(d.n + 1, 256), 256, 0, ctx->stream, d);
    LP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LP_HIP(ctx, hipGetLastError());
    return LP_OPTIMAL;
}

int lp_simplex_crash(lp_simplex_problem* p) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_state_init, 1, 1, 0, s, d, 0.0, 0);
    LP_HIP(ctx, hipMemsetAsync(d.rowused, 0, (size_t)d.m, s));
    for (int t = 0; t < d.m; ++t) {
        hipLaunchKernelGGL(k_crash_select, 1, 1024, 0, s, d, t);
        lp_simplex_launch_update(p);
    }
    hipLaunchKernelGGL(k_crash_finish, 1, 1, 0, s, d);
    // permute rows into basis-position order (through the pristine buffer)
    hipLaunchKernelGGL(k_permute_rows, d.m + 1, 256, 0, s, d, d.T, p->dT0);
    LP_HIP(ctx, hipMemcpyAsync(d.T, p->dT0, p->tableau_bytes, hipMemcpyDeviceToDevice, s));
    SimplexState hs;
    LP_HIP(ctx, hipMemcpyAsync(&hs, d.state, sizeof(hs), hipMemcpyDeviceToHost, s));
    LP_HIP(ctx, hipStreamSynchronize(s));
    LP_HIP(ctx, hipGetLastError());
    return hs.status == kRunning ? LP_OPTIMAL : hs.status;
}

int lp_simplex_run_launch(lp_simplex_problem* p, double eps, int max_iter, lp_simplex_stats* stats) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    hipStream_t s = ctx->stream;
    const size_t shm = 2 * sizeof(double) * (size_t)(d.m + 2) + 16;
    if (shm > 48 * 1024) {   // (m > 3070: opt in to more dynamic LDS than the default limit)
        if (shm > 156 * 1024) LP_FAIL(ctx, LP_BAD_ARG, "simplex: m too large for the selector's LDS");
        LP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_simplex_select),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    }
    int launches = 0;
    LP_HIP(ctx, hipEventRecord(p->ev0, s));
    hipLaunchKernelGGL(k_state_init, 1, 1, 0, s, d, eps, max_iter);
    ++launches;
    int batch = 16;
    int status = kRunning;
    // Kernels turn into no-ops once the state leaves kRunning, so pivots are queued in
    // growing batches and the status word is polled once per batch.
    for (;;) {
        for (int k = 0; k < batch; ++k) {
            hipLaunchKernelGGL(k_simplex_select, 1, 1024, shm, s, d);
            lp_simplex_launch_update(p);
        }
        launches += 2 * batch;
        LP_HIP(ctx, hipMemcpyAsync(p->h_state, d.state, sizeof(SimplexState), hipMemcpyDeviceToHost, s));
        LP_HIP(ctx, hipStreamSynchronize(s));
        status = p->h_state->status;
        if (status != kRunning) break;
        if (batch < 256) batch *= 2;
    }
    LP_HIP(ctx, hipEventRecord(p->ev1, s));
    LP_HIP(ctx, hipEventSynchronize(p->ev1));
    LP_HIP(ctx, hipGetLastError());
    float ms = 0.f;
    LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->last_status = status;
    p->last_algo = LP_SIMPLEX_ALGO_LAUNCH;
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

int lp_simplex_extract_x(lp_simplex_problem* p, double* dx) {
    const SimplexDev& d = p->dev;
    hipStream_t s = p->ctx->stream;
    hipLaunchKernelGGL(k_extract_x, lp_ceil_div(d.n, 256), 256, 0, s, d, dx);
    hipLaunchKernelGGL(k_scatter_x, lp_ceil_div(d.m, 256), 256, 0, s, d, dx);
    return LP_OPTIMAL;
}

int lp_simplex_bench_update(lp_simplex_problem* p, int row, int col, int iters, float* ms_out) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    hipStream_t s = ctx->stream;
    if (row < 0 || row >= d.m || col < 0 || col >= d.n || iters <= 0)
        LP_FAIL(ctx, LP_BAD_ARG, "lp_bench_rank1_update: bad pivot position or iteration count");
    // Stage a valid pivot (eta column + pivot-row copy) with the crash selector's
    // arithmetic, then replay the update kernel.  Values drift (the same eta is
    // re-applied), which is irrelevant for timing; the tableau is restored afterwards.
    if (!p->dscratchT) LP_HIP(ctx, hipMalloc(&p->dscratchT, p->tableau_bytes));   // (micro-benchmarks only)
    LP_HIP(ctx, hipMemcpyAsync(p->dscratchT, d.T, p->tableau_bytes, hipMemcpyDeviceToDevice, s));
    std::vector<double> lcol((size_t)d.m + 1), prow((size_t)d.ld);
    std::vector<double> Th((size_t)(d.m + 1) * d.ld);
    LP_HIP(ctx, hipMemcpyAsync(Th.data(), d.T, p->tableau_bytes, hipMemcpyDeviceToHost, s));
    LP_HIP(ctx, hipStreamSynchronize(s));
    const double ur = Th[(size_t)row * d.ld + col];
    if (ur == 0.0) LP_FAIL(ctx, LP_BAD_ARG, "lp_bench_rank1_update: zero pivot element");
    for (int i = 0; i <= d.m; ++i)
        lcol[i] = (i == row) ? 1.0 : -1e-3 * Th[(size_t)i * d.ld + col] / ur;  // damped: stays finite
    for (int j = 0; j < d.ld; ++j) prow[j] = Th[(size_t)row * d.ld + j];
    SimplexState hs;
    std::memset(&hs, 0, sizeof(hs));
    hs.status = kRunning;
    hs.enter = col;
    hs.leave = row;
    hs.pivot_valid = 1;
    LP_HIP(ctx, hipMemcpyAsync(d.lcol, lcol.data(), sizeof(double) * lcol.size(), hipMemcpyHostToDevice, s));
    LP_HIP(ctx, hipMemcpyAsync(d.prow, prow.data(), sizeof(double) * prow.size(), hipMemcpyHostToDevice, s));
    LP_HIP(ctx, hipMemcpyAsync(d.state, &hs, sizeof(hs), hipMemcpyHostToDevice, s));
    for (int k = 0; k < 3; ++k) lp_simplex_launch_update(p);  // warm-up
    LP_HIP(ctx, hipEventRecord(p->ev0, s));
    for (int k = 0; k < iters; ++k) lp_simplex_launch_update(p);
    LP_HIP(ctx, hipEventRecord(p->ev1, s));
    LP_HIP(ctx, hipEventSynchronize(p->ev1));
    float ms = 0.f;
    LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
    if (ms_out) *ms_out = ms / (float)iters;
    LP_HIP(ctx, hipMemcpyAsync(d.T, p->dscratchT, p->tableau_bytes, hipMemcpyDeviceToDevice, s));
    hs.status = p->last_status;
    hs.iters = p->last_iters;
    hs.pivot_valid = 0;
    LP_HIP(ctx, hipMemcpyAsync(d.state, &hs, sizeof(hs), hipMemcpyHostToDevice, s));
    LP_HIP(ctx, hipStreamSynchronize(s));
    LP_HIP(ctx, hipGetLastError());
    return LP_OPTIMAL;
}
