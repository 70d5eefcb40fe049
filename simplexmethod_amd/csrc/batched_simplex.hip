// batched_simplex.hip — many independent LPs of one shape, ONE LP PER WORKGROUP
// (BASELINE.json configs[4]: 4096 LPs, m=128, n=256).
//
// Each workgroup runs the whole of Solver::solveWithBasis
// (/root/reference/src/SimplexSolover.h:408-451) for its LP with the tableau resident in
// LDS for the entire solve: HBM is touched once to load [A | b | c] and once to store the
// vertex.  The tableau is kept in CONDENSED form — only the n-m non-basic columns, xB and
// the reduced-cost row ((m+1) x (n-m+1) doubles; 133 KB at 128x256, which is why one LP
// fills one CU's 160 KB LDS).  A pivot overwrites the entering column's slot with the
// column of the variable that leaves (the eta column itself: fma(l_i, 1, 0) = l_i), so
// every stored element has exactly the bits the full-tableau update
// (oracle/lp_oracle.c: tableau_pivot) would give it.
//
// Slots are in arbitrary variable order after the first pivot, so the order-dependent
// scans of SimplexSolover.h:153-161 / :181-192 are done on (value, key) pairs
// (key = variable index for pricing, basis position for the ratio test):
//   M  = extreme value, jM = smallest key attaining it, P = extreme over keys < jM;
//   M beyond P by more than eps  =>  the sequential scan ends on jM (see
//   device_select.hpp); otherwise the scan is replayed jump by jump.
#include <cfloat>

#include "device_select.hpp"
#include "lp_internal.hpp"
#include "batched_problem.hpp"

namespace {

constexpr int kRunning = -100;

struct Published {   // what wave 0 hands to the rest of the workgroup (first 16 bytes of LDS)
    int v[4];        // [0] entering slot, [1] leaving position
};

// The sequential EPS-hysteresis scan (SimplexSolover.h:153-161 / :164-172 / :181-192) over `count`
// (value, key) entries stored in arbitrary order; entry s is read through
// get(s, value, key, eligible).  Returns the slot of the selected entry (-1 if none) and the scan's
// final value in `best`.  Keys are the original indices: the scan order is ascending key.
//
// Done by ONE wave (the tableaus of this kernel have a few hundred rows/columns at
// most: two or three entries per lane).  No LDS scratch, no barriers; the caller publishes the
// result to the other waves.  Entries are re-read through get() on every pass.
template <bool WANT_MAX, typename Get>
__device__ int wave_scan_keyed(int count, double eps, double& best, Get get) {
    const double sentinel = WANT_MAX ? -INFINITY : INFINITY;
    const int lane = threadIdx.x & 63;
    double lv = sentinel;
    int lkey = INT_MAX, lslot = -1;
    for (int s = lane; s < count; s += 64) {
        double v;
        int k;
        bool ok;
        get(s, v, k, ok);
        if (ok && ((WANT_MAX ? (v > lv) : (v < lv)) || (v == lv && k < lkey))) {
            lv = v;
            lkey = k;
            lslot = s;
        }
    }
    // (reductions on sortable keys: device_select.hpp; lv / lp never hold a NaN — a NaN entry fails
    // every comparison above and is never taken)
    const double M = lpdev::f64_from_key(lpdev::wave_ext_key<WANT_MAX>(lpdev::f64_sort_key(lv)));
    const int jM = (int)lpdev::wave_ext_u32<false>((unsigned)((lv == M && lv != sentinel) ? lkey : INT_MAX));
    best = sentinel;
    if (jM == INT_MAX) return -1;
    const unsigned long long hit = __ballot(lv == M && lkey == jM);
    const int sM = __builtin_amdgcn_readlane(lslot, (int)__builtin_ctzll(hit));
    double lp = sentinel;
    for (int s = lane; s < count; s += 64) {
        double v;
        int k;
        bool ok;
        get(s, v, k, ok);
        if (ok && k < jM) lp = lpdev::ext2<WANT_MAX>(lp, v);
    }
    const double P = lpdev::f64_from_key(lpdev::wave_ext_key<WANT_MAX>(lpdev::f64_sort_key(lp)));
    if (WANT_MAX ? (M > P + eps) : (M < P - eps)) {
        best = M;
        return sM;
    }
    // near-tie: replay the chain jump by jump (each jump: the eligible entry of smallest key beyond the threshold)
    int sel = -1;
    for (;;) {
        const double thr = WANT_MAX ? best + eps : best - eps;
        int ck = INT_MAX, cs = -1;
        double cv = 0.0;
        for (int s = lane; s < count; s += 64) {
            double v;
            int k;
            bool ok;
            get(s, v, k, ok);
            if (ok && (WANT_MAX ? (v > thr) : (v < thr)) && k < ck) {
                ck = k;
                cv = v;
                cs = s;
            }
        }
        const int kmin = (int)lpdev::wave_ext_u32<false>((unsigned)ck);
        if (kmin == INT_MAX) break;
        const int src = (int)__builtin_ctzll(__ballot(ck == kmin));
        best = lpdev::wave_bcast_f64(cv, src);
        sel = __builtin_amdgcn_readlane(cs, src);
    }
    return sel;
}

// The ratio test (:181-194) by one wave with its K = ceil(m / 64) ratios held in registers (one
// division per row instead of one per pass of wave_scan_keyed): keys are the basis positions, i.e.
// the entry index itself.  Returns the leaving position, -1 if no ratio is finite.
template <int K>
__device__ __forceinline__ int wave_ratio_select(const double (&rv)[K], int m, double eps) {
    const int lane = threadIdx.x & 63;
    double lext = INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) lext = fmin(lext, rv[k]);
    const double M = lpdev::f64_from_key(lpdev::wave_ext_key<false>(lpdev::f64_sort_key(lext)));   // (fmin dropped NaNs)
    if (!(M < INFINITY)) return -1;
    int lidx = INT_MAX;
#pragma unroll
    for (int k = K - 1; k >= 0; --k) lidx = (rv[k] == M && lane + 64 * k < m) ? lane + 64 * k : lidx;
    const int jM = (int)lpdev::wave_ext_u32<false>((unsigned)lidx);
    double lp = INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) lp = (lane + 64 * k < jM) ? fmin(lp, rv[k]) : lp;
    const double P = lpdev::f64_from_key(lpdev::wave_ext_key<false>(lpdev::f64_sort_key(lp)));
    if (M < P - eps) return jM;
    // near-tie: replay the chain jump by jump
    double best = INFINITY;
    int sel = -1;
    for (;;) {
        const double thr = best - eps;
        int cand = INT_MAX;
        double cv = INFINITY;
#pragma unroll
        for (int k = K - 1; k >= 0; --k)
            if (rv[k] < thr && lane + 64 * k < m) {
                cand = lane + 64 * k;
                cv = rv[k];
            }
        const int first = (int)lpdev::wave_ext_u32<false>((unsigned)cand);
        if (first == INT_MAX) break;
        best = lpdev::wave_bcast_f64(cv, first & 63);
        sel = first;
    }
    return sel;
}

// STAMPS (diagnostic build): cycles of every phase of a pivot as seen by wave 0 (the scanning wave)
// and by wave 1 (an updating wave) of workgroup 0, summed in registers, stored once at the end.
template <bool STAMPS>
__global__ __launch_bounds__(1024) void k_batched_simplex(BatchedDev d) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = STAMPS ? __builtin_readcyclecounter() : 0;
#define BS_STAMP(s)                                                          \
    do {                                                                     \
        if (STAMPS) {                                                        \
            const unsigned long long now_ = __builtin_readcyclecounter();    \
            acc[(s)] += now_ - tprev;                                        \
            tprev = now_;                                                    \
        }                                                                    \
    } while (0)
    const int m = d.m, n = d.n, nn = n - m, W = nn + 1, pitch = d.pitch;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lp = blockIdx.x;
    // ---- LDS carve
    Published* pubs = reinterpret_cast<Published*>(smem);
    double* T = smem + sizeof(Published) / 8;         // (m+1) x pitch
    double* prow = T + (size_t)(m + 1) * pitch;       // W
    double* lcol = prow + W;                          // m+1
    double* ratio = lcol + (m + 1);                   // m
    int* slotvar = reinterpret_cast<int*>(ratio + m); // nn : variable held by each slot
    int* basis = slotvar + nn;                        // m  : N by position
    int* posofvar = basis + m;                        // n  : scratch for the initial split

    const double* A = d.A + (size_t)lp * m * n;
    const double* b = d.b + (size_t)lp * m;
    const double* c = d.c + (size_t)lp * n;
    const int* bin = d.basis_in + (size_t)lp * m;

    // ---- initial condensed tableau for the slack identity basis (Symmetrical.cpp:169-188)
    for (int j = tid; j < n; j += nt) posofvar[j] = -1;
    __syncthreads();
    for (int t = tid; t < m; t += nt) {
        basis[t] = bin[t];
        posofvar[bin[t]] = t;
    }
    __syncthreads();
    if (tid == 0) {  // slots take the non-basic variables in ascending order
        int s = 0;
        for (int j = 0; j < n; ++j)
            if (posofvar[j] < 0) slotvar[s++] = j;
    }
    __syncthreads();
    for (int idx = tid; idx < nn * m; idx += nt) {
        const int s = idx / m, i = idx - s * m;
        T[(size_t)i * pitch + s] = A[(size_t)slotvar[s] * m + i];
    }
    for (int i = tid; i < m; i += nt) T[(size_t)i * pitch + nn] = b[i];
    for (int s = tid; s < nn; s += nt) T[(size_t)m * pitch + s] = c[slotvar[s]];
    if (tid == 0) T[(size_t)m * pitch + nn] = 0.0;
    __syncthreads();

    const double eps = d.eps;
    int iters = 0;
    int status = kRunning;
    const int wave = tid >> 6, lane = tid & 63;
    int* pub = pubs->v;   // [0] entering slot, [1] leaving position, published by wave 0
    // Pricing over the non-basic slots, keyed by variable index (:152-174): wave 0 alone.  It runs
    // for pivot k+1 WHILE the other waves apply pivot k's update to the constraint rows: wave 0
    // updates the reduced-cost row first, which is all the pricing reads.
    auto price = [&]() {
        double best;
        const double* drow = T + (size_t)m * pitch;
        auto getd = [&](int s, double& v, int& k, bool& ok) {
            v = drow[s];
            k = slotvar[s];
            ok = true;
        };
        int se0 = d.maximize ? wave_scan_keyed<true>(nn, eps, best, getd)
                             : wave_scan_keyed<false>(nn, eps, best, getd);
        const bool optimal = d.maximize ? (best <= eps) : (best >= -eps);
        if (optimal) se0 = -1;
        if (lane == 0) pub[0] = se0;
    };
    // the waves other than wave 0 update the constraint rows: ugroups threads per column
    const int unt = nt - 64, ut = tid - 64;
    const int ugroups = unt / W > 0 ? unt / W : 1;
    if (wave == 0) price();
    if (STAMPS) tprev = __builtin_readcyclecounter();
    while (true) {
        BS_STAMP(0);       // wave 0: reduced-cost row + pricing; others: rank-1 update
        __syncthreads();   // tableau complete, pub[0] published
        BS_STAMP(1);       // barrier wait
        if (iters >= d.max_iter) {  // SimplexSolover.h:429,:450
            status = LP_ITER_LIMIT;
            break;
        }
        const int se = pub[0];
        if (se < 0) {
            status = LP_OPTIMAL;
            break;
        }
        // ---- entering column, unbounded test (:176-179), ratios (:185-186) and the ratio test
        // keyed by basis position (:181-194; +inf entries are never taken): wave 0 alone (computing
        // the ratios with all threads first and scanning an LDS array was measured slower: one more
        // barrier than the divisions cost)
        if (wave == 0) {
            int r;
            if (m <= 256) {
                // up to four rows per lane: ratios computed once, kept in registers
                double rv[4];
                int any_pos = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = lane + 64 * k;
                    const double ui = (i < m) ? T[(size_t)i * pitch + se] : 0.0;
                    rv[k] = (i < m && ui > eps) ? T[(size_t)i * pitch + nn] / ui : INFINITY;
                    if (i < m && !(ui <= eps)) any_pos = 1;
                }
                r = wave_ratio_select<4>(rv, m, eps);
                if (!__any(any_pos)) r = -1;
            } else {
                int any_pos = 0;
                for (int i = lane; i < m; i += 64)
                    if (!(T[(size_t)i * pitch + se] <= eps)) any_pos = 1;
                double theta;
                auto getr = [&](int i, double& v, int& k, bool& ok) {
                    const double ui = T[(size_t)i * pitch + se];
                    v = (ui > eps) ? T[(size_t)i * pitch + nn] / ui : INFINITY;
                    k = i;
                    ok = true;
                };
                r = wave_scan_keyed<false>(m, eps, theta, getr);
                if (!__any(any_pos)) r = -1;
            }
            if (lane == 0) pub[1] = r;
        }
        BS_STAMP(2);       // wave 0: entering column, ratios, ratio test
        __syncthreads();
        BS_STAMP(3);       // barrier wait (the other waves wait here for the ratio test)
        const int r = pub[1];
        if (r < 0) {
            status = LP_UNBOUNDED;
            break;
        }
        // ---- eta column (:198-204) and a copy of the pivot row
        const double ur = T[(size_t)r * pitch + se];
        const double inv = 1.0 / ur;
        for (int j = tid; j < W; j += nt) prow[j] = T[(size_t)r * pitch + j];
        for (int i = tid; i <= m; i += nt)
            lcol[i] = (i == r) ? inv : -T[(size_t)i * pitch + se] / ur;
        BS_STAMP(4);       // eta column + pivot-row copy
        __syncthreads();
        BS_STAMP(5);       // barrier wait
        // ---- rank-1 update of every stored element; slot se receives the leaving column
        if (wave == 0) {
            // the reduced-cost row, the basis bookkeeping, then the next pivot's pricing
            const double lm = lcol[m];
            double* drow = T + (size_t)m * pitch;
            for (int j = lane; j < W; j += 64) drow[j] = (j == se) ? lm : fma(lm, prow[j], drow[j]);
            if (lane == 0) {
                const int ve = slotvar[se];
                slotvar[se] = basis[r];
                basis[r] = ve;  // N(leave_pos) = enter, :196
            }
            price();
        } else {
            // the constraint rows: a thread owns one column (its pivot-row entry stays in a register)
            // and every ugroups-th row; four rows per step so that their LDS reads are in flight
            // together.  Column se receives the leaving column (the eta column itself).
            for (int slot = ut; slot < ugroups * W; slot += unt) {   // (one slot per thread unless W > unt)
                const int j = slot % W, g = slot / W;
                const double pj = prow[j];
                const bool is_se = (j == se);
                for (int i = g; i < m; i += 4 * ugroups) {
                    double l[4], old[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int iu = i + u * ugroups;
                        const int ic = iu < m ? iu : 0;
                        l[u] = lcol[ic];
                        old[u] = T[(size_t)ic * pitch + j];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int iu = i + u * ugroups;
                        const double t = is_se ? l[u] : (iu == r) ? pj * l[u] : fma(l[u], pj, old[u]);
                        if (iu < m) T[(size_t)iu * pitch + j] = t;
                    }
                }
            }
        }
        ++iters;
    }
    __syncthreads();
    // ---- outputs: x(N(t)) = xB(t), zeros elsewhere (:131-132); basis; counters
    double* x = d.x + (size_t)lp * n;
    for (int j = tid; j < n; j += nt) x[j] = 0.0;
    __syncthreads();
    for (int t = tid; t < m; t += nt) {
        x[basis[t]] = T[(size_t)t * pitch + nn];
        d.basis_out[(size_t)lp * m + t] = basis[t];
    }
    if (tid == 0) {
        d.iters[lp] = iters;
        d.status[lp] = status;
    }
    if (STAMPS && d.stamps && lp == 0 && (tid == 0 || tid == 64)) {
        for (int q = 0; q < 6; ++q) d.stamps[(tid ? 8 : 0) + q] = acc[q];
        d.stamps[(tid ? 8 : 0) + 6] = (unsigned long long)iters;
    }
#undef BS_STAMP
}

}  // namespace

size_t lp_batched_lds_bytes(int m, int n, int* pitch_out) {
    const int nn = n - m, W = nn + 1;
    const int pitch = (W & 1) ? W : W + 1;  // odd pitch: conflict-free column reads
    if (pitch_out) *pitch_out = pitch;
    size_t dbl = sizeof(Published) / 8 + (size_t)(m + 1) * pitch + W + (m + 1) + m;
    size_t bytes = dbl * 8 + sizeof(int) * (size_t)(nn + m + n);
    return (bytes + 15) & ~(size_t)15;
}

int lp_batched_launch(lp_context* ctx, const BatchedDev& d) {
    const size_t shm = lp_batched_lds_bytes(d.m, d.n, nullptr);
    if (d.stamps) {
        LP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_batched_simplex<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(k_batched_simplex<true>, d.batch, 1024, shm, ctx->stream, d);
        return LP_OPTIMAL;
    }
    LP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_batched_simplex<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(k_batched_simplex<false>, d.batch, 1024, shm, ctx->stream, d);
    return LP_OPTIMAL;
}
