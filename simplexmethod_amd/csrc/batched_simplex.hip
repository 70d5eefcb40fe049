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

// Uncached-by-the-compiler LDS word accesses for the in-workgroup hand-over.  A `volatile int*` made from an LDS
// pointer is a GENERIC volatile access: the compiler emitted flat_load/flat_store with system scope (sc0 sc1) and
// waited for vmcnt and lgkmcnt — every look at the published pivot number cost hundreds of cycles.
__device__ __forceinline__ int lds_peek(const int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_poke(int* p, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
}

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
    // (M beats the extreme P of the entries in front by more than eps iff it beats every lane's share
    // of them: fl(v + eps) is monotone in v — one ballot instead of a 64-bit key reduction)
    if (__ballot(!lpdev::beats<WANT_MAX>(M, lp, eps)) == 0ULL) {
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
    // (short-cut reductions of device_select.hpp: one 6-step pass over the high words; when a single lane
    // holds the extreme — the common case — its low word and its index come by v_readlane instead of two more passes)
    unsigned long long hit;
    const double M = lpdev::f64_from_key(lpdev::wave_ext_key_n<false, 64>(lpdev::f64_sort_key(lext), &hit));   // (fmin dropped NaNs)
    if (!(M < INFINITY)) return -1;
    int lidx = INT_MAX;
#pragma unroll
    for (int k = K - 1; k >= 0; --k) lidx = (rv[k] == M && lane + 64 * k < m) ? lane + 64 * k : lidx;
    const int jM = ((hit & (hit - 1)) == 0ULL) ? __builtin_amdgcn_readlane(lidx, (int)__builtin_ctzll(hit))
                                               : (int)lpdev::wave_ext_u32<false>((unsigned)lidx);
    double lp = INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) lp = (lane + 64 * k < jM) ? fmin(lp, rv[k]) : lp;
    if (__ballot(!lpdev::beats<false>(M, lp, eps)) == 0ULL) return jM;   // (one ballot instead of reducing P)
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

// Pricing (:152-174) by one wave over up to 64*K entries held in REGISTERS: entry (lane + 64*q) has
// value dv[q] and key kv[q] (the variable index: the scan order is ascending key; INT_MAX = no
// entry).  Same chain semantics as wave_scan_keyed, without re-reading LDS on every pass.  Returns
// the selected slot (-1 if none) and the scan's final value in `best`.
template <bool WANT_MAX, int K>
__device__ __forceinline__ int wave_price_select(const double (&dv)[K], const int (&kv)[K], double eps, double& best) {
    const int lane = threadIdx.x & 63;
    const double sentinel = WANT_MAX ? -INFINITY : INFINITY;
    double lv = sentinel;
    int lkey = INT_MAX, lslot = -1;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const bool ok = kv[q] != INT_MAX;
        if (ok && ((WANT_MAX ? (dv[q] > lv) : (dv[q] < lv)) || (dv[q] == lv && kv[q] < lkey))) {
            lv = dv[q];
            lkey = kv[q];
            lslot = lane + 64 * q;
        }
    }
    // (short-cut reductions: one pass over the high words of the keys; a single lane holding the extreme gives
    // its variable index and slot by v_readlane — two six-step passes fewer than reducing value, then index)
    unsigned long long hits;
    const double M = lpdev::f64_from_key(lpdev::wave_ext_key_n<WANT_MAX, 64>(lpdev::f64_sort_key(lv), &hits));
    best = sentinel;
    if (M == sentinel) return -1;
    int jM, src;
    if ((hits & (hits - 1)) == 0ULL) {
        src = (int)__builtin_ctzll(hits);
        jM = __builtin_amdgcn_readlane(lkey, src);
    } else {
        jM = (int)lpdev::wave_ext_u32<false>((unsigned)(lv == M ? lkey : INT_MAX));
        src = (int)__builtin_ctzll(__ballot(lv == M && lkey == jM));
    }
    if (jM == INT_MAX) return -1;
    const int sM = __builtin_amdgcn_readlane(lslot, src);
    double lp = sentinel;
#pragma unroll
    for (int q = 0; q < K; ++q)
        if (kv[q] < jM) lp = lpdev::ext2<WANT_MAX>(lp, dv[q]);
    // (M beats the extreme P of the entries in front by more than eps iff it beats every lane's share
    // of them: fl(v + eps) is monotone in v — one ballot instead of a 64-bit key reduction)
    if (__ballot(!lpdev::beats<WANT_MAX>(M, lp, eps)) == 0ULL) {
        best = M;
        return sM;
    }
    int sel = -1;   // near-tie: replay the chain jump by jump
    for (;;) {
        const double thr = WANT_MAX ? best + eps : best - eps;
        int ck = INT_MAX, cs = -1;
        double cv = 0.0;
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (kv[q] != INT_MAX && (WANT_MAX ? (dv[q] > thr) : (dv[q] < thr)) && kv[q] < ck) {
                ck = kv[q];
                cv = dv[q];
                cs = lane + 64 * q;
            }
        const int kmin = (int)lpdev::wave_ext_u32<false>((unsigned)ck);
        if (kmin == INT_MAX) break;
        const int src = (int)__builtin_ctzll(__ballot(ck == kmin));
        best = lpdev::wave_bcast_f64(cv, src);
        sel = __builtin_amdgcn_readlane(cs, src);
    }
    return sel;
}

// ---------------------------------------------------------------------------------------------
// Register-resident form of the same kernel (what lp_batched_launch selects when the shape fits):
// the constraint rows of the condensed tableau live in REGISTERS.  Updating thread (column j, row
// group g) owns rows g, g + G, g + 2G, ... of column j for the whole solve — the assignment the
// LDS kernel above already uses, so nothing about the arithmetic changes — and LDS only carries
// what has to cross threads: the entering column and xB (read by wave 0's ratio test), the eta
// column, the pivot row and the reduced-cost row.  Per pivot the LDS kernel moves the whole
// tableau through LDS three times (400 KB at 128 x 256: 7.4 k of its 10.4 k cycles,
// profiles/r02_batched_stamps.txt); here a thread reads RPT eta entries (broadcast) and does RPT
// fused multiply-adds.  Bit-identical results (same operands, same operations per element).
// ---------------------------------------------------------------------------------------------
// Rows are padded to G * RPT: entries past m hold zeros in the eta column (fma(0, p, 0) = 0), so the
// per-row loops carry no predicates (44 live lane masks would spill ~600 SGPRs).
// Staging area of the initial tableau (register form): columns of A are read whole and coalesced (a column is
// m contiguous doubles), CH at a time, into LDS with an odd row stride, and every updating thread takes its rows
// from there — its own 43 reads straight from A were 64 different cache lines per instruction.
__host__ __device__ inline int batched_stage_stride(int m) { return m | 1; }
__host__ __device__ inline int batched_stage_cols(int m) {
    const int ch = 4096 / batched_stage_stride(m);   // <= 32 KB
    return ch < 1 ? 1 : (ch > 32 ? 32 : ch);
}

template <int NT>
__host__ __device__ inline int batched_reg_rpt(int m, int n) {   // rows per updating thread, 0 = does not fit
    const int W = n - m + 1;
    if (W < 2 || W > NT - 64) return 0;
    const int G = (NT - 64) / W;
    return (m + G - 1) / G;
}

// (NT = 512 asks for 4 waves per SIMD = two workgroups per CU: two LPs share a CU, and one LP's
// one-wave scans run under the other's update)
// DREG: wave 0 keeps the reduced-cost row in registers (12 more VGPRs: not in the 512-thread form,
// whose 128-VGPR budget is already short).
// STAMPS (diagnostic build, LP_BATCHED_STAMPS=reg): cycles of every phase of a pivot as seen by wave 0
// (the scanning wave) and wave 1 (an updating wave) of workgroup 0, summed in registers.
template <int NT, int RPT, bool DREG, bool STAMPS = false>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 1) void k_batched_simplex_reg(BatchedDev d) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
#define BR_STAMP(s)                                                          \
    do {                                                                     \
        if (STAMPS) {                                                        \
            const unsigned long long now_ = __builtin_readcyclecounter();    \
            acc[(s)] += now_ - tprev;                                        \
            tprev = now_;                                                    \
        }                                                                    \
    } while (0)
    const int m = d.m, n = d.n, nn = n - m, W = nn + 1;
    const int tid = threadIdx.x;
    const int lp = blockIdx.x;
    const int G = (NT - 64) / W;
    const int mp = G * RPT;                           // padded row count (>= m)
    // ---- LDS carve
    Published* pubs = reinterpret_cast<Published*>(smem);
    double* drow = smem + sizeof(Published) / 8;      // W    : reduced-cost row (entry nn = -objective)
    double* prow = drow + W;                          // W    : pivot row before scaling
    double* lcol = prow + W;                          // mp+1 : eta column (entry mp: reduced-cost row), zeros past m
    double* ucol = lcol + (mp + 1);                   // mp   : entering column
    double* xcol = ucol + mp;                         // mp   : xB
    int* slotvar = reinterpret_cast<int*>(xcol + mp); // nn   : variable held by each slot
    int* basis = slotvar + nn;                        // m    : N by position
    int* posofvar = basis + m;                        // n    : scratch for the initial split

    const double* A = d.A + (size_t)lp * m * n;
    const double* b = d.b + (size_t)lp * m;
    const double* c = d.c + (size_t)lp * n;
    const int* bin = d.basis_in + (size_t)lp * m;

    // ---- initial condensed tableau for the slack identity basis (Symmetrical.cpp:169-188)
    for (int q = tid; q < n; q += NT) posofvar[q] = -1;
    for (int q = tid; q <= mp; q += NT) lcol[q] = 0.0;
    if (tid == 0) pubs->v[2] = -1;   // pivot number of the published entering slot (none yet)
    __syncthreads();
    for (int q = tid; q < m; q += NT) {
        basis[q] = bin[q];
        posofvar[bin[q]] = q;
    }
    __syncthreads();
    if (tid < 64) {  // slots take the non-basic variables in ascending order (wave 0: ballot compaction)
        int s = 0;
        for (int base = 0; base < n; base += 64) {
            const int q = base + tid;
            const bool nb = q < n && posofvar[q] < 0;
            const unsigned long long mask = __ballot(nb);
            if (nb) slotvar[s + __popcll(mask & ((1ULL << tid) - 1ULL))] = q;
            s += __popcll(mask);
        }
    }
    __syncthreads();
    // updating threads: tid >= 64; column j, row group g of G; rows g*RPT .. g*RPT + RPT-1 (a contiguous
    // block: the k-th row is a compile-time offset from one LDS address per array)
    const int wave = tid >> 6, lane = tid & 63;
    const int ut = tid - 64;
    // Wave-aligned assignment when the non-basic columns fill whole waves (nn a multiple of 64, e.g. 128): every
    // updating wave holds ONE row group and 64 consecutive columns, and the right-hand-side column's G owners sit
    // together in a wave of their own.  With columns simply numbered through, the waves that contained the
    // right-hand-side column (43 single-lane stores of xB per pivot) and a row-group boundary (two eta addresses per
    // read, two branches of the pivot-row scaling) took 3.6-4.3 k cycles for an update the others did in 2.0-2.3 k,
    // and the whole workgroup waited for them at the loop's barrier.
    const int CW = nn / 64;
    const bool aligned = (nn % 64) == 0 && G * CW + 1 <= (NT - 64) / 64 && m <= 256;
    int j, g;
    bool upd;
    if (aligned) {
        const int uw = wave - 1;
        if (uw >= 0 && uw < G * CW) {
            g = uw / CW;
            j = (uw % CW) * 64 + lane;
            upd = true;
        } else {   // (wave 0 scans; the wave behind the column waves owns the right-hand side, see xbw below)
            g = G;
            j = nn;
            upd = false;
        }
    } else {
        j = ut >= 0 ? ut % W : 0;
        g = ut >= 0 ? ut / W : G;
        upd = ut >= 0 && g < G;
    }
    // The right-hand-side column (xB) in aligned mode: ONE wave, rows lane*RX .. lane*RX + RX-1 per lane (RX =
    // ceil(m/64) <= 4) — its update is RX fmas per lane and ONE store per row pair, where three lanes with 43 rows each
    // (the column-thread layout) needed 43 single-lane stores and were the slowest wave of every pivot.
    const bool xbw = aligned && wave - 1 == G * CW;   // (a role of its own below: its state must not be live in the column waves' loop)
    const int RX = (m + 63) / 64;
    double t[RPT];
    {
        double* stage = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(posofvar + n) + 15) & ~(uintptr_t)15);
        const int SP = batched_stage_stride(m), CH = batched_stage_cols(m);
        for (int c0 = 0; c0 < W; c0 += CH) {
            for (int e = tid; e < CH * m; e += NT) {   // coalesced along the rows of a column
                const int cc = e / m, i = e - cc * m;
                if (c0 + cc < W) stage[cc * SP + i] = (c0 + cc < nn) ? A[(size_t)slotvar[c0 + cc] * m + i] : b[i];
            }
            __syncthreads();
            if (upd && j >= c0 && j < c0 + CH) {
                const double* col = stage + (j - c0) * SP;
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const int i = g * RPT + k;
                    t[k] = i < m ? col[i] : 0.0;
                }
            }
            __syncthreads();
        }
        if (!upd) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) t[k] = 0.0;
        }
    }
    for (int q = tid; q < mp; q += NT) {
        xcol[q] = q < m ? b[q] : 0.0;
        ucol[q] = 0.0;
    }
    for (int q = tid; q < nn; q += NT) drow[q] = c[slotvar[q]];
    if (tid == 0) drow[nn] = 0.0;
    __syncthreads();

    const double eps = d.eps;
    int iters = 0;
    int status = kRunning;
    int* pub = pubs->v;   // [0] entering slot, [1] leaving position, published by wave 0
    // Pricing over the non-basic slots, keyed by variable index (:152-174): wave 0 alone, for pivot
    // k+1 WHILE the other waves apply pivot k's update to their registers.  For nn <= 256 wave 0 keeps
    // the reduced-cost row and the slots' keys in REGISTERS (entry lane + 64*q), so a pivot's pricing
    // reads no LDS but the pivot row; LDS drow[] then only carries the entering slot's value (the
    // -objective entry of the row is not kept: the objective is evaluated from the vertex on the host).
    const bool dreg = DREG && nn <= 256;
    double dv[4] = {0.0, 0.0, 0.0, 0.0};
    int kv[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
    if (wave == 0 && dreg) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sl = lane + 64 * q;
            if (sl < nn) {
                dv[q] = drow[sl];
                kv[q] = slotvar[sl];
            }
        }
    }
    bool priced_before_loop = true;   // (the first pricing belongs to pivot 0, the others to pivot iters + 1)
    auto price = [&]() {
        double best;
        int se0;
        if (dreg && nn <= 128) {   // (two entries per lane: half of the four-per-lane form's work)
            const double dv2[2] = {dv[0], dv[1]};
            const int kv2[2] = {kv[0], kv[1]};
            se0 = d.maximize ? wave_price_select<true, 2>(dv2, kv2, eps, best)
                             : wave_price_select<false, 2>(dv2, kv2, eps, best);
        } else if (dreg) {
            se0 = d.maximize ? wave_price_select<true, 4>(dv, kv, eps, best)
                             : wave_price_select<false, 4>(dv, kv, eps, best);
        } else {
            auto getd = [&](int sl, double& v, int& k, bool& ok) {
                v = drow[sl];
                k = slotvar[sl];
                ok = true;
            };
            se0 = d.maximize ? wave_scan_keyed<true>(nn, eps, best, getd)
                             : wave_scan_keyed<false>(nn, eps, best, getd);
        }
        const bool optimal = d.maximize ? (best <= eps) : (best >= -eps);
        if (optimal) se0 = -1;
        if (dreg && se0 >= 0) {   // the entering slot's reduced cost, for the eta column's entry of row m
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lane + 64 * q == se0) drow[se0] = dv[q];
        }
        if (lane == 0) {
            // the entering slot, then the pivot number it belongs to: the updating waves wait for the
            // number (LDS writes of one wave land in order) and hand the slot's column over before the
            // next barrier
            lds_poke(pub, se0);
            lds_poke(pub + 2, iters + (priced_before_loop ? 0 : 1));
        }
    };
    // The pivot loop exists twice, once per ROLE, with the same sequence of workgroup barriers (a barrier
    // counts arrivals, not program locations): wave 0 scans (ratio test, reduced costs, pricing) and the
    // other waves hold the tableau rows.  In one common loop the 88 VGPRs of t[] were live through the
    // scanning wave's code as well; with 128 VGPRs per thread (two workgroups per CU) the allocator sent
    // LDS addresses and four rows to SCRATCH, and their global-memory round trips sat inside the ratio
    // test and the rank-1 update (profiles/r02_batched_stamps.txt).
    if (STAMPS) tprev = __builtin_readcyclecounter();
    if (wave == 0) {
        price();
        priced_before_loop = false;
        while (true) {
            __syncthreads();   // (1) registers, xcol and the entering column complete, pub[0] published
            BR_STAMP(7);
            if (iters >= d.max_iter) {  // SimplexSolover.h:429,:450
                status = LP_ITER_LIMIT;
                break;
            }
            const int se = pub[0];
            if (se < 0) {
                status = LP_OPTIMAL;
                break;
            }
            BR_STAMP(0);
            BR_STAMP(1);
            // ---- unbounded test (:179), ratios (:185-186) and the ratio test keyed by basis position
            // (:181-194; +inf entries are never taken); the entering column is in ucol already
            int r;
            if (m <= 128) {   // (two ratios per lane: half of the four-per-lane form's work for m <= 128)
                double rv[2];
                int any_pos = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = lane + 64 * k;
                    const double ui = (i < m) ? ucol[i] : 0.0;
                    rv[k] = (i < m && ui > eps) ? xcol[i] / ui : INFINITY;
                    if (i < m && !(ui <= eps)) any_pos = 1;
                }
                r = wave_ratio_select<2>(rv, m, eps);
                if (!__any(any_pos)) r = -1;
            } else if (m <= 256) {
                double rv[4];
                int any_pos = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = lane + 64 * k;
                    const double ui = (i < m) ? ucol[i] : 0.0;
                    rv[k] = (i < m && ui > eps) ? xcol[i] / ui : INFINITY;
                    if (i < m && !(ui <= eps)) any_pos = 1;
                }
                r = wave_ratio_select<4>(rv, m, eps);
                if (!__any(any_pos)) r = -1;
            } else {
                int any_pos = 0;
                for (int i = lane; i < m; i += 64)
                    if (!(ucol[i] <= eps)) any_pos = 1;
                double theta;
                auto getr = [&](int i, double& v, int& k, bool& ok) {
                    const double ui = ucol[i];
                    v = (ui > eps) ? xcol[i] / ui : INFINITY;
                    k = i;
                    ok = true;
                };
                r = wave_scan_keyed<false>(m, eps, theta, getr);
                if (!__any(any_pos)) r = -1;
            }
            if (lane == 0) pub[1] = r;
            BR_STAMP(2);
            __syncthreads();   // (3) pub[1] published
            BR_STAMP(3);
            if (r < 0) {
                status = LP_UNBOUNDED;
                break;
            }
            // ---- this wave's share of the eta column (:198-204)
            // (one division per entry: F(r,r) = 1/u_r and F(i,r) = -u_i/u_r share the divisor, :201-204)
            const double ur = ucol[r];
            for (int i = tid; i < m; i += NT) lcol[i] = lpdev::div_midrange((i == r) ? 1.0 : -ucol[i], ur);   // (the division's bits, a third of its instructions)
            if (tid == 0) lcol[mp] = lpdev::div_midrange(-drow[se], ur);
            BR_STAMP(4);
            __syncthreads();   // (4) eta column and pivot row complete
            BR_STAMP(5);
            // ---- the reduced-cost row, the basis bookkeeping, then the next pivot's pricing
            const double lm = lcol[mp];
            const int vleave = basis[r];
            if (dreg) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= 2 && nn <= 128) break;   // (uniform: entries 128.. do not exist)
                    const int sl = lane + 64 * q;
                    if (sl < nn) {
                        dv[q] = (sl == se) ? lm : fma(lm, prow[sl], dv[q]);
                        if (sl == se) kv[q] = vleave;   // slot se now holds the leaving variable
                    }
                }
            } else {
                for (int q = lane; q < W; q += 64) drow[q] = (q == se) ? lm : fma(lm, prow[q], drow[q]);
            }
            if (lane == 0) {
                const int ve = slotvar[se];
                slotvar[se] = vleave;
                basis[r] = ve;  // N(leave_pos) = enter, :196
            }
            price();
            BR_STAMP(6);
            ++iters;
        }
    } else if (xbw) {
        // ---- the right-hand-side column (xB) in aligned mode: ONE wave, rows lane*RX .. lane*RX + RX-1 per lane (RX =
        // ceil(m/64) <= 4) — its update is RX fmas per lane and one store per row, where three lanes with 43 rows each
        // (the column-thread layout) needed 43 single-lane stores and were the slowest wave of every pivot.  Same
        // barrier sequence as the other two roles; same operations per element as a column thread's (:198-204).
        double tx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane * RX + q;
            tx[q] = (q < RX && i < m) ? b[i] : 0.0;
        }
        while (true) {
            __syncthreads();   // (1)
            BR_STAMP(7);
            if (iters >= d.max_iter) {
                status = LP_ITER_LIMIT;
                break;
            }
            if (pub[0] < 0) {
                status = LP_OPTIMAL;
                break;
            }
            __syncthreads();   // (3)
            const int r = pub[1];
            if (r < 0) {
                status = LP_UNBOUNDED;
                break;
            }
            // (this wave's threads hold no share of the eta column: tid >= m)
            if (lane == r / RX) {
                const int qr = r - lane * RX;
                prow[nn] = qr == 0 ? tx[0] : qr == 1 ? tx[1] : qr == 2 ? tx[2] : tx[3];
            }
            __syncthreads();   // (4)
            const double pj = prow[nn];
            const double sc = pj * lcol[r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = lane * RX + q;
                if (q < RX && i < m) {
                    tx[q] = (i == r) ? sc : fma(lcol[i], pj, tx[q]);
                    xcol[i] = tx[q];
                }
            }
            BR_STAMP(6);
            ++iters;
        }
    } else {
        // ---- the entering column leaves its owners' registers (:176) as soon as wave 0 has priced the
        // pivot — no phase and no barrier of its own: the updating waves wait for the pivot number
        // behind their rank-1 update (wave 0 prices meanwhile) and the owners store before barrier (1)
        auto hand_over = [&]() {
            while (lds_peek(pub + 2) != iters) __builtin_amdgcn_s_sleep(1);
            const int se_next = lds_peek(pub);
            if (upd && se_next >= 0 && j == se_next) {
                // one lane per row group stores its RPT rows: this is the serial tail of every pivot (the ratio test
                // waits for it), so exactly one ds_write_b64 with an immediate offset per row — the compiler's form
                // copied every pair of rows into temporaries first (4 moves + 1 ds_write2_b64 per pair; a hand-written
                // ds_write2_b64 per pair measured no faster than single stores)
                const unsigned ua = (unsigned)(size_t)(ucol + g * RPT);
#define BR_ST(K) if ((K) < RPT) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ua), "v"(t[(K) < RPT ? (K) : 0]), "n"((K) * 8) : "memory");
                BR_ST(0) BR_ST(1) BR_ST(2) BR_ST(3) BR_ST(4) BR_ST(5) BR_ST(6) BR_ST(7) BR_ST(8) BR_ST(9) BR_ST(10)
                BR_ST(11) BR_ST(12) BR_ST(13) BR_ST(14) BR_ST(15) BR_ST(16) BR_ST(17) BR_ST(18) BR_ST(19) BR_ST(20)
                BR_ST(21) BR_ST(22) BR_ST(23) BR_ST(24) BR_ST(25) BR_ST(26) BR_ST(27) BR_ST(28) BR_ST(29) BR_ST(30)
                BR_ST(31) BR_ST(32) BR_ST(33) BR_ST(34) BR_ST(35) BR_ST(36) BR_ST(37) BR_ST(38) BR_ST(39) BR_ST(40)
                BR_ST(41) BR_ST(42) BR_ST(43)
#undef BR_ST
                static_assert(RPT <= 44, "extend the list above");
            }
        };
        hand_over();
        while (true) {
            __syncthreads();   // (1)
            BR_STAMP(7);
            if (iters >= d.max_iter) {
                status = LP_ITER_LIMIT;
                break;
            }
            const int se = pub[0];
            if (se < 0) {
                status = LP_OPTIMAL;
                break;
            }
            BR_STAMP(0);
            BR_STAMP(2);       // (wave 0 runs the ratio test)
            __syncthreads();   // (3)
            BR_STAMP(3);
            const int r = pub[1];
            if (r < 0) {
                status = LP_UNBOUNDED;
                break;
            }
            // ---- eta column (:198-204) and the pivot row, from the owners' registers
            // (one division per entry: F(r,r) = 1/u_r and F(i,r) = -u_i/u_r share the divisor, :201-204)
            const double ur = ucol[r];
            for (int i = tid; i < m; i += NT) lcol[i] = lpdev::div_midrange((i == r) ? 1.0 : -ucol[i], ur);   // (the division's bits, a third of its instructions)
            const int gr = r / RPT, kr = r % RPT;
            if (upd && g == gr) {
                // (kr is uniform: a switch reaches the one row with a scalar branch tree — 44 predicated
                // stores were a third of this phase)
                double v = 0.0;
                switch (kr) {
#define BR_CASE(K) case K: v = t[(K) < RPT ? (K) : 0]; break;
                    BR_CASE(0) BR_CASE(1) BR_CASE(2) BR_CASE(3) BR_CASE(4) BR_CASE(5) BR_CASE(6) BR_CASE(7)
                    BR_CASE(8) BR_CASE(9) BR_CASE(10) BR_CASE(11) BR_CASE(12) BR_CASE(13) BR_CASE(14) BR_CASE(15)
                    BR_CASE(16) BR_CASE(17) BR_CASE(18) BR_CASE(19) BR_CASE(20) BR_CASE(21) BR_CASE(22) BR_CASE(23)
                    BR_CASE(24) BR_CASE(25) BR_CASE(26) BR_CASE(27) BR_CASE(28) BR_CASE(29) BR_CASE(30) BR_CASE(31)
                    BR_CASE(32) BR_CASE(33) BR_CASE(34) BR_CASE(35) BR_CASE(36) BR_CASE(37) BR_CASE(38) BR_CASE(39)
                    BR_CASE(40) BR_CASE(41) BR_CASE(42) BR_CASE(43)
#undef BR_CASE
                    default: break;
                }
                static_assert(RPT <= 44, "extend the switch above");
                prow[j] = v;
            }
            BR_STAMP(4);
            __syncthreads();   // (4)
            BR_STAMP(5);
            // ---- rank-1 update; slot se receives the leaving column (the eta column itself)
            if (upd) {
                const double pj = prow[j];
                if (j == se) {
#pragma unroll
                    for (int k = 0; k < RPT; ++k) t[k] = lcol[g * RPT + k];
                } else {
#pragma unroll
                    for (int k = 0; k < RPT; ++k) t[k] = fma(lcol[g * RPT + k], pj, t[k]);
                    if (g == gr) {   // the pivot row itself is scaled, not eliminated
#pragma unroll
                        for (int k = 0; k < RPT; ++k)
                            if (k == kr) t[k] = pj * lcol[r];
                    }
                }
                if (j == nn) {   // xB for the next ratio test and the final vertex
#pragma unroll
                    for (int k = 0; k < RPT; ++k) xcol[g * RPT + k] = t[k];
                }
            }
            BR_STAMP(6);
            ++iters;
            hand_over();
            BR_STAMP(1);   // (diagnostic builds: the hand-over's share of the wait in front of barrier (1))
        }
    }
    __syncthreads();
    // ---- outputs: x(N(t)) = xB(t), zeros elsewhere (:131-132); basis; counters
    double* x = d.x + (size_t)lp * n;
    for (int q = tid; q < n; q += NT) x[q] = 0.0;
    __syncthreads();
    for (int q = tid; q < m; q += NT) {
        x[basis[q]] = xcol[q];
        d.basis_out[(size_t)lp * m + q] = basis[q];
    }
    if (tid == 0) {
        d.iters[lp] = iters;
        d.status[lp] = status;
    }
    if (STAMPS && d.stamps && lp == 0 && (tid & 63) == 0) {   // every wave: its phase 6 and its wait at barrier (1)
        d.stamps[32 + 2 * (tid >> 6)] = acc[6];
        d.stamps[33 + 2 * (tid >> 6)] = acc[7];
        d.stamps[48 + (tid >> 6)] = acc[1];   // (column waves: the entering column's hand-over)
    }
    if (STAMPS && d.stamps && lp == 0 && (tid == 0 || tid == 64)) {
        for (int q = 0; q < 8; ++q) d.stamps[(tid ? 16 : 0) + q] = acc[q];
        d.stamps[(tid ? 16 : 0) + 8] = (unsigned long long)iters;
    }
#undef BR_STAMP
}

template <int NT, int RPT, bool DREG, bool STAMPS = false>
static int batched_reg_launch(lp_context* ctx, const BatchedDev& d) {
    const int nn = d.n - d.m, W = nn + 1, G = (NT - 64) / W, mp = G * RPT;
    const size_t dbl = sizeof(Published) / 8 + 2 * (size_t)W + (size_t)(mp + 1) + 2 * (size_t)mp;
    size_t shm = dbl * 8 + sizeof(int) * (size_t)(nn + d.m + d.n);
    shm = (shm + 15) & ~(size_t)15;
    shm += 16 + sizeof(double) * (size_t)batched_stage_cols(d.m) * (size_t)batched_stage_stride(d.m);   // staging of the initial tableau
    LP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_batched_simplex_reg<NT, RPT, DREG, STAMPS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((k_batched_simplex_reg<NT, RPT, DREG, STAMPS>), d.batch, NT, shm, ctx->stream, d);
    return LP_OPTIMAL;
}

size_t lp_batched_lds_bytes(int m, int n, int* pitch_out) {
    const int nn = n - m, W = nn + 1;
    const int pitch = (W & 1) ? W : W + 1;  // odd pitch: conflict-free column reads
    if (pitch_out) *pitch_out = pitch;
    size_t dbl = sizeof(Published) / 8 + (size_t)(m + 1) * pitch + W + (m + 1) + m;
    size_t bytes = dbl * 8 + sizeof(int) * (size_t)(nn + m + n);
    return (bytes + 15) & ~(size_t)15;
}

int lp_batched_launch(lp_context* ctx, const BatchedDev& d) {
    if (d.stamps && d.stamps_reg) {   // diagnostic build of the default form of BASELINE configs[4]'s shape class
        if (batched_reg_rpt<512>(d.m, d.n) >= 1 && batched_reg_rpt<512>(d.m, d.n) <= 44)
            return batched_reg_launch<512, 44, true, true>(ctx, d);
        LP_FAIL(ctx, LP_BAD_ARG, "LP_BATCHED_STAMPS=reg: the shape does not take the 512-thread register form");
    }
    if (!d.stamps) {
        // Register-resident form, smallest row array that holds the shape.  Shapes with many rows per
        // thread take the 512-thread form first: two workgroups (two LPs) then share a CU and one LP's
        // one-wave scans run under the other's update (128 x 256: 3.24 ms against 3.63 ms for the
        // 1024-thread form and 4.02 ms for the LDS form).
        const int rpt = batched_reg_rpt<1024>(d.m, d.n), rpt2 = batched_reg_rpt<512>(d.m, d.n);
        if (rpt >= 1 && rpt <= 4) return batched_reg_launch<1024, 4, true>(ctx, d);
        if (rpt >= 1 && rpt <= 12) return batched_reg_launch<1024, 12, true>(ctx, d);
        if (rpt2 >= 1 && rpt2 <= 44) return batched_reg_launch<512, 44, true>(ctx, d);
        if (rpt >= 1 && rpt <= 20) return batched_reg_launch<1024, 20, true>(ctx, d);
    }
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
