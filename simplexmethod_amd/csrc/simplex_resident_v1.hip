// simplex_resident_v1.hip — (round-2 form, kept for A/B while the round-3 kernel settles: LP_RESIDENT_V1=1)
// single-LP tableau simplex with the tableau RESIDENT ON CHIP
// (LP_SIMPLEX_ALGO_RESIDENT; what LP_SIMPLEX_ALGO_AUTO selects whenever the shape fits).
//
// Same pivot rules as the other two paths (/root/reference/src/SimplexSolover.h:152-196) and the
// same bits in every tableau element (each element takes the same fma per pivot), but the tableau
// never moves: G co-resident workgroups hold RS_CPT columns each in REGISTERS (thread i = tableau
// row i; the xB column is replicated in every workgroup) for the whole solve, one launch per solve.
// A 512 x 1024 tableau is 32 workgroups x 131 KB of registers — one XCD of the MI355X.
//
// One pivot = ONE all-to-all hop between the workgroups (measured floor 1.1 us on one XCD,
// scripts/ubench_handoff.hip):
//   publish  every workgroup prices its own columns (Dantzig chain summary: extreme M_k, its first
//            index, P_k = extreme of what precedes it), runs the ratio test (:181-192) on ITS
//            candidate column speculatively, and publishes {M_k, P_k, u_r, column, row} plus the
//            candidate column itself;
//   consume  every workgroup reads all G records and replays the reference's scan over them —
//            identical inputs, identical decision everywhere, no leader — then reads the winner's
//            column and applies the rank-1 update to its own registers (its part of the pivot row
//            is local).
//   Near-ties (the hysteresis of :157 / :168 cannot be decided from the summaries) take an exact
//   slow path: the scan is replayed over all n published reduced costs and the owner of the
//   entering column publishes it in a second hop.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, R2): every shared 8 bytes is an
// {epoch tag, 32-bit half of a double} granule written by ONE 16-byte store (two granules) and
// read by sc1 loads that bypass the reader's L1; a reader spins until every tag equals the
// pivot's epoch, so no flags, fences or drains are needed.  Stores are write-through (sc1) unless
// a census at kernel start shows all participants on one XCD, whose shared L2 then serves plain
// stores (2x faster hop; placement is observed, never assumed).  Every spin is bounded: on a
// timeout the solve reports failure and the host reruns it on the look-ahead path.
#include "device_select.hpp"
#include "lp_internal.hpp"
#include "simplex_problem.hpp"

namespace {

constexpr int kRunning = -100;
constexpr int kResidentFailed = -101;   // internal: a hand-off timed out (never leaves this file)
constexpr int RS_CPT = 32;              // tableau columns per workgroup
constexpr int RS_MAX_G = 256;           // one workgroup per CU
constexpr unsigned long long kSpinLimitTicks = 20000000ull;   // 200 ms of the 100 MHz real-time clock

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v16d __attribute__((ext_vector_type(16)));

enum { MODE_PIVOT = 0, MODE_OPTIMAL = 1, MODE_UNBOUNDED = 2, MODE_SLOW = 3, MODE_FAIL = 4 };

struct Ctl {   // decision of the current pivot, written by wave 0, read by everyone after a barrier
    int mode, kst, e, r;
    int oldb, fail, plain, pad;
    double ur, dE;
    double inv, lm;   // 1/u_r and -T[m][e]/u_r, computed once by the winner's workgroup
};

// ---- granules -------------------------------------------------------------------------------
__device__ __forceinline__ v4i g_pack(unsigned ep, double v) {
    const long long b = __double_as_longlong(v);
    v4i g = {(int)ep, (int)(b & 0xFFFFFFFFLL), (int)ep, (int)(b >> 32)};
    return g;
}
__device__ __forceinline__ v4i g_pack2(unsigned ep, int a, int b) {
    v4i g = {(int)ep, a, (int)ep, b};
    return g;
}
__device__ __forceinline__ bool g_fresh(v4i g, unsigned ep) { return g.x == (int)ep && g.z == (int)ep; }
__device__ __forceinline__ double g_f64(v4i g) {
    return __longlong_as_double(((long long)g.w << 32) | (unsigned int)g.y);
}
__device__ __forceinline__ v4i ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);   // sc1: served by L2, never by this CU's L1
}
__device__ __forceinline__ void st16(v4i g, __amdgpu_buffer_rsrc_t r, unsigned off, bool plain) {
    if (plain)
        __builtin_amdgcn_raw_buffer_store_b128(g, r, off, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(g, r, off, 0, 16);  // write-through
}

// Bounded spin bookkeeping: cheap until 256 polls have failed, then the real-time clock decides.
struct Spin {
    unsigned n = 0;
    unsigned long long t0 = 0;
    __device__ __forceinline__ bool expired(__amdgpu_buffer_rsrc_t r, unsigned abort_off) {
        if ((++n & 255u) != 0) return false;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) t0 = now;
        if (now - t0 > kSpinLimitTicks) return true;
        return __builtin_amdgcn_raw_buffer_load_b32(r, abort_off, 0, 16) != 0;
    }
};

// ---- the register-resident slab: RS_CPT = 32 tableau entries of this thread's row, held in two
// 16-double vectors that are LOCAL variables of the kernel (the compiler then indexes them with
// s_set_gpr_idx: one indexed register move for a wave-uniform dynamic column, no select chain and no
// scratch; wrapped in a struct passed by reference the same code went through scratch memory).
#define RS_SLAB_GET(j) (((j) < 16) ? Ta[(j) & 15] : Tb[(j) & 15])
#define RS_SLAB_SET(j, v)            \
    do {                             \
        if ((j) < 16)                \
            Ta[(j) & 15] = (v);      \
        else                         \
            Tb[(j) & 15] = (v);      \
    } while (0)

struct Shared {
    double* prow;    // RS_CPT + 8 : this workgroup's part of the pivot row (+ xB_r at RS_CPT)
    double* ratio;   // mpad       : ratio-test values of the staged candidate (slow-path copy)
    double* u;       // mpad       : the staged candidate column
    lpdev::BlockSelScratch* sc;
    Ctl* ctl;
    int* basis;      // mpad       : N by position (every workgroup keeps its own copy)
};

__host__ __device__ inline size_t resident_lds_bytes(int mpad) {
    return sizeof(double) * ((size_t)RS_CPT + 8 + 2 * (size_t)mpad) + sizeof(lpdev::BlockSelScratch) + 80 +
           sizeof(int) * (size_t)mpad;
}

struct Comm {   // buffer descriptor and byte offsets of the hand-off areas (all inside rd.comm)
    __amdgpu_buffer_rsrc_t r;
    unsigned recA, recB, col, dpub, recS, colS, census, abort;
};

// NaN-free key of a ratio / reduced cost for the reductions (a NaN is never selected by the
// reference's `<` / `>` scans, exactly like the sentinel)
__device__ __forceinline__ double nan_to(double v, double sentinel) { return (v == v) ? v : sentinel; }

// First half of the ratio test (:181-192) on the candidate column `up` of this workgroup, and
// publication of the column: every wave leaves its slice summary in LDS (block_select_stage1).
__device__ __forceinline__ void stage_candidate(double up, double xb, bool rowok, double eps, unsigned ep,
                                                const Comm& cm, unsigned col_off, bool plain, const Shared& sh) {
    const int tid = threadIdx.x;
    if (rowok) st16(g_pack(ep, up), cm.r, col_off + (unsigned)tid * 16u, plain);
    const double ratio = (rowok && up > eps) ? nan_to(xb / up, INFINITY) : INFINITY;   // :185-186
    sh.ratio[tid] = ratio;
    sh.u[tid] = up;
    lpdev::block_select_stage1<false>(ratio, eps, sh.sc);
}

template <int CPT, bool STAMPS>
__global__ __launch_bounds__(512) void k_simplex_resident(SimplexDev d, ResidentDev rd) {
    static_assert(CPT == 32, "the slab holds 32 columns");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (blockIdx.x % (unsigned)rd.stride) return;
    const int k = (int)(blockIdx.x / (unsigned)rd.stride);
    const int G = rd.G;
    if (k >= G) return;
    SimplexState* st = d.state;
    if (st->status != kRunning) return;
    const int tid = threadIdx.x, lane = tid & 63;
    // (wave-uniform by construction; telling the compiler makes every `wave == ...` a scalar branch
    // instead of an exec-mask region)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = d.m, n = d.n, ld = d.ld;
    const int mpad = rd.mpad;
    const int W2 = (mpad > 64) ? 1 : 0;   // the wave that finishes the ratio test while wave 0 polls
    const bool rowok = tid < m;
    const int col0 = k * CPT;
    const bool maximize = d.maximize != 0;
    const double eps = st->eps;
    const int max_iter = st->max_iter;
    const unsigned long long kNegInf = lpdev::f64_sort_key(-INFINITY);

    Shared sh;
    sh.prow = smem;
    sh.ratio = sh.prow + CPT + 8;
    sh.u = sh.ratio + mpad;
    sh.sc = reinterpret_cast<lpdev::BlockSelScratch*>(sh.u + mpad);
    sh.ctl = reinterpret_cast<Ctl*>(sh.sc + 1);
    sh.basis = reinterpret_cast<int*>(reinterpret_cast<char*>(sh.ctl) + 80);

    Comm cm;
    cm.r = __builtin_amdgcn_make_buffer_rsrc(rd.comm, 0, rd.comm_bytes, 0x00020000);
    cm.recA = rd.recA_off; cm.recB = rd.recB_off; cm.col = rd.col_off; cm.dpub = rd.dpub_off;
    cm.recS = rd.recS_off; cm.colS = rd.colS_off; cm.census = rd.census_off; cm.abort = rd.abort_off;
    const unsigned col_stride = (unsigned)mpad * 16u;   // bytes of one published column

    // ---- load this workgroup's slab (thread = row) and the replicated pieces
    v16d Ta, Tb;
    {
        const double* Trow = d.T + (size_t)(rowok ? tid : 0) * ld;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            Ta[j] = (rowok && col0 + j < n) ? Trow[col0 + j] : 0.0;
            Tb[j] = (rowok && col0 + 16 + j < n) ? Trow[col0 + 16 + j] : 0.0;
        }
    }
    double xb = rowok ? d.T[(size_t)tid * ld + n] : 0.0;   // replica of column n
    // reduced costs of this workgroup's columns: lane l of EVERY wave holds column col0 + l
    const int mycol = col0 + lane;
    const bool colok = lane < CPT && mycol < n;
    bool nbl = colok && d.nonbasic[colok ? mycol : 0] != 0;
    double dl = colok ? d.T[(size_t)m * ld + mycol] : 0.0;
    double obj = d.T[(size_t)m * ld + n];
    for (int i = tid; i < m; i += mpad) sh.basis[i] = d.basis[i];
    int it = st->iters;
    int status = (it >= max_iter) ? LP_ITER_LIMIT : kRunning;   // SimplexSolver.h:429,:450

    // ---- placement census: are all participants on one XCD (then plain stores reach the shared L2)?
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        if (tid == 0) {
            st16(g_pack2(1u, (int)xcc, 0), cm.r, cm.census + (unsigned)k * 16u, false);
            sh.ctl->fail = 0;
        }
        if (wave == 0) {
            Spin spin;
            bool same = true, failed = false;
            for (;;) {
                bool ok = true;
                same = true;
                for (int q = lane; q < G; q += 64) {
                    const v4i g = ld16(cm.r, cm.census + (unsigned)q * 16u);
                    ok &= g_fresh(g, 1u);
                    same &= g.y == (int)xcc;
                }
                if (__all(ok)) break;
                if (spin.expired(cm.r, cm.abort)) {
                    failed = true;
                    break;
                }
            }
            const bool all_same = __all(same);   // (a vote inside `if (lane == 0)` would see lane 0 only)
            if (lane == 0) {
                sh.ctl->plain = (all_same && !(rd.pad0 & 1)) ? 1 : 0;
                if (failed) sh.ctl->fail = 1;   // code 1: census
            }
        }
        __syncthreads();
    }
    const bool plain = sh.ctl->plain != 0;
    if (sh.ctl->fail) status = kResidentFailed;

    // Diagnostic build only (STAMPS): cycles of every phase of workgroup 0's wave 0, summed over the
    // solve in registers and stored once at the end (a store per stamp would sit in front of every
    // later vmcnt wait and distort what it measures).
    unsigned long long acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = STAMPS ? __builtin_readcyclecounter() : 0;
#define RS_STAMP(s)                                                          \
    do {                                                                     \
        if (STAMPS) {                                                        \
            const unsigned long long now_ = __builtin_readcyclecounter();    \
            acc[(s)] += now_ - tprev;                                        \
            tprev = now_;                                                    \
        }                                                                    \
    } while (0)

    // One pivot, in the order the dependencies allow:
    //   decide(p)  -> pivot row of p through LDS -> reduced-cost row of p+1 (needs only that row)
    //   -> pricing(p+1) -> the ONE candidate column of p+1 updated ahead of the rest -> publish it,
    //   its record A, first half of its ratio test -> record B
    //   -> rank-1 update of all 32 columns, which runs while the records travel -> decide(p+1).
    unsigned ep = 0, par = 0, slot = 0;
    double pv = 0.0, up = 0.0;
    unsigned long long pkey = 0, mkey = 0, hit = 0;
    int jl = -1;
    // wave 0 issues its first poll of the records right after publishing, so that the round trip
    // runs under the rank-1 update; the values are looked at only when the update is done
    const int R = (G + 63) >> 6;      // records per lane (blocked: lane order = column order)
    const int q0 = lane * R;
    v4i pfa = {0, 0, 0, 0}, pfb = {0, 0, 0, 0};
    bool pf = false;
#define RS_PREFETCH()                                                                     \
    do {                                                                                  \
        if (wave == 0 && q0 < G) {                                                        \
            const unsigned base = cm.recA + (par * (unsigned)G + (unsigned)q0) * 32u;     \
            pfa = ld16(cm.r, base);                                                       \
            pfb = ld16(cm.r, base + 16);                                                  \
            pf = true;                                                                    \
        }                                                                                 \
    } while (0)

    // pricing summary of my columns (:152-174; minimisation scans -d with the same rule).  Every
    // wave computes it from its own replica (lane l = column l): no barrier, no LDS.
#define RS_PRICE()                                                                 \
    do {                                                                           \
        pv = nbl ? nan_to(maximize ? dl : -dl, -INFINITY) : -INFINITY;             \
        pkey = lpdev::f64_sort_key(pv);                                            \
        mkey = lpdev::wave_ext_key<true>(pkey);                                    \
        hit = __ballot(nbl && pkey == mkey && pkey != kNegInf);                    \
        jl = hit ? (int)__builtin_ctzll(hit) : -1;                                 \
    } while (0)

    // publication of my candidate (column values UP for the rows, xB values XBV): record A {M_k,
    // column} first — the consumers' decision needs nothing else from most workgroups — then the
    // column, the first half of the ratio test, and record B {P_k, u_r, leaving row} from wave W2.
#define RS_PUBLISH(UP, XBV)                                                                          \
    do {                                                                                             \
        ++ep;                                                                                        \
        par = ep & 1u;                                                                               \
        slot = par * (unsigned)G + (unsigned)k;                                                      \
        if (wave == 0) {                                                                             \
            if (lane < 2) {                                                                          \
                const v4i g = lane == 0 ? g_pack(ep, hit ? lpdev::f64_from_key(mkey) : -INFINITY)    \
                                        : g_pack2(ep, jl >= 0 ? col0 + jl : -1, 0);                  \
                st16(g, cm.r, cm.recA + slot * 32u + (unsigned)lane * 16u, plain);                   \
            }                                                                                        \
        }                                                                                            \
        if (jl >= 0) stage_candidate((UP), (XBV), rowok, eps, ep, cm, cm.col + slot * col_stride, plain, sh); \
        RS_STAMP(8);                                                                                 \
        __syncthreads();                                                                             \
        RS_STAMP(9);                                                                                \
        if (wave == W2) {                                                                            \
            /* does my maximum beat every reduced cost of mine in front of it by more than eps? */   \
            const double Mk_ = hit ? lpdev::f64_from_key(mkey) : 0.0;                                \
            const int okp = (hit && __ballot(lane < jl && !(Mk_ > pv + eps)) == 0ULL) ? 1 : 0;        \
            int rk = -2;                                                                             \
            double urk = 0.0;                                                                        \
            if (jl >= 0) {                                                                           \
                rk = lpdev::block_select_stage2<false>(sh.ratio, m, eps, sh.sc);                     \
                urk = (rk >= 0) ? sh.u[rk] : 0.0;                                                    \
            }                                                                                        \
            /* the two wave-uniform quotients of the update (F(r,r) = 1/u_r, :204, and the reduced-cost   \
               row's -d_e/u_r) are computed HERE, once, off the consumers' critical path */              \
            const double invk = (rk >= 0) ? 1.0 / urk : 0.0;                                         \
            const double lmk = (rk >= 0) ? -(maximize ? Mk_ : -Mk_) / urk : 0.0;                     \
            if (lane < 5) {                                                                          \
                const v4i g = lane == 0 ? g_pack2(ep, okp, 0)                                        \
                            : lane == 1 ? g_pack(ep, urk)                                            \
                            : lane == 2 ? g_pack2(ep, rk, 0)                                         \
                            : lane == 3 ? g_pack(ep, invk) : g_pack(ep, lmk);                        \
                st16(g, cm.r, cm.recB + slot * 96u + (unsigned)lane * 16u, plain);                   \
            }                                                                                        \
        }                                                                                            \
    } while (0)

    // (The records B are NOT polled ahead: only the winner's is ever needed, it is published last, and
    // 32 lanes x 5 granules of speculative polls per workgroup and pivot slowed the stores they were
    // waiting for — 3.33 -> 3.19 us per pivot without them.)

    if (status == kRunning) {   // prologue: candidate of the initial tableau
        RS_PRICE();
        if (jl >= 0) up = RS_SLAB_GET(jl);
        RS_PUBLISH(up, xb);
        RS_PREFETCH();
    }
    while (status == kRunning) {
        RS_STAMP(0);
        // ================= consume: everyone's records, one decision ============================
        if (wave == 0) {
            double Ml = -INFINITY;
            int el = -1, ql = -1;
            Spin spin;
            bool failed = false;
            for (;;) {
                bool ok = true;
                Ml = -INFINITY; el = -1; ql = -1;
                for (int t = 0; t < R; ++t) {
                    const int q = q0 + t;
                    if (q >= G) break;
                    const unsigned base = cm.recA + (par * (unsigned)G + (unsigned)q) * 32u;
                    v4i a, b;
                    if (t == 0 && pf) {
                        a = pfa;
                        b = pfb;
                    } else {
                        a = ld16(cm.r, base);
                        b = ld16(cm.r, base + 16);
                    }
                    ok &= g_fresh(a, ep) && g_fresh(b, ep);
                    const double Mq = g_f64(a);
                    if (b.y >= 0 && Mq > Ml) {           // strictly greater: ties keep the earlier column
                        Ml = Mq;
                        el = b.y;
                        ql = q;
                    }
                }
                pf = false;
                if (__all(ok)) break;
                if (spin.expired(cm.r, cm.abort)) {
                    failed = true;
                    const unsigned long long bad = __ballot(!ok);
                    if (lane == 0) sh.ctl->pad = bad ? (int)__builtin_ctzll(bad) * R : -1;   // first stale record
                    break;
                }
            }
            RS_STAMP(1);
            const unsigned long long Mlk = lpdev::f64_sort_key(Ml);
            const unsigned long long Mk = lpdev::wave_ext_key<true>(Mlk);
            const double M = lpdev::f64_from_key(Mk);
            const unsigned long long whit = __ballot(ql >= 0 && Mlk == Mk);
            int mode, kst = 0, e = -1, r = -1;
            double ur = 0.0, inv = 0.0, lm = 0.0;
            if (failed) {
                mode = MODE_FAIL;
                if (lane == 0) sh.ctl->fail = 2;   // code 2: record poll
            } else if (!whit || !(M > eps)) {
                mode = MODE_OPTIMAL;                     // the scan's final value is <= M <= eps (:162 / :174)
            } else {
                const int W = (int)__builtin_ctzll(whit);
                kst = __builtin_amdgcn_readlane(ql, W);
                e = __builtin_amdgcn_readlane(el, W);
                // record B of the winner {verdict on its own columns, u_r, leaving row, 1/u_r, -d_e/u_r}: its loads
                // travel while the verdict on the other workgroups' maxima is assembled
                const unsigned baseB = cm.recB + (par * (unsigned)G + (unsigned)kst) * 96u;
                v4i b0, b1, b2, b3, b4;
                b0 = ld16(cm.r, baseB);
                b1 = ld16(cm.r, baseB + 16u);
                b2 = ld16(cm.r, baseB + 32u);
                b3 = ld16(cm.r, baseB + 48u);
                b4 = ld16(cm.r, baseB + 64u);
                // Does M beat everything in front of the winner's first maximum by more than eps?  The
                // lanes before the winner's lane (one ballot), the records of the winner's own lane in
                // front of the winner (only when a lane holds several records, G > 64), and the
                // winner's own verdict on its columns in front of its maximum (record B)
                const unsigned long long near = __ballot(lane < W && !(M > Ml + eps));
                int near_lane = 0;
                if (R > 1 && lane == W) {
                    for (int t = 0; t < R; ++t) {
                        const int q = q0 + t;
                        if (q >= kst) break;
                        const v4i a = ld16(cm.r, cm.recA + (par * (unsigned)G + (unsigned)q) * 32u);
                        const v4i b = ld16(cm.r, cm.recA + (par * (unsigned)G + (unsigned)q) * 32u + 16u);
                        if (b.y >= 0 && !(M > g_f64(a) + eps)) near_lane = 1;
                    }
                }
                near_lane = __builtin_amdgcn_readlane(near_lane, W);
                Spin spinB;
                while (!(g_fresh(b0, ep) && g_fresh(b1, ep) && g_fresh(b2, ep) && g_fresh(b3, ep) && g_fresh(b4, ep))) {
                    if (spinB.expired(cm.r, cm.abort)) {
                        failed = true;
                        break;
                    }
                    b0 = ld16(cm.r, baseB);
                    b1 = ld16(cm.r, baseB + 16u);
                    b2 = ld16(cm.r, baseB + 32u);
                    b3 = ld16(cm.r, baseB + 48u);
                    b4 = ld16(cm.r, baseB + 64u);
                }
                const bool clear = near == 0ULL && near_lane == 0 && b0.y != 0;
                ur = g_f64(b1);
                r = b2.y;
                inv = g_f64(b3);
                lm = g_f64(b4);
                if (failed) {
                    mode = MODE_FAIL;
                    if (lane == 0) sh.ctl->fail = 6;   // code 6: record B of the winner
                } else if (clear) {                      // the scan must end on (M, its first index)
                    mode = (r < 0) ? MODE_UNBOUNDED : MODE_PIVOT;
                } else {
                    mode = MODE_SLOW;
                }
            }
            if (lane == 0) {
                Ctl* c = sh.ctl;
                c->mode = mode; c->kst = kst; c->e = e; c->r = r;
                c->ur = ur; c->dE = M; c->inv = inv; c->lm = lm;
                c->oldb = (mode == MODE_PIVOT) ? sh.basis[r] : -1;
            }
            RS_STAMP(2);
        }
        __syncthreads();
        RS_STAMP(3);
        // the whole decision block in one go (four 16-byte LDS reads in flight together; field by field,
        // each read's round trip was paid in turn behind its readfirstlane)
        Ctl cc = *sh.ctl;
        int mode = cc.mode;
        if (cc.fail) mode = MODE_FAIL;
        bool from_colS = false;
        if (mode == MODE_SLOW) {
            // ---- exact replay of the scan over all n published reduced costs (near-tie)
            __syncthreads();   // everyone has read the decision before wave 0 rewrites it
            if (wave == 0) {
                // every workgroup reaches this branch for the same pivot: only now are all the reduced
                // costs published (a near-tie is rare; storing them with every pivot was 512 bytes per
                // workgroup of traffic in front of the records)
                if (lane < CPT) st16(g_pack(ep, pv), cm.r, cm.dpub + (slot * CPT + (unsigned)lane) * 16u, plain);
                bool failed = false;
                double best;
                auto load = [&](int j, bool& ok) {
                    ok = true;
                    const unsigned off = cm.dpub + (par * (unsigned)G * CPT + (unsigned)j) * 16u;
                    Spin spin;
                    v4i g;
                    for (;;) {
                        g = ld16(cm.r, off);
                        if (g_fresh(g, ep)) break;
                        if (spin.expired(cm.r, cm.abort)) {
                            failed = true;
                            break;
                        }
                    }
                    return g_f64(g);
                };
                const int e = lpdev::wave_chain_select<true>(n, eps, best, load);
                failed = __any(failed);
                if (lane == 0) {
                    Ctl* c = sh.ctl;
                    c->e = e;
                    c->dE = best;
                    c->kst = e >= 0 ? e / CPT : 0;
                    c->mode = failed ? MODE_FAIL : ((e < 0 || !(best > eps)) ? MODE_OPTIMAL : MODE_SLOW);
                    if (failed) c->fail = 3;   // code 3: slow-path reduced costs
                }
            }
            __syncthreads();
            mode = sh.ctl->mode;
            if (mode == MODE_SLOW) {
                const int e = sh.ctl->e;
                const int owner = sh.ctl->kst;
                if (owner == k) {   // second hop: the owner stages the true entering column
                    const int je = __builtin_amdgcn_readfirstlane(e - col0);
                    up = RS_SLAB_GET(je);
                    stage_candidate(up, xb, rowok, eps, ep, cm, cm.colS + par * col_stride, plain, sh);
                    __syncthreads();
                    if (wave == 0) {
                        const int r2 = lpdev::block_select_stage2<false>(sh.ratio, m, eps, sh.sc);
                        const double ur2 = (r2 >= 0) ? sh.u[r2] : 0.0;
                        const double dE2 = maximize ? sh.ctl->dE : -sh.ctl->dE;
                        const double inv2 = (r2 >= 0) ? 1.0 / ur2 : 0.0;
                        const double lm2 = (r2 >= 0) ? -dE2 / ur2 : 0.0;
                        if (lane < 4) {
                            const v4i g = lane == 0 ? g_pack(ep, ur2)
                                        : lane == 1 ? g_pack2(ep, r2, 0)
                                        : lane == 2 ? g_pack(ep, inv2) : g_pack(ep, lm2);
                            st16(g, cm.r, cm.recS + par * 64u + (unsigned)lane * 16u, plain);
                        }
                    }
                }
                __syncthreads();   // everyone has read e / owner before wave 0 rewrites the decision
                if (wave == 0) {
                    Spin spin;
                    bool failed = false;
                    v4i a, b, c2, d2;
                    for (;;) {
                        a = ld16(cm.r, cm.recS + par * 64u);
                        b = ld16(cm.r, cm.recS + par * 64u + 16u);
                        c2 = ld16(cm.r, cm.recS + par * 64u + 32u);
                        d2 = ld16(cm.r, cm.recS + par * 64u + 48u);
                        if (g_fresh(a, ep) && g_fresh(b, ep) && g_fresh(c2, ep) && g_fresh(d2, ep)) break;
                        if (spin.expired(cm.r, cm.abort)) {
                            failed = true;
                            break;
                        }
                    }
                    if (lane == 0) {
                        Ctl* c = sh.ctl;
                        c->ur = g_f64(a);
                        c->r = b.y;
                        c->inv = g_f64(c2);
                        c->lm = g_f64(d2);
                        c->mode = failed ? MODE_FAIL : (b.y < 0 ? MODE_UNBOUNDED : MODE_PIVOT);
                        if (failed) c->fail = 4;   // code 4: slow-path second hop
                        c->oldb = (!failed && b.y >= 0) ? sh.basis[b.y] : -1;
                    }
                }
                __syncthreads();
                cc = *sh.ctl;
                mode = cc.mode;
                from_colS = true;
            }
        }
        if (mode != MODE_PIVOT) {
            status = mode == MODE_OPTIMAL ? LP_OPTIMAL : mode == MODE_UNBOUNDED ? LP_UNBOUNDED : kResidentFailed;
            break;
        }
        const int kst = __builtin_amdgcn_readfirstlane(cc.kst);
        const int e = __builtin_amdgcn_readfirstlane(cc.e);
        const int r = __builtin_amdgcn_readfirstlane(cc.r);
        const int oldb = __builtin_amdgcn_readfirstlane(cc.oldb);
        const double ur = cc.ur;
        // ---- my part of the pivot row (before scaling), broadcast through LDS
        if (tid == r) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                sh.prow[j] = Ta[j];
                sh.prow[16 + j] = Tb[j];
            }
            sh.prow[CPT] = xb;
        }
        // ---- entering column: the winner's published candidate (mine is still in a register; after the
        // slow path it is the owner's second-hop column).  Requested behind the pivot-row barrier.
        const bool want_col = kst != k && rowok;
        const unsigned coff = (from_colS ? cm.colS + par * col_stride
                                         : cm.col + (par * (unsigned)G + (unsigned)kst) * col_stride) +
                              (unsigned)tid * 16u;
        v4i gcol = {0, 0, 0, 0};
        const unsigned ep_col = ep;
        const double inv = cc.inv;   // F(r,r) = 1/u_r, :204
        const double lm = cc.lm;     // F row of the reduced costs: -T[m][e]/u_r
        RS_STAMP(4);
        __syncthreads();
        if (want_col) gcol = ld16(cm.r, coff);   // awaited after the next pricing
        RS_STAMP(5);
        // ---- reduced-cost row (row m of the tableau) after this pivot, replicated per wave
        if (colok) {
            dl = (mycol == e) ? 0.0 : fma(lm, sh.prow[lane], dl);
            if (mycol == e) nbl = false;
            if (mycol == oldb) nbl = true;
        }
        obj = fma(lm, sh.prow[CPT], obj);
        if (tid == 0) {
            sh.basis[r] = e;   // :196 (wave 0 read the old entry before the decision barrier)
            if (k == 0 && it < d.trace_cap) {
                d.trace_enter[it] = e;
                d.trace_leave[it] = r;
            }
        }
        ++it;
        const bool last = it >= max_iter;   // :450: this pivot is applied, no further one is chosen
        if (!last) RS_PRICE();
        RS_STAMP(6);
        // ---- the entering column has arrived by now
        double u = up;
        if (want_col) {
            Spin spin;
            while (!g_fresh(gcol, ep_col)) {
                if (spin.expired(cm.r, cm.abort)) {
                    sh.ctl->fail = 5;   // code 5: entering column (acted on at the next decision barrier)
                    break;
                }
                gcol = ld16(cm.r, coff);
            }
            u = g_f64(gcol);
        }
        const double l = -u / ur;         // F(i,r), :201 (rows other than r)
        const double xbn = (tid == r) ? xb * inv : fma(l, sh.prow[CPT], xb);
        RS_STAMP(7);
        if (!last) {
            // the candidate column of the NEXT pivot, updated ahead of the other 31 (same operation,
            // same operands as the full update below: identical bits)
            double upn = 0.0;
            if (jl >= 0) {
                const double t = RS_SLAB_GET(jl);
                upn = (tid == r) ? t * inv : fma(l, sh.prow[jl], t);
            }
            RS_PUBLISH(upn, xbn);
            up = upn;
        }
        RS_STAMP(10);
        // ---- rank-1 update of my registers (tableau_pivot: F(i,r) = -u_i/u_r, F(r,r) = 1/u_r, :198-204);
        // the records published above are travelling meanwhile
        if (rowok) {
            if (tid == r) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    Ta[j] = Ta[j] * inv;
                    Tb[j] = Tb[j] * inv;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) Ta[j] = fma(l, sh.prow[j], Ta[j]);
            }
        }
        if (!last) RS_PREFETCH();   // first poll of the records A: halfway through the update they are mostly out
        if (rowok) {
            if (tid != r) {
#pragma unroll
                for (int j = 0; j < 16; ++j) Tb[j] = fma(l, sh.prow[16 + j], Tb[j]);
            }
            if (kst == k) {   // column e becomes the unit vector
                const double unit = (tid == r) ? 1.0 : 0.0;
                switch (e - col0) {
#define RS_CASE(J)                  \
    case J: Ta[J] = unit; break;    \
    case 16 + J: Tb[J] = unit; break;
                    RS_CASE(0) RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7)
                    RS_CASE(8) RS_CASE(9) RS_CASE(10) RS_CASE(11) RS_CASE(12) RS_CASE(13) RS_CASE(14) RS_CASE(15)
#undef RS_CASE
                    default: break;
                }
            }
        }
        xb = xbn;
        RS_STAMP(11);
        if (last) status = LP_ITER_LIMIT;
    }
#undef RS_STAMP
#undef RS_PRICE
#undef RS_PUBLISH
#undef RS_PREFETCH

    if (status == kResidentFailed) {   // nothing is written back: the host reruns on another path
        if (tid == 0) {
            // first failing workgroup records where it stopped: {code, workgroup, epoch} (diagnostic)
            if (atomicCAS(reinterpret_cast<int*>(rd.comm + rd.abort_off), 0, 1) == 0) {
                st->enter = sh.ctl->fail * 1000 + k;
                st->leave = (int)ep * 1000 + sh.ctl->pad;
            }
            st->status = kResidentFailed;
        }
        return;
    }
    if (STAMPS && rd.stamps && tid == 0)
        for (int q = 0; q < 16; ++q) rd.stamps[(size_t)k * 16 + q] = acc[q];
    // ---- write the tableau back (row-major (m+1) x ld, what every other entry point reads)
    if (rowok) {
        double* Trow = d.T + (size_t)tid * ld;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (col0 + j < n) Trow[col0 + j] = Ta[j];
            if (col0 + 16 + j < n) Trow[col0 + 16 + j] = Tb[j];
        }
        if (k == 0) Trow[n] = xb;
    }
    if (wave == 0 && colok) {
        d.T[(size_t)m * ld + mycol] = dl;
        d.nonbasic[mycol] = nbl ? 1 : 0;
    }
    __syncthreads();
    if (k == 0) {
        for (int i = tid; i < m; i += mpad) d.basis[i] = sh.basis[i];
        if (tid == 0) {
            d.T[(size_t)m * ld + n] = obj;
            st->iters = it;
            st->status = status;
            st->pivot_valid = 0;
        }
    }
}

__global__ void k_resident_state_init_v1(SimplexDev d, double eps, int max_iter) {
    SimplexState* st = d.state;
    st->status = kRunning;
    st->iters = 0;
    st->max_iter = max_iter;
    st->enter = st->leave = -1;
    st->pivot_valid = 0;
    st->eps = eps;
}

}  // namespace

// Shape check + buffer plan.  Returns 1 and fills *out if the chip-resident path can run (m, n).
int lp_resident_plan_v1(int m, int n, ResidentDev* out) {
    if (m < 1 || m > 512 || n < m) return 0;   // one row per thread, 512-thread workgroups (193 VGPRs)
    const int G = (n + RS_CPT - 1) / RS_CPT;
    if (G > RS_MAX_G) return 0;
    ResidentDev r{};
    r.G = G;
    r.stride = (G <= 32) ? 8 : 1;   // <= 32 workgroups: every 8th block = one XCD under round-robin dispatch
    r.mpad = ((m + 63) / 64) * 64;
    unsigned off = 0;
    auto take = [&](size_t bytes) {
        const unsigned at = off;
        off += (unsigned)((bytes + 255) & ~(size_t)255);
        return at;
    };
    r.abort_off = take(256);
    r.census_off = take((size_t)G * 16);
    r.recA_off = take((size_t)2 * G * 32);
    r.recB_off = take((size_t)2 * G * 96);
    r.recS_off = take(2 * 64);
    r.dpub_off = take((size_t)2 * G * RS_CPT * 16);
    r.colS_off = take((size_t)2 * r.mpad * 16);
    r.col_off = take((size_t)2 * G * r.mpad * 16);
    r.comm_bytes = off;
    *out = r;
    return 1;
}

int lp_simplex_run_resident_v1(lp_simplex_problem* p, double eps, int max_iter, lp_simplex_stats* stats) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    const ResidentDev& rd = p->res;
    hipStream_t s = ctx->stream;
    if (rd.G < 1 || !rd.comm) LP_FAIL(ctx, LP_BAD_ARG, "chip-resident path unavailable for this problem");
    // > 80 KiB of LDS per workgroup: one workgroup per CU, so that G workgroups own G CUs
    size_t shm = resident_lds_bytes(rd.mpad);
    if (shm < 84 * 1024) shm = 84 * 1024;
    const bool stamped = rd.stamps != nullptr;
    const void* kfn = stamped ? reinterpret_cast<const void*>(k_simplex_resident<RS_CPT, true>)
                              : reinterpret_cast<const void*>(k_simplex_resident<RS_CPT, false>);
    LP_HIP(ctx, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    ResidentDev rdv = rd;
    if (getenv("LP_RESIDENT_FORCE_SC1")) rdv.pad0 |= 1;      // diagnostics: write-through stores on one XCD too
    if (getenv("LP_RESIDENT_SPREAD")) rdv.stride = 1;        // diagnostics: participants on all XCDs
    LP_HIP(ctx, hipEventRecord(p->ev0, s));
    hipLaunchKernelGGL(k_resident_state_init_v1, 1, 1, 0, s, d, eps, max_iter);
    LP_HIP(ctx, hipMemsetAsync(rd.comm, 0, rd.comm_bytes, s));   // every tag of every granule: epoch 0
    if (!p->res_ev0) {   // HIP events tight around the one kernel launch (lp_simplex_stats::update_ms)
        LP_HIP(ctx, hipEventCreate(&p->res_ev0));
        LP_HIP(ctx, hipEventCreate(&p->res_ev1));
    }
    LP_HIP(ctx, hipEventRecord(p->res_ev0, s));
    if (stamped)
        hipLaunchKernelGGL((k_simplex_resident<RS_CPT, true>), rdv.G * rdv.stride, rdv.mpad, shm, s, d, rdv);
    else
        hipLaunchKernelGGL((k_simplex_resident<RS_CPT, false>), rdv.G * rdv.stride, rdv.mpad, shm, s, d, rdv);
    LP_HIP(ctx, hipEventRecord(p->res_ev1, s));
    LP_HIP(ctx, hipMemcpyAsync(p->h_state, d.state, sizeof(SimplexState), hipMemcpyDeviceToHost, s));
    LP_HIP(ctx, hipEventRecord(p->ev1, s));
    LP_HIP(ctx, hipEventSynchronize(p->ev1));
    LP_HIP(ctx, hipGetLastError());
    float ms = 0.f, kms = 0.f;
    LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
    LP_HIP(ctx, hipEventElapsedTime(&kms, p->res_ev0, p->res_ev1));
    int status = p->h_state->status;
    if (status == kResidentFailed || status == kRunning) {
        // a hand-off timed out (e.g. the workgroups never became co-resident because another kernel
        // holds the CUs): nothing was written back, the look-ahead / launch path solves it instead
        char msg[200];
        snprintf(msg, sizeof(msg), "chip-resident simplex: hand-off timed out (code*1000+workgroup %d, epoch*1000+record %d), "
                 "re-running on the launch-based path", p->h_state->enter, p->h_state->leave);
        ctx->last_error = msg;
        if (getenv("LP_RESIDENT_STRICT")) return LP_BAD_ARG;   // tests: a fallback must not hide a protocol bug
        if (p->look.J >= 2) {
            int rc = lp_lookahead_prepare(p);
            if (rc) return rc;
            return lp_simplex_run_lookahead(p, eps, max_iter, stats);
        }
        return lp_simplex_run_launch(p, eps, max_iter, stats);
    }
    p->last_status = status;
    p->last_iters = p->h_state->iters;
    p->last_algo = LP_SIMPLEX_ALGO_RESIDENT;
    if (stats) {
        stats->status = status;
        stats->pivots = p->h_state->iters;
        stats->launches = 2;
        stats->solve_ms = ms;
        stats->update_ms = kms;         // the resident kernel alone: every pivot of the solve
        stats->update_launches = 1;
        stats->bytes_per_pivot = 16.0 * (double)d.m * (double)(d.n + 1);
    }
    return status;
}
