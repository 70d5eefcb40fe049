// simplex_resident.hip — single-LP tableau simplex with the tableau RESIDENT ON CHIP
// (LP_SIMPLEX_ALGO_RESIDENT; what LP_SIMPLEX_ALGO_AUTO selects whenever the shape fits).
//
// Same pivot rules as the other paths (/root/reference/src/SimplexSolover.h:152-196) and the same bits
// in every tableau element (each element takes the same fma per pivot), but the tableau never moves:
// G co-resident workgroups hold CPT columns each in REGISTERS (thread i = tableau row i; the xB column
// is replicated in every workgroup) for the whole solve, one launch per solve.
//   m <= 512 : 32 columns per workgroup, up to 512 row threads (a 512 x 1024 tableau = 32 workgroups x
//              131 KB of registers = one XCD of the MI355X)
//   m <= 960 : 16 columns per workgroup, up to 960 row threads
// Every workgroup has ONE MORE WAVE, the communication wave, which owns no rows: it polls the other
// workgroups' records and takes the decision.
//
// One pivot = ONE all-to-all hop between the workgroups, and everything a workgroup has to say leaves it
// THE MOMENT IT IS KNOWN, straight from the wave that computed it (round 4; before, the slices of the ratio
// test met in LDS behind a barrier and the communication wave packed one record per workgroup: 1300 cycles
// of the pivot's critical path):
//   publish  every row wave prices the workgroup's columns from its own replica of the reduced costs
//            (Dantzig chain summary: extreme M_k, its first index, "M_k beats everything of mine in front of
//            it by more than eps", :152-174); wave 0 stores the 16-byte PRICING RECORD {M_k, column, verdict}
//            at once.  Then every row thread stores its entry of the candidate column, updated ahead of the
//            other columns, and every row wave runs the ratio test (:181-192) on ITS 64 rows of that column
//            — speculatively: only one workgroup's candidate wins — and stores a 32-byte SLICE RECORD
//            {smallest ratio, its first row, "beats everything of the slice in front of it by more than
//            eps", the column's entry there}.
//   consume  the communication wave of every workgroup reads all G pricing records (G <= 32: one load,
//            the next sweep already in flight) and replays the reference's scan over them — this is over
//            long before the slices of that pivot are even written — then polls the 8 (or 15) slice
//            records of THE WINNER only and combines them: leaving row r, pivot element u_r.  Identical
//            inputs, identical decision everywhere, no leader.  The decision block goes to the row waves
//            through LDS behind the one barrier on the pivot's critical path; every row thread then reads
//            its entry of the winner's column and applies the rank-1 update to its registers (its part of
//            the pivot row is local).
//   Near-ties (the hysteresis of :157 / :168 / :187 cannot be decided from the summaries) take exact slow
//   paths: the pricing scan is replayed over all n published reduced costs and the owner of the entering
//   column publishes it in a second hop; the ratio scan is replayed by every workgroup over the ratios
//   it recomputes from the winner's column and its own replica of xB (same operands, same bits).
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, R2): every shared 8 bytes is an
// {epoch tag, 32-bit payload} granule written by ONE 16-byte store (two granules) and read by sc1
// loads that bypass the reader's L1; a reader spins until every tag equals the pivot's epoch, so
// no flags, fences or drains are needed.  Two parities of every record slot suffice: a workgroup
// publishes epoch p+2 only after its decision p+1, i.e. after consuming everyone's p+1; the columns have
// THREE slots, because a pricing record leaves before the publisher's row threads have received their
// entries of the previous winner's column.  Stores are write-through (sc1) unless a census at kernel
// start shows all participants on one XCD, whose shared L2 then serves plain stores (3x faster hop;
// placement is observed, never assumed).  Every spin is bounded: on a timeout the failing workgroup
// raises a chip-wide abort word and NOTHING is written back — commit or abort is ONE compare-and-swap
// on that word, so all workgroups write back or none does — and the host reruns the solve on another
// path (lp_simplex_stats::fell_back).
#include "device_select.hpp"
#include "lp_internal.hpp"
#include "simplex_problem.hpp"

#ifndef RS_MARK_A
#define RS_MARK_A (-1)
#define RS_MARK_B (-1)
#endif

namespace {

constexpr int kRunning = -100;
constexpr int kResidentFailed = -101;   // internal: a hand-off timed out (never leaves this file)
constexpr int RS_MAX_G = 256;           // one workgroup per CU
constexpr unsigned long long kSpinLimitTicks = 20000000ull;   // 200 ms of the 100 MHz real-time clock
constexpr int kOutcomeCommit = 1, kOutcomeAbort = 2;   // the abort word: 0 = running

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

enum { MODE_PIVOT = 0, MODE_OPTIMAL = 1, MODE_UNBOUNDED = 2, MODE_SLOW = 3, MODE_FAIL = 4, MODE_RSLOW = 5 };

struct CtlHead {   // decision of one pivot, written by the communication wave, read by everyone after a barrier
    int mode_kst, e, r, pad0;   // mode | winner's workgroup << 8; entering column; leaving row
    double ur, dE;              // pivot element; scan value of the entering column (maximise: d_e, minimise: -d_e)
};
struct Ctl {
    CtlHead h[2];   // by the parity of the pivot's epoch: the next decision is written while a slow wave may still be
                    // reading this one (ONE barrier per pivot; a wave cannot be two decisions behind)
    int fail, plain, stale, pad;
};
static_assert(sizeof(CtlHead) == 32 && sizeof(Ctl) == 80, "a head is read as two 16-byte LDS loads");

// ---- granules -------------------------------------------------------------------------------
// column granules: full 32-bit epoch tags
__device__ __forceinline__ v4i g_pack(unsigned ep, double v) {
    const long long b = __double_as_longlong(v);
    v4i g = {(int)ep, (int)(b & 0xFFFFFFFFLL), (int)ep, (int)(b >> 32)};
    return g;
}
__device__ __forceinline__ bool g_fresh(v4i g, unsigned ep) { return g.x == (int)ep && g.z == (int)ep; }
__device__ __forceinline__ double g_f64(v4i g) {
    return __longlong_as_double(((long long)g.w << 32) | (unsigned int)g.y);
}
__device__ __forceinline__ unsigned long long g_u64(v4i g) {
    return ((unsigned long long)(unsigned)g.w << 32) | (unsigned)g.y;
}
// record granules: the tag word carries a 16-bit tag (never 0: bit 15 set) and 16 bits of payload; the 64
// payload bits proper are a double or the SORTABLE KEY of one (lpdev::f64_sort_key: what the consumer
// reduces, so that neither side converts)
__device__ __forceinline__ unsigned r_tag(unsigned ep) { return 0x8000u | (ep & 0x7FFFu); }
__device__ __forceinline__ v4i r_pack(unsigned ep, unsigned a16, unsigned b16, unsigned long long bits) {
    const unsigned t = r_tag(ep) << 16;
    v4i g = {(int)(t | a16), (int)(unsigned)bits, (int)(t | b16), (int)(unsigned)(bits >> 32)};
    return g;
}
__device__ __forceinline__ bool r_fresh(v4i g, unsigned ep) {
    const unsigned t = r_tag(ep);
    return ((unsigned)g.x >> 16) == t && ((unsigned)g.z >> 16) == t;
}
constexpr unsigned kNoColumn = 0xFFFFu;   // pricing record: nothing eligible among my columns
constexpr unsigned kCommit = 0xFFFEu;     // pricing record: "I have finished; write back"

__device__ __forceinline__ v4i ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);   // sc1: served by L2, never by this CU's L1
}
__device__ __forceinline__ void st16(v4i g, __amdgpu_buffer_rsrc_t r, unsigned off, bool plain) {
    if (plain)
        __builtin_amdgcn_raw_buffer_store_b128(g, r, off, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(g, r, off, 0, 16);  // write-through
}

// Workgroup barrier of the pivot loop: LDS traffic only.  __syncthreads() also waits for the wave's GLOBAL memory
// operations (vmcnt(0): its fence has workgroup scope) — the acknowledgement of every row wave's column store
// (~400 cycles), the communication wave's sweep still in flight (~200) — and nothing in the loop passes data
// between the waves of a workgroup through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Bounded spin bookkeeping: cheap until 256 polls have failed, then the real-time clock decides.
struct Spin {
    unsigned n = 0;
    unsigned long long t0 = 0;
    __device__ __forceinline__ bool expired(__amdgpu_buffer_rsrc_t r, unsigned abort_off) {
        if ((++n & 255u) != 0) return false;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) t0 = now;
        if (now - t0 > kSpinLimitTicks) return true;
        return __builtin_amdgcn_raw_buffer_load_b32(r, abort_off, 0, 16) == kOutcomeAbort;
    }
};

struct Shared {
    double* prow;    // CPT + 8   : !MIRROR: this workgroup's part of the pivot row (+ xB_r at CPT), staged by thread r
    double* mirror;  // NT x (CPT + 2): MIRROR: row-readable copy of the register slab (+ xB), see RS_MIRROR_WRITE
    unsigned* readers;   // 1: row waves that have read the pivot row of the mirror, summed over the solve
    double* ratio;   // NT        : ratio-test values of the entering column (near-tie replays only)
    double* u;       // NT        : that column
    Ctl* ctl;
    SimplexDev* stash;   // the kernel's arguments for the epilogue: re-read from here, the pivot loop is ~100 SGPRs
                         // short and every uniform value kept live across it is reloaded by v_readlane chains
};

// reload of a structure the compiler shall not connect with its original (dword by dword through a volatile view)
template <typename T>
__device__ __forceinline__ T lds_reload(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized");
    T out;
    const volatile int* src = reinterpret_cast<const volatile int*>(p);
    int* dst = reinterpret_cast<int*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];
    return out;
}

// LDS layout: every array at an offset that depends only on the kernel's template parameters (the arrays are
// sized for the instantiation's largest row count NT: the workgroup has the CU to itself anyway), so that every
// LDS address in the pivot loop is an immediate and none of the base pointers occupies an SGPR
__host__ __device__ constexpr size_t resident_lds_bytes(int nt, int cpt) {
    return sizeof(double) * ((size_t)cpt + 8 + 2 * (size_t)nt) + sizeof(Ctl) + 16 + 256 /* stash */ +
           (nt > 512 ? sizeof(double) * (size_t)nt * ((size_t)cpt + 2) : 0) /* mirror */;
}

struct Decision {   // the decision block as the row waves use it
    int mode, kst, e, r;
    double ur, dE;   // (per-lane copies of wave-uniform values: they feed vector arithmetic only)
};
__device__ __forceinline__ Decision ctl_read(const CtlHead* p) {
    const v4i a = *reinterpret_cast<const v4i*>(p);
    const v2d b = *reinterpret_cast<const v2d*>(&p->ur);
    Decision o;
    const int mk = __builtin_amdgcn_readfirstlane(a.x);
    o.mode = mk & 0xFF;
    o.kst = mk >> 8;
    o.e = __builtin_amdgcn_readfirstlane(a.y);
    o.r = __builtin_amdgcn_readfirstlane(a.z);
    o.ur = b[0];
    o.dE = b[1];
    return o;
}

struct Comm {   // buffer descriptor and byte offsets of the hand-off areas (all inside rd.comm)
    __amdgpu_buffer_rsrc_t r;
    unsigned prec, srec, col, dpub, recS, colS, census, abort;
};

// NaN-free key of a ratio / reduced cost for the reductions (a NaN is never selected by the
// reference's `<` / `>` scans, exactly like the sentinel)
__device__ __forceinline__ double nan_to(double v, double sentinel) { return (v == v) ? v : sentinel; }

// The fast path of the ratio test (:181-192) works on approximate quotients q~ = xB_i * (1/u_i refined twice): |q~ - q| <=
// 2 ulp <= 2^-51 |q|.  "q_M beats q_v by more than eps" is concluded only if (q~_M + eps) + S|q~_M| < q~_v - S|q~_v| with
// S = 2^-48 (eight times the bound), and the slice's verdict also needs |q~_M| < 2^15, so that the slack stays below eps
// / 30 and no entry BEHIND the minimum can be more than eps below it either; then the reference's chain must end on the
// first index of the approximate minimum, whatever the last bits of the true quotients are.  Anything else is a "near-
// tie" and takes the exact replay with true divisions (MODE_RSLOW).  The minimum's VALUE is not used downstream: the
// pivot needs the row r and the exact column entry u_r.
constexpr double kRatioSlack = 0x1p-48, kRatioCap = 32768.0;
constexpr int RS_SLICES = 16;   // slice records per workgroup and parity: [0..15] the A granules, [16..31] the B granules

// CPT columns per workgroup; NT = upper bound of the row threads (the launch uses mpad = m rounded up to
// 64 row threads PLUS ONE COMMUNICATION WAVE: blockDim = mpad + 64).
template <int CPT, int NT>
__global__ __launch_bounds__(NT + 64) void k_simplex_resident(SimplexDev d, ResidentDev rd) {
    static_assert(CPT == 32 || CPT == 16, "the slab is two vectors of 16 or 8 doubles");
    constexpr int HALF = CPT / 2;
    constexpr int NWMAX = NT <= 512 ? 8 : 16;   // slices of the ratio test (a power of two >= row waves)
    static_assert(NWMAX <= RS_SLICES, "slice records of a workgroup");
    constexpr int KREPLAY = 4;                  // entries per lane and tile of the near-tie replay (256-row tiles)
    // How the pivot row reaches the waves (measured, round 4): with 8 row waves its owner stages it behind a second
    // barrier (2.45 us per pivot at 512 x 1024 against 2.74 with the mirror, whose 139 KB of LDS stores per pivot
    // compete with the communication wave for the time the slice records travel); with up to 15 row waves that
    // barrier costs more than the mirror (768 x 1536: 4.16 against 3.78 us).
    constexpr bool MIRROR = NT > 512;
    typedef double vslab __attribute__((ext_vector_type(HALF)));
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (blockIdx.x % (unsigned)rd.stride) return;
    const int k = (int)(blockIdx.x / (unsigned)rd.stride);
    const int G = rd.G;
    if (k >= G) return;
    SimplexState* st = d.state;
    if (st->status != kRunning) return;
    const int tid = threadIdx.x, lane = tid & 63;
    // (wave-uniform by construction; telling the compiler makes every role test a scalar branch
    // instead of an exec-mask region)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = d.m, n = d.n, ld = d.ld;
    const int mpad = rd.mpad;
    const int nrw = mpad >> 6;                 // row waves 0 .. nrw-1 (thread i = tableau row i)
    const bool is_comm = wave == nrw;          // the last wave: polls and decides
    const bool rowok = tid < m;
    const int col0 = k * CPT;
    const bool maximize = d.maximize != 0;
    const double eps = st->eps;
    const int max_iter = st->max_iter;
    constexpr unsigned long long kNegInf = 0x000FFFFFFFFFFFFFull;   // f64_sort_key(-inf)
    constexpr unsigned long long kPosInf = 0xFFF0000000000000ull;   // f64_sort_key(+inf)

    Shared sh;
    sh.prow = smem;
    sh.ratio = sh.prow + CPT + 8;
    sh.u = sh.ratio + NT;
    sh.ctl = reinterpret_cast<Ctl*>(sh.u + NT);
    sh.readers = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(sh.ctl) + sizeof(Ctl));
    sh.stash = reinterpret_cast<SimplexDev*>(reinterpret_cast<char*>(sh.ctl) + sizeof(Ctl) + 16);
    static_assert(sizeof(SimplexDev) <= 256, "resident_lds_bytes reserves 256 bytes for the stash");
    if (tid == 0) {
        *sh.stash = d;
        *sh.readers = 0u;
    }
    sh.mirror = reinterpret_cast<double*>(reinterpret_cast<char*>(sh.stash) + 256);
    constexpr int MS = CPT + 2;   // row stride of the mirror in doubles (16-byte rows; 272 / 144 bytes: b128 stores conflict-free)

    Comm cm;
    cm.r = __builtin_amdgcn_make_buffer_rsrc(rd.comm, 0, rd.comm_bytes, 0x00020000);
    cm.prec = rd.prec_off; cm.srec = rd.srec_off; cm.col = rd.col_off; cm.dpub = rd.dpub_off;
    cm.recS = rd.recS_off; cm.colS = rd.colS_off; cm.census = rd.census_off; cm.abort = rd.abort_off;
    const unsigned col_stride = (unsigned)mpad * 16u;   // bytes of one published column

    // ---- the register-resident slab: CPT tableau entries of this thread's row, held in two vectors
    // that are LOCAL variables of the kernel (the compiler then indexes them with s_set_gpr_idx: one
    // indexed register move for a wave-uniform dynamic column, no select chain and no scratch)
    vslab Ta, Tb;
    // (a BRANCH on the wave-uniform half, kept one by the empty asm: as a select the compiler first merges the two
    // halves element by element — 32 v_cndmask and 32 more live registers — and then indexes the result)
#define RS_SLAB_GET(dst, j)                          \
    do {                                             \
        if ((j) < HALF) {                            \
            dst = Ta[(j) & (HALF - 1)];              \
            asm volatile("" : "+v"(dst));            \
        } else {                                     \
            dst = Tb[(j) & (HALF - 1)];              \
            asm volatile("" : "+v"(dst));            \
        }                                            \
    } while (0)
    // The LDS mirror: every row thread keeps a copy of its row (CPT entries + xB) where the OTHER waves can read
    // it.  After a decision every wave takes its pivot-row entries straight from row r of the mirror — no staging
    // by the row's owner, no barrier in front of the pricing.  Written behind the rank-1 update, i.e. while the
    // slice records travel (139 KB of LDS stores per pivot: they have that time and no more).  Thread r may
    // overwrite row r only when every row wave has read it: a count of readers in LDS, which the wave owning r
    // looks at once — it has long been complete — before its stores.
#define RS_MIRROR_WRITE()                                                                  \
    do {                                                                                   \
        double* mr_ = sh.mirror + (size_t)tid * MS;                                        \
        _Pragma("unroll") for (int j = 0; j < HALF; j += 2) {                               \
            const v2d a_ = {Ta[j], Ta[j + 1]}, b_ = {Tb[j], Tb[j + 1]};                    \
            *reinterpret_cast<v2d*>(mr_ + j) = a_;                                         \
            *reinterpret_cast<v2d*>(mr_ + HALF + j) = b_;                                  \
        }                                                                                  \
        mr_[CPT] = xb;                                                                     \
    } while (0)
    {
        const double* Trow = d.T + (size_t)(rowok ? tid : 0) * ld;
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            Ta[j] = (rowok && col0 + j < n) ? Trow[col0 + j] : 0.0;
            Tb[j] = (rowok && col0 + HALF + j < n) ? Trow[col0 + HALF + j] : 0.0;
        }
    }
    double xb = rowok ? d.T[(size_t)tid * ld + n] : 0.0;   // replica of column n
    if (MIRROR && !is_comm) RS_MIRROR_WRITE();   // (visible to the other waves behind the census barrier)
    // reduced costs of this workgroup's columns: lane l of EVERY row wave holds column col0 + l
    const int mycol = col0 + lane;
    const bool colok = lane < CPT && mycol < n;
    bool nbl = colok && d.nonbasic[colok ? mycol : 0] != 0;
    // "my column is basic" (a column may be neither: barred from entering, lp_simplex_phase2_costs).  The column that
    // LEAVES the basis at a pivot needs no look-up of N(r): a basic column is an exact unit vector — set, not computed
    // (below and simplex_launch.hip) — so it is the one basic column whose pivot-row entry is 1.
    if (tid == 0) sh.ctl->pad = 0;
    __syncthreads();
    for (int i = tid; i < m; i += (int)blockDim.x) {
        const int bi = d.basis[i] - col0;
        if (bi >= 0 && bi < CPT) atomicOr(reinterpret_cast<unsigned*>(&sh.ctl->pad), 1u << bi);
    }
    double dl = colok ? d.T[(size_t)m * ld + mycol] : 0.0;
    double obj = d.T[(size_t)m * ld + n];
    // N by position: the communication wave's REGISTERS (lane l, element q: position 64 q + l; every workgroup
    // keeps its own copy), rewritten with a wave-uniform element index behind the decision — off the pivot's path:
    // nobody needs the leaving ENTRY (see basl below)
    constexpr int NBAS = NT / 64 <= 8 ? 8 : 16;
    typedef int vbas __attribute__((ext_vector_type(NBAS)));
    vbas basv = 0;
    if (is_comm) {
#pragma unroll
        for (int q = 0; q < NBAS; ++q) basv[q] = (q * 64 + lane < m) ? d.basis[q * 64 + lane] : -1;
    }
    int it = st->iters;
    int status = (it >= max_iter) ? LP_ITER_LIMIT : kRunning;   // SimplexSolver.h:429,:450

    // ---- placement census: are all participants on one XCD (then plain stores reach the shared L2)?
    if (is_comm) {
        __builtin_amdgcn_s_setprio(3);   // the pivot's critical path runs through this wave: it wins issue arbitration on its SIMD
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        if (lane == 0) {
            v4i g = {1, (int)xcc, 1, 0};
            st16(g, cm.r, cm.census + (unsigned)k * 16u, false);
            sh.ctl->fail = 0;
            sh.ctl->stale = -1;
        }
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
            sh.ctl->plain = (all_same && !(rd.flags & 1)) ? 1 : 0;
            if (failed || (rd.flags & 2)) sh.ctl->fail = 1;   // code 1: census (flag bit 1: injected by the tests)
        }
    }
    __syncthreads();
    bool basl = colok && ((reinterpret_cast<const unsigned*>(&sh.ctl->pad)[0] >> (lane & 31)) & 1u) != 0;
    const bool plain = sh.ctl->plain != 0;
    if (sh.ctl->fail) status = kResidentFailed;

    // Interval timer of diagnostic builds (-DRS_MARK_A=a -DRS_MARK_B=b, scripts/resident_marks.py): cycles from
    // mark a to the next mark b of the communication wave (marks 0-9) or of row wave 0 (marks 10-29), summed
    // over the solve: two clock reads per pivot instead of a stamp per phase, so the rest runs undisturbed.
    unsigned long long mk_t = 0, mk_acc = 0;
#define RS_MARK(id)                                                                    \
    do {                                                                               \
        if ((id) == RS_MARK_A) mk_t = __builtin_readcyclecounter();                    \
        if ((id) == RS_MARK_B && mk_t) {                                               \
            mk_acc += __builtin_readcyclecounter() - mk_t;                             \
            mk_t = 0;                                                                  \
        }                                                                              \
    } while (0)
#define RS_MARK_C(id) do { if (RS_MARK_A >= 0 && is_comm) RS_MARK(id); } while (0)
#define RS_MARK_R(id) do { if (RS_MARK_A >= 0 && wave == 0) RS_MARK(id); } while (0)

    unsigned ep = 0, par = 0, c3 = 0;   // epoch of the records being published / consumed, its parity, ep % 3
    double pv = 0.0;
    unsigned long long pkey = 0, mkey = 0, hit = 0;
    int jl = -1;

    // Pricing summary of my columns (:152-174; minimisation scans -d with the same rule).  Every row
    // wave computes it from its own replica (lane l = column l): no barrier, no LDS.  Wave 0 publishes it
    // at once: {M_k as a sortable key, its first column, "M_k beats every reduced cost of mine in front of it
    // by more than eps"}.
#define RS_PRICE_AND_PUBLISH()                                                     \
    do {                                                                           \
        pv = nbl ? nan_to(maximize ? dl : -dl, -INFINITY) : -INFINITY;             \
        pkey = lpdev::f64_sort_key(pv);                                            \
        mkey = lpdev::wave_ext_key_n<true, CPT>(pkey, &hit);                       \
        if (mkey == kNegInf) hit = 0;                                              \
        jl = hit ? (int)__builtin_ctzll(hit) : -1;                                 \
        if (wave == 0) {                                                           \
            const double Mk_ = lpdev::f64_from_key(mkey);                          \
            const bool okp_ = hit && __ballot(lane < jl && !(Mk_ > pv + eps)) == 0ULL; \
            if (lane == 0)                                                         \
                st16(r_pack(ep, jl >= 0 ? (unsigned)(col0 + jl) : kNoColumn, okp_ ? 1u : 0u, hit ? mkey : kNegInf), \
                     cm.r, cm.prec + (par * (unsigned)G + (unsigned)k) * 16u, plain); \
        }                                                                          \
    } while (0)

    // Candidate (column values UP for the rows, xB values XBV), row waves only: the ratio test (:181-192) on
    // this wave's 64 rows and its slice record {key of the smallest ratio, first row attaining it (+1; 0: no
    // row eligible), verdict | the column's entry there}; then the column's entry for the other workgroups.
#define RS_CANDIDATE(UP, XBV)                                                                        \
    do {                                                                                             \
        if (jl >= 0) {                                                                               \
            /* :185-186 with an APPROXIMATE quotient (reciprocal refined twice, one product: within 2 ulp of  \
               XBV / UP, ~50 cycles instead of the division's ~320): see kRatioSlack */             \
            const double rc_ = lpdev::mid_recip2(UP);                                                \
            const double ratio_ = (rowok && (UP) > eps) ? nan_to((XBV) * rc_, INFINITY) : INFINITY;  \
            const double vlow_ = (ratio_ == INFINITY) ? INFINITY : fma(-kRatioSlack, fabs(ratio_), ratio_); \
            unsigned long long rhit_;                                                                \
            const unsigned long long rk_ = lpdev::wave_ext_key_n<false, 64>(lpdev::f64_sort_key(ratio_), &rhit_); \
            const bool any_ = rk_ != kPosInf;                                                        \
            const int L_ = any_ ? (int)__builtin_ctzll(rhit_) : 0;                                   \
            const double Mv_ = lpdev::f64_from_key(rk_);                                             \
            const double Mhi_ = (Mv_ + eps) + kRatioSlack * fabs(Mv_);                               \
            const unsigned long long near_ = __ballot(lane < L_ && !(Mhi_ < vlow_));                 \
            const bool ok_ = any_ && near_ == 0ULL && fabs(Mv_) < kRatioCap;                         \
            const double uL_ = lpdev::wave_bcast_f64((UP), L_);                                      \
            if (lane < 2) {                                                                          \
                const v4i g_ = lane == 0 ? r_pack(ep, any_ ? (unsigned)(wave * 64 + L_ + 1) : 0u, ok_ ? 1u : 0u, rk_) \
                                         : r_pack(ep, 0u, 0u, (unsigned long long)__double_as_longlong(uL_)); \
                st16(g_, cm.r, cm.srec + (((par * (unsigned)G + (unsigned)k) * 2u + (unsigned)lane) * RS_SLICES + (unsigned)wave) * 16u, plain); \
            }                                                                                        \
            /* the candidate column for the others: needed only after the next decision */           \
            if (rowok) st16(g_pack(ep, (UP)), cm.r, cm.col + (c3 * (unsigned)G + (unsigned)k) * col_stride + (unsigned)tid * 16u, plain); \
        }                                                                                            \
    } while (0)
#define RS_NEXT_EPOCH()                  \
    do {                                 \
        ++ep;                            \
        par = ep & 1u;                   \
        c3 = c3 == 2u ? 0u : c3 + 1u;    \
    } while (0)

    // ======================================================================================================
    // The pivot loop exists TWICE, once per role, with the same sequence of workgroup barriers (a barrier counts
    // arrivals, not program locations): the row waves' copy keeps the slab in place in its registers, and the
    // communication wave's copy never mentions it (one loop for both roles cost the allocator a second copy of
    // the slab and 40 registers in scratch).
    //   decision barrier    the decision block is in LDS
    //   pivot-row barrier   thread r has staged row r of the tableau (and xB_r) in LDS; the communication wave only
    //                       passes through
    // Only the communication wave turns a failure flag into MODE_FAIL, and only in front of a barrier behind which
    // everybody reads the mode: a wave that saw the flag earlier than its neighbours would leave the barrier
    // sequence alone.
    // ======================================================================================================
    if (status == kRunning) RS_NEXT_EPOCH();
    if (is_comm) {
        int mode = MODE_FAIL, kst = 0, e = -1, r = -1;
        double ur = 0.0, M = 0.0;
        unsigned long long Mk;
        while (status == kRunning) {
            bool failed = false;
            kst = 0; e = -1; r = -1; ur = 0.0;
            RS_MARK_C(0);
            // (a row thread whose entering column never came says so before it gets to the decision barrier; one that
            // says so after this read is heard at the next pivot — nothing is written back either way)
            const int fail_seen = sh.ctl->fail;
            // ================= consume: everyone's pricing records -> the winner =================
            if (G <= 32) {
                // one sweep reads all the pricing records (lane q: record q); the next sweep is in flight while
                // this one is tested
                const bool live = lane < G;
                const unsigned off = cm.prec + (par * (unsigned)G + (unsigned)(live ? lane : 0)) * 16u;
                Spin spin;
                v4i a = ld16(cm.r, off);
                for (;;) {
                    const v4i a1 = ld16(cm.r, off);
                    const bool ok = !live || r_fresh(a, ep);
                    if (__all(ok)) break;
                    if (spin.expired(cm.r, cm.abort)) {
                        failed = true;
                        const unsigned long long bad = __ballot(!ok);
                        if (lane == 0) sh.ctl->stale = bad ? (int)(__builtin_ctzll(bad) & 31) : -1;   // first stale record
                        break;
                    }
                    a = a1;
                }
                RS_MARK_C(1);
                const unsigned eq = (unsigned)a.x & 0xFFFFu;
                const bool cand = live && eq < kCommit;
                const unsigned long long kq = cand ? g_u64(a) : (lane < 32 ? kNegInf : 0ULL);
                const double Mq = lpdev::f64_from_key(kq);
                unsigned long long whit;
                Mk = lpdev::wave_ext_key_n<true, 32>(kq, &whit);
                M = lpdev::f64_from_key(Mk);
                if (failed) {
                    mode = MODE_FAIL;
                    if (lane == 0) sh.ctl->fail = 2;   // code 2: record poll
                } else if (Mk == kNegInf || !(M > eps)) {
                    mode = MODE_OPTIMAL;                     // the scan's final value is <= M <= eps (:162 / :174)
                } else {
                    const int W = (int)__builtin_ctzll(whit);   // first lane = first workgroup attaining M
                    // Does M beat everything in front of the winner by more than eps?  (one ballot: fl(v + eps)
                    // is monotone in v) — and the winner's own verdict on its columns
                    const unsigned long long near = __ballot(lane < W && !(M > Mq + eps));
                    kst = W;
                    e = (int)__builtin_amdgcn_readlane((int)eq, W);
                    const unsigned okw = (unsigned)__builtin_amdgcn_readlane(a.z, W) & 0xFFFFu;
                    mode = (near == 0ULL && okw != 0) ? MODE_PIVOT : MODE_SLOW;
                }
            } else {
                // more than 32 workgroups: R consecutive records per lane (lane order = column order)
                const int R = (G + 63) >> 6;
                const int q0 = lane * R;
                unsigned long long kl = kNegInf, kfront = kNegInf;
                unsigned el = kNoColumn, okl = 0;
                int ql = -1;
                Spin spin;
                for (;;) {
                    bool ok = true;
                    kl = kNegInf; kfront = kNegInf; el = kNoColumn; okl = 0; ql = -1;
                    for (int t = 0; t < R; ++t) {
                        const int q = q0 + t;
                        if (q >= G) break;
                        const v4i a = ld16(cm.r, cm.prec + (par * (unsigned)G + (unsigned)q) * 16u);
                        ok &= r_fresh(a, ep);
                        const unsigned long long kq = g_u64(a);
                        const unsigned eq = (unsigned)a.x & 0xFFFFu;
                        if (eq < kCommit && kq > kl) {           // strictly greater: ties keep the earlier column
                            kfront = kl;                         // the extreme of this lane's records in front of it
                            kl = kq;
                            el = eq;
                            okl = (unsigned)a.z & 0xFFFFu;
                            ql = q;
                        }
                    }
                    if (__all(ok)) break;
                    if (spin.expired(cm.r, cm.abort)) {
                        failed = true;
                        const unsigned long long bad = __ballot(!ok);
                        if (lane == 0) sh.ctl->stale = bad ? (int)__builtin_ctzll(bad) * R : -1;   // first stale record
                        break;
                    }
                }
                RS_MARK_C(1);
                unsigned long long whit;
                Mk = lpdev::wave_ext_key_n<true, 64>(kl, &whit);
                M = lpdev::f64_from_key(Mk);
                if (failed) {
                    mode = MODE_FAIL;
                    if (lane == 0) sh.ctl->fail = 2;   // code 2: record poll
                } else if (Mk == kNegInf || !(M > eps)) {
                    mode = MODE_OPTIMAL;
                } else {
                    const int W = (int)__builtin_ctzll(whit);   // first lane = first workgroup attaining M
                    // the lanes before the winner's lane (a lane's extreme stands for all its records), the
                    // records of the winner's own lane in front of the winner, and the winner's own verdict
                    const unsigned long long near =
                        __ballot((lane < W && !(M > lpdev::f64_from_key(kl) + eps)) ||
                                 (lane == W && !(M > lpdev::f64_from_key(kfront) + eps)));
                    kst = __builtin_amdgcn_readlane(ql, W);
                    e = (int)__builtin_amdgcn_readlane((int)el, W);
                    const unsigned okw = (unsigned)__builtin_amdgcn_readlane((int)okl, W);
                    mode = (near == 0ULL && okw != 0) ? MODE_PIVOT : MODE_SLOW;
                }
            }
            RS_MARK_C(2);
            if (mode == MODE_PIVOT) {
                // ---- the winner's slice records: lane w its A granule {key of the smallest ratio, row + 1,
                // verdict}, lane 32 + w its B granule {the column's entry at that row}
                const int w = lane & 31;
                const bool slive = w < nrw;
                const unsigned soff = cm.srec + (((par * (unsigned)G + (unsigned)kst) * 2u + (unsigned)(lane >> 5)) * RS_SLICES +
                                                 (unsigned)(slive ? w : 0)) * 16u;
                Spin spin;
                v4i b = ld16(cm.r, soff);
                for (;;) {
                    const v4i b1 = ld16(cm.r, soff);
                    const bool ok = !slive || r_fresh(b, ep);
                    if (__all(ok)) break;
                    if (spin.expired(cm.r, cm.abort)) {
                        failed = true;
                        if (lane == 0) sh.ctl->stale = 1000 + kst;
                        break;
                    }
                    b = b1;
                }
                RS_MARK_C(3);
                const bool isA = lane < 32 && slive;
                const unsigned long long Ml = isA ? g_u64(b) : kPosInf;
                const int Jl = isA ? (int)((unsigned)b.x & 0xFFFFu) - 1 : -1;
                const int okl = (int)((unsigned)b.z & 0xFFFFu);
                const double Ul = g_f64(b);
                const double Mv = lpdev::f64_from_key(Ml);   // this slice's smallest (approximate) ratio, +inf: none
                const double vlow = (Mv == INFINITY) ? INFINITY : fma(-kRatioSlack, fabs(Mv), Mv);
                unsigned long long h2;
                const unsigned long long M2 = lpdev::wave_ext_key_n<false, NWMAX>(Ml, &h2);
                if (failed) {
                    mode = MODE_FAIL;
                    if (lane == 0) sh.ctl->fail = 8;   // code 8: slice poll
                } else if (M2 == kPosInf) {
                    mode = MODE_UNBOUNDED;   // no row of the entering column is eligible (:179)
                } else {
                    const int W2 = (int)__builtin_ctzll(h2);   // first slice attaining the minimum (row order)
                    const int okW = __builtin_amdgcn_readlane(okl, W2);
                    // (the ratios are approximate: "beats by more than eps" with the slack on both sides, see RS_CANDIDATE)
                    const double M2v = lpdev::f64_from_key(M2);
                    const double Mhi = (M2v + eps) + kRatioSlack * fabs(M2v);
                    const unsigned long long near2 = __ballot(lane < W2 && !(Mhi < vlow));
                    r = __builtin_amdgcn_readlane(Jl, W2);
                    ur = lpdev::wave_bcast_f64(Ul, W2 + 32);
                    if (!(okW && near2 == 0ULL)) mode = MODE_RSLOW;   // near-tie: every workgroup replays the chain exactly
                }
            }
            if (fail_seen) mode = MODE_FAIL;
            if (lane == 0) {
                CtlHead* c = &sh.ctl->h[par];
                const v4i head = {mode | (kst << 8), e, r, 0};
                *reinterpret_cast<v4i*>(c) = head;
                c->ur = ur; c->dE = M;
            }
            RS_MARK_C(4);
            lds_barrier();   // ---- the decision barrier
            RS_MARK_C(5);
            if (__builtin_expect(mode != MODE_PIVOT, 0)) {   // ---- everything but the plain pivot: out of the hot path's way
            if (mode == MODE_RSLOW) {
                // near-tie in the ratio test: the rows recompute the ratios of the entering column; exact replay (:181-192)
                lds_barrier();   // R1
                double best2;
                auto load2 = [&](int j, bool& ok) {
                    ok = true;
                    return sh.ratio[j];
                };
                r = lpdev::wave_chain_select<false, KREPLAY>(m, eps, best2, load2);
                mode = sh.ctl->fail ? MODE_FAIL : (r < 0 ? MODE_UNBOUNDED : MODE_PIVOT);
                if (lane == 0) {
                    CtlHead* c = &sh.ctl->h[par];
                    const v4i head = {mode | (kst << 8), e, r, 0};
                    *reinterpret_cast<v4i*>(c) = head;
                    c->ur = mode == MODE_PIVOT ? sh.u[r] : 0.0;
                }
                lds_barrier();   // R2
            }
            if (mode == MODE_SLOW) {
                // near-tie in the pricing: exact replay of the scan over all n published reduced costs
                lds_barrier();   // S0: everyone has read the decision
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
                e = lpdev::wave_chain_select<true, 4>(n, eps, best, load);
                failed = __any(failed);
                kst = e >= 0 ? e / CPT : 0;
                M = best;
                mode = failed ? MODE_FAIL : ((e < 0 || !(best > eps)) ? MODE_OPTIMAL : MODE_SLOW);
                if (lane == 0) {
                    CtlHead* c = &sh.ctl->h[par];
                    const v4i head = {mode | (kst << 8), e, -1, 0};
                    *reinterpret_cast<v4i*>(c) = head;
                    c->dE = best;
                    if (failed) sh.ctl->fail = 3;   // code 3: slow-path reduced costs
                }
                lds_barrier();   // S1
                if (mode == MODE_SLOW) {
                    if (kst == k) {   // the owner of the entering column stages it (second hop) and runs its ratio test
                        lds_barrier();   // O1: the rows' ratios are in LDS
                        double best2;
                        auto load2 = [&](int j, bool& ok) {
                            ok = true;
                            return sh.ratio[j];
                        };
                        const int r2 = lpdev::wave_chain_select<false, KREPLAY>(m, eps, best2, load2);
                        if (lane == 0) sh.ctl->h[par].r = r2;
                        lds_barrier();   // O2: thread r2 publishes {r2, u_r2}
                    }
                    Spin spin;
                    v4i a;
                    for (;;) {
                        a = ld16(cm.r, cm.recS + par * 16u);
                        if (r_fresh(a, ep)) break;
                        if (spin.expired(cm.r, cm.abort)) {
                            failed = true;
                            break;
                        }
                    }
                    r = (int)((unsigned)a.x & 0xFFFFu) - 1;
                    mode = failed ? MODE_FAIL : (r < 0 ? MODE_UNBOUNDED : MODE_PIVOT);
                    if (lane == 0) {
                        CtlHead* c = &sh.ctl->h[par];
                        const v4i head = {mode | (kst << 8), e, r, 0};
                        *reinterpret_cast<v4i*>(c) = head;
                        c->ur = g_f64(a);
                        if (failed) sh.ctl->fail = 4;   // code 4: slow-path second hop
                    }
                    lds_barrier();   // S3
                }
            }
            if (mode != MODE_PIVOT) {
                status = mode == MODE_OPTIMAL ? LP_OPTIMAL : mode == MODE_UNBOUNDED ? LP_UNBOUNDED : kResidentFailed;
                break;
            }
            }
            ++it;
            RS_NEXT_EPOCH();
            if (!MIRROR) lds_barrier();   // ---- the pivot-row barrier (the rows' business)
            if (lane == (r & 63)) basv[(r >> 6) & (NBAS - 1)] = e;   // N(leave_pos) = enter, :196
            if (k == 0 && lane == 0) {   // (workgroup 0 only: the trace pointers come from the stash — three LDS reads behind
                                         // the barrier's memory clobber — not from SGPRs held all along)
                const SimplexDev* sd = sh.stash;
                if (it - 1 < sd->trace_cap) {
                    sd->trace_enter[it - 1] = e;
                    sd->trace_leave[it - 1] = r;
                }
            }
            if (it >= max_iter) status = LP_ITER_LIMIT;   // :450: that pivot is applied, no further one is chosen
        }
        // (this wave owns no rows: fresh values, so that the registers its copy of the slab occupied are not kept
        // alive — idle — through the row waves' loop for the sake of the common epilogue)
        Ta = 0.0;
        Tb = 0.0;
        xb = 0.0;
    } else {
        double up = 0.0;
        if (status == kRunning) {   // prologue: candidate of the initial tableau
            RS_PRICE_AND_PUBLISH();
            if (jl >= 0) RS_SLAB_GET(up, jl);
            RS_CANDIDATE(up, xb);
        }
        while (status == kRunning) {
            RS_MARK_R(20);
            lds_barrier();   // ---- the decision barrier
            RS_MARK_R(10);
            Decision cc = ctl_read(&sh.ctl->h[par]);
            bool have_col = false;
            v4i gcol = {0, 0, 0, 0};
            const unsigned ep_col = ep;
            // the winner's column (mine is still in a register); awaited after the next pricing
            unsigned coff = cm.col + (c3 * (unsigned)G + (unsigned)cc.kst) * col_stride + (unsigned)tid * 16u;
            if (__builtin_expect(cc.mode == MODE_PIVOT, 1)) {
                if (cc.kst != k && rowok) gcol = ld16(cm.r, coff);
            } else {   // ---- everything but the plain pivot: out of the hot path's way
            if (cc.mode == MODE_RSLOW) {
                if (cc.kst != k && rowok) gcol = ld16(cm.r, coff);
                // ---- near-tie in the ratio test: the ratios of the entering column, recomputed here from the winner's
                // published column and my replica of xB — the winner's own operands, so every workgroup replays the
                // same chain
                double u_ = up;
                if (cc.kst != k && rowok) {
                    Spin spin;
                    while (!g_fresh(gcol, ep_col)) {
                        if (spin.expired(cm.r, cm.abort)) {
                            sh.ctl->fail = 5;
                            break;
                        }
                        gcol = ld16(cm.r, coff);
                    }
                    u_ = g_f64(gcol);
                }
                have_col = true;
                sh.ratio[tid] = (rowok && u_ > eps) ? nan_to(xb / u_, INFINITY) : INFINITY;
                sh.u[tid] = u_;
                lds_barrier();   // R1
                lds_barrier();   // R2
                cc = ctl_read(&sh.ctl->h[par]);
            }
            if (cc.mode == MODE_SLOW) {
                // ---- near-tie in the pricing: every workgroup reaches this branch for the same pivot; only now are all
                // the reduced costs published (storing them with every pivot was 512 bytes per workgroup of traffic in
                // front of the records)
                lds_barrier();   // S0
                if (wave == 0 && lane < CPT)
                    st16(g_pack(ep, pv), cm.r, cm.dpub + ((par * (unsigned)G + (unsigned)k) * CPT + (unsigned)lane) * 16u, plain);
                lds_barrier();   // S1
                cc = ctl_read(&sh.ctl->h[par]);
                if (cc.mode == MODE_SLOW) {
                    if (cc.kst == k) {   // second hop: the owner stages the true entering column
                        const int je = __builtin_amdgcn_readfirstlane(cc.e - col0);
                        RS_SLAB_GET(up, je);
                        if (rowok) st16(g_pack(ep, up), cm.r, cm.colS + par * col_stride + (unsigned)tid * 16u, plain);
                        sh.ratio[tid] = (rowok && up > eps) ? nan_to(xb / up, INFINITY) : INFINITY;
                        lds_barrier();   // O1
                        lds_barrier();   // O2
                        const int r2 = sh.ctl->h[par].r;
                        if (tid == (r2 >= 0 ? r2 : 0))
                            st16(r_pack(ep, (unsigned)(r2 + 1), 0u, (unsigned long long)__double_as_longlong(r2 >= 0 ? up : 0.0)), cm.r,
                                 cm.recS + par * 16u, plain);
                    }
                    lds_barrier();   // S3
                    cc = ctl_read(&sh.ctl->h[par]);
                    coff = cm.colS + par * col_stride + (unsigned)tid * 16u;
                    if (cc.mode == MODE_PIVOT && cc.kst != k && rowok) gcol = ld16(cm.r, coff);
                }
            }
            if (cc.mode != MODE_PIVOT) {
                status = cc.mode == MODE_OPTIMAL ? LP_OPTIMAL : cc.mode == MODE_UNBOUNDED ? LP_UNBOUNDED : kResidentFailed;
                break;
            }
            }
            const int kst = cc.kst, e = cc.e, r = cc.r;
            // ---- the pivot row, lane j its entry j (lane CPT: xB_r); the quotients by u_r are computed meanwhile.
            //   MIRROR: ONE LDS read per wave, straight from row r of the mirror (no barrier);
            //   else  : thread r stages its registers — 17 stores of one lane — behind a second barrier.
            double pl = 0.0;
            if (MIRROR) {
                pl = sh.mirror[(size_t)r * MS + (lane <= CPT ? lane : 0)];
            } else if (tid == r) {
#pragma unroll
                for (int j = 0; j < HALF; j += 2) {
                    const v2d a_ = {Ta[j], Ta[j + 1]}, b_ = {Tb[j], Tb[j + 1]};
                    *reinterpret_cast<v2d*>(sh.prow + j) = a_;
                    *reinterpret_cast<v2d*>(sh.prow + HALF + j) = b_;
                }
                sh.prow[CPT] = xb;
            }
            RS_MARK_R(11);
            ++it;
            const bool last = it >= max_iter;   // :450: this pivot is applied, no further one is chosen
            RS_NEXT_EPOCH();                    // what is published from here on belongs to the next decision
            // (quotients by u_r: its reciprocal refined once for all three of them — lpdev::mid_div, straight-line —
            // and the plain divisions on a cold path if an operand leaves the range in which that is the division's own
            // instruction sequence)
            const bool fastq = __all(lpdev::mid_range(cc.ur) && lpdev::mid_range(cc.dE));
            const double r2u = lpdev::mid_recip2(cc.ur);
            const double numm = -(maximize ? cc.dE : -cc.dE);
            double inv = fma(fma(-cc.ur, r2u, 1.0), r2u, r2u);   // (mid_div with numerator 1: its quotient is r2u itself)
            double lm = lpdev::mid_div(numm, cc.ur, r2u);
            if (__builtin_expect(!fastq, 0)) {
                inv = 1.0 / cc.ur;
                lm = numm / cc.ur;
            }
            if (!MIRROR) {
                lds_barrier();   // ---- the pivot-row barrier
                pl = sh.prow[lane <= CPT ? lane : 0];
            }
            RS_MARK_R(16);
            // reduced-cost row (row m of the tableau) after this pivot, replicated per wave; the wave-uniform entries
            // needed below (xB_r and the candidate's) come out of pl's registers by v_readlane
            const double pxb = lpdev::wave_bcast_f64(pl, CPT);
            if (MIRROR && lane == 0)   // "my wave has read row r"
                __hip_atomic_fetch_add(sh.readers, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (colok) {
                dl = (mycol == e) ? 0.0 : fma(lm, pl, dl);
                if (basl && pl == 1.0) {   // N(r) leaves the basis (:196)
                    basl = false;
                    nbl = true;
                }
                if (mycol == e) {
                    nbl = false;
                    basl = true;
                }
            }
            obj = fma(lm, pxb, obj);
            RS_MARK_R(12);
            if (!last) RS_PRICE_AND_PUBLISH();
            RS_MARK_R(13);
            // ---- the entering column has arrived by now: F(i,r), :201 (rows other than r)
            double l;
            if (kst != k && rowok) {
                if (!have_col) {
                    Spin spin;
                    while (!g_fresh(gcol, ep_col)) {
                        if (spin.expired(cm.r, cm.abort)) {
                            sh.ctl->fail = 5;   // code 5: entering column (acted on at the next decision / the commit)
                            break;
                        }
                        gcol = ld16(cm.r, coff);
                    }
                }
                l = -g_f64(gcol);
            } else {
                l = -up;
            }
            {
                const double lq = lpdev::mid_div(l, cc.ur, r2u);
                const bool lok = lpdev::mid_range_or_zero(l);
                if (__builtin_expect(!(fastq && __all(lok)), 0))
                    l = l / cc.ur;
                else
                    l = lq;
            }
            const double xbn = (tid == r) ? xb * inv : fma(l, pxb, xb);
            RS_MARK_R(14);
            double upn = 0.0;
            if (!last) {
                if (jl >= 0) {
                    // the candidate column of the NEXT pivot, updated ahead of the others (same operation,
                    // same operands as the full update below: identical bits)
                    double t;
                    RS_SLAB_GET(t, jl);
                    upn = (tid == r) ? t * inv : fma(l, lpdev::wave_bcast_f64(pl, jl), t);
                }
                RS_CANDIDATE(upn, xbn);
            }
            RS_MARK_R(15);
            // ---- rank-1 update of my registers (tableau_pivot: F(i,r) = -u_i/u_r, F(r,r) = 1/u_r, :198-204).
            // ONE fma per element for every row, in place: row r's own entries ARE the pivot row, so its
            // T_rj * (1/u_r) is fma(1/u_r, prow_j, -0.0) — the same product, and adding -0.0 changes no bit of it.
            if (tid == r) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) {
                    Ta[j] = -0.0;
                    Tb[j] = -0.0;
                }
            }
            const double le = (tid == r) ? inv : l;
            if (MIRROR) {   // the pivot row out of pl's registers (v_readlane): no LDS, no barrier
#pragma unroll
                for (int j = 0; j < HALF; ++j) Ta[j] = fma(le, lpdev::wave_bcast_f64(pl, j), Ta[j]);
#pragma unroll
                for (int j = 0; j < HALF; ++j) Tb[j] = fma(le, lpdev::wave_bcast_f64(pl, HALF + j), Tb[j]);
            } else {        // ... from its staged copy (broadcast reads)
#pragma unroll
                for (int j = 0; j < HALF; ++j) Ta[j] = fma(le, sh.prow[j], Ta[j]);
#pragma unroll
                for (int j = 0; j < HALF; ++j) Tb[j] = fma(le, sh.prow[HALF + j], Tb[j]);
            }
            if (kst == k) {   // column e becomes the unit vector (selects on a wave-uniform condition, in place: an indexed
                              // register move made the compiler copy the slab to and fro; one workgroup per pivot gets here)
                const double unit = (tid == r) ? 1.0 : 0.0;
                const int je = __builtin_amdgcn_readfirstlane(e - col0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) {
                    Ta[j] = (je == j) ? unit : Ta[j];
                    Tb[j] = (je == HALF + j) ? unit : Tb[j];
                }
            }
            xb = xbn;
            up = upn;
            if (MIRROR) {
                if (wave == (r >> 6)) {   // my wave owns row r: every row wave must have read it from the mirror (long since)
                    const unsigned want = (unsigned)nrw * (unsigned)it;
                    Spin spin;
                    while ((int)(__hip_atomic_load(sh.readers, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - want) < 0) {
                        if (spin.expired(cm.r, cm.abort)) {
                            sh.ctl->fail = 9;   // code 9: mirror readers
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                RS_MIRROR_WRITE();
            }
            RS_MARK_R(17);
            if (last) status = LP_ITER_LIMIT;
        }
    }
#undef RS_PRICE_AND_PUBLISH
#undef RS_CANDIDATE
#undef RS_NEXT_EPOCH

    // ================= commit: the tableau goes home only if EVERY workgroup got here =================
    // One more hop in the pricing-record stream (the next epoch): "I have finished".  A workgroup that has seen
    // everybody's PROPOSES commit, one that failed anywhere — also one whose entering column timed out on the
    // very last pivot — or whose wait for the others expired proposes abort: a compare-and-swap on the abort
    // word, and the first proposal is everyone's outcome (a commit proposal exists only if every workgroup
    // finished, so a failed workgroup can never lose to one).  All write back or none does.
    __syncthreads();   // (a row thread's late fail = 5 is visible to everyone below)
    if (status != kResidentFailed && sh.ctl->fail) status = kResidentFailed;
    __syncthreads();
    if (is_comm) {
        bool saw_all = false;
        if (status != kResidentFailed) {
            ++ep;
            par = ep & 1u;
            if (lane == 0) st16(r_pack(ep, kCommit, 0u, 0ULL), cm.r, cm.prec + (par * (unsigned)G + (unsigned)k) * 16u, plain);
            Spin spin;
            saw_all = true;
            for (;;) {
                bool ok = true;
                for (int q = lane; q < G; q += 64) {
                    const v4i a = ld16(cm.r, cm.prec + (par * (unsigned)G + (unsigned)q) * 16u);
                    ok &= r_fresh(a, ep) && ((unsigned)a.x & 0xFFFFu) == kCommit;
                }
                if (__all(ok)) break;
                if (spin.expired(cm.r, cm.abort)) {
                    saw_all = false;
                    if (lane == 0 && sh.ctl->fail == 0) sh.ctl->fail = 7;   // code 7: commit
                    break;
                }
            }
        }
        if (lane == 0) {
            const int proposal = saw_all ? kOutcomeCommit : kOutcomeAbort;
            const int old = atomicCAS(reinterpret_cast<int*>(rd.comm + rd.abort_off), 0, proposal);
            const int outcome = old ? old : proposal;
            if (old == 0 && proposal == kOutcomeAbort) {
                // the first failing workgroup records where it stopped: {code, workgroup, epoch} (diagnostic)
                SimplexState* stf = lds_reload(sh.stash).state;
                stf->enter = sh.ctl->fail * 1000 + k;
                stf->leave = (int)ep * 1000 + sh.ctl->stale;
            }
            sh.ctl->pad = outcome;
        }
    }
    __syncthreads();
    if (sh.ctl->pad != kOutcomeCommit) {   // nothing is written back: the host reruns on another path
        if (tid == 0) lds_reload(sh.stash).state->status = kResidentFailed;
        return;
    }
    if (RS_MARK_A >= 0 && lane == 0 && (RS_MARK_A < 10 ? is_comm : wave == 0)) {
        v4i g = {(int)(unsigned)mk_acc, (int)(unsigned)(mk_acc >> 32), it, 0};
        st16(g, cm.r, cm.census + (unsigned)k * 16u, false);
    }
    // ---- write the tableau back (row-major (m+1) x ld, what every other entry point reads)
    const SimplexDev de = lds_reload(sh.stash);
    if (tid < de.m) {
        double* Trow = de.T + (size_t)tid * de.ld;
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            if (col0 + j < de.n) Trow[col0 + j] = Ta[j];
            if (col0 + HALF + j < de.n) Trow[col0 + HALF + j] = Tb[j];
        }
        if (k == 0) Trow[de.n] = xb;
    }
    if (wave == 0 && lane < CPT && col0 + lane < de.n) {
        de.T[(size_t)de.m * de.ld + col0 + lane] = dl;
        de.nonbasic[col0 + lane] = nbl ? 1 : 0;
    }
    if (k == 0) {
        if (is_comm) {
#pragma unroll
            for (int q = 0; q < NBAS; ++q)
                if (q * 64 + lane < de.m) de.basis[q * 64 + lane] = basv[q];
        }
        if (tid == 0) {
            de.T[(size_t)de.m * de.ld + de.n] = obj;
            de.state->iters = it;
            de.state->status = status;
            de.state->pivot_valid = 0;
        }
    }
#undef RS_SLAB_GET
#undef RS_MIRROR_WRITE
}

__global__ void k_resident_state_init(SimplexDev d, double eps, int max_iter) {
    SimplexState* st = d.state;
    st->status = kRunning;
    st->iters = 0;
    st->max_iter = max_iter;
    st->enter = st->leave = -1;
    st->pivot_valid = 0;
    st->eps = eps;
}

template <int CPT, int NT>
void launch_resident(const SimplexDev& d, const ResidentDev& rd, size_t shm, hipStream_t s, hipError_t* attr_err) {
    *attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_simplex_resident<CPT, NT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (*attr_err == hipSuccess)
        hipLaunchKernelGGL((k_simplex_resident<CPT, NT>), rd.G * rd.stride, rd.mpad + 64, shm, s, d, rd);
}

// ---- self-test of lpdev::mid_div against the compiler's division (lp_debug_division) ----
__global__ void k_debug_division(const double* __restrict__ num, const double* __restrict__ den, int n,
                                 double* __restrict__ fast, double* __restrict__ plain) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = num[i], y = den[i];
    // the kernel's own rule: the fast sequence inside the range, the plain division outside
    fast[i] = (lpdev::mid_range(y) && lpdev::mid_range_or_zero(x)) ? lpdev::mid_div(x, y, lpdev::mid_recip2(y)) : x / y;
    asm volatile("" : "+v"(x), "+v"(y));   // (two separate computations, whatever the optimiser thinks of them)
    plain[i] = x / y;
}

}  // namespace

// Shape check + buffer plan.  Returns 1 and fills *out if the chip-resident path can run (m, n).
int lp_resident_plan(int m, int n, ResidentDev* out) {
    if (m < 1 || m > 960 || n < m) return 0;    // one row per thread, plus the communication wave: <= 1024 threads
    const int cpt = m <= 512 ? 32 : 16;         // 512 row threads x 32 columns (<= 168 VGPRs at 9 waves) or 960 x 16 (<= 128)
    const int G = (n + cpt - 1) / cpt;
    if (G > RS_MAX_G) return 0;
    if (n > 0xFFF0) return 0;                   // the record carries the column in 16 bits
    ResidentDev r{};
    r.G = G;
    r.cpt = cpt;
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
    r.prec_off = take((size_t)2 * G * 16);
    r.srec_off = take((size_t)2 * G * 2 * RS_SLICES * 16);
    r.recS_off = take(2 * 16);
    r.dpub_off = take((size_t)2 * G * cpt * 16);
    r.colS_off = take((size_t)2 * r.mpad * 16);
    r.col_off = take((size_t)3 * G * r.mpad * 16);
    r.comm_bytes = off;
    *out = r;
    return 1;
}

int lp_simplex_run_resident(lp_simplex_problem* p, double eps, int max_iter, lp_simplex_stats* stats) {
    lp_context* ctx = p->ctx;
    const SimplexDev& d = p->dev;
    const ResidentDev& rd = p->res;
    hipStream_t s = ctx->stream;
    if (rd.G < 1 || !rd.comm) LP_FAIL(ctx, LP_BAD_ARG, "chip-resident path unavailable for this problem");
    // > 80 KiB of LDS per workgroup: one workgroup per CU, so that G workgroups own G CUs
    size_t shm = resident_lds_bytes(rd.cpt == 32 ? 512 : 960, rd.cpt);   // (the instantiation's NT: see resident_lds_bytes)
    if (shm < 84 * 1024) shm = 84 * 1024;
    ResidentDev rdv = rd;
    if (getenv("LP_RESIDENT_FORCE_SC1")) rdv.flags |= 1;     // tests: write-through stores on one XCD too
    if (getenv("LP_RESIDENT_SPREAD")) rdv.stride = 1;        // tests: participants on all XCDs
    if (getenv("LP_RESIDENT_INJECT_FAILURE")) rdv.flags |= 2;   // tests: the census reports a failure
    LP_HIP(ctx, hipEventRecord(p->ev0, s));
    hipLaunchKernelGGL(k_resident_state_init, 1, 1, 0, s, d, eps, max_iter);
    LP_HIP(ctx, hipMemsetAsync(rd.comm, 0, rd.comm_bytes, s));   // every tag of every granule: epoch 0
    if (!p->res_ev0) {   // HIP events tight around the one kernel launch (lp_simplex_stats::update_ms)
        LP_HIP(ctx, hipEventCreate(&p->res_ev0));
        LP_HIP(ctx, hipEventCreate(&p->res_ev1));
    }
    LP_HIP(ctx, hipEventRecord(p->res_ev0, s));
    hipError_t attr_err = hipSuccess;
    if (rd.cpt == 32)
        launch_resident<32, 512>(d, rdv, shm, s, &attr_err);
    else
        launch_resident<16, 960>(d, rdv, shm, s, &attr_err);
    LP_HIP(ctx, attr_err);
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
        if (getenv("LP_RESIDENT_DEBUG")) {   // the pricing-record area as the failure left it
            std::vector<int> rec((size_t)2 * rd.G * 4);
            if (hipMemcpy(rec.data(), rd.comm + rd.prec_off, rec.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                for (int par = 0; par < 2; ++par)
                    for (int q = 0; q < rd.G; ++q) {
                        const int* w = &rec[((size_t)par * rd.G + q) * 4];
                        fprintf(stderr, "[resident debug] parity %d pricing record %3d: %08x %08x %08x %08x\n", par, q, w[0], w[1], w[2], w[3]);
                    }
            }
        }
        if (getenv("LP_RESIDENT_STRICT")) return LP_BAD_ARG;   // tests: a fallback must not hide a protocol bug
        int rc;
        if (p->look.J >= 2) {
            rc = lp_lookahead_prepare(p);
            if (rc) return rc;
            rc = lp_simplex_run_lookahead(p, eps, max_iter, stats);
        } else {
            rc = lp_simplex_run_launch(p, eps, max_iter, stats);
        }
        if (rc >= 0 && stats) stats->solve_ms += ms;   // the caller waited for the timed-out launch too
        return rc;
    }
#if RS_MARK_A >= 0   // diagnostic builds: the interval RS_MARK_A -> RS_MARK_B per workgroup
    {
        std::vector<int> cg((size_t)rd.G * 4);
        if (hipMemcpy(cg.data(), rd.comm + rd.census_off, cg.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            double sum = 0, mx = 0;
            for (int q = 0; q < rd.G; ++q) {
                const double c = (double)(((unsigned long long)(unsigned)cg[q * 4 + 1] << 32) | (unsigned)cg[q * 4]) /
                                 (double)(cg[q * 4 + 2] > 0 ? cg[q * 4 + 2] : 1);
                sum += c;
                mx = c > mx ? c : mx;
            }
            fprintf(stderr, "[resident marks] %d -> %d: workgroup 0 %.0f, mean %.0f, max %.0f cycles per pivot\n", RS_MARK_A, RS_MARK_B,
                    (double)(((unsigned long long)(unsigned)cg[1] << 32) | (unsigned)cg[0]) / (double)(cg[2] > 0 ? cg[2] : 1),
                    sum / rd.G, mx);
        }
    }
#endif
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

int lp_simplex_debug_division(lp_context* ctx, const double* num, const double* den, int n, double* fast_out, double* plain_out) {
    double *dx = nullptr, *dy = nullptr, *df = nullptr, *dp = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n;
    hipError_t e = hipMalloc(&dx, bytes);
    if (e == hipSuccess) e = hipMalloc(&dy, bytes);
    if (e == hipSuccess) e = hipMalloc(&df, bytes);
    if (e == hipSuccess) e = hipMalloc(&dp, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(dx, num, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dy, den, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_division, (unsigned)lp_ceil_div(n, 256), 256, 0, ctx->stream, dx, dy, n, df, dp);
        e = hipMemcpyAsync(fast_out, df, bytes, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(plain_out, dp, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dx);
    (void)hipFree(dy);
    (void)hipFree(df);
    (void)hipFree(dp);
    LP_HIP(ctx, e);
    return LP_OPTIMAL;
}
