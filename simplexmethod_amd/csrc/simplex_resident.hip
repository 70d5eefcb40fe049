// simplex_resident.hip — single-LP tableau simplex with the tableau RESIDENT ON CHIP
// (LP_SIMPLEX_ALGO_RESIDENT; what LP_SIMPLEX_ALGO_AUTO selects whenever the shape fits).
//
// Same pivot rules as the other two paths (/root/reference/src/SimplexSolover.h:152-196) and the
// same bits in every tableau element (each element takes the same fma per pivot), but the tableau
// never moves: G co-resident workgroups hold CPT columns each in REGISTERS (thread i = tableau
// row i; the xB column is replicated in every workgroup) for the whole solve, one launch per solve.
//   m <= 512 : 32 columns per workgroup, up to 512 row threads (a 512 x 1024 tableau = 32 workgroups x
//              131 KB of registers = one XCD of the MI355X)
//   m <= 960 : 16 columns per workgroup, up to 960 row threads
// Every workgroup has ONE MORE WAVE, the communication wave, which owns no rows: it polls the other
// workgroups' records, takes the decision, finishes the ratio test and publishes.  A wave that also
// updates rows cannot poll while it updates, and the hop between "my record is out" and "I know
// everybody's" costs 1000 cycles when a wave does nothing else (scripts/ubench_hop_pure.hip) against
// 3000 when the polling wave first has its 64 rows to update (the previous form of this kernel).
//
// One pivot = ONE all-to-all hop between the workgroups:
//   publish  every workgroup prices its own columns (Dantzig chain summary: extreme M_k, its first
//            index, "M_k beats everything of mine in front of it by more than eps"), runs the ratio
//            test (:181-192) on ITS candidate column speculatively — every row wave its 64 rows, the
//            communication wave the combination — and publishes ONE 32-byte record {M_k, column,
//            verdict, leaving row, u_r}; behind it, off the critical path, the row threads publish the
//            eta column -u_i/u_r of the candidate (:201), one value per thread;
//   consume  the communication wave of every workgroup reads all G records (G <= 32: one 64-lane
//            load, the next one already in flight) and replays the reference's scan over them —
//            identical inputs, identical decision everywhere, no leader; every row thread then reads
//            its entry of the winner's eta column and applies the rank-1 update to its own registers
//            (its part of the pivot row is local).
//   Near-ties (the hysteresis of :157 / :168 cannot be decided from the summaries) take an exact
//   slow path: the scan is replayed over all n published reduced costs and the owner of the
//   entering column publishes it in a second hop.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, R2): every shared 8 bytes is an
// {epoch tag, 32-bit payload} granule written by ONE 16-byte store (two granules) and read by sc1
// loads that bypass the reader's L1; a reader spins until every tag equals the pivot's epoch, so
// no flags, fences or drains are needed.  Two parities of every slot suffice: a workgroup publishes
// epoch p+2 only after consuming everyone's p+1.  Stores are write-through (sc1) unless a census at
// kernel start shows all participants on one XCD, whose shared L2 then serves plain stores (3x
// faster hop; placement is observed, never assumed).  Every spin is bounded: on a timeout the
// failing workgroup raises a chip-wide abort word and NOTHING is written back — the tableau goes
// home only behind one last hop in which every workgroup has seen every other one finish — and the
// host reruns the solve on another path (lp_simplex_stats::fell_back).
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

typedef int v4i __attribute__((ext_vector_type(4)));

enum { MODE_PIVOT = 0, MODE_OPTIMAL = 1, MODE_UNBOUNDED = 2, MODE_SLOW = 3, MODE_FAIL = 4 };

struct Ctl {   // decision of the current pivot, written by wave 0, read by everyone after a barrier
    int mode, kst, e, r;
    double ur, dE;        // pivot element; scan value of the entering column (maximise: d_e, minimise: -d_e)
    int fail, plain, pad0, pad1;
};
static_assert(sizeof(Ctl) == 48, "Ctl is read as three 16-byte LDS loads");

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
// record granules: the tag word carries a 16-bit tag (never 0: bit 15 set) and 16 bits of payload
__device__ __forceinline__ unsigned r_tag(unsigned ep) { return 0x8000u | (ep & 0x7FFFu); }
__device__ __forceinline__ v4i r_pack(unsigned ep, unsigned a16, unsigned b16, double v) {
    const long long b = __double_as_longlong(v);
    const unsigned t = r_tag(ep) << 16;
    v4i g = {(int)(t | a16), (int)(b & 0xFFFFFFFFLL), (int)(t | b16), (int)(b >> 32)};
    return g;
}
__device__ __forceinline__ bool r_fresh(v4i g, unsigned ep) {
    const unsigned t = r_tag(ep);
    return ((unsigned)g.x >> 16) == t && ((unsigned)g.z >> 16) == t;
}
constexpr unsigned kNoColumn = 0xFFFFu;   // record: nothing eligible among my columns
constexpr unsigned kCommit = 0xFFFEu;     // record: "I have finished; write back"

__device__ __forceinline__ v4i ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);   // sc1: served by L2, never by this CU's L1
}
__device__ __forceinline__ void st16(v4i g, __amdgpu_buffer_rsrc_t r, unsigned off, bool plain) {
    if (plain)
        __builtin_amdgcn_raw_buffer_store_b128(g, r, off, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(g, r, off, 0, 16);  // write-through
}

// Two granules in LDS, re-read on every call (each 8-byte half carries its own tag, so the halves may
// come from different writes; the caller checks both tags)
__device__ __forceinline__ v4i lds_granules(const v4i* p) {
    const volatile unsigned long long* q = reinterpret_cast<const volatile unsigned long long*>(p);
    const unsigned long long a = q[0], b = q[1];
    v4i g = {(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)(unsigned)b, (int)(unsigned)(b >> 32)};
    return g;
}

// Workgroup barrier of the pivot loop: LDS traffic only.  __syncthreads() also waits for the wave's GLOBAL memory
// operations (vmcnt(0): its fence has workgroup scope) — at the ratio barrier that was the acknowledgement of every
// row wave's column store (~400 cycles), at the decision barrier the communication wave's sweep still in flight
// (~200) — and nothing in the loop passes data between the waves of a workgroup through global memory.
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
        return __builtin_amdgcn_raw_buffer_load_b32(r, abort_off, 0, 16) != 0;
    }
};

struct Shared {
    double* prow;    // CPT + 8   : this workgroup's part of the pivot row (+ xB_r at CPT)
    double* ratio;   // mpad      : ratio-test values of the staged candidate (near-tie replay)
    double* u;       // mpad      : the staged candidate column
    struct SelScratch* sel;
    Ctl* ctl;
    v4i* pub;        // {epoch, u_r.lo, epoch, u_r.hi} of my candidate, from the communication wave
    int* basis;      // mpad      : N by position (every workgroup keeps its own copy)
    double* mirror;  // mpad x (CPT + 2): row-readable copy of the register slab (+ xB), see RS_MIRROR_WRITE
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
// LDS address in the pivot loop is an immediate and none of the nine base pointers occupies an SGPR
__host__ __device__ constexpr size_t resident_lds_bytes(int nt, int cpt) {
    return sizeof(double) * ((size_t)cpt + 8 + 2 * (size_t)nt) + 2048 /* SelScratch */ + sizeof(Ctl) + 16 +
           sizeof(int) * (size_t)nt + 256 /* stash */ + sizeof(double) * (size_t)nt * ((size_t)cpt + 2) /* mirror */;
}

// Read of the decision block's head {mode, kst, e, r} and of the failure flag (the pivot element and the
// scan value are read where the rare paths need them)
__device__ __forceinline__ Ctl ctl_read_head(const Ctl* p) {
    const v4i a = *reinterpret_cast<const v4i*>(p);
    const double ur = p->ur, dE = p->dE;
    const int f = p->fail;
    Ctl o;
    o.mode = __builtin_amdgcn_readfirstlane(a.x);
    o.kst = __builtin_amdgcn_readfirstlane(a.y);
    o.e = __builtin_amdgcn_readfirstlane(a.z);
    o.r = __builtin_amdgcn_readfirstlane(a.w);
    o.ur = ur; o.dE = dE;   // (per-lane copies of wave-uniform values: they feed vector arithmetic only)
    o.fail = __builtin_amdgcn_readfirstlane(f);
    o.plain = 0; o.pad0 = 0; o.pad1 = 0;
    return o;
}

struct Comm {   // buffer descriptor and byte offsets of the hand-off areas (all inside rd.comm)
    __amdgpu_buffer_rsrc_t r;
    unsigned rec, col, dpub, recS, colS, census, abort;
};

// NaN-free key of a ratio / reduced cost for the reductions (a NaN is never selected by the
// reference's `<` / `>` scans, exactly like the sentinel)
__device__ __forceinline__ double nan_to(double v, double sentinel) { return (v == v) ? v : sentinel; }

struct SelScratch {   // LDS: ratio-test slice summaries (one per row wave) and the pricing summary of the candidate
    // (64 entries each: entries past the row waves keep the identity written once at kernel start, so the
    // communication wave loads them with every lane and no lane mask)
    unsigned long long M[64];   // sortable key of the slice's smallest ratio
    double U[64];               // the candidate column's entry at the slice's first minimum
    int ok[64];                 // "the slice's minimum beats everything of the slice in front of it by more than eps"
    int J[64];                  // row of the slice's first minimum (INT_MAX: nothing eligible)
    unsigned long long pM;      // sortable key of M_k, the extreme reduced cost of my columns
    int pJ, pOk;                // its first column (local index, -1: none eligible); verdict on my columns in front of it
};
static_assert(sizeof(SelScratch) <= 2048 && sizeof(SelScratch) % 16 == 0, "resident_lds_bytes reserves 2048 bytes, 16-byte aligned");

// CPT columns per workgroup; NT = upper bound of the row threads (the launch uses mpad = m rounded up to
// 64 row threads PLUS ONE COMMUNICATION WAVE: blockDim = mpad + 64).
// !PUBL (the default): the published column is u_i itself, stored before the ratio test, and the consumers
//       divide by the record's u_r; PUBL (A/B, LP_RESIDENT_PUBL=1): the eta column -u_i/u_r, published by
//       the rows behind their rank-1 update once the communication wave has handed them u_r (the consumers'
//       division comes off the critical path, but the whole pivot measured 0.15 us longer).
template <int CPT, int NT, bool STAMPS, bool PUBL>
__global__ __launch_bounds__(NT + 64) void k_simplex_resident(SimplexDev d, ResidentDev rd) {
    static_assert(CPT == 32 || CPT == 16, "the slab is two vectors of 16 or 8 doubles");
    constexpr int HALF = CPT / 2;
    constexpr int NWMAX = NT <= 512 ? 8 : 16;   // slices of the ratio test (a power of two >= row waves)
    constexpr int KREPLAY = 4;                  // entries per lane and tile of the near-tie replay (256-row tiles)
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
    const bool is_comm = wave == nrw;          // the last wave: polls, decides, finishes the ratio test, publishes
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
    sh.sel = reinterpret_cast<SelScratch*>(sh.u + NT);
    sh.ctl = reinterpret_cast<Ctl*>(reinterpret_cast<char*>(sh.sel) + 2048);
    sh.pub = reinterpret_cast<v4i*>(sh.ctl + 1);
    sh.basis = reinterpret_cast<int*>(sh.pub + 1);
    sh.stash = reinterpret_cast<SimplexDev*>(sh.basis + NT);
    static_assert(sizeof(SimplexDev) <= 256, "resident_lds_bytes reserves 256 bytes for the stash");
    if (tid == 0) *sh.stash = d;
    sh.mirror = reinterpret_cast<double*>(reinterpret_cast<char*>(sh.stash) + 256);
    constexpr int MS = CPT + 2;   // row stride of the mirror in doubles (16-byte rows; 272 / 144 bytes: b128 stores conflict-free)

    Comm cm;
    cm.r = __builtin_amdgcn_make_buffer_rsrc(rd.comm, 0, rd.comm_bytes, 0x00020000);
    cm.rec = rd.rec_off; cm.col = rd.col_off; cm.dpub = rd.dpub_off;
    cm.recS = rd.recS_off; cm.colS = rd.colS_off; cm.census = rd.census_off; cm.abort = rd.abort_off;
    const unsigned col_stride = (unsigned)mpad * 16u;   // bytes of one published column

    // ---- the register-resident slab: CPT tableau entries of this thread's row, held in two vectors
    // that are LOCAL variables of the kernel (the compiler then indexes them with s_set_gpr_idx: one
    // indexed register move for a wave-uniform dynamic column, no select chain and no scratch)
    vslab Ta, Tb;
#define RS_SLAB_GET(j) (((j) < HALF) ? Ta[(j) & (HALF - 1)] : Tb[(j) & (HALF - 1)])
    // The LDS mirror: every row thread keeps a copy of its row (CPT entries + xB) where the OTHER waves can read
    // it.  After a decision every wave takes its pivot-row entries straight from row r of the mirror — no staging
    // by the row's owner, no barrier in front of the pricing.  Written behind the rank-1 update, i.e. while the
    // records travel; read only between the decision barrier and the ratio barrier.
    typedef double v2d __attribute__((ext_vector_type(2)));
#define RS_MIRROR_WRITE()                                                                  \
    do {                                                                                   \
        if (!is_comm) {                                                                    \
            double* mr_ = sh.mirror + (size_t)tid * MS;                                    \
            _Pragma("unroll") for (int j = 0; j < HALF; j += 2) {                           \
                const v2d a_ = {Ta[j], Ta[j + 1]}, b_ = {Tb[j], Tb[j + 1]};                \
                *reinterpret_cast<v2d*>(mr_ + j) = a_;                                     \
                *reinterpret_cast<v2d*>(mr_ + HALF + j) = b_;                              \
            }                                                                              \
            mr_[CPT] = xb;                                                                 \
        }                                                                                  \
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
    RS_MIRROR_WRITE();   // (visible to the other waves behind the census barrier)
    // reduced costs of this workgroup's columns: lane l of EVERY row wave holds column col0 + l
    const int mycol = col0 + lane;
    const bool colok = lane < CPT && mycol < n;
    bool nbl = colok && d.nonbasic[colok ? mycol : 0] != 0;
    double dl = colok ? d.T[(size_t)m * ld + mycol] : 0.0;
    double obj = d.T[(size_t)m * ld + n];
    for (int i = tid; i < m; i += mpad + 64) sh.basis[i] = d.basis[i];
    int it = st->iters;
    int status = (it >= max_iter) ? LP_ITER_LIMIT : kRunning;   // SimplexSolver.h:429,:450

    // ---- placement census: are all participants on one XCD (then plain stores reach the shared L2)?
    if (is_comm) {
        __builtin_amdgcn_s_setprio(3);   // the pivot's critical path runs through this wave: it wins issue arbitration on its SIMD
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        sh.sel->M[lane] = kPosInf;   // identity of the ratio test's slice table
        sh.sel->U[lane] = 0.0;
        sh.sel->ok[lane] = 0;
        sh.sel->J[lane] = INT_MAX;
        if (lane == 0) {
            v4i g = {1, (int)xcc, 1, 0};
            st16(g, cm.r, cm.census + (unsigned)k * 16u, false);
            sh.ctl->fail = 0;
            sh.ctl->pad0 = -1;
            v4i z = {0, 0, 0, 0};
            *sh.pub = z;
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
        if (STAMPS && rd.stamps && lane == 0) rd.stamps[16 * 256 + (size_t)k * 16 + 15] = 1000u + xcc * 10u + (all_same ? 1u : 0u);
        if (lane == 0) {
            sh.ctl->plain = (all_same && !(rd.flags & 1)) ? 1 : 0;
            if (failed || (rd.flags & 2)) sh.ctl->fail = 1;   // code 1: census (flag bit 1: injected by the tests)
        }
    }
    __syncthreads();
    const bool plain = sh.ctl->plain != 0;
    if (sh.ctl->fail) status = kResidentFailed;

    // Diagnostic build only (STAMPS): cycles of every phase of the communication wave (slots 0-5) and of
    // row wave 0 (slots 6-15), summed over the solve in registers and stored once at the end (a store per
    // stamp would sit in front of every later vmcnt wait and distort what it measures).
    unsigned long long acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = STAMPS ? __builtin_readcyclecounter() : 0;
#define RS_STAMP(s)                                                          \
    do {                                                                     \
        if (STAMPS) {                                                        \
            const unsigned long long now_ = __builtin_readcyclecounter();    \
            acc[(s)] += now_ - tprev;                                        \
            tprev = now_;                                                    \
            /* progress marker of this wave (read by the host after a timed-out run only) */ \
            if (rd.stamps && lane == 0) rd.stamps[16 * 256 + (size_t)k * 16 + wave] = (unsigned long long)ep * 100u + (s); \
        }                                                                    \
    } while (0)
#define RS_STAMP_C(s) do { if (is_comm) RS_STAMP(s); } while (0)
#define RS_STAMP_R(s) do { if (!is_comm) RS_STAMP(s); } while (0)
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

    unsigned ep = 0, par = 0, slot = 0;
    double pv = 0.0;
    unsigned long long pkey = 0, mkey = 0, hit = 0;
    int jl = -1;
    double lpub = 0.0;    // my entry of the eta column of my own candidate (what I published)

    // pricing summary of my columns (:152-174; minimisation scans -d with the same rule).  Every row
    // wave computes it from its own replica (lane l = column l): no barrier, no LDS.
#define RS_PRICE()                                                                 \
    do {                                                                           \
        pv = nbl ? nan_to(maximize ? dl : -dl, -INFINITY) : -INFINITY;             \
        pkey = lpdev::f64_sort_key(pv);                                            \
        mkey = lpdev::wave_ext_key_n<true, CPT>(pkey, &hit);                       \
        if (mkey == kNegInf) hit = 0;                                              \
        jl = hit ? (int)__builtin_ctzll(hit) : -1;                                 \
    } while (0)

    // Candidate (column values UP for the rows, xB values XBV).  Row waves: first half of the ratio test
    // (:181-192; every wave leaves the summary of its 64 rows in LDS), wave 0 adds the pricing summary.
    // A barrier.  The communication wave finishes the ratio test and publishes the record {M_k, column,
    // verdict on my columns in front of it, leaving row, u_r}; through LDS it hands u_r to the row waves,
    // which publish the eta column behind their part of the rank-1 update (RS_PUBLISH_COLUMN).
#define RS_CANDIDATE(UP, XBV)                                                                        \
    do {                                                                                             \
        ++ep;                                                                                        \
        par = ep & 1u;                                                                               \
        slot = par * (unsigned)G + (unsigned)k;                                                      \
        if (!is_comm) {                                                                              \
            if (jl >= 0) {                                                                           \
                const double ratio_ = (rowok && (UP) > eps) ? nan_to((XBV) / (UP), INFINITY) : INFINITY;   /* :185-186 */ \
                sh.ratio[tid] = ratio_;                                                              \
                sh.u[tid] = (UP);                                                                    \
                unsigned long long rhit_;                                                            \
                const unsigned long long rk_ = lpdev::wave_ext_key_n<false, 64>(lpdev::f64_sort_key(ratio_), &rhit_); \
                const bool any_ = rk_ != kPosInf;                                                    \
                const int L_ = any_ ? (int)__builtin_ctzll(rhit_) : 0;                               \
                const unsigned long long near_ =                                                     \
                    __ballot(lane < L_ && !lpdev::beats<false>(lpdev::f64_from_key(rk_), ratio_, eps)); \
                const double uL_ = lpdev::wave_bcast_f64((UP), L_);                                  \
                if (lane == 0) {                                                                     \
                    sh.sel->M[wave] = rk_;                                                           \
                    sh.sel->U[wave] = uL_;                                                           \
                    sh.sel->ok[wave] = (any_ && near_ == 0ULL) ? 1 : 0;                              \
                    sh.sel->J[wave] = any_ ? wave * 64 + L_ : INT_MAX;                               \
                }                                                                                    \
            }                                                                                        \
            if (wave == 0) {                                                                         \
                /* does my maximum beat every reduced cost of mine in front of it by more than eps? */ \
                const double Mk_ = hit ? lpdev::f64_from_key(mkey) : -INFINITY;                      \
                const bool okp_ = hit && __ballot(lane < jl && !(Mk_ > pv + eps)) == 0ULL;           \
                if (lane == 0) {                                                                     \
                    sh.sel->pM = hit ? mkey : kNegInf;                                               \
                    sh.sel->pJ = jl;                                                                 \
                    sh.sel->pOk = okp_ ? 1 : 0;                                                      \
                }                                                                                    \
            }                                                                                        \
            /* the candidate column for the others goes out LAST: a store in front of the ratio test made the      \
               compiler wait for its completion (vmcnt(0), ~400 cycles) before it reused the store's registers */ \
            if (!PUBL && jl >= 0 && rowok) st16(g_pack(ep, (UP)), cm.r, cm.col + slot * col_stride + (unsigned)tid * 16u, plain); \
        }                                                                                            \
        RS_STAMP_R(11);                                                                              \
        RS_STAMP_C(2);                                                                               \
        RS_MARK_R(16);                                                                               \
        lds_barrier();                                                                               \
        RS_STAMP_R(12);                                                                              \
        RS_MARK_R(17);                                                                               \
        RS_MARK_C(5);                                                                                \
        if (is_comm) {                                                                               \
            if (lane == 0 && bk_r >= 0) {   /* N(leave_pos) = enter, :196: every row wave has read the old entry */ \
                sh.basis[bk_r] = bk_e;                                                               \
                bk_r = -1;                                                                           \
            }                                                                                        \
            const int pJ_ = sh.sel->pJ;                                                              \
            const int pOk_ = sh.sel->pOk;                                                            \
            const unsigned long long pM_ = sh.sel->pM;                                               \
            int rk = -1;                                                                             \
            double urk = 0.0;                                                                        \
            if (pJ_ >= 0) {                                                                          \
                const unsigned long long Ml_ = sh.sel->M[lane];                                      \
                const int okl_ = sh.sel->ok[lane];                                                   \
                const int Jl_ = sh.sel->J[lane];                                                     \
                const double Ul_ = sh.sel->U[lane];                                                  \
                if (RS_MARK_A >= 30) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); RS_MARK(30); } \
                unsigned long long h2_;                                                              \
                const unsigned long long M2_ = lpdev::wave_ext_key_n<false, NWMAX>(Ml_, &h2_);       \
                RS_MARK(31);                                                                         \
                if (M2_ != kPosInf) {                                                                \
                    const int W_ = (int)__builtin_ctzll(h2_);   /* first slice attaining the minimum (row order) */ \
                    const int jM_ = __builtin_amdgcn_readlane(Jl_, W_);                              \
                    const int okW_ = __builtin_amdgcn_readlane(okl_, W_);                            \
                    const double uW_ = lpdev::wave_bcast_f64(Ul_, W_);                               \
                    const unsigned long long near2_ = __ballot(                                      \
                        lane < W_ && !lpdev::beats<false>(lpdev::f64_from_key(M2_), lpdev::f64_from_key(Ml_), eps)); \
                    if (okW_ && near2_ == 0ULL) {                                                    \
                        rk = jM_;                                                                    \
                        urk = uW_;                                                                   \
                    } else {   /* near-tie: exact replay over the LDS copy of the ratios */          \
                        double best_;                                                                \
                        auto load_ = [&](int j, bool& ok) {                                          \
                            ok = true;                                                               \
                            return sh.ratio[j];                                                      \
                        };                                                                           \
                        rk = lpdev::wave_chain_select<false, KREPLAY>(m, eps, best_, load_);         \
                        urk = (rk >= 0) ? sh.u[rk] : 0.0;                                            \
                    }                                                                                \
                }                                                                                    \
            }                                                                                        \
            RS_MARK(32);                                                                             \
            if (lane < 2) {                                                                          \
                const v4i g = lane == 0 ? r_pack(ep, pJ_ >= 0 ? (unsigned)(col0 + pJ_) : kNoColumn,  \
                                                 (pOk_ ? 0x8000u : 0u) | (unsigned)(rk + 1), lpdev::f64_from_key(pM_)) \
                                        : r_pack(ep, 0u, 0u, urk);                                   \
                st16(g, cm.r, cm.rec + slot * 32u + (unsigned)lane * 16u, plain);                    \
            }                                                                                        \
            if (PUBL && lane == 0) *sh.pub = g_pack(ep, urk);   /* read behind RS_PUBLISH_COLUMN's barrier */ \
            RS_STAMP(3);                                                                             \
            RS_MARK_C(6);                                                                            \
        }                                                                                            \
    } while (0)

    // eta column of my candidate, one entry per row thread: F(i,r) = -u_i/u_r (:201) — exactly the value
    // the consumers would compute from u_i and the record's u_r, computed once here, off their path.
    // u_r comes from the communication wave through LDS behind a workgroup barrier (every wave executes
    // this macro).  NOT a spin on the LDS word: the communication wave is the youngest wave of its SIMD,
    // and older waves spinning there starved it — its record came out 200 ms late or never.
#define RS_PUBLISH_COLUMN(UP)                                                                        \
    do {                                                                                             \
        if (PUBL && !is_comm && jl >= 0) {                                                           \
            /* u_r has been in LDS for hundreds of cycles when the rows get here (their update is longer than  \
               the communication wave's half of the ratio test); should it not be, they SLEEP between looks: \
               older waves spinning on an LDS word starved the communication wave, the youngest of its SIMD */ \
            v4i pg_ = lds_granules(sh.pub);                                                          \
            for (unsigned spins_ = 0; !g_fresh(pg_, ep) && spins_ < (1u << 20); ++spins_) {          \
                __builtin_amdgcn_s_sleep(8);                                                         \
                pg_ = lds_granules(sh.pub);                                                          \
            }                                                                                        \
            if (!g_fresh(pg_, ep)) sh.ctl->fail = 6;   /* code 6: u_r never came */                  \
            lpub = -(UP) / g_f64(pg_);                                                               \
            if (rowok) st16(g_pack(ep, lpub), cm.r, cm.col + slot * col_stride + (unsigned)tid * 16u, plain); \
        }                                                                                            \
    } while (0)

    int bk_e = -1, bk_r = -1;
    double up = 0.0;
    if (status == kRunning) {   // prologue: candidate of the initial tableau
        if (!is_comm) {
            RS_PRICE();
            if (jl >= 0) up = RS_SLAB_GET(jl);
        }
        RS_CANDIDATE(up, xb);
        RS_PUBLISH_COLUMN(up);
    }
    while (status == kRunning) {
        RS_STAMP_C(4);
        RS_STAMP_R(14);
        RS_MARK_C(0);
        RS_MARK_R(20);
        // ================= consume: everyone's records, one decision (communication wave) ============
        if (is_comm) {
            int mode, kst = 0, e = -1, r = -1;
            double ur = 0.0, M;
            unsigned long long Mk;
            bool failed = false;
            if (G <= 32) {
                // one 64-lane sweep reads all the records: lane q the first granule {M_q, column, verdict,
                // leaving row} of record q, lane 32 + q its second {u_r}; the next sweep is in flight while
                // this one is tested
                const int q = lane & 31;
                const bool live = q < G;
                const unsigned off = cm.rec + (par * (unsigned)G + (unsigned)(live ? q : 0)) * 32u + (unsigned)(lane >> 5) * 16u;
                Spin spin;
                v4i a = ld16(cm.r, off);
                for (;;) {
                    const v4i a1 = ld16(cm.r, off);
                    const bool ok = !live || r_fresh(a, ep);
                    if (__all(ok)) break;
                    if (spin.expired(cm.r, cm.abort)) {
                        failed = true;
                        const unsigned long long bad = __ballot(!ok);
                        if (lane == 0) sh.ctl->pad0 = bad ? (int)(__builtin_ctzll(bad) & 31) : -1;   // first stale record
                        break;
                    }
                    a = a1;
                }
                RS_STAMP(0);
                RS_MARK_C(1);
                const unsigned eq = (unsigned)a.x & 0xFFFFu;
                const bool cand = lane < 32 && live && eq < kCommit;
                const double Mq = cand ? g_f64(a) : -INFINITY;
                unsigned long long whit;
                Mk = lpdev::wave_ext_key_n<true, 32>(lane < 32 ? lpdev::f64_sort_key(Mq) : 0ULL, &whit);
                M = lpdev::f64_from_key(Mk);
                RS_MARK(34);
                if (failed) {
                    mode = MODE_FAIL;
                    if (lane == 0) sh.ctl->fail = 2;   // code 2: record poll
                } else if (Mk == kNegInf || !(M > eps)) {
                    mode = MODE_OPTIMAL;                     // the scan's final value is <= M <= eps (:162 / :174)
                } else {
                    const int W = (int)__builtin_ctzll(whit);   // first lane = first workgroup attaining M
                    // Does M beat everything in front of the winner by more than eps?  (one ballot: fl(v + eps)
                    // is monotone in v) — and the winner's own verdict on its columns (bit 15)
                    const unsigned long long near = __ballot(lane < W && !(M > Mq + eps));
                    kst = W;
                    e = (int)__builtin_amdgcn_readlane((int)eq, W);
                    const unsigned misc = (unsigned)__builtin_amdgcn_readlane(a.z, W) & 0xFFFFu;
                    ur = lpdev::wave_bcast_f64(g_f64(a), W + 32);
                    r = (int)(misc & 0x7FFu) - 1;
                    const bool clear = near == 0ULL && (misc & 0x8000u) != 0;
                    mode = !clear ? MODE_SLOW : (r < 0 ? MODE_UNBOUNDED : MODE_PIVOT);   // :179
                }
            } else {
                // more than 32 workgroups: R consecutive records per lane (lane order = column order)
                const int R = (G + 63) >> 6;
                const int q0 = lane * R;
                double Ml = -INFINITY, Mfront = -INFINITY;
                unsigned el = kNoColumn, miscl = 0;
                int ql = -1;
                double ul = 0.0;
                Spin spin;
                for (;;) {
                    bool ok = true;
                    Ml = -INFINITY; Mfront = -INFINITY; el = kNoColumn; miscl = 0; ql = -1; ul = 0.0;
                    for (int t = 0; t < R; ++t) {
                        const int q = q0 + t;
                        if (q >= G) break;
                        const unsigned base = cm.rec + (par * (unsigned)G + (unsigned)q) * 32u;
                        const v4i a = ld16(cm.r, base), b = ld16(cm.r, base + 16);
                        ok &= r_fresh(a, ep) && r_fresh(b, ep);
                        const double Mq = g_f64(a);
                        const unsigned eq = (unsigned)a.x & 0xFFFFu;
                        if (eq < kCommit && Mq > Ml) {           // strictly greater: ties keep the earlier column
                            Mfront = Ml;                         // the extreme of this lane's records in front of it
                            Ml = Mq;
                            el = eq;
                            miscl = (unsigned)a.z & 0xFFFFu;
                            ul = g_f64(b);
                            ql = q;
                        }
                    }
                    if (__all(ok)) break;
                    if (spin.expired(cm.r, cm.abort)) {
                        failed = true;
                        const unsigned long long bad = __ballot(!ok);
                        if (lane == 0) sh.ctl->pad0 = bad ? (int)__builtin_ctzll(bad) * R : -1;   // first stale record
                        break;
                    }
                }
                RS_STAMP(0);
                RS_MARK_C(1);
                unsigned long long whit;
                Mk = lpdev::wave_ext_key_n<true, 64>(lpdev::f64_sort_key(Ml), &whit);
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
                        __ballot((lane < W && !(M > Ml + eps)) || (lane == W && !(M > Mfront + eps)));
                    kst = __builtin_amdgcn_readlane(ql, W);
                    e = (int)__builtin_amdgcn_readlane((int)el, W);
                    const unsigned misc = (unsigned)__builtin_amdgcn_readlane((int)miscl, W);
                    ur = lpdev::wave_bcast_f64(ul, W);
                    r = (int)(misc & 0x7FFu) - 1;
                    const bool clear = near == 0ULL && (misc & 0x8000u) != 0;
                    mode = !clear ? MODE_SLOW : (r < 0 ? MODE_UNBOUNDED : MODE_PIVOT);   // :179
                }
            }
            if (lane == 0) {
                Ctl* c = sh.ctl;
                const v4i head = {mode, kst, e, r};
                *reinterpret_cast<v4i*>(c) = head;
                c->ur = ur; c->dE = M;
            }
            RS_STAMP(1);
            RS_MARK_C(2);
        }
        lds_barrier();   // the decision barrier
        RS_STAMP_R(6);
        RS_MARK_C(3);
        RS_MARK_R(10);
        // the whole decision block in one go (three 16-byte LDS reads in flight together), by ONE lane: a
        // broadcast read costs the LDS all 64 lanes' bandwidth, and every wave asks at this moment
        Ctl cc = ctl_read_head(sh.ctl);
        int mode = cc.mode;
        if (cc.fail) mode = MODE_FAIL;
        bool from_colS = false;
        if (mode == MODE_SLOW) {
            // ---- exact replay of the scan over all n published reduced costs (near-tie)
            __syncthreads();   // everyone has read the decision before the communication wave rewrites it
            // every workgroup reaches this branch for the same pivot: only now are all the reduced
            // costs published (a near-tie is rare; storing them with every pivot was 512 bytes per
            // workgroup of traffic in front of the records)
            if (wave == 0 && lane < CPT) st16(g_pack(ep, pv), cm.r, cm.dpub + (slot * CPT + (unsigned)lane) * 16u, plain);
            if (is_comm) {
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
                const int e = lpdev::wave_chain_select<true, 4>(n, eps, best, load);
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
                    if (!is_comm) {
                        const int je = __builtin_amdgcn_readfirstlane(e - col0);
                        up = RS_SLAB_GET(je);
                        if (rowok) st16(g_pack(ep, up), cm.r, cm.colS + par * col_stride + (unsigned)tid * 16u, plain);
                        const double ratio = (rowok && up > eps) ? nan_to(xb / up, INFINITY) : INFINITY;
                        sh.ratio[tid] = ratio;
                    }
                    __syncthreads();
                    if (is_comm) {   // (rare path: the whole chain on the LDS copy)
                        double best2;
                        auto load2 = [&](int j, bool& ok) {
                            ok = true;
                            return sh.ratio[j];
                        };
                        const int r2 = lpdev::wave_chain_select<false, KREPLAY>(m, eps, best2, load2);
                        // (the row threads' copy of the column: thread r2 stored it to colS as well)
                        if (lane == 0) sh.ctl->r = r2;
                    }
                    __syncthreads();
                    const int r2 = sh.ctl->r;
                    if (tid == (r2 >= 0 ? r2 : 0) && !is_comm) st16(r_pack(ep, (unsigned)(r2 + 1), 0u, r2 >= 0 ? up : 0.0), cm.r, cm.recS + par * 16u, plain);
                }
                __syncthreads();   // everyone has read e / owner before the communication wave rewrites the decision
                if (is_comm) {
                    Spin spin;
                    bool failed = false;
                    v4i a;
                    for (;;) {
                        a = ld16(cm.r, cm.recS + par * 16u);
                        if (r_fresh(a, ep)) break;
                        if (spin.expired(cm.r, cm.abort)) {
                            failed = true;
                            break;
                        }
                    }
                    if (lane == 0) {
                        Ctl* c = sh.ctl;
                        const int r2 = (int)((unsigned)a.x & 0xFFFFu) - 1;
                        c->ur = g_f64(a);
                        c->r = r2;
                        c->mode = failed ? MODE_FAIL : (r2 < 0 ? MODE_UNBOUNDED : MODE_PIVOT);
                        if (failed) c->fail = 4;   // code 4: slow-path second hop
                    }
                }
                __syncthreads();
                cc = ctl_read_head(sh.ctl);
                mode = cc.mode;
                from_colS = true;
            }
        }
        if (mode != MODE_PIVOT) {
            status = mode == MODE_OPTIMAL ? LP_OPTIMAL : mode == MODE_UNBOUNDED ? LP_UNBOUNDED : kResidentFailed;
            break;
        }
        const int kst = cc.kst, e = cc.e, r = cc.r;
        RS_MARK_R(11);
        // ---- entering column: the winner's published eta column (mine is still in a register; after
        // the slow path it is the owner's second-hop column of u_i)
        const bool want_col = kst != k && rowok;
        const unsigned coff = (from_colS ? cm.colS + par * col_stride
                                         : cm.col + (par * (unsigned)G + (unsigned)kst) * col_stride) +
                              (unsigned)tid * 16u;
        v4i gcol = {0, 0, 0, 0};
        const unsigned ep_col = ep;
        if (want_col) gcol = ld16(cm.r, coff);   // awaited after the next pricing
        const int oldb = sh.basis[r];    // (rewritten by the communication wave behind the ratio barrier)
        RS_STAMP_R(7);
        RS_MARK_R(12);
        RS_STAMP_R(8);
        RS_MARK_C(4);
        RS_MARK_R(13);
        ++it;
        const bool last = it >= max_iter;   // :450: this pivot is applied, no further one is chosen
        double xbn = 0.0, upn = 0.0, l = 0.0, inv = 0.0;
        if (is_comm) {
            // this wave's idle window (nothing to do until the ratio barrier): the pivot row goes from the mirror
            // to the place the rows' rank-1 update reads it from — thread r overwrites its mirror row during that
            // update — and the trace is kept
            if (lane <= CPT) sh.prow[lane] = sh.mirror[(size_t)r * MS + lane];
            bk_e = e;
            bk_r = r;
            if (k == 0 && lane == 0) {   // (workgroup 0 only: the trace pointers come from the stash, not from SGPRs held all along)
                const SimplexDev dt = lds_reload(sh.stash);
                if (it - 1 < dt.trace_cap) {
                    dt.trace_enter[it - 1] = e;
                    dt.trace_leave[it - 1] = r;
                }
            }
        } else {
            // ---- reduced-cost row (row m of the tableau) after this pivot, replicated per wave.  ONE LDS read
            // per wave, straight from row r of the mirror: lane j takes pivot-row entry j (lane CPT: xB_r); the
            // wave-uniform entries needed below (xB_r and the candidate's) come out of these registers by
            // v_readlane.  The two quotients of the update, F(r,r) = 1/u_r (:204) and the reduced-cost row's
            // -d_e/u_r, are computed while that read is in flight.
            const double pl = sh.mirror[(size_t)r * MS + (lane <= CPT ? lane : 0)];
            inv = 1.0 / cc.ur;
            const double lm = -(maximize ? cc.dE : -cc.dE) / cc.ur;
            const double pxb = lpdev::wave_bcast_f64(pl, CPT);
            if (colok) {
                dl = (mycol == e) ? 0.0 : fma(lm, pl, dl);
                if (mycol == e) nbl = false;
                if (mycol == oldb) nbl = true;
            }
            obj = fma(lm, pxb, obj);
            if (!last) RS_PRICE();
            RS_STAMP(9);
            RS_MARK_R(14);
            // ---- the entering column has arrived by now: F(i,r), :201 (rows other than r)
            if (want_col) {
                Spin spin;
                while (!g_fresh(gcol, ep_col)) {
                    if (spin.expired(cm.r, cm.abort)) {
                        sh.ctl->fail = 5;   // code 5: entering column (acted on at the next decision barrier / the commit)
                        break;
                    }
                    gcol = ld16(cm.r, coff);
                }
                l = (PUBL && !from_colS) ? g_f64(gcol) : -g_f64(gcol) / sh.ctl->ur;
            } else {
                l = (PUBL && !from_colS) ? lpub : -up / sh.ctl->ur;
            }
            xbn = (tid == r) ? xb * inv : fma(l, pxb, xb);
            RS_STAMP(10);
            RS_MARK_R(15);
            if (!last && jl >= 0) {
                // the candidate column of the NEXT pivot, updated ahead of the others (same operation,
                // same operands as the full update below: identical bits)
                const double t = RS_SLAB_GET(jl);
                upn = (tid == r) ? t * inv : fma(l, lpdev::wave_bcast_f64(pl, jl), t);
            }
        }
        if (!last) {
            RS_CANDIDATE(upn, xbn);
        }
        if (!is_comm) {
            // ---- rank-1 update of my registers (tableau_pivot: F(i,r) = -u_i/u_r, F(r,r) = 1/u_r, :198-204);
            // the record published above is travelling meanwhile
            if (rowok) {
                if (tid == r) {
#pragma unroll
                    for (int j = 0; j < HALF; ++j) {
                        Ta[j] = Ta[j] * inv;
                        Tb[j] = Tb[j] * inv;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < HALF; ++j) Ta[j] = fma(l, sh.prow[j], Ta[j]);
#pragma unroll
                    for (int j = 0; j < HALF; ++j) Tb[j] = fma(l, sh.prow[HALF + j], Tb[j]);
                }
                if (kst == k) {   // column e becomes the unit vector
                    const double unit = (tid == r) ? 1.0 : 0.0;
                    switch (e - col0) {
#define RS_CASE(J)                                        \
    case J: if (J < HALF) Ta[J & (HALF - 1)] = unit; break;          \
    case HALF + J: if (J < HALF) Tb[J & (HALF - 1)] = unit; break;
                        RS_CASE(0) RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7)
#undef RS_CASE
                        default: break;
                    }
                    if (HALF == 16) {
                        switch (e - col0) {
#define RS_CASE(J)                                        \
    case J: Ta[J & (HALF - 1)] = unit; break;             \
    case 16 + J: Tb[J & (HALF - 1)] = unit; break;
                            RS_CASE(8) RS_CASE(9) RS_CASE(10) RS_CASE(11) RS_CASE(12) RS_CASE(13) RS_CASE(14) RS_CASE(15)
#undef RS_CASE
                            default: break;
                        }
                    }
                }
            }
            if (!last) RS_PUBLISH_COLUMN(upn);
            xb = xbn;
            up = upn;
            RS_MIRROR_WRITE();
            RS_STAMP(13);
        }
        if (last) status = LP_ITER_LIMIT;
    }
#undef RS_PRICE
#undef RS_CANDIDATE
#undef RS_PUBLISH_COLUMN

    // ================= commit: the tableau goes home only if EVERY workgroup got here =================
    // One more hop in the ordinary record stream (epoch ep + 1): "I have finished".  A workgroup that
    // failed anywhere — also one whose entering column timed out on the very last pivot — raises the
    // abort word instead, which every spin of the others observes; nobody writes anything back then.
    __syncthreads();   // (a row thread's late fail = 5 is visible to everyone below)
    if (is_comm && lane == 0 && bk_r >= 0) sh.basis[bk_r] = bk_e;   // (the last pivot before the iteration limit)
    __syncthreads();
    if (status != kResidentFailed && sh.ctl->fail) status = kResidentFailed;
    if (status != kResidentFailed) {
        ++ep;
        par = ep & 1u;
        slot = par * (unsigned)G + (unsigned)k;
        if (is_comm) {
            if (lane < 2) st16(r_pack(ep, kCommit, 0u, 0.0), cm.r, cm.rec + slot * 32u + (unsigned)lane * 16u, plain);
            Spin spin;
            bool failed = false;
            for (;;) {
                bool ok = true;
                for (int q = lane; q < G; q += 64) {
                    const unsigned base = cm.rec + (par * (unsigned)G + (unsigned)q) * 32u;
                    const v4i a = ld16(cm.r, base), b = ld16(cm.r, base + 16);
                    ok &= r_fresh(a, ep) && r_fresh(b, ep) && ((unsigned)a.x & 0xFFFFu) == kCommit;
                }
                if (__all(ok)) break;
                if (spin.expired(cm.r, cm.abort)) {
                    failed = true;
                    break;
                }
            }
            if (failed && lane == 0) sh.ctl->fail = 7;   // code 7: commit
        }
        __syncthreads();
        if (sh.ctl->fail) status = kResidentFailed;
    }
    RS_STAMP_C(5);
    RS_STAMP_R(15);
#undef RS_STAMP
#undef RS_STAMP_C
#undef RS_STAMP_R
    if (status == kResidentFailed) {   // nothing is written back: the host reruns on another path
        if (tid == 0) {
            SimplexState* st = lds_reload(sh.stash).state;
            // first failing workgroup records where it stopped: {code, workgroup, epoch} (diagnostic)
            if (atomicCAS(reinterpret_cast<int*>(rd.comm + rd.abort_off), 0, 1) == 0) {
                st->enter = sh.ctl->fail * 1000 + k;
                st->leave = (int)ep * 1000 + sh.ctl->pad0;
            }
            st->status = kResidentFailed;
        }
        return;
    }
    if (RS_MARK_A >= 0 && lane == 0 && ((RS_MARK_A < 10 || RS_MARK_A >= 30) ? is_comm : wave == 0)) {
        v4i g = {(int)(unsigned)mk_acc, (int)(unsigned)(mk_acc >> 32), it, 0};
        st16(g, cm.r, cm.census + (unsigned)k * 16u, false);
    }
    if (STAMPS && rd.stamps && (tid == 0 || (is_comm && lane == 0))) {
        const int q0 = is_comm ? 0 : 6, q1 = is_comm ? 6 : 16;
        for (int q = q0; q < q1; ++q) rd.stamps[(size_t)k * 16 + q] = acc[q];
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
        for (int i = tid; i < de.m; i += (int)blockDim.x) de.basis[i] = sh.basis[i];
        if (tid == 0) {
            de.T[(size_t)de.m * de.ld + de.n] = obj;
            de.state->iters = it;
            de.state->status = status;
            de.state->pivot_valid = 0;
        }
    }
#undef RS_SLAB_GET
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
void launch_resident(const SimplexDev& d, const ResidentDev& rd, size_t shm, hipStream_t s, bool stamped, bool pubu,
                     hipError_t* attr_err) {
#define RS_GO(ST, PL)                                                                                                    \
    do {                                                                                                                 \
        *attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_simplex_resident<CPT, NT, ST, PL>),             \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                           \
        if (*attr_err == hipSuccess)                                                                                     \
            hipLaunchKernelGGL((k_simplex_resident<CPT, NT, ST, PL>), rd.G * rd.stride, rd.mpad + 64, shm, s, d, rd);         \
    } while (0)
    if (stamped)
        RS_GO(true, false);
    else if (pubu)
        RS_GO(false, false);
    else
        RS_GO(false, true);
#undef RS_GO
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
    r.rec_off = take((size_t)2 * G * 32);
    r.recS_off = take(2 * 16);
    r.dpub_off = take((size_t)2 * G * cpt * 16);
    r.colS_off = take((size_t)2 * r.mpad * 16);
    r.col_off = take((size_t)2 * G * r.mpad * 16);
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
    const bool stamped = rd.stamps != nullptr;
    ResidentDev rdv = rd;
    if (getenv("LP_RESIDENT_FORCE_SC1")) rdv.flags |= 1;     // diagnostics: write-through stores on one XCD too
    if (getenv("LP_RESIDENT_SPREAD")) rdv.stride = 1;        // diagnostics: participants on all XCDs
    if (getenv("LP_RESIDENT_INJECT_FAILURE")) rdv.flags |= 2;   // tests: the census reports a failure
    const bool pubu = getenv("LP_RESIDENT_PUBL") == nullptr; // default: publish u_i (consumers divide); A/B: the eta column
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
        launch_resident<32, 512>(d, rdv, shm, s, stamped, pubu, &attr_err);
    else
        launch_resident<16, 960>(d, rdv, shm, s, stamped, pubu, &attr_err);
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
        if (getenv("LP_RESIDENT_DEBUG")) {   // the record area as the failure left it
            std::vector<int> rec((size_t)2 * rd.G * 8);
            if (hipMemcpy(rec.data(), rd.comm + rd.rec_off, rec.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                for (int par = 0; par < 2; ++par)
                    for (int q = 0; q < rd.G; ++q) {
                        const int* w = &rec[((size_t)par * rd.G + q) * 8];
                        fprintf(stderr, "[resident debug] parity %d record %3d: %08x %08x %08x %08x | %08x %08x %08x %08x\n", par, q,
                                w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
                    }
            }
        }
        if (stamped && getenv("LP_RESIDENT_DEBUG")) {   // diagnostic instantiation: where every wave of every workgroup stopped
            std::vector<unsigned long long> prog(16 * 256);
            if (hipMemcpy(prog.data(), rd.stamps + 16 * 256, prog.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                for (int q = 0; q < rd.G; ++q) {
                    fprintf(stderr, "[resident debug] workgroup %3d, last stamp of waves 0..%d (epoch*100 + slot):", q, rd.mpad / 64);
                    for (int w = 0; w <= rd.mpad / 64; ++w) fprintf(stderr, " %llu", prog[(size_t)q * 16 + w]);
                    fprintf(stderr, "  census %llu", prog[(size_t)q * 16 + 15]);
                    fprintf(stderr, "\n");
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
    if (RS_MARK_A >= 0 && getenv("LP_RESIDENT_MARKS")) {   // diagnostic builds: the interval RS_MARK_A -> RS_MARK_B per workgroup
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
