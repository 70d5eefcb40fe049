// device_select.hpp — wave64 primitives for the order-dependent pivot rules.
//
// The reference scans candidates sequentially with an EPS hysteresis
// (/root/reference/src/SimplexSolover.h:153-161 pricing, :181-192 ratio test):
//     best = -inf;  for j ascending: if (v_j > best + eps) { best = v_j; sel = j; }
// That is not an associative reduction, but it is a chain of "records": once
// `best` holds v_p, the next accepted entry is the FIRST j > p with
// v_j > v_p + eps, and no entry before p can qualify again (it was either
// accepted with a smaller value or rejected against a smaller threshold).
//
// One wave replays the scan exactly:
//   fast path  — let M be the maximum and jM its first index, P the maximum of the
//     incoming `best` and of every entry before jM.  The scan's value v just before
//     jM satisfies P - eps <= v <= P, so if M > P + eps the entry jM is accepted
//     whatever v is, and nothing after it can be (all entries <= M): answer (M, jM)
//     with three wave reductions.
//   slow path  — (a near-tie, |M - P| <= eps, e.g. degenerate vertices) the chain is
//     replayed jump by jump, each jump a parallel "first index above threshold".
// Reductions use DPP row operations + row_bcast (gfx9 family), not ds_bpermute.
#pragma once

#include <hip/hip_runtime.h>

#include <climits>

namespace lpdev {

// ---- DPP wave reductions (result valid in every lane) -----------------------
#define LP_DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]
#define LP_DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]
#define LP_DPP_ROW_HALF_MIRROR 0x141
#define LP_DPP_ROW_MIRROR 0x140
#define LP_DPP_ROW_BCAST15 0x142
#define LP_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xFFFFFFFFLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ int wave_min_i32(int v) {
    int o;
    o = dpp_i32<LP_DPP_QUAD_XOR1, 0xF>(v); v = o < v ? o : v;
    o = dpp_i32<LP_DPP_QUAD_XOR2, 0xF>(v); v = o < v ? o : v;
    o = dpp_i32<LP_DPP_ROW_HALF_MIRROR, 0xF>(v); v = o < v ? o : v;
    o = dpp_i32<LP_DPP_ROW_MIRROR, 0xF>(v); v = o < v ? o : v;
    o = dpp_i32<LP_DPP_ROW_BCAST15, 0xA>(v); v = o < v ? o : v;
    o = dpp_i32<LP_DPP_ROW_BCAST31, 0xC>(v); v = o < v ? o : v;
    return __builtin_amdgcn_readlane(v, 63);
}

template <bool WANT_MAX>
__device__ __forceinline__ double wave_ext_f64(double v) {
    double o;
    // v_max_f64 / v_min_f64: one instruction per step (inputs are never NaN here: ineligible
    // entries were replaced by the +-inf sentinel before the reduction)
#define LP_STEP(CTRL, MASK)                              \
    o = dpp_f64<CTRL, MASK>(v);                          \
    v = WANT_MAX ? fmax(v, o) : fmin(v, o);
    LP_STEP(LP_DPP_QUAD_XOR1, 0xF)
    LP_STEP(LP_DPP_QUAD_XOR2, 0xF)
    LP_STEP(LP_DPP_ROW_HALF_MIRROR, 0xF)
    LP_STEP(LP_DPP_ROW_MIRROR, 0xF)
    LP_STEP(LP_DPP_ROW_BCAST15, 0xA)
    LP_STEP(LP_DPP_ROW_BCAST31, 0xC)
#undef LP_STEP
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double wave_bcast_f64(double v, int src_lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFLL), src_lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Exact replay of the sequential chain over `len` entries read through `load`
// (load(j, ok) returns v_j and sets ok=false for ineligible j), executed by ONE
// full wave (all 64 lanes must call, with identical arguments).  Returns the
// selected index or -1; `best` ends as the chain's final value (+-inf if nothing
// was eligible).  WANT_MAX: the :153-161 form (v > best + eps); otherwise the
// :164-172 / :181-192 form (v < best - eps).
template <bool WANT_MAX, int K = 16, typename Load>   // K: entries per lane per tile (a power of two)
__device__ int wave_chain_select(int len, double eps, double& best, Load load) {
    constexpr int TILE = 64 * K;
    const int lane = threadIdx.x & 63;
    const double sentinel = WANT_MAX ? -INFINITY : INFINITY;
    best = sentinel;
    int sel = -1;
    for (int base = 0; base < len; base += TILE) {
        // (only one wave runs here, so what counts is the length of the dependent instruction
        // chains: extremes are balanced v_max/v_min trees, indices are found afterwards)
        double val[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = base + k * 64 + lane;
            // unconditional (clamped) load so that all K loads are in flight together
            bool ok = false;
            double v = load(j < len ? j : len - 1, ok);
            val[k] = (ok && j < len) ? v : sentinel;
        }
        auto ext = [](double a, double b) { return WANT_MAX ? fmax(a, b) : fmin(a, b); };
        auto tree = [&](const double (&x)[K]) {   // balanced: depth log2 K
            static_assert((K & (K - 1)) == 0, "K must be a power of two");
            double t[K];
#pragma unroll
            for (int i = 0; i < K; ++i) t[i] = x[i];
#pragma unroll
            for (int w = K / 2; w >= 1; w /= 2)
#pragma unroll
                for (int i = 0; i < w; ++i) t[i] = ext(t[i], t[i + w]);
            return t[0];
        };
        const double lext = tree(val);  // this lane's extreme value (NaN entries never win: fmax/fmin drop them)
        // ---- fast path
        const double M = wave_ext_f64<WANT_MAX>(lext);
        const double thr_in = WANT_MAX ? best + eps : best - eps;
        if (!(WANT_MAX ? (M > thr_in) : (M < thr_in))) continue;  // nothing here is accepted
        int lidx = INT_MAX;             // first index at which this lane holds M
#pragma unroll
        for (int k = K - 1; k >= 0; --k) lidx = (val[k] == M) ? base + k * 64 + lane : lidx;
        const int jM = wave_min_i32(lidx);
        double before[K];               // entries before jM
#pragma unroll
        for (int k = 0; k < K; ++k) before[k] = (base + k * 64 + lane < jM) ? val[k] : sentinel;
        double P = wave_ext_f64<WANT_MAX>(tree(before));
        if (WANT_MAX ? (best > P) : (best < P)) P = best;
        if (WANT_MAX ? (M > P + eps) : (M < P - eps)) {
            best = M;
            sel = jM;
            continue;
        }
        // ---- slow path: replay the chain jump by jump (near-ties within eps)
        for (;;) {
            const double thr = WANT_MAX ? best + eps : best - eps;
            int cand = INT_MAX;
            double cv = sentinel;
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                const bool q = WANT_MAX ? (val[k] > thr) : (val[k] < thr);
                if (q) {
                    cand = base + k * 64 + lane;
                    cv = val[k];
                }
            }
            const int first = wave_min_i32(cand);
            if (first == INT_MAX) break;
            best = wave_bcast_f64(cv, (first - base) & 63);
            sel = first;
        }
    }
    return sel;
}

// ---------------------------------------------------------------------------
// Reductions on SORTABLE KEYS.  A double maps to a 64-bit key whose unsigned order is the numeric
// order (-inf lowest, +inf highest; the caller replaces NaNs by its sentinel first), and a wave
// extreme of keys is two passes of six 32-bit DPP min/max instructions (high words, then low
// words among the lanes that tie on the high word): v_max_u32 / v_min_u32 take the DPP operand
// directly, where an fp64 extreme needs two DPP moves and a v_max_f64 per step — less than half
// the dependent latency, which is all a lone reducing wave pays.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long f64_sort_key(double v) {
    const long long b = __double_as_longlong(v);
    return (unsigned long long)(b ^ ((b >> 63) | (long long)0x8000000000000000ull));
}
__device__ __forceinline__ double f64_from_key(unsigned long long k) {
    const long long b = (long long)k;
    return __longlong_as_double((b < 0) ? (b ^ (long long)0x8000000000000000ull) : ~b);
}

// Extreme of a 32-bit unsigned value over the 64 lanes (all lanes must be active); uniform result.
// One asm block: the DPP hazard (a VALU write followed by a DPP read of the same VGPR needs two
// wait states) is covered by the s_nop 1 between steps.
template <bool WANT_MAX>
__device__ __forceinline__ unsigned wave_ext_u32(unsigned v) {
    if (WANT_MAX) {
        asm volatile(
            "s_nop 1\n\t"
            "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
            "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
            : "+v"(v));
    } else {
        asm volatile(
            "s_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
            : "+v"(v));
    }
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <bool WANT_MAX>
__device__ __forceinline__ unsigned long long wave_ext_key(unsigned long long key) {
    const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    const unsigned mhi = wave_ext_u32<WANT_MAX>(hi);
    const unsigned lo2 = (hi == mhi) ? lo : (WANT_MAX ? 0u : 0xFFFFFFFFu);
    const unsigned mlo = wave_ext_u32<WANT_MAX>(lo2);
    return ((unsigned long long)mhi << 32) | mlo;
}

__device__ __forceinline__ unsigned long long wave_bcast_u64(unsigned long long v, int src_lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

// ---------------------------------------------------------------------------
// Workgroup-wide form of the chain scan, ONE entry per thread: thread t holds entry t in `v`
// (threads past the end and ineligible entries hold the sentinel, NaNs replaced by it) and has
// also written it to vals[t] in LDS (read only by the slow path).
//   block_select_stage1: every wave reduces its 64 entries to (M_w, first index j_w, ok_w) and stores
//     them in LDS; ok_w = "M_w beats every entry of the wave in front of j_w by more than eps".  The
//     caller then executes a workgroup barrier.
//   block_select_stage2: ONE full wave (any) combines the <= 16 slices: global M, the first wave
//     w* attaining it.  If ok_w* holds and M beats M_w by more than eps for every w < w*, the scan
//     must end on (M, j_w*) as in wave_chain_select's fast path; otherwise that wave replays the
//     whole chain.
// "M beats P = ext(entries in front) by more than eps" is tested entry by entry with one ballot
// instead of reducing P: fl(v + eps) is monotone in v, so M > fl(P + eps) iff M > fl(v + eps) for
// every v in front — the same verdict, one 64-bit key reduction (~150 cycles of dependent DPP
// steps on the pivot's critical path) less per stage.
// ---------------------------------------------------------------------------
struct BlockSelScratch {  // 16-B aligned, lives in LDS
    unsigned long long M[16];   // sortable keys
    int ok[16];
    int J[16];
};

template <bool WANT_MAX>
__device__ __forceinline__ bool beats(double M, double v, double eps) {
    return WANT_MAX ? (M > v + eps) : (M < v - eps);
}

template <bool WANT_MAX>
__device__ __forceinline__ void block_select_stage1(double v, double eps, BlockSelScratch* sc) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long skey = f64_sort_key(WANT_MAX ? -INFINITY : INFINITY);
    const unsigned long long key = f64_sort_key(v);
    const unsigned long long mk = wave_ext_key<WANT_MAX>(key);
    const unsigned long long hit = __ballot(key == mk && key != skey);
    const int L = hit ? (int)__builtin_ctzll(hit) : 64;
    const unsigned long long near = __ballot(lane < L && !beats<WANT_MAX>(f64_from_key(mk), v, eps));
    if (lane == 0) {
        sc->M[wave] = hit ? mk : skey;
        sc->ok[wave] = (hit && near == 0ULL) ? 1 : 0;
        sc->J[wave] = hit ? wave * 64 + L : INT_MAX;
    }
}

template <bool WANT_MAX>
__device__ __forceinline__ int block_select_stage2(const double* vals, int len, double eps,
                                                   const BlockSelScratch* sc) {
    const int lane = threadIdx.x & 63, nwaves = (int)(blockDim.x >> 6);
    const unsigned long long skey = f64_sort_key(WANT_MAX ? -INFINITY : INFINITY);
    const bool has = lane < nwaves;
    const unsigned long long Ml = has ? sc->M[lane] : skey;
    const int okl = has ? sc->ok[lane] : 0;
    const int Jl = has ? sc->J[lane] : INT_MAX;
    const unsigned long long M = wave_ext_key<WANT_MAX>(Ml);
    const unsigned long long whit = __ballot(has && Ml == M && Jl != INT_MAX);
    if (!whit) return -1;
    const int W = (int)__builtin_ctzll(whit);
    const int jM = __builtin_amdgcn_readlane(Jl, W);
    const int okW = __builtin_amdgcn_readlane(okl, W);
    const double Md = f64_from_key(M);
    const unsigned long long near = __ballot(lane < W && !beats<WANT_MAX>(Md, f64_from_key(Ml), eps));
    if (okW && near == 0ULL) return jM;
    double best;   // near-tie: exact replay over the LDS copy
    auto load = [&](int j, bool& ok) {
        ok = true;
        return vals[j];
    };
    return wave_chain_select<WANT_MAX>(len, eps, best, load);
}

// ---------------------------------------------------------------------------
// Workgroup-wide form of the chain scan over ANY number of entries (the one-workgroup selectors of
// simplex_launch.hip / simplex_overlap.hip at large n, m: a lone wave walking 8192 reduced costs in tiles of
// 1024 was 10-20 us of every pivot).  Every thread of the workgroup calls it.  produce(j) returns entry j with
// ineligible entries as the sentinel; NaNs are replaced by the sentinel here (the chain never takes either).
// STORE: the entries are also written to vals[j] (global or LDS scratch of >= len doubles); otherwise vals
// already holds them.  Result as wave_chain_select's: the chain's last accepted index (or -1) and value.
//   pass 1  thread t scans entries t, t + T, ...: its extreme and the first index holding it; waves and then the
//           workgroup combine (M, jM = first index of the overall extreme).
//   pass 2  "M beats every entry in front of jM by more than eps" — then the sequential chain must end on
//           (M, jM) (wave_chain_select's fast path, one tile = everything) — else wave 0 replays the chain.
// ---------------------------------------------------------------------------
struct BlockChainScratch {   // LDS
    double M[16];
    int J[16];
    int sel;
    double best;
};

template <bool WANT_MAX, bool STORE, typename Produce>
__device__ __forceinline__ int block_chain_select(int len, double eps, double& best, Produce produce, double* vals,
                                                  BlockChainScratch* sc) {
    const int tid = threadIdx.x, T = (int)blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const double sentinel = WANT_MAX ? -INFINITY : INFINITY;
    double lv = sentinel;
    int lj = INT_MAX;
    for (int j = tid; j < len; j += T) {
        double v = produce(j);
        v = (v == v) ? v : sentinel;
        if (STORE) vals[j] = v;
        if (WANT_MAX ? (v > lv) : (v < lv)) {   // j ascends: the first index of the thread's extreme
            lv = v;
            lj = j;
        }
    }
    const double Mw = wave_ext_f64<WANT_MAX>(lv);
    const int jw = wave_min_i32((lv == Mw) ? lj : INT_MAX);
    if (lane == 0) {
        sc->M[wave] = Mw;
        sc->J[wave] = jw;
    }
    __syncthreads();
    double M = sentinel;
    for (int w = 0; w < nwaves; ++w) M = WANT_MAX ? fmax(M, sc->M[w]) : fmin(M, sc->M[w]);
    int jM = INT_MAX;
    for (int w = 0; w < nwaves; ++w)
        if (sc->M[w] == M && sc->J[w] < jM) jM = sc->J[w];
    if (jM == INT_MAX) {   // nothing eligible
        best = sentinel;
        __syncthreads();   // (sc may be reused by the caller's next scan)
        return -1;
    }
    int near = 0;
    for (int j = tid; j < jM; j += T)
        if (!beats<WANT_MAX>(M, vals[j], eps)) near = 1;
    if (!__syncthreads_or(near)) {
        best = M;
        return jM;
    }
    if (wave == 0) {   // near-ties within eps in front of the extreme: the exact replay
        double b;
        auto load = [&](int j, bool& ok) {
            ok = true;
            return vals[j];
        };
        const int r = wave_chain_select<WANT_MAX>(len, eps, b, load);
        if (lane == 0) {
            sc->sel = r;
            sc->best = b;
        }
    }
    __syncthreads();
    best = sc->best;
    const int r = sc->sel;
    __syncthreads();
    return r;
}

template <bool WANT_MAX>
__device__ __forceinline__ double ext2(double a, double b) {
    return WANT_MAX ? (a > b ? a : b) : (a < b ? a : b);
}

// ---------------------------------------------------------------------------
// Narrow and short-cut forms (round 3).  A lone reducing wave pays every dependent instruction at
// ~8 cycles, so (1) a reduction over the first N lanes only (N a power of two; lanes >= N must hold
// the identity) stops after log2 N DPP steps and reads lane N - 1; (2) a 64-bit key extreme reduces
// the HIGH words first and, when exactly one lane holds the extreme high word (the common case: two
// candidates agreeing in sign, exponent and 20 mantissa bits are rare), takes the low word straight
// from that lane — one 6-step pass instead of two.  Ties on the high word take the second pass, so
// the result is the exact extreme either way.
// ---------------------------------------------------------------------------
#define LP_DPP_STEP(OP, CTRL) \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 " CTRL " bank_mask:0xf" : "+v"(v))
template <bool WANT_MAX, int N>
__device__ __forceinline__ unsigned wave_ext_u32_n(unsigned v) {
    static_assert(N == 2 || N == 4 || N == 8 || N == 16 || N == 32 || N == 64, "N: a power of two, 2..64");
    if (WANT_MAX) {
        LP_DPP_STEP("v_max_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf");
        if (N >= 4) LP_DPP_STEP("v_max_u32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf");
        if (N >= 8) LP_DPP_STEP("v_max_u32_dpp", "row_half_mirror row_mask:0xf");
        if (N >= 16) LP_DPP_STEP("v_max_u32_dpp", "row_mirror row_mask:0xf");
        if (N >= 32) LP_DPP_STEP("v_max_u32_dpp", "row_bcast:15 row_mask:0xa");
        if (N >= 64) LP_DPP_STEP("v_max_u32_dpp", "row_bcast:31 row_mask:0xc");
    } else {
        LP_DPP_STEP("v_min_u32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf");
        if (N >= 4) LP_DPP_STEP("v_min_u32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf");
        if (N >= 8) LP_DPP_STEP("v_min_u32_dpp", "row_half_mirror row_mask:0xf");
        if (N >= 16) LP_DPP_STEP("v_min_u32_dpp", "row_mirror row_mask:0xf");
        if (N >= 32) LP_DPP_STEP("v_min_u32_dpp", "row_bcast:15 row_mask:0xa");
        if (N >= 64) LP_DPP_STEP("v_min_u32_dpp", "row_bcast:31 row_mask:0xc");
    }
    asm volatile("s_nop 1" ::: "memory");
    return (unsigned)__builtin_amdgcn_readlane((int)v, N - 1);
}
#undef LP_DPP_STEP

// Extreme of 64-bit sortable keys over the first N lanes (lanes >= N: the identity key).  Returns the
// extreme key (wave-uniform); *hits = the lanes holding it.
template <bool WANT_MAX, int N>
__device__ __forceinline__ unsigned long long wave_ext_key_n(unsigned long long key, unsigned long long* hits) {
    const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    const unsigned mhi = wave_ext_u32_n<WANT_MAX, N>(hi);
    const unsigned long long cand = __ballot(hi == mhi);
    unsigned mlo;
    if ((cand & (cand - 1)) == 0ULL) {   // exactly one lane (cand != 0: the extreme is somebody's)
        mlo = (unsigned)__builtin_amdgcn_readlane((int)lo, (int)__builtin_ctzll(cand));
        *hits = cand;
    } else {
        const unsigned lo2 = (hi == mhi) ? lo : (WANT_MAX ? 0u : 0xFFFFFFFFu);
        mlo = wave_ext_u32_n<WANT_MAX, N>(lo2);
        *hits = __ballot(hi == mhi && lo == mlo);
    }
    return ((unsigned long long)mhi << 32) | mlo;
}

// block_select_stage1 / stage2 on those: slices of 64 entries (one per thread), NWMAX = upper bound
// of the number of waves (power of two).  Same verdicts as the forms above.
template <bool WANT_MAX>
__device__ __forceinline__ void block_select_stage1_n(double v, double eps, BlockSelScratch* sc) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned long long skey = f64_sort_key(WANT_MAX ? -INFINITY : INFINITY);
    const unsigned long long key = f64_sort_key(v);
    unsigned long long hit;
    const unsigned long long mk = wave_ext_key_n<WANT_MAX, 64>(key, &hit);
    const bool any = mk != skey;
    const int L = any ? (int)__builtin_ctzll(hit) : 64;
    const unsigned long long near = __ballot(lane < L && !beats<WANT_MAX>(f64_from_key(mk), v, eps));
    if (lane == 0) {
        sc->M[wave] = mk;
        sc->ok[wave] = (any && near == 0ULL) ? 1 : 0;
        sc->J[wave] = any ? wave * 64 + L : INT_MAX;
    }
}

template <bool WANT_MAX, int NWMAX, int K>
__device__ __forceinline__ int block_select_stage2_n(const double* vals, int len, double eps,
                                                     const BlockSelScratch* sc) {
    const int lane = threadIdx.x & 63, nwaves = (int)(blockDim.x >> 6);
    const unsigned long long skey = f64_sort_key(WANT_MAX ? -INFINITY : INFINITY);
    const bool has = lane < nwaves;
    const unsigned long long Ml = has ? sc->M[lane & 15] : skey;
    const int okl = has ? sc->ok[lane & 15] : 0;
    const int Jl = has ? sc->J[lane & 15] : INT_MAX;
    unsigned long long hit;
    const unsigned long long M = wave_ext_key_n<WANT_MAX, NWMAX>(Ml, &hit);
    if (M == skey) return -1;
    const int W = (int)__builtin_ctzll(hit);   // first slice attaining the extreme (slices are in row order)
    const int jM = __builtin_amdgcn_readlane(Jl, W);
    const int okW = __builtin_amdgcn_readlane(okl, W);
    const double Md = f64_from_key(M);
    const unsigned long long near = __ballot(lane < W && !beats<WANT_MAX>(Md, f64_from_key(Ml), eps));
    if (okW && near == 0ULL) return jM;
    double best;   // near-tie: exact replay over the LDS copy
    auto load = [&](int j, bool& ok) {
        ok = true;
        return vals[j];
    };
    return wave_chain_select<WANT_MAX, K>(len, eps, best, load);
}

// ---------------------------------------------------------------------------
// Quotients by a wave-uniform denominator (round 4: the pivot element u_r divides three times per pivot — 1/u_r,
// -d_e/u_r, -u_i/u_r — and an fp64 division is ~35 dependent instructions, ~320 cycles of a lone wave's time).
// The compiler's division is: scale both operands (v_div_scale_f64 x 2), reciprocal of the denominator refined
// twice, quotient, residual, correction (v_div_fmas_f64), special cases (v_div_fixup_f64).  For operands whose
// magnitudes lie in [2^-500, 2^501) nothing is scaled and nothing is special: v_div_fmas is then a plain fma and the
// sequence below IS that sequence, instruction for instruction — the same bits as `num / den`
// (tests/test_gpu_simplex.py::test_midrange_division_matches_division; enum_leaf.hip's recip_midrange is its
// num = 1 case).  The refined reciprocal depends on the denominator alone: it is computed once, while the
// numerators are still on their way.  A zero numerator keeps its quotient num * r2 (a zero of the right sign; the
// correction step would turn -0 into +0).  Operands outside the range: the caller divides plainly.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double mid_recip2(double den) {
    const double r0 = __builtin_amdgcn_rcp(den);
    const double r1 = fma(fma(-den, r0, 1.0), r0, r0);
    return fma(fma(-den, r1, 1.0), r1, r1);
}
__device__ __forceinline__ double mid_div(double num, double den, double r2) {
    const double q = num * r2;
    const double e = fma(-den, q, num);
    const double q2 = fma(e, r2, q);
    return (num == 0.0) ? q : q2;
}
__device__ __forceinline__ bool mid_range(double v) {   // 2^-500 <= |v| < 2^501 (not zero, denormal, inf, NaN)
    const unsigned ex = ((unsigned)__double2hiint(v) >> 20) & 0x7FFu;
    return ex - 523u <= 1000u;
}
__device__ __forceinline__ bool mid_range_or_zero(double v) { return v == 0.0 || mid_range(v); }

// num / den for every active lane — the same bits — by the sequence above while every active lane's operands are
// inside its range, by the plain division otherwise (one wave-uniform branch)
__device__ __forceinline__ double div_midrange(double num, double den) {
    const double q = mid_div(num, den, mid_recip2(den));
    if (__builtin_expect(!__all(mid_range(den) && mid_range_or_zero(num)), 0)) return num / den;
    return q;
}

}  // namespace lpdev
