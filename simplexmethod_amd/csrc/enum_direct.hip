// enum_direct.hip — vertex enumeration, one independent m x m solve per basis subset
// (LP_ENUM_ALGO_DIRECT).
//
// EnumerationSolver is an empty stub in the reference (src/EnumerationSolver.h:3-10;
// spec README.md:27,40-42); the per-basis step it needs is what
// Canonical::GetBasicSolution / IsFeasibleBasis / Evaluate do for one basis
// (src/ProblemTypes/Canonical.cpp:165-197, :79-87).  Semantics: SURVEY.md §8 row E1,
// operation order: oracle/lp_oracle.c (orc_enum_subset) — replayed here bit for bit.
//
// Mapping: G lanes (16 or 32) of a wave own one subset; lane i holds row i of
// W = [A[:,S] | b] in registers (column index static).  Partial pivoting is a
// G-lane butterfly arg-max; the pivot row is broadcast lane->group; rows are never
// moved (a lane is simply marked used).  A (<= 4 KiB) sits in LDS with an odd row
// stride so the row gather is bank-conflict-free.  Each group walks a contiguous
// range of combination ranks: unrank once, then lexicographic successors.
#include <cfloat>

#include "enum_problem.hpp"

namespace {

struct GroupBest {
    double score;  // best score seen by this group (score = z if maximize else -z)
};

template <int G>
struct SubsetSolve {
    double xrow;     // used lanes: value of the variable pivoted in this row (back-substituted)
    double xa, xb;   // values of the last two columns (2x2 block), uniform in the group
    int P[G];        // P[t] = group-relative lane that became the pivot row of step t < m-2
    int sa, sb;      // the last two columns of the subset
    bool singular;
    bool feasible;
};

// Lexicographic unranking of `rank` into the sorted subset S (combinatorial number system).
template <int G>
__device__ __forceinline__ void unrank_subset(const EnumDev& d, unsigned long long rank, int (&S)[G]) {
    int a = 0;
#pragma unroll
    for (int t = 0; t < G; ++t) {
        S[t] = 0;
        if (t < d.m) {
            int j = a;
            for (;; ++j) {
                const unsigned long long cnt = d.binom[(d.n - 1 - j) * kBinomK + (d.m - 1 - t)];
                if (rank < cnt) break;
                rank -= cnt;
            }
            S[t] = j;
            a = j + 1;
        }
    }
}

template <int G>
__device__ __forceinline__ void next_subset(int m, int n, int (&S)[G]) {
    int tpos = -1;
#pragma unroll
    for (int t = 0; t < G; ++t)
        if (t < m && S[t] < n - m + t) tpos = t;
#pragma unroll
    for (int t = 0; t < G; ++t) {
        if (t == tpos)
            S[t] += 1;
        else if (t > tpos && t < m && t > 0)
            S[t] = S[t - 1] + 1;
    }
}

// The build-defined per-subset solve; see orc_enum_subset (oracle/lp_oracle.c) for the
// operation order this replays: Gauss-Jordan with partial pivoting on the first m-2 columns,
// then the 2x2 block of the last two columns on the two unused rows, then back-substitution.
template <int G>
__device__ __forceinline__ void solve_subset(const EnumDev& d, const double* sA, const double* sb,
                                             const int (&S)[G], int gl /* lane in group */,
                                             SubsetSolve<G>& out) {
    const int m = d.m;
    const int g = m - 2;  // Gauss-Jordan steps (m == 1 is handled at the end)
    const bool active = gl < m;
    int sa = 0, sbc = 0;
#pragma unroll
    for (int t = 0; t < G; ++t) {
        if (t == m - 2) sa = S[t];
        if (t == m - 1) sbc = S[t];
    }
    double W[G];
#pragma unroll
    for (int t = 0; t < G; ++t) W[t] = (active && t < g) ? sA[gl * d.lda + S[t]] : 0.0;
    double Wa = (active && m >= 2) ? sA[gl * d.lda + sa] : 0.0;
    double Wb = active ? sA[gl * d.lda + sbc] : 0.0;
    double rhs = active ? sb[gl] : 0.0;
    bool used = !active;
    bool sing = false;
    double minp = INFINITY, maxp = 0.0;
#pragma unroll
    for (int t = 0; t < G; ++t) {
        out.P[t] = 0;
        if (t < g) {
            // (a NaN entry is never the maximum — oracle: `a > big` is false for it; here it must not
            // enter the butterfly either, where it would survive every comparison)
            const double aw = fabs(W[t]);
            double a = (used || !(aw >= 0.0)) ? -1.0 : aw;
            int idx = gl;
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) {
                const double oa = __shfl_xor(a, off, G);
                const int oi = __shfl_xor(idx, off, G);
                if (oa > a || (oa == a && oi < idx)) {
                    a = oa;
                    idx = oi;
                }
            }
            const int p = idx;
            if (!(a > 0.0)) sing = true;
            minp = fmin(minp, a);
            maxp = fmax(maxp, a);
            const double piv = __shfl(W[t], p, G);
            const double inv = 1.0 / piv;
            const double l = -(W[t] * inv);
            const bool isp = (gl == p);
#pragma unroll
            for (int c = t + 1; c < G; ++c) {
                if (c < g) {
                    const double pc = __shfl(W[c], p, G);
                    W[c] = isp ? pc * inv : fma(l, pc, W[c]);
                }
            }
            const double pa = __shfl(Wa, p, G);
            Wa = isp ? pa * inv : fma(l, pa, Wa);
            const double pb = __shfl(Wb, p, G);
            Wb = isp ? pb * inv : fma(l, pb, Wb);
            const double pr = __shfl(rhs, p, G);
            rhs = isp ? pr * inv : fma(l, pr, rhs);
            if (isp) used = true;
            out.P[t] = p;
        }
    }
    // ---- the two unused rows r1 < r2 and the 2x2 block
    const int lane = threadIdx.x & 63;
    const int gbase = lane & ~(G - 1);
    const unsigned long long gmask = (G == 64) ? ~0ULL : (((1ULL << G) - 1ULL) << gbase);
    const unsigned long long um = (__ballot(!used) & gmask) >> gbase;
    double xa = 0.0, xb = 0.0, big1 = 0.0, big2 = 0.0;
    if (m >= 2) {
        const int r1 = um ? (int)__builtin_ctzll(um) : 0;
        const unsigned long long um2 = um & (um - 1);
        const int r2 = um2 ? (int)__builtin_ctzll(um2) : r1;
        const double a1 = __shfl(Wa, r1, G), a2 = __shfl(Wa, r2, G);
        const double b1 = __shfl(Wb, r1, G), b2 = __shfl(Wb, r2, G);
        const double h1 = __shfl(rhs, r1, G), h2 = __shfl(rhs, r2, G);
        const bool second = fabs(a2) > fabs(a1);
        const double pa = second ? a2 : a1, pb = second ? b2 : b1, ph = second ? h2 : h1;
        const double qa = second ? a1 : a2, qb = second ? b1 : b2, qh = second ? h1 : h2;
        big1 = fabs(pa);
        const double inv1 = 1.0 / pa;
        const double l = -(qa * inv1);
        const double wqb = fma(l, pb, qb);
        const double rq = fma(l, ph, qh);
        big2 = fabs(wqb);
        const double inv2 = 1.0 / wqb;
        xb = rq * inv2;
        xa = fma(-pb, xb, ph) * inv1;
        if (!(big1 > 0.0) || !(big2 > 0.0)) sing = true;
        minp = fmin(minp, fmin(big1, big2));
        maxp = fmax(maxp, fmax(big1, big2));
        rhs = fma(-Wb, xb, fma(-Wa, xa, rhs));  // back-substitution (meaningful on used rows)
    } else {  // m == 1: a single pivot
        const double piv = __shfl(Wb, 0, G), h = __shfl(rhs, 0, G);
        big1 = fabs(piv);
        if (!(big1 > 0.0)) sing = true;
        minp = maxp = big1;
        xb = h * (1.0 / piv);
    }
    if (minp <= DBL_EPSILON * (double)m * maxp) sing = true;
    out.singular = sing;
    out.xrow = rhs;
    out.xa = xa;
    out.xb = xb;
    out.sa = sa;
    out.sb = sbc;
    const bool isused = used && active;
    bool ok = !isused || (rhs >= -1e-9);  // Canonical.cpp:171; NaN is infeasible
    const unsigned long long bal = __ballot(ok);
    out.feasible = ((bal & gmask) == gmask) && (xb >= -1e-9) && (m < 2 || xa >= -1e-9);
}

template <int G>
__device__ __forceinline__ double subset_objective(const EnumDev& d, const double* sc,
                                                   const int (&S)[G], const SubsetSolve<G>& s) {
    double z = 0.0;
#pragma unroll
    for (int t = 0; t < G; ++t) {
        if (t < d.m - 2) {
            const double xv = __shfl(s.xrow, s.P[t], G);
            z = fma(sc[S[t]], xv, z);  // Canonical.cpp:86, ascending column order
        }
    }
    if (d.m >= 2) z = fma(sc[s.sa], s.xa, z);
    z = fma(sc[s.sb], s.xb, z);
    return z;
}

__device__ __forceinline__ void stage_problem(const EnumDev& d, double* sA, double* sb, double* sc) {
    for (int k = threadIdx.x; k < d.m * d.lda; k += blockDim.x) sA[k] = d.A[k];
    for (int k = threadIdx.x; k < d.m; k += blockDim.x) sb[k] = d.b[k];
    for (int k = threadIdx.x; k < d.n; k += blockDim.x) sc[k] = d.c[k];
    __syncthreads();
}

// MODE 0: pass 1 (best score + counts + per-chunk best); MODE 1: pass 2 (first rank
// with score >= star - tol).
template <int G, int MODE>
__global__ __launch_bounds__(256) void k_enum_direct(EnumDev d, unsigned long long begin,
                                                     unsigned long long end,
                                                     unsigned long long per_chunk, double star,
                                                     double tol) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sA = smem;
    double* sb = sA + d.m * d.lda;
    double* sc = sb + d.m;
    unsigned long long* sred = reinterpret_cast<unsigned long long*>(sc + d.n);  // 4 words
    if (threadIdx.x < 4) sred[threadIdx.x] = (threadIdx.x == 0) ? lp_f64_key(-INFINITY) : 0ULL;
    stage_problem(d, sA, sb, sc);

    const int gl = threadIdx.x & (G - 1);
    const unsigned long long chunk =
        (unsigned long long)blockIdx.x * (blockDim.x / G) + (threadIdx.x / G);
    unsigned long long k0 = begin + chunk * per_chunk;
    unsigned long long k1 = k0 + per_chunk;
    if (k1 > end) k1 = end;

    double best = -INFINITY;
    unsigned long long cnt0 = 0, cnt1 = 0, cnt2 = 0;
    unsigned long long first = ~0ULL;
    if (k0 < k1) {  // uniform per group; groups of one wave may diverge here
        int S[G];
        unrank_subset<G>(d, k0, S);
        for (unsigned long long k = k0; k < k1; ++k) {
            SubsetSolve<G> s;
            solve_subset<G>(d, sA, sb, S, gl, s);
            if (s.singular) {
                ++cnt2;
            } else if (!s.feasible) {
                ++cnt1;
            } else {
                ++cnt0;
                const double z = subset_objective<G>(d, sc, S, s);
                const double score = d.maximize ? z : -z;
                if (MODE == 0) {
                    if (score > best) best = score;
                } else {
                    if (score >= star - tol && first == ~0ULL) first = k;
                }
            }
            next_subset<G>(d.m, d.n, S);
        }
    }
    if (MODE == 0) {
        if (gl == 0) {
            if (k0 < end) d.chunk_best[chunk] = best;
            atomicMax(&sred[0], lp_f64_key(best));
            if (cnt0) atomicAdd(&sred[1], cnt0);
            if (cnt1) atomicAdd(&sred[2], cnt1);
            if (cnt2) atomicAdd(&sred[3], cnt2);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicMax(&d.result->best_key, sred[0]);
            if (sred[1]) atomicAdd(&d.result->counts[0], sred[1]);
            if (sred[2]) atomicAdd(&d.result->counts[1], sred[2]);
            if (sred[3]) atomicAdd(&d.result->counts[2], sred[3]);
        }
    } else {
        if (gl == 0 && first != ~0ULL) atomicMin(&d.result->first_rank, first);
    }
}

// One subset, one group: writes xB (by sorted column), the subset, objective, verdict.
template <int G>
__global__ __launch_bounds__(64) void k_enum_vertex(EnumDev d, unsigned long long rank, double* vx,
                                                    int* vi) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sA = smem;
    double* sb = sA + d.m * d.lda;
    double* sc = sb + d.m;
    stage_problem(d, sA, sb, sc);
    if (threadIdx.x >= G) return;
    const int gl = threadIdx.x;
    int S[G];
    unrank_subset<G>(d, rank, S);
    SubsetSolve<G> s;
    solve_subset<G>(d, sA, sb, S, gl, s);
    const double z = subset_objective<G>(d, sc, S, s);
#pragma unroll
    for (int t = 0; t < G; ++t) {
        if (t < d.m) {
            double xv = __shfl(s.xrow, s.P[t], G);
            if (t == d.m - 2) xv = s.xa;
            if (t == d.m - 1) xv = s.xb;
            if (gl == 0) {
                vx[t] = xv;
                vi[t] = S[t];
            }
        }
    }
    if (gl == 0) {
        vx[kEnumMaxM] = z;
        vi[kEnumMaxM] = s.singular ? LP_SUBSET_SINGULAR
                                   : (s.feasible ? LP_SUBSET_FEASIBLE : LP_SUBSET_INFEASIBLE);
    }
}

// Objectives of listed ranks (the shared-prefix path's feasible subsets): one group per entry.
template <int G>
__global__ __launch_bounds__(256) void k_enum_eval_list(EnumDev d, const unsigned long long* list,
                                                        unsigned long long count,
                                                        const unsigned long long* count_ptr,
                                                        unsigned long long cap, double* scores) {
    // count_ptr != null: the list was filled by kernels queued just before this one and its length
    // still lives on the device (an over-full list is reported by the host afterwards)
    if (count_ptr) count = *count_ptr < cap ? *count_ptr : cap;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sA = smem;
    double* sb = sA + d.m * d.lda;
    double* sc = sb + d.m;
    stage_problem(d, sA, sb, sc);
    const int gl = threadIdx.x & (G - 1);
    const unsigned long long groups = (unsigned long long)gridDim.x * (blockDim.x / G);
    unsigned long long e = (unsigned long long)blockIdx.x * (blockDim.x / G) + (threadIdx.x / G);
    double best = -INFINITY;
    for (; e < count; e += groups) {
        int S[G];
        unrank_subset<G>(d, list[e], S);
        SubsetSolve<G> s;
        solve_subset<G>(d, sA, sb, S, gl, s);
        const double z = subset_objective<G>(d, sc, S, s);
        // listed subsets are feasible by construction; a verdict mismatch would be a bug and
        // shows up as -inf here
        const double score = (!s.singular && s.feasible) ? (d.maximize ? z : -z) : -INFINITY;
        if (gl == 0) scores[e] = score;
        if (score > best) best = score;
    }
    if (gl == 0 && best > -INFINITY) atomicMax(&d.result->best_key, lp_f64_key(best));
}

// list == null: the scores are indexed by rank - base (the dense form of a degenerate LP's pass)
__global__ void k_enum_list_first(EnumDev d, const unsigned long long* list, unsigned long long count,
                                  const unsigned long long* count_ptr, unsigned long long cap,
                                  const double* scores, double star, double tol, unsigned long long base,
                                  int star_on_device) {
    // count_ptr != null: queued behind the evaluation kernel — the list length is read on the device;
    // star_on_device: so is the best score (the reference value of the tie rule)
    if (count_ptr) count = *count_ptr < cap ? *count_ptr : cap;
    if (star_on_device) star = lp_key_f64(d.result->best_key);
    unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long first = ~0ULL;
    for (; e < count; e += stride) {
        const unsigned long long rk = list ? list[e] : base + e;
        if (scores[e] >= star - tol && rk < first) first = rk;
    }
    if (first != ~0ULL) atomicMin(&d.result->first_rank, first);
}

size_t enum_smem_bytes(const EnumDev& d) {
    return sizeof(double) * (size_t)(d.m * d.lda + d.m + d.n) + 4 * sizeof(unsigned long long) + 16;
}

}  // namespace

template <int MODE>
static int launch_direct(lp_enum_problem* p, uint64_t begin, uint64_t end, double star, double tol,
                         uint64_t* per_chunk_out, int* chunks_out) {
    lp_context* ctx = p->ctx;
    const EnumDev& d = p->dev;
    const int G = d.m <= 16 ? 16 : 32;
    const int block = 256;
    const int groups_per_block = block / G;
    const uint64_t count = end - begin;
    // enough groups to fill the chip several times over, with ~64 subsets per group to amortise the
    // unranking and the block's copy of A — but a SMALL range is spread over every CU first: a group's
    // subsets are a serial chain (~10 us each at m = 16), and 64 of them in a row were a 0.7-1.7 ms floor
    // under every small problem (C(13,6) = 1716 subsets, the reference's own examples)
    uint64_t want_groups = (uint64_t)ctx->num_cus * 8 * groups_per_block;
    if (want_groups > (uint64_t)p->chunk_cap) want_groups = p->chunk_cap;
    uint64_t per_chunk = lp_ceil_div<uint64_t>(count, want_groups);
    if (per_chunk < 64) {
        per_chunk = lp_ceil_div<uint64_t>(count, (uint64_t)ctx->num_cus * groups_per_block);   // one block per CU first
        if (per_chunk > 64) per_chunk = 64;
        if (per_chunk < 1) per_chunk = 1;
    }
    const uint64_t chunks = lp_ceil_div<uint64_t>(count, per_chunk);
    const unsigned grid = (unsigned)lp_ceil_div<uint64_t>(chunks, groups_per_block);
    const size_t shm = enum_smem_bytes(d);
    if (G == 16)
        hipLaunchKernelGGL((k_enum_direct<16, MODE>), grid, block, shm, ctx->stream, d, begin, end,
                           per_chunk, star, tol);
    else
        hipLaunchKernelGGL((k_enum_direct<32, MODE>), grid, block, shm, ctx->stream, d, begin, end,
                           per_chunk, star, tol);
    if (per_chunk_out) *per_chunk_out = per_chunk;
    if (chunks_out) *chunks_out = (int)chunks;
    return LP_OPTIMAL;
}

static int reset_result(lp_enum_problem* p) {
    EnumResult r;
    std::memset(&r, 0, sizeof(r));
    r.best_key = lp_f64_key(-INFINITY);
    r.first_rank = ~0ULL;
    *p->h_result = r;
    LP_HIP(p->ctx, hipMemcpyAsync(p->dev.result, p->h_result, sizeof(r), hipMemcpyHostToDevice,
                                  p->ctx->stream));
    return LP_OPTIMAL;
}

static int fetch_result(lp_enum_problem* p) {
    LP_HIP(p->ctx, hipMemcpyAsync(p->h_result, p->dev.result, sizeof(EnumResult),
                                  hipMemcpyDeviceToHost, p->ctx->stream));
    LP_HIP(p->ctx, hipStreamSynchronize(p->ctx->stream));
    LP_HIP(p->ctx, hipGetLastError());
    return LP_OPTIMAL;
}

int lp_enum_direct_range(lp_enum_problem* p, uint64_t begin, uint64_t end, double* score_best,
                         uint64_t counts[3], lp_enum_stats* stats) {
    lp_context* ctx = p->ctx;
    int rc = reset_result(p);
    if (rc) return rc;
    uint64_t per_chunk = 0;
    int chunks = 0;
    LP_HIP(ctx, hipEventRecord(p->ev0, ctx->stream));
    if (end > begin) launch_direct<0>(p, begin, end, 0.0, 0.0, &per_chunk, &chunks);
    LP_HIP(ctx, hipEventRecord(p->ev1, ctx->stream));
    rc = fetch_result(p);
    if (rc) return rc;
    float ms = 0.f;
    LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
    p->last_begin = begin;
    p->last_end = end;
    p->last_per_chunk = per_chunk;
    p->last_chunks = chunks;
    *score_best = lp_key_f64(p->h_result->best_key);
    for (int k = 0; k < 3; ++k) counts[k] = p->h_result->counts[k];
    if (stats) {
        stats->kernel_ms = ms;
        stats->subsets = end - begin;
        stats->launches = end > begin ? 1 : 0;
    }
    return LP_OPTIMAL;
}

int lp_enum_direct_first(lp_enum_problem* p, uint64_t begin, uint64_t end, double score_star,
                         double tol, uint64_t* rank_out) {
    lp_context* ctx = p->ctx;
    *rank_out = UINT64_MAX;
    if (end <= begin) return LP_OPTIMAL;
    // Narrow to the first chunk of the cached pass 1 whose best score qualifies.
    if (begin == p->last_begin && end == p->last_end && p->last_chunks > 0) {
        p->h_chunk_best.resize((size_t)p->last_chunks);
        LP_HIP(ctx, hipMemcpyAsync(p->h_chunk_best.data(), p->dev.chunk_best,
                                   sizeof(double) * (size_t)p->last_chunks, hipMemcpyDeviceToHost,
                                   ctx->stream));
        LP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int g = 0;
        while (g < p->last_chunks && !(p->h_chunk_best[(size_t)g] >= score_star - tol)) ++g;
        if (g == p->last_chunks) return LP_OPTIMAL;
        const uint64_t nb = begin + (uint64_t)g * p->last_per_chunk;
        uint64_t ne = nb + p->last_per_chunk;
        if (ne > end) ne = end;
        begin = nb;
        end = ne;
    }
    int rc = reset_result(p);
    if (rc) return rc;
    launch_direct<1>(p, begin, end, score_star, tol, nullptr, nullptr);
    rc = fetch_result(p);
    if (rc) return rc;
    // the narrowing above invalidated the chunk cache geometry for a second narrowing
    *rank_out = p->h_result->first_rank;
    return LP_OPTIMAL;
}

int lp_enum_direct_vertex(lp_enum_problem* p, uint64_t rank, double* xB, int* subset, double* z,
                          int* verdict) {
    lp_context* ctx = p->ctx;
    const EnumDev& d = p->dev;
    const size_t shm = enum_smem_bytes(d);
    if (d.m <= 16)
        hipLaunchKernelGGL((k_enum_vertex<16>), 1, 64, shm, ctx->stream, d, rank, p->dvx, p->dvi);
    else
        hipLaunchKernelGGL((k_enum_vertex<32>), 1, 64, shm, ctx->stream, d, rank, p->dvx, p->dvi);
    double hx[kEnumMaxM + 1];
    int hi[kEnumMaxM + 1];
    LP_HIP(ctx, hipMemcpyAsync(hx, p->dvx, sizeof(hx), hipMemcpyDeviceToHost, ctx->stream));
    LP_HIP(ctx, hipMemcpyAsync(hi, p->dvi, sizeof(hi), hipMemcpyDeviceToHost, ctx->stream));
    LP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LP_HIP(ctx, hipGetLastError());
    for (int t = 0; t < d.m; ++t) {
        xB[t] = hx[t];
        subset[t] = hi[t];
    }
    *z = hx[kEnumMaxM];
    *verdict = hi[kEnumMaxM];
    return LP_OPTIMAL;
}

// Queues, behind the kernels that fill the feasible list, its evaluation and the tie rule against
// the list's own best score (tolerance tol) — no host round trip: the caller synchronises once and
// finds best_key and first_rank in the result block.
int lp_enum_queue_list_tail(lp_enum_problem* p, double tol, const double* records) {
    lp_context* ctx = p->ctx;
    const EnumDev& d = p->dev;
    const PrefixDev& pd = p->prefix;
    const size_t shm = enum_smem_bytes(d);
    const unsigned grid = (unsigned)ctx->num_cus * 2;
    if (records)
        lp_enum_queue_record_eval(p, records);
    else if (d.m <= 16)
        hipLaunchKernelGGL((k_enum_eval_list<16>), grid, 256, shm, ctx->stream, d, pd.list, 0ULL,
                           (const unsigned long long*)pd.list_count, (unsigned long long)pd.list_cap, pd.scores);
    else
        hipLaunchKernelGGL((k_enum_eval_list<32>), grid, 256, shm, ctx->stream, d, pd.list, 0ULL,
                           (const unsigned long long*)pd.list_count, (unsigned long long)pd.list_cap, pd.scores);
    hipLaunchKernelGGL(k_enum_list_first, (unsigned)ctx->num_cus * 4, 256, 0, ctx->stream, d, pd.list, 0ULL,
                       (const unsigned long long*)pd.list_count, (unsigned long long)pd.list_cap, pd.scores, 0.0, tol,
                       0ULL, 1);
    return LP_OPTIMAL;
}

// Dense form: tie rule over the rank-indexed scores of [begin, end), against the device's best score.
int lp_enum_queue_dense_tail(lp_enum_problem* p, double tol, uint64_t begin, uint64_t end) {
    lp_context* ctx = p->ctx;
    hipLaunchKernelGGL(k_enum_list_first, (unsigned)ctx->num_cus * 4, 256, 0, ctx->stream, p->dev,
                       (const unsigned long long*)nullptr, (unsigned long long)(end - begin),
                       (const unsigned long long*)nullptr, 0ULL, p->prefix.dense_scores, 0.0, tol,
                       (unsigned long long)begin, 1);
    return LP_OPTIMAL;
}

int lp_enum_list_first(lp_enum_problem* p, double score_star, double tol, uint64_t* rank_out) {
    lp_context* ctx = p->ctx;
    *rank_out = UINT64_MAX;
    if (p->list_n == 0) return LP_OPTIMAL;
    int rc = reset_result(p);
    if (rc) return rc;
    const uint64_t entries = p->dense_active ? p->list_end - p->list_begin : p->list_n;
    const unsigned grid = (unsigned)std::min<uint64_t>(lp_ceil_div<uint64_t>(entries, 256), 1024);
    if (p->dense_active)
        hipLaunchKernelGGL(k_enum_list_first, grid, 256, 0, ctx->stream, p->dev, (const unsigned long long*)nullptr,
                           (unsigned long long)(p->list_end - p->list_begin), (const unsigned long long*)nullptr, 0ULL,
                           p->prefix.dense_scores, score_star, tol, (unsigned long long)p->list_begin, 0);
    else
        hipLaunchKernelGGL(k_enum_list_first, grid, 256, 0, ctx->stream, p->dev, p->prefix.list,
                           (unsigned long long)p->list_n, (const unsigned long long*)nullptr, 0ULL, p->prefix.scores,
                           score_star, tol, 0ULL, 0);
    rc = fetch_result(p);
    if (rc) return rc;
    *rank_out = p->h_result->first_rank;
    return LP_OPTIMAL;
}
