// enum_problem.hpp — device-side state of one vertex enumeration.
#pragma once

#include "lp_internal.hpp"

constexpr int kEnumMaxN = 64;   // ranks must fit u64: C(64,32) < 2^64
constexpr int kEnumMaxM = 32;
constexpr int kBinomK = kEnumMaxM + 2;  // columns of the binomial table

// Monotone double -> u64 key (so atomicMax on the key is max on the double).
__host__ __device__ inline unsigned long long lp_f64_key(double v) {
    unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (unsigned long long)__double_as_longlong(v);
#else
    std::memcpy(&b, &v, sizeof(b));
#endif
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ inline double lp_key_f64(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)b);
#else
    double v;
    std::memcpy(&v, &b, sizeof(v));
    return v;
#endif
}

// Results of one pass, accumulated with device-scope atomics.
struct EnumResult {
    unsigned long long best_key;    // key of the best score (score = z if maximize else -z)
    unsigned long long counts[3];   // feasible, infeasible, singular
    unsigned long long first_rank;  // pass 2: smallest qualifying rank
    unsigned long long range_flag;  // leaf kernels, fast reciprocal: some pivot left its range — the pass is repeated with plain divisions
    unsigned long long pad[2];
};

// What the host reads back after a shared-prefix pass, in ONE device block and one pinned mirror (one copy per
// pass instead of four: each small device-to-host copy is a 5-20 us blit on the stream, a tenth of the fixed cost
// of an 8-way shard).  EnumDev::result, PrefixDev::list_count / overflow / level_counts point into the device block.
struct EnumPassBlock {
    EnumResult result;
    unsigned long long list_count;
    int overflow;
    int pad;
    int level_counts[32];
};

struct EnumDev {
    int m, n, lda;                  // lda = n + 1 (odd stride: conflict-free row gathers)
    int maximize;
    int rs;                         // row stride of 32-row prefix records: m rounded up to even (enum_tree.hpp)
    int pad0;
    const double* A;                // m x lda row-major copy of the canonical A (pad column 0)
    const double* b;                // m
    const double* c;                // n
    const unsigned long long* binom;  // (kEnumMaxN+1) x kBinomK table, C(i,k)
    EnumResult* result;
    double* chunk_best;             // per-chunk best score of the last pass 1
};

// Shared-prefix path (enum_prefix.hip)
struct PrefixDev {
    int* level_counts;                // [32]: records of each tree level (level 0 = 1), device side
    int* overflow;                    // != 0: a buffer was too small, the caller falls back
    int* root_cursor;                 // [2] work cursors (regular / thin leaf kernel, sweep groups)
    unsigned long long* list;         // ranks of feasible subsets
    int* list_rec;                    // record (last breadth-first level) each entry was found under
    unsigned long long* list_count;
    unsigned long long list_cap;
    unsigned long long list_abort;    // leaf kernels stop drawing items once the list count exceeds this
    double* scores;                   // objective score of each list entry (after evaluation)
    double* dense_scores;             // dense form (degenerate LPs): score of every subset of the range, by
    unsigned long long dense_cap;     // rank - begin (-inf: not feasible); no list
    int4* items;                      // leaf-kernel work items, table 0: (record, child column, first subset, rank offset)
    int4* items2;                     // table 1 (two-level kernel): (record, child | j2 << 8, first subset, rank offset)
    int* item_count;                  // [2]
    int item_cap, item_cap2;
    const unsigned* comb4;            // same for 4-subsets (third level of the leaf kernel)
    const unsigned* comb5;            // same for 5-subsets (second level of the leaf kernel)
    const unsigned* comb6;            // [32 offsets][entries]: all 6-subsets of R columns in lex order,
                                      // 5 bits per index; entry of leaf l of R columns = comb6[comb6[R] + l]
};

struct lp_enum_problem {
    lp_context* ctx = nullptr;
    bool complete = false;           // every allocation of lp_enum_upload succeeded (a shell worth keeping)
    EnumDev dev{};
    double* dA = nullptr;
    double* db = nullptr;
    double* dc = nullptr;
    unsigned long long* dbinom = nullptr;
    std::vector<double> hA, hb, hc;  // host copies (column-major A) for argument checks only
    EnumPassBlock* d_pass = nullptr; // the device block behind dev.result, prefix.list_count / overflow / level_counts
    EnumPassBlock* h_pass = nullptr; // pinned mirror; the four h_* pointers below point into it
    EnumResult* h_result = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // chunking of the last lp_enum_range call (pass 2 narrows to the first qualifying chunk)
    uint64_t last_begin = 0, last_end = 0, last_per_chunk = 0;
    int last_chunks = 0;
    int chunk_cap = 0;
    std::vector<double> h_chunk_best;
    // vertex scratch
    double* dvx = nullptr;  // kEnumMaxM xB values + 1 objective
    int* dvi = nullptr;     // kEnumMaxM subset + 1 verdict
    // shared-prefix path
    PrefixDev prefix{};
    double* prefix_buf[2] = {nullptr, nullptr};
    size_t prefix_buf_bytes[2] = {0, 0};
    int* h_item_count = nullptr;               // pinned
    int* h_level_counts = nullptr;             // pinned copy of the 32 level counts
    unsigned long long* h_list_count = nullptr;  // pinned
    int* h_overflow = nullptr;                 // pinned
    bool list_valid = false;                   // the feasible list of the last prefix pass 1 is usable
    uint64_t list_begin = 0, list_end = 0, list_n = 0;
    int last_algo = 0;
    uint64_t split_hint = 0;
    // dense form of a pass (enum_prefix.hip): chosen when a pass finds more than a third of its range
    // feasible; dense_hint keeps later passes of this problem from listing first
    bool dense_active = false, dense_hint = false;
    // the leaf kernels divide plainly (enum_leaf.hip: leaf_verdict): set for good once a pass of the fast
    // kernels met pivots outside the fast reciprocal's exponent range, or by LP_ENUM_EXACT_DIV=1 (A/B, tests)
    bool exact_div = false;
    // A range whose feasible subsets do not fit the list (degenerate LPs: up to every non-singular
    // basis is feasible) is enumerated in sub-ranges, one list at a time; pass 2 re-runs only the
    // sub-ranges whose best score can hold the winner.
    struct PrefixChunk {
        uint64_t begin, end;
        double best;     // best score of the sub-range (-inf: no feasible subset)
        bool direct;     // the sub-range ran on the direct kernel (no list)
    };
    std::vector<PrefixChunk> pchunks;
    bool pchunks_valid = false;
    uint64_t pchunks_begin = 0, pchunks_end = 0;
    // shard of the last lp_enum_solve_sharded call (enum_sharded.hip)
    int shard_rank = -1, shard_world = -1;
    uint64_t shard_lo = 0, shard_hi = 0;
    // tie rule already applied on the device against the range's own best score (prefix path)
    bool spec_valid = false;
    double spec_star = 0.0, spec_tol = 0.0;
    uint64_t spec_first = ~0ULL;
};

// enum_direct.hip
int lp_enum_direct_range(lp_enum_problem* p, uint64_t begin, uint64_t end, double* score_best,
                         uint64_t counts[3], lp_enum_stats* stats);
int lp_enum_direct_first(lp_enum_problem* p, uint64_t begin, uint64_t end, double score_star,
                         double tol, uint64_t* rank_out);
int lp_enum_direct_vertex(lp_enum_problem* p, uint64_t rank, double* xB, int* subset, double* z,
                          int* verdict);
// smallest listed rank whose score is within tol of score_star (UINT64_MAX if none)
int lp_enum_list_first(lp_enum_problem* p, double score_star, double tol, uint64_t* rank_out);
// evaluation + tie rule against the list's own best, queued without a host round trip; records !=
// null: the depth m-7 records of the pass (the entries are evaluated from them, enum_leaf.hip)
int lp_enum_queue_list_tail(lp_enum_problem* p, double tol, const double* records);
int lp_enum_queue_dense_tail(lp_enum_problem* p, double tol, uint64_t begin, uint64_t end);
void lp_enum_queue_record_eval(lp_enum_problem* p, const double* records);

// enum_leaf.hip: one lane per subset below the records of the last breadth-first level
// (shape: lp_enum_prefix_shape — 1 = the tuned kernels, 2 / 3 = the general kernel on 16- / 32-row records)
// dense: every subset's score goes to prefix.dense_scores[rank - begin] (general kernel), no list
int lp_enum_launch_leaves(lp_enum_problem* p, const double* roots, int bound, int level, bool fused,
                          int shape, bool dense, uint64_t begin, uint64_t end);

// enum_prefix.hip
bool lp_enum_prefix_supported(const lp_enum_problem* p);
int lp_enum_prefix_shape(const lp_enum_problem* p);
// enum_leaf.hip: recip_midrange(x[i]) and 1.0 / x[i] computed on the device (lp_debug_reciprocal)
int lp_enum_debug_reciprocal(lp_context* ctx, const double* x, int n, double* fast_out, double* plain_out);
// LP_ITER_LIMIT = "could not run here (memory / a level buffer too small), use the direct path";
// kEnumListOverflow = the feasible list was too small: *h_list_count holds the number of feasible
// subsets of the range, the caller splits the range (capi.hip: enum_prefix_chunked)
constexpr int kEnumListOverflow = 1001;
// kEnumRangeTooWide = the range has more depth m-7 nodes than the level buffers hold: the caller
// splits it into about p->split_hint parts
constexpr int kEnumRangeTooWide = 1002;
int lp_enum_prefix_range(lp_enum_problem* p, uint64_t begin, uint64_t end, double* score_best,
                         uint64_t counts[3], lp_enum_stats* stats, bool dense = false);
