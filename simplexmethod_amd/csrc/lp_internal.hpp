// lp_internal.hpp — shared host-side plumbing for the HIP hot path (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/simplexmethod_amd.h"

struct lp_context {
    int device = -1;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cus = 0;
    size_t total_mem = 0;   // device memory (queried on first use)
    std::string last_error;
    // Large device buffers released by freed problems, kept for the next one (hipMalloc / hipFree of
    // the enumeration's multi-GB level buffers cost milliseconds: more than a small solve).
    std::vector<std::pair<void*, size_t>> pool;
    // shape-independent subset tables of the enumeration's leaf kernels, built once per context
    unsigned* dcomb6 = nullptr;
    unsigned* dcomb5 = nullptr;
    unsigned* dcomb4 = nullptr;
    // two more streams + events (created on first use): the enumeration's independent leaf kernels run
    // side by side so that one kernel's tail is filled by the next one's head
    hipStream_t aux_stream[2] = {nullptr, nullptr};
    hipEvent_t aux_event[3] = {nullptr, nullptr, nullptr};
    // freed enumeration problems with every allocation intact (their ~20 device / pinned allocations
    // and events cost more than a C(28,14) solve); lp_enum_upload refills one instead of allocating
    std::vector<void*> enum_shells;
    // pinned state blocks and event sets of freed simplex problems (hipHostMalloc / hipEventCreate
    // cost more than a small solve)
    struct HostBundle {
        void* pinned = nullptr;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    };
    std::vector<HostBundle> bundles;
};

// Best-fit buffer of at least `bytes` from the context's pool (not more than twice as large), else
// hipMalloc.  *got receives the actual size.
inline hipError_t lp_pool_alloc(lp_context* ctx, void** out, size_t bytes, size_t* got) {
    int best = -1;
    for (int k = 0; k < (int)ctx->pool.size(); ++k) {
        const size_t have = ctx->pool[(size_t)k].second;
        if (have >= bytes && have <= 2 * bytes + (1u << 20) &&
            (best < 0 || have < ctx->pool[(size_t)best].second))
            best = k;
    }
    if (best >= 0) {
        *out = ctx->pool[(size_t)best].first;
        *got = ctx->pool[(size_t)best].second;
        ctx->pool.erase(ctx->pool.begin() + best);
        return hipSuccess;
    }
    *got = bytes;
    return hipMalloc(out, bytes);
}
inline void lp_pool_release(lp_context* ctx, void* p, size_t bytes) {
    if (!p) return;
    constexpr size_t kMaxPooled = 8;
    if (bytes < (1u << 20) || ctx->pool.size() >= kMaxPooled) {
        (void)hipFree(p);
        return;
    }
    ctx->pool.emplace_back(p, bytes);
}

// HIP call check: records the message in the context and returns -(hipError_t).
#define LP_HIP(ctx, expr)                                                                  \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            char _buf[512];                                                                \
            snprintf(_buf, sizeof(_buf), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,     \
                     hipGetErrorString(_e));                                               \
            (ctx)->last_error = _buf;                                                      \
            return -(int)_e;                                                               \
        }                                                                                  \
    } while (0)

#define LP_FAIL(ctx, code, msg)      \
    do {                             \
        (ctx)->last_error = (msg);   \
        return (code);               \
    } while (0)

template <typename T>
static inline T lp_ceil_div(T a, T b) {
    return (a + b - 1) / b;
}

// Host-side combinatorics shared by the enumeration paths (exact u64; 0 = overflow).
uint64_t lp_host_binom(int n, int k);
// rank, among the t-subsets of {0 .. n-m+t-1}, of the first t elements of the rank-th m-subset of
// {0 .. n-1} (enum_prefix.hip)
uint64_t lp_host_prefix_rank(int n, int m, uint64_t rank, int t);
