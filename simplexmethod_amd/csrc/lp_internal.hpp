// lp_internal.hpp — shared host-side plumbing for the HIP hot path (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/simplexmethod_amd.h"

struct lp_context {
    int device = -1;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cus = 0;
    std::string last_error;
};

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
