// device_select.hpp — wave64 primitives for the order-dependent pivot rules.
//
// The reference scans candidates sequentially with an EPS hysteresis
// (/root/reference/src/SimplexSolover.h:153-161 pricing, :181-192 ratio test):
//     best = -inf;  for j ascending: if (v_j > best + eps) { best = v_j; sel = j; }
// That is not an associative reduction, but it is a chain of "records": once
// `best` holds v_p, the next accepted entry is the FIRST j > p with
// v_j > v_p + eps, and no entry before p can qualify again (it was either
// accepted with a smaller value or rejected against a smaller threshold).  One
// wave therefore replays the scan exactly as a short sequence of jumps, each a
// parallel "first index above threshold" (expected ~ln(len) jumps).
#pragma once

#include <hip/hip_runtime.h>

#include <climits>

namespace lpdev {

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double wave_bcast_f64(double v, int src_lane) {
    return __shfl(v, src_lane, 64);
}

// Exact replay of the sequential chain over `len` entries read through `load`
// (load(j, ok) returns v_j and sets ok=false for ineligible j), executed by ONE
// full wave (all 64 lanes must call).  Returns the selected index or -1; `best`
// ends as the chain's final value (+-inf if nothing was eligible).
// WANT_MAX: the :153-161 form (v > best + eps); otherwise the :164-172 / :181-192
// form (v < best - eps).
template <bool WANT_MAX, typename Load>
__device__ int wave_chain_select(int len, double eps, double& best, Load load) {
    constexpr int K = 16;            // entries per lane per tile
    constexpr int TILE = 64 * K;
    const int lane = threadIdx.x & 63;
    const double sentinel = WANT_MAX ? -INFINITY : INFINITY;
    best = sentinel;
    int sel = -1;
    for (int base = 0; base < len; base += TILE) {
        double val[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = base + k * 64 + lane;
            bool ok = false;
            double v = sentinel;
            if (j < len) v = load(j, ok);
            val[k] = ok ? v : sentinel;
        }
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

}  // namespace lpdev
