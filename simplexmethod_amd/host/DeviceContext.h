// DeviceContext.h — process-wide lp_context per HIP device, shared by the solver wrappers.
#pragma once

#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "simplexmethod_amd.h"

namespace lpgpu {

// Throws std::runtime_error when the device cannot be used (there is no CPU fallback).
inline lp_context* context(int device = 0) {
    static std::mutex mu;
    static std::map<int, lp_context*> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
    lp_context* ctx = nullptr;
    const int rc = lp_context_create(device, nullptr, &ctx);
    if (rc != LP_OPTIMAL || !ctx)
        throw std::runtime_error(std::string("simplexmethod_amd: cannot create a context on HIP device ") +
                                 std::to_string(device) + ": " + lp_last_error(nullptr) +
                                 " (status " + std::to_string(rc) + ")");
    cache[device] = ctx;
    return ctx;
}

// Maps an ABI status to the exception the reference would have thrown
// (/root/reference/src/SimplexSolover.h:101,:126,:443,:450; Canonical.cpp:27-46).
inline void throw_for_status(int status, lp_context* ctx) {
    switch (status) {
        case LP_OPTIMAL: return;
        case LP_UNBOUNDED: throw std::runtime_error("Objective function is unbounded");
        case LP_ITER_LIMIT: throw std::runtime_error("Iteration limit reached");
        case LP_SINGULAR: throw std::runtime_error("Singular basis matrix");
        case LP_INFEASIBLE: throw std::runtime_error("No feasible basis");
        case LP_BAD_ARG:
            throw std::invalid_argument(std::string("bad argument: ") + (ctx ? lp_last_error(ctx) : ""));
        default:
            throw std::runtime_error(std::string("HIP runtime failure: ") +
                                     (ctx ? lp_last_error(ctx) : "") + " (status " +
                                     std::to_string(status) + ")");
    }
}

}  // namespace lpgpu
