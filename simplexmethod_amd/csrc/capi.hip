// capi.hip — the extern "C" boundary (include/simplexmethod_amd.h): argument checks that
// mirror the reference's constructors, uploads/downloads, and dispatch to the kernels.
#include <chrono>
#include <cmath>
#include <cstdio>

#include "batched_problem.hpp"
#include "enum_problem.hpp"
#include "lp_internal.hpp"
#include "simplex_problem.hpp"

uint64_t lp_host_binom(int n, int k) {
    if (k < 0 || k > n) return 0;
    if (k > n - k) k = n - k;
    unsigned __int128 r = 1;
    for (int i = 1; i <= k; ++i) {
        r = r * (unsigned)(n - k + i) / (unsigned)i;
        if (r > (unsigned __int128)UINT64_MAX) return 0;
    }
    return (uint64_t)r;
}

extern "C" {

int lp_abi_version(void) { return LP_ABI_VERSION; }

int lp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* lp_status_string(int status) {
    switch (status) {
        case LP_OPTIMAL: return "optimal";
        case LP_UNBOUNDED: return "objective unbounded";
        case LP_ITER_LIMIT: return "iteration limit reached";
        case LP_SINGULAR: return "singular basis matrix";
        case LP_INFEASIBLE: return "no feasible basis";
        case LP_BAD_ARG: return "bad argument";
        default: return status < 0 ? "HIP runtime error" : "unknown status";
    }
}

static thread_local std::string g_create_error;

int lp_context_create(int device, void* stream, lp_context** ctx_out) {
    if (!ctx_out) return LP_BAD_ARG;
    *ctx_out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = "no HIP device visible (this library has no CPU fallback)";
        return e != hipSuccess ? -(int)e : -(int)hipErrorNoDevice;
    }
    if (device < 0 || device >= count) return LP_BAD_ARG;
    e = hipSetDevice(device);
    if (e != hipSuccess) return -(int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return -(int)e;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName +
                         ", kernels are built for gfx950 only";
        return -(int)hipErrorNoBinaryForGpu;
    }
    lp_context* ctx = new lp_context();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return -(int)e;
        }
        ctx->owns_stream = true;
    }
    *ctx_out = ctx;
    return LP_OPTIMAL;
}

static void enum_destroy(lp_enum_problem* p);   // (below: releases a kept enumeration shell)

void lp_context_destroy(lp_context* ctx) {
    if (!ctx) return;
    for (void* q : ctx->enum_shells) enum_destroy(static_cast<lp_enum_problem*>(q));
    ctx->enum_shells.clear();
    for (auto& b : ctx->pool) (void)hipFree(b.first);
    for (auto& hb : ctx->bundles) {
        (void)hipHostFree(hb.pinned);
        for (hipEvent_t e : hb.ev)
            if (e) (void)hipEventDestroy(e);
    }
    (void)hipFree(ctx->dcomb6);
    (void)hipFree(ctx->dcomb5);
    (void)hipFree(ctx->dcomb4);
    for (hipStream_t a : ctx->aux_stream)
        if (a) (void)hipStreamDestroy(a);
    for (hipEvent_t e : ctx->aux_event)
        if (e) (void)hipEventDestroy(e);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* lp_last_error(const lp_context* ctx) {
    return ctx ? ctx->last_error.c_str() : g_create_error.c_str();
}

int lp_context_sync(lp_context* ctx) {
    if (!ctx) return LP_BAD_ARG;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    LP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LP_OPTIMAL;
}

// ===========================================================================
// simplex
// ===========================================================================

// Canonical's constructor checks (Canonical.cpp:27-46) + SetOriginalVariablesCount (:156-163).
static int check_canonical(lp_context* ctx, const double* A, int m, int n, const double* b,
                           const double* c, const int* basis, int n_orig) {
    if (!A || !b || !c || !basis) LP_FAIL(ctx, LP_BAD_ARG, "null problem array");
    if (m <= 0 || n <= 0) LP_FAIL(ctx, LP_BAD_ARG, "empty problem");
    if (n < m) LP_FAIL(ctx, LP_BAD_ARG, "fewer columns than rows");
    if (n_orig <= 0 || n_orig > n) LP_FAIL(ctx, LP_BAD_ARG, "bad original variable count");
    for (int t = 0; t < m; ++t)
        if (basis[t] < 0 || basis[t] >= n) LP_FAIL(ctx, LP_BAD_ARG, "basis index out of range");
    return LP_OPTIMAL;
}

void lp_simplex_free(lp_simplex_problem* p) {
    if (!p) return;
    lp_context* ctx = p->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);   // the arena goes back to the pool: nothing may still use it
    lp_pool_release(ctx, p->arena, p->arena_bytes);
    (void)hipFree(p->dscratchT);
    (void)hipFree(p->ov_T);
    (void)hipFree(p->ov_vec);
    (void)hipFree(p->look.stamps);
    if (p->h_state) {   // pinned block + events: kept for the next problem of this context
        lp_context::HostBundle hb;
        hb.pinned = p->h_state;
        hb.ev[0] = p->ev0; hb.ev[1] = p->ev1; hb.ev[2] = p->res_ev0; hb.ev[3] = p->res_ev1;
        if (ctx->bundles.size() < 8) {
            ctx->bundles.push_back(hb);
        } else {
            (void)hipHostFree(hb.pinned);
            for (hipEvent_t e : hb.ev)
                if (e) (void)hipEventDestroy(e);
        }
    }
    for (hipEvent_t e : p->upd_events) (void)hipEventDestroy(e);
    delete p;
}

namespace {
// T (rows x ld, row-major) <- [A | b] with the cost row c underneath, from the column-major A the
// caller holds (Eigen's layout): a tiled transpose on the device instead of a strided host loop.
__global__ __launch_bounds__(256) void k_build_tableau(const double* __restrict__ Acol, const double* __restrict__ b,
                                                       const double* __restrict__ c, double* __restrict__ T,
                                                       int m, int n, int ld) {
    __shared__ double tile[32][33];
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int q = ty; q < 32; q += 8) {   // read: consecutive threads along i (contiguous in column-major A)
        const int j = j0 + q, i = i0 + tx;
        tile[q][tx] = (j < n && i < m) ? Acol[(size_t)j * m + i] : 0.0;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {   // write: consecutive threads along j (contiguous in row-major T)
        const int i = i0 + q, j = j0 + tx;
        if (i < m && j < n) T[(size_t)i * ld + j] = tile[tx][q];
    }
    if (blockIdx.x == 0) {   // column n (b), padding, and (first row of blocks) the cost row
        for (int q = threadIdx.x; q < 32; q += 256) {
            const int i = i0 + q;
            if (i < m) {
                T[(size_t)i * ld + n] = b[i];
                for (int j = n + 1; j < ld; ++j) T[(size_t)i * ld + j] = 0.0;
            }
        }
    }
    if (blockIdx.y == 0) {
        for (int q = threadIdx.x; q < 32; q += 256) {
            const int j = j0 + q;
            if (j < n) T[(size_t)m * ld + j] = c[j];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int j = n; j < ld; ++j) T[(size_t)m * ld + j] = 0.0;
    }
}
}  // namespace

int lp_simplex_upload(lp_context* ctx, const double* A, int m, int n, const double* b,
                      const double* c, const int* basis_in, int maximize, int n_orig,
                      lp_simplex_problem** problem_out) {
    if (!ctx || !problem_out) return LP_BAD_ARG;
    *problem_out = nullptr;
    int rc = check_canonical(ctx, A, m, n, b, c, basis_in, n_orig);
    if (rc) return rc;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    lp_simplex_problem* p = new lp_simplex_problem();
    p->ctx = ctx;
    p->n_orig = n_orig;
    p->h_c.assign(c, c + n);
    SimplexDev& d = p->dev;
    d.m = m;
    d.n = n;
    d.ld = ((n + 1 + 7) / 8) * 8;
    d.maximize = maximize ? 1 : 0;
    d.trace_cap = 16384;
    const size_t rows = (size_t)m + 1;
    p->tableau_bytes = sizeof(double) * rows * (size_t)d.ld;
#define LP_TRY(expr)                        \
    do {                                    \
        hipError_t _e = (expr);             \
        if (_e != hipSuccess) {             \
            ctx->last_error = #expr;        \
            lp_simplex_free(p);             \
            return -(int)_e;                \
        }                                   \
    } while (0)
    // ---- one arena for everything on the device (taken from / returned to the context's pool:
    // a repeated one-shot solve of the same shape allocates nothing)
    LookDev& la = p->look;
    la.J = lp_lookahead_pick_j(m, n);
    la.rows_pad = ((m + 1 + 7) / 8) * 8;
    const size_t J = (size_t)(la.J > 0 ? la.J : 1);
    const bool resident = lp_resident_plan(m, n, &p->res) != 0;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t staging_bytes = std::max(p->tableau_bytes, sizeof(double) * ((size_t)m * n + m + n) + 256);
    const size_t oT = take(p->tableau_bytes), oT0 = take(staging_bytes);
    const size_t olcol = take(sizeof(double) * rows), oprow = take(sizeof(double) * (size_t)d.ld);
    const size_t obasis = take(sizeof(int) * (size_t)m), obasis0 = take(sizeof(int) * (size_t)m);
    const size_t onb = take((size_t)n), onb0 = take((size_t)n), oused = take((size_t)m);
    const size_t orowpos = take(sizeof(int) * (size_t)m);
    const size_t otre = take(sizeof(int) * (size_t)d.trace_cap), otrl = take(sizeof(int) * (size_t)d.trace_cap);
    const size_t ostate = take(sizeof(SimplexState)), odx = take(sizeof(double) * (size_t)n);
    const size_t oetaL = take(sizeof(double) * J * (size_t)la.rows_pad), oetaP = take(sizeof(double) * J * (size_t)d.ld);
    const size_t odvec = take(sizeof(double) * (size_t)d.ld), orhs = take(sizeof(double) * (size_t)la.rows_pad);
    const size_t opiv = take(sizeof(int) * 2 * J), ocount = take(sizeof(int));
    const size_t ocomm = resident ? take(p->res.comm_bytes) : 0;
    {
        size_t got = 0;
        LP_TRY(lp_pool_alloc(ctx, &p->arena, off, &got));
        p->arena_bytes = got;
    }
    char* base = static_cast<char*>(p->arena);
    d.T = reinterpret_cast<double*>(base + oT);
    p->dT0 = reinterpret_cast<double*>(base + oT0);
    d.lcol = reinterpret_cast<double*>(base + olcol);
    d.prow = reinterpret_cast<double*>(base + oprow);
    d.basis = reinterpret_cast<int*>(base + obasis);
    p->dbasis0 = reinterpret_cast<int*>(base + obasis0);
    d.nonbasic = reinterpret_cast<unsigned char*>(base + onb);
    p->dnonbasic0 = reinterpret_cast<unsigned char*>(base + onb0);
    d.rowused = reinterpret_cast<unsigned char*>(base + oused);
    d.rowpos = reinterpret_cast<int*>(base + orowpos);
    d.trace_enter = reinterpret_cast<int*>(base + otre);
    d.trace_leave = reinterpret_cast<int*>(base + otrl);
    d.state = reinterpret_cast<SimplexState*>(base + ostate);
    p->dx = reinterpret_cast<double*>(base + odx);
    la.etaL = reinterpret_cast<double*>(base + oetaL);
    la.etaP = reinterpret_cast<double*>(base + oetaP);
    la.dvec = reinterpret_cast<double*>(base + odvec);
    la.rhs = reinterpret_cast<double*>(base + orhs);
    la.piv = reinterpret_cast<int*>(base + opiv);
    la.count = reinterpret_cast<int*>(base + ocount);
    if (resident) p->res.comm = base + ocomm;
    if (!ctx->bundles.empty()) {
        const lp_context::HostBundle hb = ctx->bundles.back();
        ctx->bundles.pop_back();
        p->h_state = static_cast<SimplexState*>(hb.pinned);
        p->ev0 = hb.ev[0]; p->ev1 = hb.ev[1]; p->res_ev0 = hb.ev[2]; p->res_ev1 = hb.ev[3];
    } else {
        LP_TRY(hipHostMalloc(&p->h_state, sizeof(SimplexState) + 64));
        LP_TRY(hipEventCreate(&p->ev0));
        LP_TRY(hipEventCreate(&p->ev1));
    }
    hipStream_t s = ctx->stream;
    LP_TRY(hipMemsetAsync(la.count, 0, sizeof(int), s));

    // ---- initial tableau [A | b] with the cost row c underneath, row-major: A goes up as the caller
    // holds it (column-major) into the staging area and is transposed on the device
    double* dA = p->dT0;
    double* db = dA + (size_t)m * n;
    double* dc = db + m;
    std::vector<unsigned char> nonbasic((size_t)n, 1);
    for (int t = 0; t < m; ++t) nonbasic[(size_t)basis_in[t]] = 0;
    LP_TRY(hipMemcpyAsync(dA, A, sizeof(double) * (size_t)m * n, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(db, b, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(dc, c, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_build_tableau, dim3(lp_ceil_div(n, 32), lp_ceil_div(m, 32)), 256, 0, s, dA, db, dc, d.T, m, n,
                       d.ld);
    LP_TRY(hipMemcpyAsync(d.basis, basis_in, sizeof(int) * (size_t)m, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(p->dbasis0, basis_in, sizeof(int) * (size_t)m, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(d.nonbasic, nonbasic.data(), (size_t)n, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(p->dnonbasic0, nonbasic.data(), (size_t)n, hipMemcpyHostToDevice, s));
    LP_TRY(hipStreamSynchronize(s));
    LP_TRY(hipGetLastError());
#undef LP_TRY

    // computeBFS (SimplexSolover.h:423): nothing to do for the slack identity basis with
    // zero basic costs (Symmetrical::ToCanonical, Symmetrical.cpp:169-188); otherwise m
    // Gauss-Jordan pivots on the device.
    bool identity = true, zero_costs = true;
    for (int t = 0; t < m && identity; ++t) {
        if (c[basis_in[t]] != 0.0) zero_costs = false;
        for (int i = 0; i < m && identity; ++i)
            if (A[(size_t)basis_in[t] * m + i] != ((i == t) ? 1.0 : 0.0)) identity = false;
    }
    p->init_status = LP_OPTIMAL;
    if (identity && !zero_costs) {
        // unit-vector basis with costs (the artificial basis of a phase-I problem): the crash pivots
        // only touch the reduced-cost row — one pass instead of m rank-1 updates
        rc = lp_simplex_price_out_identity(p);
        if (rc) {
            lp_simplex_free(p);
            return rc;
        }
    } else if (!identity) {
        rc = lp_simplex_crash(p);
        if (rc < 0) {
            lp_simplex_free(p);
            return rc;
        }
        p->init_status = rc;
    }
    hipError_t e = hipMemcpyAsync(p->dT0, d.T, p->tableau_bytes, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        lp_simplex_free(p);
        return -(int)e;
    }
    *problem_out = p;
    return LP_OPTIMAL;
}

int lp_simplex_reset(lp_simplex_problem* p) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    const SimplexDev& d = p->dev;
    hipStream_t s = ctx->stream;
    LP_HIP(ctx, hipMemcpyAsync(d.T, p->dT0, p->tableau_bytes, hipMemcpyDeviceToDevice, s));
    LP_HIP(ctx, hipMemcpyAsync(d.basis, p->dbasis0, sizeof(int) * (size_t)d.m, hipMemcpyDeviceToDevice, s));
    LP_HIP(ctx, hipMemcpyAsync(d.nonbasic, p->dnonbasic0, (size_t)d.n, hipMemcpyDeviceToDevice, s));
    LP_HIP(ctx, hipStreamSynchronize(s));
    p->last_status = -100;
    p->last_iters = 0;
    return LP_OPTIMAL;
}

int lp_simplex_run(lp_simplex_problem* p, double eps, int max_iter, int algo,
                   lp_simplex_stats* stats_out) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    if (stats_out) std::memset(stats_out, 0, sizeof(*stats_out));
    if (p->init_status != LP_OPTIMAL) {  // "Singular basis matrix", SimplexSolover.h:125-126
        if (stats_out) stats_out->status = p->init_status;
        p->last_status = p->init_status;
        return p->init_status;
    }
    const bool asked_auto = algo == LP_SIMPLEX_ALGO_AUTO;
    if (algo == LP_SIMPLEX_ALGO_AUTO)
        algo = p->res.G >= 1 ? LP_SIMPLEX_ALGO_RESIDENT
               : p->look.J >= 3 ? LP_SIMPLEX_ALGO_LOOKAHEAD   // (depth 2, 1536 x 3072: 20.0 us per pivot against the overlapped path's 18.4)
               : lp_overlap_auto(p->dev.m) ? LP_SIMPLEX_ALGO_OVERLAP : LP_SIMPLEX_ALGO_LAUNCH;
    if (algo == LP_SIMPLEX_ALGO_OVERLAP && asked_auto && lp_overlap_prepare(p) != LP_OPTIMAL) {
        (void)hipGetLastError();   // no memory for the second tableau buffer: the launch pair per pivot
        ctx->last_error.clear();
        algo = LP_SIMPLEX_ALGO_LAUNCH;
    }
    const int asked = algo;
    p->last_algo = algo;   // (every path overwrites it with the algorithm that answered; an early error return reports the one asked for)
    int rc;
    switch (algo) {
        case LP_SIMPLEX_ALGO_RESIDENT:
            if (p->res.G < 1)
                LP_FAIL(ctx, LP_BAD_ARG, "chip-resident simplex needs m <= 960 and ceil(n / columns per workgroup) <= 256 workgroups");
            rc = lp_simplex_run_resident(p, eps, max_iter, stats_out);
            break;
        case LP_SIMPLEX_ALGO_LAUNCH:
            rc = lp_simplex_run_launch(p, eps, max_iter, stats_out);
            break;
        case LP_SIMPLEX_ALGO_OVERLAP:
            rc = lp_simplex_run_overlap(p, eps, max_iter, stats_out);
            break;
        case LP_SIMPLEX_ALGO_LOOKAHEAD: {
            if (p->look.J < 1)
                LP_FAIL(ctx, LP_BAD_ARG, "look-ahead selector does not fit LDS for this m, n");
            rc = lp_lookahead_prepare(p);
            if (rc) return rc;
            rc = lp_simplex_run_lookahead(p, eps, max_iter, stats_out);
            break;
        }
        default:
            LP_FAIL(ctx, LP_BAD_ARG, "unknown simplex algorithm id");
    }
    if (stats_out) {   // which algorithm produced the answer (a chip-resident hand-off that timed out is re-run)
        stats_out->algo_used = p->last_algo;
        stats_out->fell_back = (asked == LP_SIMPLEX_ALGO_RESIDENT && p->last_algo != LP_SIMPLEX_ALGO_RESIDENT) ? 1 : 0;
    }
    return rc;
}

int lp_simplex_profile(lp_simplex_problem* p, int on) {
    if (!p) return LP_BAD_ARG;
    p->profile_updates = on != 0;
    return LP_OPTIMAL;
}

int lp_simplex_download(lp_simplex_problem* p, double* x_out, int* basis_out, double* obj_out,
                        int* trace_enter, int* trace_leave, int trace_cap, double* tableau_out) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    const SimplexDev& d = p->dev;
    hipStream_t s = ctx->stream;
    std::vector<double> x((size_t)d.n, 0.0);
    if (x_out || obj_out) {
        lp_simplex_extract_x(p, p->dx);
        LP_HIP(ctx, hipMemcpyAsync(x.data(), p->dx, sizeof(double) * (size_t)d.n, hipMemcpyDeviceToHost, s));
    }
    if (basis_out)
        LP_HIP(ctx, hipMemcpyAsync(basis_out, d.basis, sizeof(int) * (size_t)d.m, hipMemcpyDeviceToHost, s));
    const int k = std::min(trace_cap, std::min(p->last_iters, d.trace_cap));
    if (trace_enter && k > 0)
        LP_HIP(ctx, hipMemcpyAsync(trace_enter, d.trace_enter, sizeof(int) * (size_t)k, hipMemcpyDeviceToHost, s));
    if (trace_leave && k > 0)
        LP_HIP(ctx, hipMemcpyAsync(trace_leave, d.trace_leave, sizeof(int) * (size_t)k, hipMemcpyDeviceToHost, s));
    std::vector<double> T;
    if (tableau_out) {
        T.resize(((size_t)d.m + 1) * (size_t)d.ld);
        LP_HIP(ctx, hipMemcpyAsync(T.data(), d.T, p->tableau_bytes, hipMemcpyDeviceToHost, s));
    }
    LP_HIP(ctx, hipStreamSynchronize(s));
    LP_HIP(ctx, hipGetLastError());
    if (x_out)  // x.head(n_orig), SimplexSolover.h:435-438
        for (int j = 0; j < p->n_orig; ++j) x_out[j] = x[(size_t)j];
    if (obj_out) {  // Canonical::Evaluate, Canonical.cpp:86
        double z = 0.0;
        for (int j = 0; j < d.n; ++j) z += p->h_c[(size_t)j] * x[(size_t)j];
        *obj_out = z;
    }
    if (tableau_out)
        for (int i = 0; i <= d.m; ++i)
            std::memcpy(tableau_out + (size_t)i * (d.n + 1), T.data() + (size_t)i * d.ld,
                        sizeof(double) * (size_t)(d.n + 1));
    return LP_OPTIMAL;
}

int lp_simplex_solve(lp_context* ctx, const double* A, int m, int n, const double* b,
                     const double* c, const int* basis_in, int maximize, int n_orig, double eps,
                     int max_iter, double* x_out, int* basis_out, double* obj_out, int* iters_out) {
    if (!ctx) return LP_BAD_ARG;
    if (!x_out) LP_FAIL(ctx, LP_BAD_ARG, "x_out is null");
    lp_simplex_problem* p = nullptr;
    int rc = lp_simplex_upload(ctx, A, m, n, b, c, basis_in, maximize, n_orig, &p);
    if (rc) return rc;
    lp_simplex_stats st;
    rc = lp_simplex_run(p, eps, max_iter, LP_SIMPLEX_ALGO_AUTO, &st);
    if (iters_out) *iters_out = st.pivots;
    if (rc == LP_OPTIMAL) {
        rc = lp_simplex_download(p, x_out, basis_out, obj_out, nullptr, nullptr, 0, nullptr);
    } else if (rc > 0 && basis_out) {
        (void)lp_simplex_download(p, nullptr, basis_out, nullptr, nullptr, nullptr, 0, nullptr);
    }
    lp_simplex_free(p);
    return rc;
}

int lp_simplex_row(lp_simplex_problem* p, int row, double* out) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    if (!out || row < 0 || row > p->dev.m) LP_FAIL(ctx, LP_BAD_ARG, "lp_simplex_row: bad row or null output");
    LP_HIP(ctx, hipSetDevice(ctx->device));
    LP_HIP(ctx, hipMemcpyAsync(out, p->dev.T + (size_t)row * p->dev.ld, sizeof(double) * (size_t)(p->dev.n + 1),
                               hipMemcpyDeviceToHost, ctx->stream));
    LP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LP_OPTIMAL;
}

int lp_simplex_force_pivot(lp_simplex_problem* p, int row, int col) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    if (row < 0 || row >= p->dev.m || col < 0 || col >= p->dev.n)
        LP_FAIL(ctx, LP_BAD_ARG, "lp_simplex_force_pivot: position outside the tableau");
    if (p->init_status != LP_OPTIMAL) LP_FAIL(ctx, LP_SINGULAR, "lp_simplex_force_pivot: the initial basis was singular");
    LP_HIP(ctx, hipSetDevice(ctx->device));
    return lp_simplex_force(p, row, col);
}

// Two-phase simplex: the host logic of the flow, every pivot on the GPU.
int lp_simplex_two_phase(lp_context* ctx, const double* A, int m, int n, const double* b,
                         const double* c, int maximize, int n_orig, double eps, int max_iter,
                         double* x_out, int* basis_out, double* obj_out, int* iters_out) {
    if (!ctx) return LP_BAD_ARG;
    if (!A || !b || !c || !x_out) LP_FAIL(ctx, LP_BAD_ARG, "lp_simplex_two_phase: null argument");
    if (m <= 0 || n < m || n_orig <= 0 || n_orig > n) LP_FAIL(ctx, LP_BAD_ARG, "lp_simplex_two_phase: bad dimensions");
    const int na = n + m;
    std::vector<double> A1((size_t)m * na, 0.0), b1((size_t)m), c1((size_t)na, 0.0), xa((size_t)na);
    std::vector<int> N((size_t)m);
    int it[3] = {0, 0, 0};
    if (iters_out) std::memcpy(iters_out, it, sizeof(it));
    // -DLP_TWO_PHASE_TRACE (diagnostic builds, scripts/two_phase_trace.py): wall time of each stage on stderr
#ifdef LP_TWO_PHASE_TRACE
    const bool trace = true;
#else
    const bool trace = false;
#endif
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto stage = [&](const char* name) {
        if (!trace) return;
        const auto t = now();
        fprintf(stderr, "[two_phase] %-28s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // make_b_nonneg (:61-68) and createAuxiliaryProblem (:70-95)
    std::vector<char> flip((size_t)m);
    for (int i = 0; i < m; ++i) {
        flip[i] = b[i] < -eps;
        b1[i] = flip[i] ? -b[i] : b[i];
        A1[(size_t)(n + i) * m + i] = 1.0;
    }
    for (int j = 0; j < n; ++j) {   // (column by column: both matrices are column-major)
        const double* src = A + (size_t)j * m;
        double* dst = A1.data() + (size_t)j * m;
        for (int i = 0; i < m; ++i) dst[i] = flip[i] ? -src[i] : src[i];
    }
    for (int j = n; j < na; ++j) c1[j] = 1.0;
    for (int t = 0; t < m; ++t) N[t] = n + t;
    stage("auxiliary problem (host)");
    // ---- phase I: minimise the sum of the artificials
    lp_simplex_problem* p = nullptr;
    int rc = lp_simplex_upload(ctx, A1.data(), m, na, b1.data(), c1.data(), N.data(), 0, na, &p);
    if (rc) return rc;
    stage("upload");
    lp_simplex_stats st;
    rc = lp_simplex_run(p, eps, max_iter, LP_SIMPLEX_ALGO_AUTO, &st);
    it[0] = st.pivots;
    stage("phase I run");
    if (rc == LP_OPTIMAL) rc = lp_simplex_download(p, xa.data(), N.data(), nullptr, nullptr, nullptr, 0, nullptr);
    stage("phase I download");
    if (rc == LP_OPTIMAL) {
        double sum = 0.0;  // :347-350
        for (int i = 0; i < m; ++i) sum += xa[(size_t)n + i];
        if (sum > eps) {  // :352-353
            rc = LP_INFEASIBLE;
            ctx->last_error = "two-phase: the problem has no feasible solution (phase I optimum > eps)";
        }
    }
    if (rc == LP_OPTIMAL) {
        // replaceArtificialColumns (:331-381): every artificial still basic (at level 0) leaves for
        // the first non-basic original column with |T[pos][cand]| > eps, chosen on the device; the
        // positions are known from the phase-I basis, so all pivots are queued behind one another
        std::vector<int> positions;
        for (int pos = 0; pos < m; ++pos)
            if (N[pos] >= n) positions.push_back(pos);
        if (!positions.empty()) {
            rc = lp_simplex_driveout(p, positions.data(), (int)positions.size(), n, eps, &it[1]);
            if (rc == LP_SINGULAR)   // :372-380: linearly dependent constraints
                ctx->last_error = "two-phase: an artificial variable cannot leave the basis (linearly dependent constraints)";
        }
    }
    // ---- phase II (:383-404) continues on the phase-I tableau: original costs priced out over the
    // current basis, artificial columns barred — no re-inversion of the basis from [A' | b']
    stage("drive-out");
    if (rc == LP_OPTIMAL) rc = lp_simplex_phase2_costs(p, c, n, maximize, n_orig);
    stage("phase II costs");
    if (rc == LP_OPTIMAL) {
        rc = lp_simplex_run(p, eps, max_iter, LP_SIMPLEX_ALGO_AUTO, &st);
        it[2] = st.pivots;
        stage("phase II run");
        if (rc == LP_OPTIMAL)
            rc = lp_simplex_download(p, x_out, N.data(), obj_out, nullptr, nullptr, 0, nullptr);
        else if (rc > 0)
            (void)lp_simplex_download(p, nullptr, N.data(), nullptr, nullptr, nullptr, 0, nullptr);
    } else if (rc == LP_SINGULAR || rc == LP_INFEASIBLE) {
        (void)lp_simplex_download(p, nullptr, N.data(), nullptr, nullptr, nullptr, 0, nullptr);
    }
    stage("phase II download");
    lp_simplex_free(p);
    stage("free");
    if (basis_out) std::memcpy(basis_out, N.data(), sizeof(int) * (size_t)m);
    if (iters_out) std::memcpy(iters_out, it, sizeof(it));
    return rc;
}

// Diagnostic: switches the look-ahead selector's per-phase cycle stamps on (cap_pivots > 0)
// and, after a run, copies them out: 8 stamps per pivot (s_memtime ticks).
int lp_debug_simplex_stamps(lp_simplex_problem* p, int cap_pivots, unsigned long long* out) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    if (!p->look.stamps) {
        if (cap_pivots <= 0) return LP_OPTIMAL;
        const size_t bytes = sizeof(unsigned long long) * 8 * (size_t)(cap_pivots + 64);
        LP_HIP(ctx, hipMalloc(&p->look.stamps, bytes));
        LP_HIP(ctx, hipMemset(p->look.stamps, 0, bytes));
        return LP_OPTIMAL;
    }
    if (out)
        LP_HIP(ctx, hipMemcpy(out, p->look.stamps, sizeof(unsigned long long) * 8 * (size_t)cap_pivots, hipMemcpyDeviceToHost));
    return LP_OPTIMAL;
}

int lp_bench_rank1_update(lp_simplex_problem* p, int row, int col, int iters,
                          float* ms_per_launch_out) {
    if (!p) return LP_BAD_ARG;
    LP_HIP(p->ctx, hipSetDevice(p->ctx->device));
    return lp_simplex_bench_update(p, row, col, iters, ms_per_launch_out);
}

int lp_bench_rankj_update(lp_simplex_problem* p, int iters, float* ms_per_launch_out,
                          int* pivots_per_launch_out) {
    if (!p) return LP_BAD_ARG;
    LP_HIP(p->ctx, hipSetDevice(p->ctx->device));
    return lp_lookahead_bench_update(p, iters, ms_per_launch_out, pivots_per_launch_out);
}

// ===========================================================================
// enumeration
// ===========================================================================

uint64_t lp_binom(int n, int k) { return lp_host_binom(n, k); }

// Cost-balanced cut of the rank space (same rule as simplexmethod_amd/dist.py:
// balanced_shard_bounds): cost(x) = x + kShardRecordCost * (depth m-7 tree nodes before subset x).
static const uint64_t kShardRecordCost = 160;
int lp_enum_shard_bounds(int n, int m, int shard, int shards, uint64_t* begin_out, uint64_t* end_out) {
    if (!begin_out || !end_out || shards <= 0 || shard < 0 || shard >= shards) return LP_BAD_ARG;
    if (m <= 0 || n < m || n > kEnumMaxN || m > kEnumMaxM) return LP_BAD_ARG;
    const uint64_t total = lp_host_binom(n, m);
    if (total == 0) return LP_BAD_ARG;
    const int d0 = m - 7;
    // (the cost model is that of the tuned kernels' box; the general kernel's shapes — m > 16 or
    // n - m > 16 — small trees and the direct kernel get equal-size cuts)
    const bool prefix_shape = m >= 7 && m <= 16 && n - m >= 2 && n - m <= 16;
    if (shards == 1 || !prefix_shape || total < (1ULL << 20)) {
        // (dist.py uses total * k // world here; same partition property, sizes differ by <= 1)
        *begin_out = (uint64_t)((unsigned __int128)total * (unsigned)shard / (unsigned)shards);
        *end_out = (uint64_t)((unsigned __int128)total * (unsigned)(shard + 1) / (unsigned)shards);
        return LP_OPTIMAL;
    }
    auto cost = [&](uint64_t x) -> unsigned __int128 {
        if (x >= total) return (unsigned __int128)total + (unsigned __int128)kShardRecordCost * lp_host_binom(n - 7, d0);
        return (unsigned __int128)x + (unsigned __int128)kShardRecordCost * lp_host_prefix_rank(n, m, x, d0);
    };
    const unsigned __int128 full = cost(total);
    auto cut = [&](int k) -> uint64_t {
        if (k <= 0) return 0;
        if (k >= shards) return total;
        const unsigned __int128 target = full * (unsigned)k / (unsigned)shards;
        uint64_t lo = 0, hi = total;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (cost(mid) < target) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    *begin_out = cut(shard);
    *end_out = cut(shard + 1);
    return LP_OPTIMAL;
}

// The feasible list of the shared-prefix path: rank, record index and score of each entry.
static void enum_list_release(lp_enum_problem* p) {
    PrefixDev& pd = p->prefix;
    lp_pool_release(p->ctx, pd.list, sizeof(unsigned long long) * pd.list_cap);
    lp_pool_release(p->ctx, pd.scores, sizeof(double) * pd.list_cap);
    lp_pool_release(p->ctx, pd.list_rec, sizeof(int) * pd.list_cap);
    pd.list = nullptr;
    pd.scores = nullptr;
    pd.list_rec = nullptr;
    pd.list_cap = 0;
}
static hipError_t enum_list_alloc(lp_enum_problem* p, unsigned long long cap) {
    PrefixDev& pd = p->prefix;
    size_t got = 0;
    hipError_t e = lp_pool_alloc(p->ctx, (void**)&pd.list, sizeof(unsigned long long) * cap, &got);
    if (e == hipSuccess) e = lp_pool_alloc(p->ctx, (void**)&pd.scores, sizeof(double) * cap, &got);
    if (e == hipSuccess) e = lp_pool_alloc(p->ctx, (void**)&pd.list_rec, sizeof(int) * cap, &got);
    if (e != hipSuccess) {
        // the sizes lp_pool_release is told must be those of the failed request
        pd.list_cap = cap;
        enum_list_release(p);
        return e;
    }
    pd.list_cap = cap;
    return hipSuccess;
}

static void enum_destroy(lp_enum_problem* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipFree(p->dA); (void)hipFree(p->db); (void)hipFree(p->dc); (void)hipFree(p->dbinom);
    (void)hipFree(p->d_pass); (void)hipFree(p->dev.chunk_best);
    (void)hipFree(p->dvx); (void)hipFree(p->dvi);
    (void)hipFree(p->prefix.root_cursor);
    enum_list_release(p);
    lp_pool_release(p->ctx, p->prefix.dense_scores, sizeof(double) * p->prefix.dense_cap);
    lp_pool_release(p->ctx, p->prefix.items, sizeof(int4) * (size_t)p->prefix.item_cap);
    lp_pool_release(p->ctx, p->prefix.items2, sizeof(int4) * (size_t)p->prefix.item_cap2);
    (void)hipFree(p->prefix.item_count);
    if (p->h_item_count) (void)hipHostFree(p->h_item_count);
    lp_pool_release(p->ctx, p->prefix_buf[0], p->prefix_buf_bytes[0]);
    lp_pool_release(p->ctx, p->prefix_buf[1], p->prefix_buf_bytes[1]);
    if (p->h_pass) (void)hipHostFree(p->h_pass);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    delete p;
}

// A freed problem keeps its allocations (all sized for the largest shape) in the context for the next
// lp_enum_upload; beyond two kept shells it is really released.
int lp_enum_exact_division(const lp_enum_problem* p) { return (p && p->exact_div) ? 1 : 0; }

int lp_debug_reciprocal(lp_context* ctx, const double* x, int n, double* fast_out, double* plain_out) {
    if (!ctx || !x || !fast_out || !plain_out || n <= 0) return LP_BAD_ARG;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    return lp_enum_debug_reciprocal(ctx, x, n, fast_out, plain_out);
}

int lp_debug_division(lp_context* ctx, const double* num, const double* den, int n, double* fast_out, double* plain_out) {
    if (!ctx || !num || !den || !fast_out || !plain_out || n <= 0) return LP_BAD_ARG;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    return lp_simplex_debug_division(ctx, num, den, n, fast_out, plain_out);
}

void lp_enum_free(lp_enum_problem* p) {
    if (!p) return;
    lp_context* ctx = p->ctx;
    if (p->complete && ctx->enum_shells.size() < 2) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->enum_shells.push_back(p);
        return;
    }
    enum_destroy(p);
}

int lp_enum_upload(lp_context* ctx, const double* A, int m, int n, const double* b,
                   const double* c, int maximize, lp_enum_problem** problem_out) {
    if (!ctx || !problem_out) return LP_BAD_ARG;
    *problem_out = nullptr;
    if (!A || !b || !c) LP_FAIL(ctx, LP_BAD_ARG, "null problem array");
    if (m <= 0 || n < m) LP_FAIL(ctx, LP_BAD_ARG, "need 0 < m <= n");
    if (n > kEnumMaxN || m > kEnumMaxM)
        LP_FAIL(ctx, LP_BAD_ARG, "enumeration supports n <= 64 and m <= 32 (ranks must fit 64 bits)");
    if (lp_host_binom(n, m) == 0) LP_FAIL(ctx, LP_BAD_ARG, "C(n,m) overflows 64 bits");
    LP_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long want_cap = 1ULL << 22;
    bool forced_cap = false;
    if (const char* e = getenv("LP_ENUM_LIST_CAP")) {   // tests: force the sub-range path on small problems
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 64 && v < want_cap) want_cap = v;
        forced_cap = true;
    } else if (const char* e2 = getenv("LP_ENUM_LIST_START")) {   // tests: a small first list that may grow
        const unsigned long long v = strtoull(e2, nullptr, 10);
        if (v >= 64 && v < want_cap) want_cap = v;
    }
    lp_enum_problem* p = nullptr;
    while (!p && !ctx->enum_shells.empty()) {   // a kept shell: every allocation is already there
        lp_enum_problem* q = static_cast<lp_enum_problem*>(ctx->enum_shells.back());
        ctx->enum_shells.pop_back();
        // (a list that grew for a degenerate problem is kept unless it is far larger than the default)
        const bool fits = forced_cap ? q->prefix.list_cap == want_cap
                                     : (q->prefix.list_cap >= want_cap && q->prefix.list_cap <= 64 * want_cap);
        if (fits) p = q; else enum_destroy(q);
    }
    const bool fresh = p == nullptr;
    if (fresh) {
        p = new lp_enum_problem();
        p->ctx = ctx;
    } else {   // forget what the previous problem left behind
        p->list_valid = p->spec_valid = p->pchunks_valid = false;
        p->dense_active = p->dense_hint = false;
        p->exact_div = false;
        p->pchunks.clear();
        p->shard_rank = p->shard_world = -1;
        p->last_begin = p->last_end = p->last_per_chunk = 0;
        p->last_chunks = 0;
        p->last_algo = 0;
        p->list_n = 0;
    }
    p->complete = false;
    EnumDev& d = p->dev;
    d.m = m;
    d.n = n;
    d.lda = n + 1;
    d.rs = m > 16 ? ((m + 1) & ~1) : 16;   // row stride of the shared-prefix records (enum_tree.hpp: rec_rs)
    d.pad0 = 0;
    d.maximize = maximize ? 1 : 0;
    p->chunk_cap = 1 << 17;
    std::vector<double> Arow((size_t)m * d.lda, 0.0);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) Arow[(size_t)i * d.lda + j] = A[(size_t)j * m + i];
    std::vector<unsigned long long> binom((size_t)(kEnumMaxN + 1) * kBinomK, 0ULL);
    for (int i = 0; i <= kEnumMaxN; ++i)
        for (int k = 0; k < kBinomK; ++k) binom[(size_t)i * kBinomK + k] = lp_host_binom(i, k);
#define LP_TRY(expr)                        \
    do {                                    \
        hipError_t _e = (expr);             \
        if (_e != hipSuccess) {             \
            ctx->last_error = #expr;        \
            lp_enum_free(p);                \
            return -(int)_e;                \
        }                                   \
    } while (0)
    hipStream_t s = ctx->stream;
    if (fresh) {   // sized for the largest shape (m <= 32, n <= 64): a shell serves any later problem
        LP_TRY(hipMalloc(&p->dA, sizeof(double) * (size_t)kEnumMaxM * (kEnumMaxN + 1)));
        LP_TRY(hipMalloc(&p->db, sizeof(double) * (size_t)kEnumMaxM));
        LP_TRY(hipMalloc(&p->dc, sizeof(double) * (size_t)kEnumMaxN));
        LP_TRY(hipMalloc(&p->dbinom, sizeof(unsigned long long) * binom.size()));
        LP_TRY(hipMalloc(&p->d_pass, sizeof(EnumPassBlock)));
        d.result = &p->d_pass->result;
        LP_TRY(hipMalloc(&d.chunk_best, sizeof(double) * (size_t)(p->chunk_cap + 64)));
        LP_TRY(hipMalloc(&p->dvx, sizeof(double) * (kEnumMaxM + 1)));
        LP_TRY(hipMalloc(&p->dvi, sizeof(int) * (kEnumMaxM + 1)));
        LP_TRY(hipHostMalloc(&p->h_pass, sizeof(EnumPassBlock)));
        std::memset(p->h_pass, 0, sizeof(EnumPassBlock));
        p->h_result = &p->h_pass->result;
        p->h_list_count = &p->h_pass->list_count;
        p->h_overflow = &p->h_pass->overflow;
        p->h_level_counts = p->h_pass->level_counts;
        LP_TRY(hipEventCreate(&p->ev0));
        LP_TRY(hipEventCreate(&p->ev1));
        LP_TRY(hipMemcpyAsync(p->dbinom, binom.data(), sizeof(unsigned long long) * binom.size(),
                              hipMemcpyHostToDevice, s));
    }
    LP_TRY(hipMemcpyAsync(p->dA, Arow.data(), sizeof(double) * Arow.size(), hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(p->db, b, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(p->dc, c, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    if (fresh) {   // shared-prefix path: small control words, the feasible list
        PrefixDev& pd = p->prefix;
        pd.level_counts = p->d_pass->level_counts;
        LP_TRY(hipMalloc(&pd.item_count, 2 * sizeof(int)));
        LP_TRY(hipHostMalloc(&p->h_item_count, sizeof(int)));
        pd.overflow = &p->d_pass->overflow;
        LP_TRY(hipMalloc(&pd.root_cursor, 2 * sizeof(int)));
        LP_TRY(enum_list_alloc(p, want_cap));
        pd.list_count = &p->d_pass->list_count;
        if (!ctx->dcomb6 || !ctx->dcomb5 || !ctx->dcomb4) {   // (shape-independent: once per context)
            // leaf kernel: every 6-subset of R <= 22 columns in lexicographic order, 5 bits per index
            std::vector<unsigned> comb6(32, 0u);
            for (int R = 6; R <= 22; ++R) {
                comb6[(size_t)R] = (unsigned)comb6.size();
                int s6[6] = {0, 1, 2, 3, 4, 5};
                for (;;) {
                    unsigned pk = 0;
                    for (int t = 0; t < 6; ++t) pk |= (unsigned)s6[t] << (5 * t);
                    comb6.push_back(pk);
                    int t = 5;
                    while (t >= 0 && s6[t] == R - 6 + t) --t;
                    if (t < 0) break;
                    ++s6[t];
                    for (int u = t + 1; u < 6; ++u) s6[u] = s6[u - 1] + 1;
                }
            }
            LP_TRY(hipMalloc(&ctx->dcomb6, sizeof(unsigned) * comb6.size()));
            LP_TRY(hipMemcpyAsync(ctx->dcomb6, comb6.data(), sizeof(unsigned) * comb6.size(), hipMemcpyHostToDevice, s));
            // second level of the leaf kernel: every 5-subset of R <= 21 columns, same packing
            std::vector<unsigned> comb5(32, 0u);
            for (int R = 5; R <= 21; ++R) {
                comb5[(size_t)R] = (unsigned)comb5.size();
                int s5[5] = {0, 1, 2, 3, 4};
                for (;;) {
                    unsigned pk = 0;
                    for (int t = 0; t < 5; ++t) pk |= (unsigned)s5[t] << (5 * t);
                    comb5.push_back(pk);
                    int t = 4;
                    while (t >= 0 && s5[t] == R - 5 + t) --t;
                    if (t < 0) break;
                    ++s5[t];
                    for (int u = t + 1; u < 5; ++u) s5[u] = s5[u - 1] + 1;
                }
            }
            LP_TRY(hipMalloc(&ctx->dcomb5, sizeof(unsigned) * comb5.size()));
            LP_TRY(hipMemcpyAsync(ctx->dcomb5, comb5.data(), sizeof(unsigned) * comb5.size(), hipMemcpyHostToDevice, s));
            // third level: every 4-subset of R <= 20 columns
            std::vector<unsigned> comb4(32, 0u);
            for (int R = 4; R <= 20; ++R) {
                comb4[(size_t)R] = (unsigned)comb4.size();
                int s4[4] = {0, 1, 2, 3};
                for (;;) {
                    unsigned pk = 0;
                    for (int t = 0; t < 4; ++t) pk |= (unsigned)s4[t] << (5 * t);
                    comb4.push_back(pk);
                    int t = 3;
                    while (t >= 0 && s4[t] == R - 4 + t) --t;
                    if (t < 0) break;
                    ++s4[t];
                    for (int u = t + 1; u < 4; ++u) s4[u] = s4[u - 1] + 1;
                }
            }
            LP_TRY(hipMalloc(&ctx->dcomb4, sizeof(unsigned) * comb4.size()));
            LP_TRY(hipMemcpyAsync(ctx->dcomb4, comb4.data(), sizeof(unsigned) * comb4.size(), hipMemcpyHostToDevice, s));
            LP_TRY(hipStreamSynchronize(s));  // comb6 / comb5 / comb4 are locals
        }
        pd.comb6 = ctx->dcomb6;
        pd.comb5 = ctx->dcomb5;
        pd.comb4 = ctx->dcomb4;
    }
    LP_TRY(hipStreamSynchronize(s));
#undef LP_TRY
    d.A = p->dA;
    d.b = p->db;
    d.c = p->dc;
    d.binom = p->dbinom;
    p->complete = true;
    *problem_out = p;
    return LP_OPTIMAL;
}

static int check_range(lp_enum_problem* p, uint64_t begin, uint64_t end) {
    const uint64_t total = lp_host_binom(p->dev.n, p->dev.m);
    if (begin > end || end > total) LP_FAIL(p->ctx, LP_BAD_ARG, "rank range outside [0, C(n,m)]");
    return LP_OPTIMAL;
}

// Makes the feasible list hold `nfeas` entries (20 bytes each) if the device has the memory.
static bool enum_list_grow(lp_enum_problem* p, uint64_t nfeas) {
    if (getenv("LP_ENUM_LIST_CAP")) return false;   // tests pin the list to exercise the sub-range path
    const unsigned long long old_cap = p->prefix.list_cap;
    const unsigned long long want = nfeas + nfeas / 16 + 4096;
    if (want <= old_cap) return false;
    (void)hipSetDevice(p->ctx->device);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    if ((size_t)want * 20 + (size_t(2) << 30) > free_b) return false;
    enum_list_release(p);
    if (enum_list_alloc(p, want) == hipSuccess) return true;
    (void)hipGetLastError();
    if (enum_list_alloc(p, old_cap) != hipSuccess) p->complete = false;   // (cannot happen: it was just released)
    return false;
}

// One shared-prefix pass over [begin, end).  A list that overflows (a degenerate LP: up to every
// non-singular basis is feasible) either gives way to the dense form (a large part of the range
// feasible) or is re-allocated for the count the pass reported, if the device has the memory, and the
// pass runs once more.
static int enum_prefix_pass(lp_enum_problem* p, uint64_t begin, uint64_t end, double* score, uint64_t counts[3],
                            lp_enum_stats* stats) {
    const bool pinned = getenv("LP_ENUM_LIST_CAP") != nullptr;   // tests pin the list to exercise the sub-range path
    const bool may_dense = p->dev.m >= 7 && !pinned;
    auto again = [&](bool dense) {   // one more pass, times and launches added up
        lp_enum_stats first{};
        if (stats) first = *stats;
        const int rc = lp_enum_prefix_range(p, begin, end, score, counts, stats, dense);
        if (stats) {
            stats->kernel_ms += first.kernel_ms;
            stats->launches += first.launches;
        }
        return rc;
    };
    int rc = lp_enum_prefix_range(p, begin, end, score, counts, stats, p->dense_hint && may_dense);
    if (p->dense_active) {
        if (rc == LP_OPTIMAL && counts[0] * 8 < end - begin) p->dense_hint = false;   // not that degenerate after all
        return rc;
    }
    if (rc == kEnumListOverflow && may_dense &&
        (*p->h_list_count > p->prefix.list_abort || *p->h_list_count * 3 > end - begin)) {
        // more than a third of the range is feasible (the pass reported the count), or the pass stopped
        // early on a list 16x over capacity: the dense form — no list, every subset's score by rank —
        // and later passes of this problem start there
        p->dense_hint = true;
        rc = again(true);
        if (p->dense_active) return rc;
    }
    if (rc == kEnumListOverflow && enum_list_grow(p, *p->h_list_count)) rc = again(false);
    return rc;
}

// Shared-prefix enumeration of a range in sub-ranges: because its depth m-7 nodes do not fit the level
// buffers (large shapes: C(n-7, m-7) records), or because its feasible subsets overflow a list that
// cannot grow.  A sub-range that still does not fit is split again.  Counts add, the best score is
// the maximum; every sub-range keeps its best score for pass 2.
static int enum_prefix_chunked(lp_enum_problem* p, uint64_t begin, uint64_t end, uint64_t parts0,
                               double* score_best, uint64_t counts[3], lp_enum_stats* stats) {
    p->pchunks.clear();
    p->pchunks_valid = false;
    struct Part { uint64_t b, e; };
    std::vector<Part> todo;
    auto split = [&](uint64_t b, uint64_t e, uint64_t parts) {   // pushes in DEscending order (stack)
        if (parts < 2) parts = 2;
        if (parts > e - b) parts = e - b;
        for (uint64_t k = parts; k-- > 0;) {
            const uint64_t pb = b + (e - b) / parts * k + std::min<uint64_t>(k, (e - b) % parts);
            const uint64_t pe = b + (e - b) / parts * (k + 1) + std::min<uint64_t>(k + 1, (e - b) % parts);
            todo.push_back({pb, pe});
        }
    };
    split(begin, end, parts0);
    double best = -INFINITY;
    float ms = 0.f;
    int launches = 0;
    for (int k = 0; k < 3; ++k) counts[k] = 0;
    while (!todo.empty()) {
        const Part part = todo.back();
        todo.pop_back();
        double sc = -INFINITY;
        uint64_t cn[3] = {0, 0, 0};
        lp_enum_stats st{};
        int rc = enum_prefix_pass(p, part.b, part.e, &sc, cn, &st);
        if (rc == kEnumListOverflow && part.e - part.b > 1 && *p->h_list_count * 2 <= part.e - part.b) {
            split(part.b, part.e, *p->h_list_count / (p->prefix.list_cap / 2) + 1);
            continue;
        }
        if (rc == kEnumRangeTooWide && part.e - part.b > 1) {
            split(part.b, part.e, p->split_hint + 1);
            continue;
        }
        bool direct = false;
        if (rc == LP_ITER_LIMIT || rc == kEnumListOverflow || rc == kEnumRangeTooWide) {   // no memory for the level buffers
            rc = lp_enum_direct_range(p, part.b, part.e, &sc, cn, &st);
            direct = true;
        }
        if (rc) return rc;
        p->pchunks.push_back({part.b, part.e, sc, direct});
        if (sc > best) best = sc;
        for (int k = 0; k < 3; ++k) counts[k] += cn[k];
        ms += st.kernel_ms;
        launches += st.launches;
    }
    p->list_valid = false;
    p->spec_valid = false;
    p->pchunks_valid = true;
    p->pchunks_begin = begin;
    p->pchunks_end = end;
    *score_best = best;
    if (stats) {
        stats->kernel_ms = ms;
        stats->subsets = end - begin;
        stats->launches = launches;
    }
    return LP_OPTIMAL;
}

int lp_enum_range(lp_enum_problem* p, uint64_t rank_begin, uint64_t rank_end, int algo,
                  double* zbest_out, uint64_t* counts_out, lp_enum_stats* stats_out) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = check_range(p, rank_begin, rank_end);
    if (rc) return rc;
    double score = -INFINITY;
    uint64_t counts[3] = {0, 0, 0};
    p->list_valid = false;
    p->spec_valid = false;
    p->pchunks_valid = false;
    if (algo != LP_ENUM_ALGO_AUTO && algo != LP_ENUM_ALGO_DIRECT && algo != LP_ENUM_ALGO_PREFIX)
        LP_FAIL(ctx, LP_BAD_ARG, "unknown enumeration algorithm id");
    if (rank_begin == rank_end) {   // an empty shard (more processes than subsets): nothing to launch
        if (zbest_out) *zbest_out = p->dev.maximize ? -INFINITY : INFINITY;
        if (counts_out)
            for (int k = 0; k < 3; ++k) counts_out[k] = 0;
        if (stats_out) std::memset(stats_out, 0, sizeof(*stats_out));
        return LP_INFEASIBLE;
    }
    if (algo == LP_ENUM_ALGO_AUTO) {
        // the shared-prefix path pays for its breadth-first levels (0.13-0.25 ms of launches) from ~2^15
        // subsets on; with 32-row records (m > 16) from the start — the direct kernel's 32-lane form is
        // five times slower per subset than its 16-lane form (scripts/enum_threshold.py)
        const int shape = lp_enum_prefix_shape(p);
        const uint64_t least = shape == 3 ? (1ULL << 8) : (1ULL << 15);
        algo = (shape != 0 && rank_end - rank_begin >= least) ? LP_ENUM_ALGO_PREFIX : LP_ENUM_ALGO_DIRECT;
    }
    switch (algo) {
        case LP_ENUM_ALGO_PREFIX:
            if (!lp_enum_prefix_supported(p))
                LP_FAIL(ctx, LP_BAD_ARG, "shared-prefix enumeration needs 6 <= m <= 32 and n-m >= 2 (m >= 7 beyond 16 x 16; n-m <= 32 for m > 16)");
            rc = enum_prefix_pass(p, rank_begin, rank_end, &score, counts, stats_out);
            if (rc == kEnumRangeTooWide) {
                // more depth m-7 nodes than the level buffers hold: sub-ranges, a quarter over the
                // exact ratio (equal rank counts do not hold equal node counts)
                rc = enum_prefix_chunked(p, rank_begin, rank_end, p->split_hint + p->split_hint / 4 + 1, &score, counts,
                                         stats_out);
            } else if (rc == kEnumListOverflow) {
                // no memory for a list that long (or LP_ENUM_LIST_CAP pins its size).  Without the list
                // every feasible subset would be solved again from scratch for its objective, so once
                // more than half of the range is feasible the shared prefixes save nothing: that range
                // goes to the direct kernel as a whole; otherwise it is enumerated in sub-ranges, one
                // list at a time.
                if (*p->h_list_count * 2 > rank_end - rank_begin)
                    rc = LP_ITER_LIMIT;
                else
                    rc = enum_prefix_chunked(p, rank_begin, rank_end, *p->h_list_count / (p->prefix.list_cap / 2) + 1,
                                             &score, counts, stats_out);
            }
            if (rc != LP_ITER_LIMIT) break;
            // no memory for the level buffers of this problem: direct path
            [[fallthrough]];
        case LP_ENUM_ALGO_DIRECT:
            rc = lp_enum_direct_range(p, rank_begin, rank_end, &score, counts, stats_out);
            break;
        default:
            LP_FAIL(ctx, LP_BAD_ARG, "unknown enumeration algorithm id");
    }
    if (rc) return rc;
    if (zbest_out) *zbest_out = p->dev.maximize ? score : -score;
    if (counts_out)
        for (int k = 0; k < 3; ++k) counts_out[k] = counts[k];
    return counts[0] ? LP_OPTIMAL : LP_INFEASIBLE;
}

int lp_enum_first_within(lp_enum_problem* p, uint64_t rank_begin, uint64_t rank_end, double zstar,
                         double tol, uint64_t* rank_out) {
    if (!p || !rank_out) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = check_range(p, rank_begin, rank_end);
    if (rc) return rc;
    const double star = p->dev.maximize ? zstar : -zstar;
    if (p->list_valid && p->list_begin == rank_begin && p->list_end == rank_end) {
        if (p->spec_valid && star == p->spec_star && tol == p->spec_tol) {
            *rank_out = p->spec_first;  // already applied on the device by the range pass
            return LP_OPTIMAL;
        }
        return lp_enum_list_first(p, star, tol, rank_out);  // every feasible subset is listed
    }
    if (p->pchunks_valid && p->pchunks_begin == rank_begin && p->pchunks_end == rank_end) {
        // the range was enumerated in sub-ranges (ascending): the first one that holds a qualifying
        // subset is re-run to rebuild its list; the others are skipped on their best score
        const std::vector<lp_enum_problem::PrefixChunk> chunks = p->pchunks;
        *rank_out = UINT64_MAX;
        for (const auto& ch : chunks) {
            if (!(ch.best >= star - tol)) continue;
            uint64_t first = UINT64_MAX;
            if (ch.direct) {
                rc = lp_enum_direct_first(p, ch.begin, ch.end, star, tol, &first);
            } else {
                double sc;
                uint64_t cn[3];
                rc = enum_prefix_pass(p, ch.begin, ch.end, &sc, cn, nullptr);
                if (rc == LP_OPTIMAL) rc = lp_enum_list_first(p, star, tol, &first);
                else if (rc == LP_ITER_LIMIT || rc == kEnumListOverflow || rc == kEnumRangeTooWide)
                    rc = lp_enum_direct_first(p, ch.begin, ch.end, star, tol, &first);
            }
            if (rc) return rc;
            if (first != UINT64_MAX) {
                *rank_out = first;
                break;
            }
        }
        p->pchunks = chunks;   // (the re-runs do not disturb the record of the range pass)
        p->pchunks_valid = true;
        p->pchunks_begin = rank_begin;
        p->pchunks_end = rank_end;
        return LP_OPTIMAL;
    }
    return lp_enum_direct_first(p, rank_begin, rank_end, star, tol, rank_out);
}

int lp_enum_vertex(lp_enum_problem* p, uint64_t rank, int n_orig, double* x_out, int* basis_out,
                   double* obj_out, int* verdict_out) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    const EnumDev& d = p->dev;
    if (rank >= lp_host_binom(d.n, d.m)) LP_FAIL(ctx, LP_BAD_ARG, "rank >= C(n,m)");
    if (n_orig <= 0 || n_orig > d.n) LP_FAIL(ctx, LP_BAD_ARG, "bad original variable count");
    double xB[kEnumMaxM], z;
    int S[kEnumMaxM], verdict;
    int rc = lp_enum_direct_vertex(p, rank, xB, S, &z, &verdict);
    if (rc) return rc;
    if (x_out) {
        for (int j = 0; j < n_orig; ++j) x_out[j] = 0.0;
        for (int t = 0; t < d.m; ++t)
            if (S[t] < n_orig) x_out[S[t]] = xB[t];
    }
    if (basis_out)
        for (int t = 0; t < d.m; ++t) basis_out[t] = S[t];
    if (obj_out) *obj_out = z;
    if (verdict_out) *verdict_out = verdict;
    return LP_OPTIMAL;
}

int lp_enum_solve(lp_context* ctx, const double* A, int m, int n, const double* b,
                  const double* c, int maximize, int n_orig, double* x_out, int* basis_out,
                  uint64_t* rank_out, double* obj_out, uint64_t* counts_out) {
    if (!ctx) return LP_BAD_ARG;
    if (n_orig <= 0 || n_orig > n) LP_FAIL(ctx, LP_BAD_ARG, "bad original variable count");
    lp_enum_problem* p = nullptr;
    int rc = lp_enum_upload(ctx, A, m, n, b, c, maximize, &p);
    if (rc) return rc;
    const uint64_t total = lp_host_binom(n, m);
    double zstar = 0.0;
    rc = lp_enum_range(p, 0, total, LP_ENUM_ALGO_AUTO, &zstar, counts_out, nullptr);
    if (rc == LP_OPTIMAL) {
        uint64_t rank = UINT64_MAX;
        rc = lp_enum_first_within(p, 0, total, zstar, 1e-9, &rank);
        if (rc == LP_OPTIMAL && rank == UINT64_MAX) {
            ctx->last_error = "pass 2 found no rank within tolerance of the pass-1 optimum";
            rc = LP_INFEASIBLE;
        }
        if (rc == LP_OPTIMAL) {
            int verdict = 0;
            rc = lp_enum_vertex(p, rank, n_orig, x_out, basis_out, obj_out, &verdict);
            if (rank_out) *rank_out = rank;
        }
    }
    lp_enum_free(p);
    return rc;
}

// ===========================================================================
// batched simplex — one LP per workgroup (batched_simplex.hip).  LPs whose initial
// basis is not the slack identity, or whose condensed tableau does not fit one CU's
// LDS, go through the single-LP path one after another instead.
// ===========================================================================

struct lp_batched_problem {
    lp_context* ctx = nullptr;
    int batch = 0, m = 0, n = 0, n_orig = 0;
    bool resident = false;              // true: LDS-resident kernel; false: per-LP fallback
    BatchedDev dev{};
    double *dA = nullptr, *db = nullptr, *dc = nullptr, *dx = nullptr;
    int *dbasis_in = nullptr, *dbasis_out = nullptr, *diters = nullptr, *dstatus = nullptr;
    std::vector<double> h_c;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<lp_simplex_problem*> lps;  // fallback
    std::vector<int> status, iters;
};

void lp_batched_free(lp_batched_problem* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    for (auto* q : p->lps) lp_simplex_free(q);
    (void)hipFree(p->dA); (void)hipFree(p->db); (void)hipFree(p->dc); (void)hipFree(p->dx);
    (void)hipFree(p->dbasis_in); (void)hipFree(p->dbasis_out); (void)hipFree(p->diters);
    (void)hipFree(p->dstatus); (void)hipFree(p->dev.stamps);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    delete p;
}

int lp_batched_shard_bounds(int batch, int shard, int shards, int* lo, int* hi) {
    if (batch < 0 || shards < 1 || shard < 0 || shard >= shards || !lo || !hi) return LP_BAD_ARG;
    *lo = (int)((long long)batch * shard / shards);
    *hi = (int)((long long)batch * (shard + 1) / shards);
    return LP_OPTIMAL;
}

int lp_batched_upload(lp_context* ctx, int batch, const double* A, int m, int n, const double* b,
                      const double* c, const int* basis_in, int maximize, int n_orig,
                      lp_batched_problem** problem_out) {
    if (!ctx || !problem_out) return LP_BAD_ARG;
    *problem_out = nullptr;
    if (batch <= 0) LP_FAIL(ctx, LP_BAD_ARG, "batch must be positive");
    for (int k = 0; k < batch; ++k) {
        int rc = check_canonical(ctx, A ? A + (size_t)k * m * n : nullptr, m, n,
                                 b ? b + (size_t)k * m : nullptr, c ? c + (size_t)k * n : nullptr,
                                 basis_in ? basis_in + (size_t)k * m : nullptr, n_orig);
        if (rc) return rc;
    }
    LP_HIP(ctx, hipSetDevice(ctx->device));
    lp_batched_problem* p = new lp_batched_problem();
    p->ctx = ctx;
    p->batch = batch;
    p->m = m;
    p->n = n;
    p->n_orig = n_orig;
    p->status.assign((size_t)batch, -100);
    p->iters.assign((size_t)batch, 0);
    p->h_c.assign(c, c + (size_t)batch * n);
    // resident path needs: slack identity basis with zero basic costs in every LP, n > m,
    // and the condensed tableau within one CU's LDS
    bool identity = n > m;
    for (int k = 0; k < batch && identity; ++k) {
        const double* Ak = A + (size_t)k * m * n;
        const double* ck = c + (size_t)k * n;
        const int* bk = basis_in + (size_t)k * m;
        for (int t = 0; t < m && identity; ++t) {
            if (ck[bk[t]] != 0.0) identity = false;
            for (int i = 0; i < m && identity; ++i)
                if (Ak[(size_t)bk[t] * m + i] != ((i == t) ? 1.0 : 0.0)) identity = false;
        }
    }
    int pitch = 0;
    const size_t lds = lp_batched_lds_bytes(m, n, &pitch);
    p->resident = identity && lds <= 160 * 1024;
    if (!p->resident) {
        for (int k = 0; k < batch; ++k) {
            lp_simplex_problem* q = nullptr;
            int rc = lp_simplex_upload(ctx, A + (size_t)k * m * n, m, n, b + (size_t)k * m,
                                       c + (size_t)k * n, basis_in + (size_t)k * m, maximize,
                                       n_orig, &q);
            if (rc) {
                lp_batched_free(p);
                return rc;
            }
            p->lps.push_back(q);
        }
        *problem_out = p;
        return LP_OPTIMAL;
    }
#define LP_TRY(expr)                        \
    do {                                    \
        hipError_t _e = (expr);             \
        if (_e != hipSuccess) {             \
            ctx->last_error = #expr;        \
            lp_batched_free(p);             \
            return -(int)_e;                \
        }                                   \
    } while (0)
    hipStream_t s = ctx->stream;
    const size_t B = (size_t)batch;
    LP_TRY(hipMalloc(&p->dA, sizeof(double) * B * m * n));
    LP_TRY(hipMalloc(&p->db, sizeof(double) * B * m));
    LP_TRY(hipMalloc(&p->dc, sizeof(double) * B * n));
    LP_TRY(hipMalloc(&p->dx, sizeof(double) * B * n));
    LP_TRY(hipMalloc(&p->dbasis_in, sizeof(int) * B * m));
    LP_TRY(hipMalloc(&p->dbasis_out, sizeof(int) * B * m));
    LP_TRY(hipMalloc(&p->diters, sizeof(int) * B));
    LP_TRY(hipMalloc(&p->dstatus, sizeof(int) * B));
    LP_TRY(hipEventCreate(&p->ev0));
    LP_TRY(hipEventCreate(&p->ev1));
    LP_TRY(hipMemcpyAsync(p->dA, A, sizeof(double) * B * m * n, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(p->db, b, sizeof(double) * B * m, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(p->dc, c, sizeof(double) * B * n, hipMemcpyHostToDevice, s));
    LP_TRY(hipMemcpyAsync(p->dbasis_in, basis_in, sizeof(int) * B * m, hipMemcpyHostToDevice, s));
    LP_TRY(hipStreamSynchronize(s));
#undef LP_TRY
    BatchedDev& d = p->dev;
    d.batch = batch;
    d.m = m;
    d.n = n;
    d.pitch = pitch;
    d.maximize = maximize ? 1 : 0;
    d.A = p->dA;
    d.b = p->db;
    d.c = p->dc;
    d.basis_in = p->dbasis_in;
    d.x = p->dx;
    d.basis_out = p->dbasis_out;
    d.iters = p->diters;
    d.status = p->dstatus;
    *problem_out = p;
    return LP_OPTIMAL;
}

int lp_batched_run(lp_batched_problem* p, double eps, int max_iter, float* ms_out) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    if (p->resident) {
        p->dev.eps = eps;
        p->dev.max_iter = max_iter;
        if (const char* sv = getenv("LP_BATCHED_STAMPS"); sv && !p->dev.stamps) {   // diagnostic build of the kernel (scripts/stamp_batched.py)
            LP_HIP(ctx, hipMalloc(&p->dev.stamps, sizeof(unsigned long long) * 64));   // 32 phase sums + 2 per wave (16 waves)
            LP_HIP(ctx, hipMemset(p->dev.stamps, 0, sizeof(unsigned long long) * 64));
            p->dev.stamps_reg = std::strcmp(sv, "reg") == 0;
        }
        LP_HIP(ctx, hipEventRecord(p->ev0, ctx->stream));
        int rc = lp_batched_launch(ctx, p->dev);
        if (rc) return rc;
        LP_HIP(ctx, hipEventRecord(p->ev1, ctx->stream));
        LP_HIP(ctx, hipEventSynchronize(p->ev1));
        LP_HIP(ctx, hipGetLastError());
        float ms = 0.f;
        LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
        if (ms_out) *ms_out = ms;
        if (p->dev.stamps) {
            unsigned long long h[64];
            LP_HIP(ctx, hipMemcpy(h, p->dev.stamps, sizeof(h), hipMemcpyDeviceToHost));
            if (p->dev.stamps_reg) {
                fprintf(stderr, "[batched stamps, register form] per wave: pricing | update phase, then the wait at the loop's barrier (incl. the entering column's hand-over):");
                for (int w = 0; w < 8; ++w)
                    fprintf(stderr, "  w%d %.0f+%.0f", w, (double)h[32 + 2 * w] / (double)(h[8] ? h[8] : 1), (double)h[33 + 2 * w] / (double)(h[8] ? h[8] : 1));
                fprintf(stderr, "\n[batched stamps, register form] hand-over per wave:");
                for (int w = 0; w < 8; ++w) fprintf(stderr, "  w%d %.0f", w, (double)h[48 + w] / (double)(h[8] ? h[8] : 1));
                fprintf(stderr, "\n");
                const char* names[8] = {"entering column -> LDS", "barrier", "ratio test | (idle)", "barrier",
                                        "eta column + pivot row -> LDS", "barrier",
                                        "reduced costs + pricing | rank-1 update", "barrier"};
                fprintf(stderr, "[batched stamps, register form] workgroup 0, %llu pivots, %.3f ms: cycles per pivot, wave 0 | wave 1\n", h[8], ms);
                for (int q = 0; q < 8; ++q)
                    fprintf(stderr, "[batched stamps]   %-42s %8.0f | %8.0f\n", names[q], (double)h[q] / (double)(h[8] ? h[8] : 1),
                            (double)h[16 + q] / (double)(h[24] ? h[24] : 1));
            } else {
                const char* names[6] = {"reduced costs + pricing | rank-1 update", "barrier", "ratio test | (idle)", "barrier",
                                        "eta column + pivot-row copy", "barrier"};
                fprintf(stderr, "[batched stamps] workgroup 0, %llu pivots, %.3f ms: cycles per pivot, wave 0 | wave 1\n", h[6], ms);
                for (int q = 0; q < 6; ++q)
                    fprintf(stderr, "[batched stamps]   %-42s %8.0f | %8.0f\n", names[q], (double)h[q] / (double)(h[6] ? h[6] : 1),
                            (double)h[8 + q] / (double)(h[14] ? h[14] : 1));
            }
        }
        return LP_OPTIMAL;
    }
    float total = 0.f;
    for (int k = 0; k < p->batch; ++k) {
        lp_simplex_stats st;
        int rc = lp_simplex_reset(p->lps[(size_t)k]);
        if (rc) return rc;
        rc = lp_simplex_run(p->lps[(size_t)k], eps, max_iter, LP_SIMPLEX_ALGO_AUTO, &st);
        if (rc < 0) return rc;
        p->status[(size_t)k] = rc;
        p->iters[(size_t)k] = st.pivots;
        total += st.solve_ms;
    }
    if (ms_out) *ms_out = total;
    return LP_OPTIMAL;
}

int lp_batched_download(lp_batched_problem* p, double* x_out, int* basis_out, double* obj_out,
                        int* iters_out, int* status_out) {
    if (!p) return LP_BAD_ARG;
    lp_context* ctx = p->ctx;
    LP_HIP(ctx, hipSetDevice(ctx->device));
    if (p->resident) {
        const size_t B = (size_t)p->batch;
        std::vector<double> x(B * p->n);
        hipStream_t s = ctx->stream;
        LP_HIP(ctx, hipMemcpyAsync(x.data(), p->dx, sizeof(double) * B * p->n, hipMemcpyDeviceToHost, s));
        LP_HIP(ctx, hipMemcpyAsync(p->status.data(), p->dstatus, sizeof(int) * B, hipMemcpyDeviceToHost, s));
        LP_HIP(ctx, hipMemcpyAsync(p->iters.data(), p->diters, sizeof(int) * B, hipMemcpyDeviceToHost, s));
        if (basis_out)
            LP_HIP(ctx, hipMemcpyAsync(basis_out, p->dbasis_out, sizeof(int) * B * p->m, hipMemcpyDeviceToHost, s));
        LP_HIP(ctx, hipStreamSynchronize(s));
        for (int k = 0; k < p->batch; ++k) {
            const bool ok = p->status[(size_t)k] == LP_OPTIMAL;
            const double* xk = x.data() + (size_t)k * p->n;
            if (x_out && ok)  // x.head(n_orig), SimplexSolover.h:435-438
                for (int j = 0; j < p->n_orig; ++j) x_out[(size_t)k * p->n_orig + j] = xk[j];
            if (obj_out && ok) {  // Canonical::Evaluate, Canonical.cpp:86
                double z = 0.0;
                const double* ck = p->h_c.data() + (size_t)k * p->n;
                for (int j = 0; j < p->n; ++j) z += ck[j] * xk[j];
                obj_out[k] = z;
            }
            if (iters_out) iters_out[k] = p->iters[(size_t)k];
            if (status_out) status_out[k] = p->status[(size_t)k];
        }
        return LP_OPTIMAL;
    }
    for (int k = 0; k < p->batch; ++k) {
        const bool ok = p->status[(size_t)k] == LP_OPTIMAL;
        int rc = lp_simplex_download(p->lps[(size_t)k],
                                     (x_out && ok) ? x_out + (size_t)k * p->n_orig : nullptr,
                                     basis_out ? basis_out + (size_t)k * p->m : nullptr,
                                     (obj_out && ok) ? obj_out + k : nullptr, nullptr, nullptr, 0,
                                     nullptr);
        if (rc) return rc;
        if (iters_out) iters_out[k] = p->iters[(size_t)k];
        if (status_out) status_out[k] = p->status[(size_t)k];
    }
    return LP_OPTIMAL;
}

int lp_simplex_solve_batched(lp_context* ctx, int batch, const double* A, int m, int n,
                             const double* b, const double* c, const int* basis_in, int maximize,
                             int n_orig, double eps, int max_iter, double* x_out, int* basis_out,
                             double* obj_out, int* iters_out, int* status_out) {
    lp_batched_problem* p = nullptr;
    int rc = lp_batched_upload(ctx, batch, A, m, n, b, c, basis_in, maximize, n_orig, &p);
    if (rc) return rc;
    rc = lp_batched_run(p, eps, max_iter, nullptr);
    if (rc == LP_OPTIMAL) rc = lp_batched_download(p, x_out, basis_out, obj_out, iters_out, status_out);
    lp_batched_free(p);
    return rc;
}

}  // extern "C"
