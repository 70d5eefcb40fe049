// enum_sharded.hip — vertex enumeration sharded over the GPUs of a node, from the C ABI.
//
// the combination-rank space [0, C(n,m)) is cut into `world` contiguous ranges of
// equal estimated cost (lp_enum_shard_bounds); every participant (one per GPU: a process, or a
// host thread of one process) enumerates its range on its own device with no data-path
// collective, and the only exchange is ONE all-gather of a 48-byte record per participant —
// {best score, smallest rank within 1e-9 of it, the three counts, status} — over RCCL/xGMI.  A
// participant whose own best IS the global optimum has already applied the tie rule against the
// right value, so the answer is the smallest of those ranks; only when another shard holds a
// different vertex within 1e-9 below the optimum is a second all-gather (recomputed ranks) needed.
// Every participant sees the same records and takes the same branch; the answer does not depend
// on where the cuts are (tie rule of SURVEY.md 8 row E1).  Same protocol as
// simplexmethod_amd/dist.py (torch.distributed), here for C++ hosts.
//
// RCCL is resolved with dlopen when the first communicator is created: a single-GPU user never
// loads the 570 MB library, and a process that already has an RCCL (PyTorch's) shares it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <mutex>

#include "enum_problem.hpp"

namespace {

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string error;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            api.error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
        if (!api.ok) api.error = "librccl lacks an expected symbol";
    });
    return api;
}

constexpr int kRecWords = 6;   // score bits, first rank, counts[3], status

// Exchange between host threads of ONE process (participants may share a device: more shards
// than GPUs).  A generation counter makes the group reusable call after call.
struct LocalGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, arrived = 0, refs = 0;
    unsigned long long generation = 0;
    std::vector<long long> slots, snapshot;
};

}  // namespace

struct lp_comm {
    int rank = 0, world = 1;
    lp_context* ctx = nullptr;
    // RCCL backend
    ncclComm_t nccl = nullptr;
    long long* dsend = nullptr;
    long long* drecv = nullptr;
    long long* hbuf = nullptr;   // pinned: kRecWords send words + world * kRecWords receive words
    // in-process backend
    LocalGroup* group = nullptr;
};

static int comm_allgather(lp_comm* c, const long long* rec, int words, long long* all) {
    if (!c || c->world == 1) {
        std::memcpy(all, rec, sizeof(long long) * (size_t)words);
        return LP_OPTIMAL;
    }
    if (c->group) {
        LocalGroup* g = c->group;
        std::unique_lock<std::mutex> lock(g->mu);
        std::memcpy(g->slots.data() + (size_t)c->rank * kRecWords, rec, sizeof(long long) * (size_t)words);
        const unsigned long long gen = g->generation;
        if (++g->arrived == g->world) {
            g->snapshot = g->slots;
            g->arrived = 0;
            ++g->generation;
            g->cv.notify_all();
        } else {
            g->cv.wait(lock, [&] { return g->generation != gen; });
        }
        for (int r = 0; r < c->world; ++r)
            std::memcpy(all + (size_t)r * words, g->snapshot.data() + (size_t)r * kRecWords,
                        sizeof(long long) * (size_t)words);
        return LP_OPTIMAL;
    }
    lp_context* ctx = c->ctx;
    RcclApi& api = rccl();
    LP_HIP(ctx, hipSetDevice(ctx->device));
    std::memcpy(c->hbuf, rec, sizeof(long long) * (size_t)words);
    LP_HIP(ctx, hipMemcpyAsync(c->dsend, c->hbuf, sizeof(long long) * (size_t)words, hipMemcpyHostToDevice, ctx->stream));
    const ncclResult_t nr = api.AllGather(c->dsend, c->drecv, (size_t)words, ncclInt64, c->nccl, ctx->stream);
    if (nr != ncclSuccess) {
        ctx->last_error = std::string("ncclAllGather: ") + api.GetErrorString(nr);
        return -1000 - (int)nr;
    }
    LP_HIP(ctx, hipMemcpyAsync(c->hbuf + kRecWords, c->drecv, sizeof(long long) * (size_t)words * c->world,
                               hipMemcpyDeviceToHost, ctx->stream));
    LP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(all, c->hbuf + kRecWords, sizeof(long long) * (size_t)words * c->world);
    return LP_OPTIMAL;
}

extern "C" {

int lp_comm_unique_id(void* id_out) {
    if (!id_out) return LP_BAD_ARG;
    RcclApi& api = rccl();
    if (!api.ok) return LP_BAD_ARG;
    ncclUniqueId id;
    if (api.GetUniqueId(&id) != ncclSuccess) return LP_BAD_ARG;
    std::memcpy(id_out, &id, sizeof(id));
    return LP_OPTIMAL;
}

int lp_comm_create_rccl(lp_context* ctx, int rank, int world, const void* unique_id, lp_comm** comm_out) {
    if (!ctx || !comm_out) return LP_BAD_ARG;
    *comm_out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !unique_id) LP_FAIL(ctx, LP_BAD_ARG, "lp_comm_create_rccl: bad rank / world / id");
    RcclApi& api = rccl();
    if (!api.ok) LP_FAIL(ctx, LP_BAD_ARG, api.error);
    LP_HIP(ctx, hipSetDevice(ctx->device));
    lp_comm* c = new lp_comm();
    c->rank = rank;
    c->world = world;
    c->ctx = ctx;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t nr = api.CommInitRank(&c->nccl, world, id, rank);   // blocks until every rank has joined
    if (nr != ncclSuccess) {
        ctx->last_error = std::string("ncclCommInitRank: ") + api.GetErrorString(nr);
        delete c;
        return -1000 - (int)nr;
    }
    hipError_t e = hipMalloc(&c->dsend, sizeof(long long) * kRecWords);
    if (e == hipSuccess) e = hipMalloc(&c->drecv, sizeof(long long) * kRecWords * (size_t)world);
    if (e == hipSuccess) e = hipHostMalloc(&c->hbuf, sizeof(long long) * kRecWords * (size_t)(world + 1));
    if (e != hipSuccess) {
        ctx->last_error = "lp_comm_create_rccl: allocation failed";
        lp_comm_destroy(c);
        return -(int)e;
    }
    *comm_out = c;
    return LP_OPTIMAL;
}

int lp_comm_create_local(int world, lp_comm** comms_out) {
    if (world < 1 || !comms_out) return LP_BAD_ARG;
    LocalGroup* g = new LocalGroup();
    g->world = world;
    g->refs = world;
    g->slots.assign((size_t)world * kRecWords, 0);
    g->snapshot = g->slots;
    for (int r = 0; r < world; ++r) {
        lp_comm* c = new lp_comm();
        c->rank = r;
        c->world = world;
        c->group = g;
        comms_out[r] = c;
    }
    return LP_OPTIMAL;
}

int lp_comm_rank(const lp_comm* c) { return c ? c->rank : 0; }
int lp_comm_world(const lp_comm* c) { return c ? c->world : 1; }

void lp_comm_destroy(lp_comm* c) {
    if (!c) return;
    if (c->group) {
        bool last;
        {
            std::lock_guard<std::mutex> lock(c->group->mu);
            last = --c->group->refs == 0;
        }
        if (last) delete c->group;
    }
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    (void)hipFree(c->dsend);
    (void)hipFree(c->drecv);
    if (c->hbuf) (void)hipHostFree(c->hbuf);
    delete c;
}

// A participant that cannot take part in the enumeration (its context, upload or communicator set-up
// failed) still owes the group its record: the others are waiting in the exchange.  Contributes a
// failed record (status != LP_OPTIMAL) to the ONE all-gather and returns that status; every other
// participant's lp_enum_solve_sharded then returns it too.
int lp_enum_shard_abstain(lp_comm* comm, int status) {
    if (status == LP_OPTIMAL) status = LP_BAD_ARG;
    if (!comm) return status;
    const int world = lp_comm_world(comm);
    long long rec[kRecWords] = {0, 0x7FFFFFFFFFFFFFFFLL, 0, 0, 0, status};
    const double none = -INFINITY;
    std::memcpy(&rec[0], &none, sizeof(double));
    std::vector<long long> all((size_t)world * kRecWords);
    const int rc = comm_allgather(comm, rec, kRecWords, all.data());
    return rc ? rc : status;
}

int lp_enum_solve_sharded(lp_comm* comm, lp_enum_problem* p, int n_orig, double* x_out, int* basis_out,
                          uint64_t* rank_out, double* obj_out, uint64_t* counts_out) {
    if (!p) return lp_enum_shard_abstain(comm, LP_BAD_ARG);   // nobody is left waiting in the exchange
    lp_context* ctx = p->ctx;
    const int rank = lp_comm_rank(comm), world = lp_comm_world(comm);
    const EnumDev& d = p->dev;
    constexpr double kTol = 1e-9;   // Solver::EPS, /root/reference/src/SimplexSolover.h:13
    constexpr long long kNone = 0x7FFFFFFFFFFFFFFFLL;
    // this participant's shard: the cost-balanced cut is host combinatorics (a binary search over
    // prefix ranks), computed once per (rank, world) and kept with the problem
    uint64_t lo = 0, hi = 0;
    int status = LP_OPTIMAL;
    if (p->shard_rank == rank && p->shard_world == world) {
        lo = p->shard_lo;
        hi = p->shard_hi;
    } else {
        status = lp_enum_shard_bounds(d.n, d.m, rank, world, &lo, &hi);
        if (status == LP_OPTIMAL) {
            p->shard_rank = rank;
            p->shard_world = world;
            p->shard_lo = lo;
            p->shard_hi = hi;
        }
    }
    // ---- my shard: pass 1, and the tie rule against my own best (no collective so far)
    double z = 0.0, score = -INFINITY;
    uint64_t counts[3] = {0, 0, 0}, first = UINT64_MAX;
    if (status == LP_OPTIMAL) {
        status = lp_enum_range(p, lo, hi, LP_ENUM_ALGO_AUTO, &z, counts, nullptr);
        if (status == LP_OPTIMAL) {
            score = d.maximize ? z : -z;
            if (!(score == score)) score = -INFINITY;
            if (score != -INFINITY) status = lp_enum_first_within(p, lo, hi, z, kTol, &first);
        } else if (status == LP_INFEASIBLE) {
            status = LP_OPTIMAL;   // no feasible subset in THIS shard
        }
    }
    // ---- the one exchange (a failed participant still takes part: nobody is left waiting)
    long long rec[kRecWords];
    std::memcpy(&rec[0], &score, sizeof(double));
    rec[1] = first > (uint64_t)kNone ? kNone : (long long)first;
    for (int k = 0; k < 3; ++k) rec[2 + k] = (long long)counts[k];
    rec[5] = status;
    std::vector<long long> all((size_t)world * kRecWords);
    int rc = comm_allgather(comm, rec, kRecWords, all.data());
    if (rc) return rc;
    uint64_t gcounts[3] = {0, 0, 0};
    double gscore = -INFINITY;
    for (int r = 0; r < world; ++r) {
        const long long* a = all.data() + (size_t)r * kRecWords;
        if (a[5] != LP_OPTIMAL) {
            if (rank != r) ctx->last_error = "lp_enum_solve_sharded: participant " + std::to_string(r) + " failed";
            return (int)a[5];
        }
        double s;
        std::memcpy(&s, &a[0], sizeof(double));
        if (s > gscore) gscore = s;
        for (int k = 0; k < 3; ++k) gcounts[k] += (uint64_t)a[2 + k];
    }
    if (counts_out)
        for (int k = 0; k < 3; ++k) counts_out[k] = gcounts[k];
    if (gscore == -INFINITY) {
        ctx->last_error = "enumeration: no feasible basis in any shard";
        return LP_INFEASIBLE;
    }
    const double zstar = d.maximize ? gscore : -gscore;
    long long grank = kNone;
    bool near = false;
    for (int r = 0; r < world; ++r) {
        const long long* a = all.data() + (size_t)r * kRecWords;
        double s;
        std::memcpy(&s, &a[0], sizeof(double));
        if (s == gscore && a[1] < grank) grank = a[1];
        if (s < gscore && s >= gscore - kTol) near = true;
    }
    if (near) {   // rare: another shard's best is a different vertex within the tolerance of the optimum
        uint64_t redo = UINT64_MAX;
        int st2 = LP_OPTIMAL;
        if (score >= gscore - kTol) st2 = lp_enum_first_within(p, lo, hi, zstar, kTol, &redo);
        long long rec2[2] = {redo > (uint64_t)kNone ? kNone : (long long)redo, st2};
        std::vector<long long> all2((size_t)world * 2);
        rc = comm_allgather(comm, rec2, 2, all2.data());
        if (rc) return rc;
        grank = kNone;
        for (int r = 0; r < world; ++r) {
            if (all2[(size_t)r * 2 + 1] != LP_OPTIMAL) return (int)all2[(size_t)r * 2 + 1];
            if (all2[(size_t)r * 2] < grank) grank = all2[(size_t)r * 2];
        }
    }
    if (grank == kNone) {
        ctx->last_error = "lp_enum_solve_sharded: no rank within tolerance of the optimum";
        return LP_INFEASIBLE;
    }
    if (rank_out) *rank_out = (uint64_t)grank;
    if (!x_out && !basis_out) {   // the optimum itself travelled with the records: no vertex kernel needed
        if (obj_out) *obj_out = zstar;
        return LP_OPTIMAL;
    }
    // the problem is replicated: every participant evaluates the winning vertex itself
    int verdict = 0;
    return lp_enum_vertex(p, (uint64_t)grank, n_orig, x_out, basis_out, obj_out, &verdict);
}

}  // extern "C"
