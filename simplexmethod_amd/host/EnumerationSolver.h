// EnumerationSolver.h — vertex enumeration over all C(n,m) bases.
//
// The reference only declares an empty class (/root/reference/src/EnumerationSolver.h:3-10);
// its specification is README.md:27 ("solve the same problem by enumerating extreme points")
// and README.md:40-42 (it cross-checks the simplex solver).  The API below has the shape of
// Solver: construct from a Canonical, solve() returns x.head(n_orig).  Semantics: SURVEY.md §8
// row E1 (see include/simplexmethod_amd.h).  Unboundedness cannot be detected by enumeration.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "Canonical.h"
#include "DeviceContext.h"

class EnumerationSolver {
public:
    struct Result {
        lpla::VectorXd x;          // x.head(n_orig) of the winning vertex
        std::vector<int> basis;    // its basis, sorted ascending
        double objective = 0.0;
        uint64_t rank = 0;         // lexicographic rank of the winning subset
        uint64_t feasible = 0, infeasible = 0, singular = 0;
        int status = LP_OPTIMAL;
    };

    explicit EnumerationSolver(const Canonical& problem, int device = 0)
        : _problem(problem), _device(device) {}

    // Throws std::runtime_error("No feasible basis") when no basis is feasible.
    lpla::VectorXd solve() { return solve_ex().x; }

    enum Exchange { EXCHANGE_AUTO = 0, EXCHANGE_RCCL = 1, EXCHANGE_LOCAL = 2 };

    // n_gpus > 1: the rank space is cut into n_gpus contiguous shards of equal estimated cost
    // (lp_enum_shard_bounds), one host thread, one lp_context and one replica of the problem per
    // shard; shard g runs on HIP device (device + g) % lp_device_count(), so a box with fewer GPUs
    // than shards still exercises the sharded path.  The incumbent is exchanged by
    // lp_enum_solve_sharded: ONE all-gather of a 48-byte record per shard — over RCCL/xGMI when every
    // shard has a device of its own (EXCHANGE_AUTO), through host memory when shards share devices.
    // The answer is identical for every n_gpus (tie rule of SURVEY.md 8 row E1).
    Result solve_ex(int n_gpus = 1, bool throw_on_failure = true, Exchange exchange = EXCHANGE_AUTO) {
        const lpla::MatrixXd& A = _problem.GetConstraintsMatrix();
        const lpla::VectorXd& b = _problem.GetRightHandSide();
        const lpla::VectorXd& c = _problem.GetObjectiveCoefficients();
        const int n_orig = _problem.GetOriginalVariablesCount();
        const int m = (int)A.rows(), n = (int)A.cols();
        const int maximize = _problem.IsMaximization() ? 1 : 0;
        Result r;
        r.x = lpla::VectorXd::Zero(n_orig);
        r.basis.assign((size_t)m, -1);
        if (n_gpus <= 1 && exchange != EXCHANGE_RCCL) {
            lp_context* ctx = lpgpu::context(_device);
            uint64_t counts[3] = {0, 0, 0};
            r.status = lp_enum_solve(ctx, A.data(), m, n, b.data(), c.data(), maximize, n_orig,
                                     r.x.data(), r.basis.data(), &r.rank, &r.objective, counts);
            r.feasible = counts[0];
            r.infeasible = counts[1];
            r.singular = counts[2];
            if (throw_on_failure) lpgpu::throw_for_status(r.status, ctx);
            return r;
        }
        if (n_gpus < 1) n_gpus = 1;
        if (lp_binom(n, m) == 0) throw std::invalid_argument("C(n,m) does not fit 64 bits");
        const int ndev = lp_device_count();
        if (ndev < 1) throw std::runtime_error("simplexmethod_amd: no HIP device (there is no CPU fallback)");
        const bool rccl = exchange == EXCHANGE_RCCL || (exchange == EXCHANGE_AUTO && n_gpus <= ndev);
        if (exchange == EXCHANGE_RCCL && n_gpus > ndev)
            throw std::invalid_argument("RCCL exchange needs one device per shard");
        struct Shard {
            lp_context* ctx = nullptr;
            lp_enum_problem* p = nullptr;
            lp_comm* comm = nullptr;
            Result res;
            int rc = LP_OPTIMAL;
            std::string error;
        };
        std::vector<Shard> sh((size_t)n_gpus);
        std::vector<lp_comm*> local((size_t)n_gpus, nullptr);
        auto release = [&]() {
            for (auto& s : sh) {
                lp_enum_free(s.p);
                lp_comm_destroy(s.comm);
                lp_context_destroy(s.ctx);
                s.p = nullptr; s.comm = nullptr; s.ctx = nullptr;
            }
        };
        // Set-up, BEFORE anybody can be waiting in a collective: every shard's context (device binding +
        // stream of its own: shards may share a device) and replica of the problem.  A device that cannot
        // be opened, or an upload that fails, ends the whole call here — no thread has been started and
        // nobody has entered ncclCommInitRank, so there is nobody to leave behind.
        for (int g = 0; g < n_gpus; ++g) {
            Shard& s = sh[(size_t)g];
            s.res.x = lpla::VectorXd::Zero(n_orig);
            s.res.basis.assign((size_t)m, -1);
            s.rc = lp_context_create(shard_device(g, ndev), nullptr, &s.ctx);
            if (s.rc == LP_OPTIMAL) s.rc = lp_enum_upload(s.ctx, A.data(), m, n, b.data(), c.data(), maximize, &s.p);
            if (s.rc != LP_OPTIMAL) {
                const int rc = s.rc;
                const std::string why = "shard " + std::to_string(g) + " (device " + std::to_string(shard_device(g, ndev)) +
                                        "): " + (s.ctx ? lp_last_error(s.ctx) : "lp_context_create failed");
                release();
                r.status = rc;
                if (!throw_on_failure) return r;
                if (rc == LP_BAD_ARG) throw std::invalid_argument("bad argument: " + why);
                throw std::runtime_error("sharded enumeration: set-up failed (status " + std::to_string(rc) + "): " + why);
            }
        }
        unsigned char id[128] = {0};
        if (rccl) {
            if (lp_comm_unique_id(id) != LP_OPTIMAL) {
                release();
                throw std::runtime_error("simplexmethod_amd: RCCL is not available");
            }
        } else if (lp_comm_create_local(n_gpus, local.data()) != LP_OPTIMAL) {
            release();
            throw std::runtime_error("simplexmethod_amd: lp_comm_create_local failed");
        }
        auto run = [&](int g) {
            Shard& s = sh[(size_t)g];
            if (rccl) {   // collective: every thread joins (every context exists)
                lp_comm* cm = nullptr;
                s.rc = lp_comm_create_rccl(s.ctx, g, n_gpus, id, &cm);
                s.comm = cm;
            } else {
                s.comm = local[(size_t)g];
            }
#ifdef LP_HOST_TEST_HOOKS
            if (s.rc == LP_OPTIMAL && g == _debug_fail_shard) s.rc = LP_BAD_ARG;
#endif
            if (s.rc == LP_OPTIMAL) {
                uint64_t counts[3] = {0, 0, 0};
                s.rc = lp_enum_solve_sharded(s.comm, s.p, n_orig, s.res.x.data(), s.res.basis.data(), &s.res.rank,
                                             &s.res.objective, counts);
                s.res.feasible = counts[0];
                s.res.infeasible = counts[1];
                s.res.singular = counts[2];
            } else {
                // a shard that cannot enumerate still owes the others its record: they are waiting in
                // the exchange (a NULL communicator — RCCL set-up failed — has nobody to tell)
                const std::string why = lp_last_error(s.ctx);
                s.rc = lp_enum_shard_abstain(s.comm, s.rc);
                s.error = why;
            }
            if (s.rc != LP_OPTIMAL && s.error.empty()) s.error = lp_last_error(s.ctx);
            s.res.status = s.rc;
        };
        std::vector<std::thread> th;
        th.reserve((size_t)n_gpus);   // (the only allocation between lp_comm_create_local and the joins)
        for (int g = 0; g < n_gpus; ++g) th.emplace_back(run, g);
        for (auto& t : th) t.join();
        int rc = LP_OPTIMAL;
        std::string error;
        for (auto& s : sh)   // the first failing shard decides (every shard fails together after the exchange)
            if (s.rc != LP_OPTIMAL && rc == LP_OPTIMAL) {
                rc = s.rc;
                error = s.error;
            }
        r = sh[0].res;
        r.status = rc;
        release();
        if (rc != LP_OPTIMAL && throw_on_failure) {
            if (rc == LP_INFEASIBLE) throw std::runtime_error("No feasible basis");
            if (rc == LP_BAD_ARG) throw std::invalid_argument("bad argument: " + error);
            throw std::runtime_error("sharded enumeration failed (status " + std::to_string(rc) + "): " + error);
        }
        return r;
    }

#ifdef LP_HOST_TEST_HOOKS
    // Test builds only (tests/cpp, -DLP_HOST_TEST_HOOKS).  debug_fail_shard: shard g fails after the set-up, as
    // if its pass had failed (it must still join the exchange and every shard must come back with its status).
    // debug_shard_device: shard g is bound to HIP device `device` (e.g. one that does not exist).
    void debug_fail_shard(int g) { _debug_fail_shard = g; }
    void debug_shard_device(int g, int device) { _debug_dev_shard = g; _debug_dev = device; }
#endif

private:
    // shard g runs on HIP device (device + g) % device count
    int shard_device(int g, int ndev) const {
#ifdef LP_HOST_TEST_HOOKS
        if (g == _debug_dev_shard) return _debug_dev;
#endif
        return (_device + g) % ndev;
    }
    Canonical _problem;
    int _device;
#ifdef LP_HOST_TEST_HOOKS
    int _debug_fail_shard = -1, _debug_dev_shard = -1, _debug_dev = 0;
#endif
};
