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

    // n_gpus > 1: the rank space is cut into n_gpus contiguous shards of equal estimated cost
    // (lp_enum_shard_bounds), one host thread and one
    // HIP device per shard; the incumbent is reduced on the host (a single process needs no
    // RCCL — the one-process-per-GPU form over RCCL is simplexmethod_amd/dist.py).
    Result solve_ex(int n_gpus = 1, bool throw_on_failure = true) {
        const lpla::MatrixXd& A = _problem.GetConstraintsMatrix();
        const lpla::VectorXd& b = _problem.GetRightHandSide();
        const lpla::VectorXd& c = _problem.GetObjectiveCoefficients();
        const int n_orig = _problem.GetOriginalVariablesCount();
        const int m = (int)A.rows(), n = (int)A.cols();
        const int maximize = _problem.IsMaximization() ? 1 : 0;
        Result r;
        r.x = lpla::VectorXd::Zero(n_orig);
        r.basis.assign((size_t)m, -1);
        if (n_gpus <= 1) {
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
        const uint64_t total = lp_binom(n, m);
        if (total == 0) throw std::invalid_argument("C(n,m) does not fit 64 bits");
        struct Shard {
            lp_context* ctx = nullptr;
            lp_enum_problem* p = nullptr;
            uint64_t lo = 0, hi = 0, counts[3] = {0, 0, 0}, first = UINT64_MAX;
            double z = 0.0;
            int rc = LP_OPTIMAL;
        };
        std::vector<Shard> sh((size_t)n_gpus);
        for (int g = 0; g < n_gpus; ++g) {
            sh[(size_t)g].ctx = lpgpu::context(_device + g);
            if (lp_enum_shard_bounds(n, m, g, n_gpus, &sh[(size_t)g].lo, &sh[(size_t)g].hi) != LP_OPTIMAL)
                throw std::invalid_argument("lp_enum_shard_bounds: bad problem shape");
        }
        auto pass1 = [&](int g) {
            Shard& s = sh[(size_t)g];
            s.rc = lp_enum_upload(s.ctx, A.data(), m, n, b.data(), c.data(), maximize, &s.p);
            if (s.rc == LP_OPTIMAL)
                s.rc = lp_enum_range(s.p, s.lo, s.hi, LP_ENUM_ALGO_AUTO, &s.z, s.counts, nullptr);
        };
        std::vector<std::thread> th;
        for (int g = 0; g < n_gpus; ++g) th.emplace_back(pass1, g);
        for (auto& t : th) t.join();
        bool any = false;
        double zstar = 0.0;
        for (auto& s : sh) {
            if (s.rc != LP_OPTIMAL && s.rc != LP_INFEASIBLE) {
                for (auto& q : sh) lp_enum_free(q.p);
                lpgpu::throw_for_status(s.rc, s.ctx);
            }
            r.feasible += s.counts[0];
            r.infeasible += s.counts[1];
            r.singular += s.counts[2];
            if (s.rc == LP_OPTIMAL && (!any || (maximize ? s.z > zstar : s.z < zstar))) {
                zstar = s.z;
                any = true;
            }
        }
        r.status = any ? LP_OPTIMAL : LP_INFEASIBLE;
        if (any) {
            th.clear();
            auto pass2 = [&](int g) {
                Shard& s = sh[(size_t)g];
                if (s.rc == LP_OPTIMAL)
                    (void)lp_enum_first_within(s.p, s.lo, s.hi, zstar, 1e-9, &s.first);
            };
            for (int g = 0; g < n_gpus; ++g) th.emplace_back(pass2, g);
            for (auto& t : th) t.join();
            uint64_t best = UINT64_MAX;
            for (auto& s : sh) best = std::min(best, s.first);
            r.rank = best;
            int verdict = 0;
            (void)lp_enum_vertex(sh[0].p, best, n_orig, r.x.data(), r.basis.data(), &r.objective, &verdict);
        }
        for (auto& s : sh) lp_enum_free(s.p);
        if (throw_on_failure) lpgpu::throw_for_status(r.status, sh[0].ctx);
        return r;
    }

private:
    Canonical _problem;
    int _device;
};
