// SimplexSolover.h — class Solver, the drop-in for /root/reference/src/SimplexSolover.h:10-452
// (the header keeps the reference's file name, misspelling included, so `#include
// "SimplexSolover.h"` keeps working).  Same constructor, same `solve()` signature and return
// value (x.head(n_orig), :435-439), same exception types; the arithmetic runs on the MI355X
// through lp_simplex_solve (include/simplexmethod_amd.h).
//
// Only the live path of the reference is reproduced: solve() -> solveWithBasis (:288-296,
// :408-451).  The two-phase / artificial-basis code (:15-95, :211-265, :331-406) is
// unreachable through the reference's public API (Canonical's constructor rejects an empty
// basis, Canonical.cpp:35-38) and is out of scope.
#pragma once

#include <stdexcept>
#include <vector>

#include "Canonical.h"
#include "DeviceContext.h"

class Solver {
public:
    static constexpr double EPS = 1e-9;     // SimplexSolover.h:13
    static constexpr int MAX_ITER = 10000;  // SimplexSolover.h:426

    struct Result {            // what the reference computes but never returns (N is local, :419)
        lpla::VectorXd x;      // x.head(n_orig)
        std::vector<int> basis;  // final basis by position
        double objective = 0.0;  // Canonical::Evaluate of the full vertex
        int iterations = 0;
        int status = LP_OPTIMAL;
    };

    explicit Solver(const Canonical& problem, int device = 0) : _problem(problem), _device(device) {}

    // Throws std::runtime_error (unbounded / iteration limit / singular basis) like the
    // reference (:126, :443, :450).
    lpla::VectorXd solve() { return solve_ex().x; }

    Result solve_ex(bool throw_on_failure = true) {
        const lpla::MatrixXd& A = _problem.GetConstraintsMatrix();      // :409-414
        const lpla::VectorXd& b = _problem.GetRightHandSide();
        const lpla::VectorXd& c = _problem.GetObjectiveCoefficients();
        const std::vector<int>& basis = _problem.GetBasisIndices();
        const int n_orig = _problem.GetOriginalVariablesCount();
        const int m = (int)A.rows(), n = (int)A.cols();
        lp_context* ctx = lpgpu::context(_device);
        Result r;
        r.x = lpla::VectorXd::Zero(n_orig);
        r.basis.assign((size_t)m, -1);
        r.status = lp_simplex_solve(ctx, A.data(), m, n, b.data(), c.data(), basis.data(),
                                    _problem.IsMaximization() ? 1 : 0, n_orig, EPS, MAX_ITER,
                                    r.x.data(), r.basis.data(), &r.objective, &r.iterations);
        if (throw_on_failure) lpgpu::throw_for_status(r.status, ctx);
        return r;
    }

private:
    Canonical _problem;  // deep copy, as in the reference (:285)
    int _device;
};
