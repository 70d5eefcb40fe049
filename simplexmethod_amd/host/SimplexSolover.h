// SimplexSolover.h — class Solver, the drop-in for /root/reference/src/SimplexSolover.h:10-452
// (the header keeps the reference's file name, misspelling included, so `#include
// "SimplexSolover.h"` keeps working).  Same constructor, same `solve()` signature and return
// value (x.head(n_orig), :435-439), same exception types; the arithmetic runs on the MI355X
// through lp_simplex_solve (include/simplexmethod_amd.h).
//
// solve() reproduces the live path of the reference: solve() -> solveWithBasis (:288-296,
// :408-451).  The reference's two-phase / artificial-basis code (:15-95, :211-265, :331-406) is
// unreachable through its public API (Canonical's constructor rejects an empty basis,
// Canonical.cpp:35-38) and internally inconsistent; twoPhaseSimplex() below is
// that flow designed afresh on the same GPU pivot kernels: it ignores the
// problem's basis indices and needs no feasible starting basis.
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

    // Two-phase simplex for problems without a usable starting basis (Symmetrical min problems,
    // negative b): the reference's private twoPhaseSimplex (:383-406) with make_b_nonneg (:61-68),
    // createAuxiliaryProblem (:70-95) and replaceArtificialColumns (:331-381), every pivot on the
    // GPU (lp_simplex_two_phase).  The problem's basis indices are ignored.  Throws
    // std::runtime_error like the reference's sketch: no feasible solution (:352-353), linearly
    // dependent constraints (:372-380), unbounded, iteration limit.
    lpla::VectorXd twoPhaseSimplex() { return twoPhaseSimplex_ex().x; }

    Result twoPhaseSimplex_ex(bool throw_on_failure = true, int phase_iterations[3] = nullptr) {
        const lpla::MatrixXd& A = _problem.GetConstraintsMatrix();
        const lpla::VectorXd& b = _problem.GetRightHandSide();
        const lpla::VectorXd& c = _problem.GetObjectiveCoefficients();
        const int n_orig = _problem.GetOriginalVariablesCount();
        const int m = (int)A.rows(), n = (int)A.cols();
        lp_context* ctx = lpgpu::context(_device);
        Result r;
        r.x = lpla::VectorXd::Zero(n_orig);
        r.basis.assign((size_t)m, -1);
        int it[3] = {0, 0, 0};
        r.status = lp_simplex_two_phase(ctx, A.data(), m, n, b.data(), c.data(),
                                        _problem.IsMaximization() ? 1 : 0, n_orig, EPS, MAX_ITER,
                                        r.x.data(), r.basis.data(), &r.objective, it);
        r.iterations = it[0] + it[1] + it[2];
        if (phase_iterations)
            for (int k = 0; k < 3; ++k) phase_iterations[k] = it[k];
        if (throw_on_failure) lpgpu::throw_for_status(r.status, ctx);
        return r;
    }

private:
    Canonical _problem;  // deep copy, as in the reference (:285)
    int _device;
};
