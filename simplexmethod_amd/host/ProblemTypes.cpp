// ProblemTypes.cpp — Canonical and Symmetrical (host-side problem containers).
#include <cmath>
#include <iostream>
#include <stdexcept>

#include "Canonical.h"
#include "Symmetrical.h"

using lpla::MatrixXd;
using lpla::VectorXd;

// ---------------------------------------------------------------------------- Canonical

Canonical::Canonical(const MatrixXd& A, const VectorXd& b, const VectorXd& c,
                     const std::vector<int>& basisIndices, bool minimize)
    : A_(A), b_(b), c_(c), basis_(basisIndices), minimize_(minimize),
      originalVariablesCount_((int)c.size()) {
    if (A_.rows() != b_.size()) throw std::invalid_argument("Canonical: rows(A) != size(b)");
    if (A_.cols() != c_.size()) throw std::invalid_argument("Canonical: cols(A) != size(c)");
    if ((long)basis_.size() != A_.rows())
        throw std::invalid_argument("Canonical: basis size != rows(A)");
    for (int idx : basis_)
        if (idx < 0 || idx >= A_.cols())
            throw std::invalid_argument("Canonical: basis index out of range");
}

double Canonical::Evaluate(const VectorXd& solution) const {
    if (solution.size() != c_.size())
        throw std::invalid_argument("Evaluate: solution size != number of variables");
    return c_.dot(solution);
}

void Canonical::SetOriginalVariablesCount(int count) {
    if (count <= 0 || count > c_.size())
        throw std::invalid_argument("Canonical: bad original variable count");
    originalVariablesCount_ = count;
}

// B x_B = b by Householder QR with column pivoting (the reference calls Eigen's
// colPivHouseholderQr().solve, Canonical.cpp:189); the result is scattered into a length-n
// vector.  A singular B is not reported here either (a truncated solution comes back), which
// is why EnumerationSolver tests singularity itself.
VectorXd Canonical::GetBasicSolution() const {
    const long m = A_.rows(), n = A_.cols();
    VectorXd x = VectorXd::Zero(n);
    MatrixXd Q(m, m);
    for (long t = 0; t < m; ++t)
        for (long i = 0; i < m; ++i) Q(i, t) = A_(i, basis_[(size_t)t]);
    std::vector<double> rhs((size_t)m), y((size_t)m, 0.0);
    std::vector<long> perm((size_t)m);
    for (long i = 0; i < m; ++i) {
        rhs[(size_t)i] = b_[i];
        perm[(size_t)i] = i;
    }
    double maxpivot = 0.0;
    for (long k = 0; k < m; ++k) {
        long best = k;
        double bn = -1.0;
        for (long j = k; j < m; ++j) {
            double s = 0.0;
            for (long i = k; i < m; ++i) s += Q(i, j) * Q(i, j);
            if (s > bn) { bn = s; best = j; }
        }
        if (best != k) {
            for (long i = 0; i < m; ++i) std::swap(Q(i, k), Q(i, best));
            std::swap(perm[(size_t)k], perm[(size_t)best]);
        }
        double tail = 0.0;
        for (long i = k + 1; i < m; ++i) tail += Q(i, k) * Q(i, k);
        const double c0 = Q(k, k);
        double beta = c0, tau = 0.0;
        if (tail != 0.0) {
            beta = std::sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            for (long i = k + 1; i < m; ++i) Q(i, k) /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        Q(k, k) = beta;
        if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
        if (tau != 0.0) {
            for (long j = k + 1; j < m; ++j) {
                double w = Q(k, j);
                for (long i = k + 1; i < m; ++i) w += Q(i, k) * Q(i, j);
                w *= tau;
                Q(k, j) -= w;
                for (long i = k + 1; i < m; ++i) Q(i, j) -= w * Q(i, k);
            }
            double w = rhs[(size_t)k];
            for (long i = k + 1; i < m; ++i) w += Q(i, k) * rhs[(size_t)i];
            w *= tau;
            rhs[(size_t)k] -= w;
            for (long i = k + 1; i < m; ++i) rhs[(size_t)i] -= w * Q(i, k);
        }
    }
    const double thr = maxpivot * 2.220446049250313e-16 * (double)m;
    long rank = 0;
    for (long k = 0; k < m; ++k)
        if (std::fabs(Q(k, k)) > thr) ++rank;
    for (long k = rank - 1; k >= 0; --k) {
        double s = rhs[(size_t)k];
        for (long j = k + 1; j < rank; ++j) s -= Q(k, j) * y[(size_t)j];
        y[(size_t)k] = s / Q(k, k);
    }
    for (long k = 0; k < m; ++k) x[basis_[(size_t)perm[(size_t)k]]] = y[(size_t)k];
    return x;
}

bool Canonical::IsFeasibleBasis() const {
    const VectorXd x = GetBasicSolution();
    for (long i = 0; i < x.size(); ++i)
        if (x[i] < -1e-9) return false;
    return true;
}

void Canonical::Print() const {
    std::cout << "Canonical LP: " << (minimize_ ? "minimize" : "maximize") << " c.x, A x = b, x >= 0; "
              << A_.rows() << " rows, " << A_.cols() << " columns (" << originalVariablesCount_
              << " original)\n  basis:";
    for (int j : basis_) std::cout << " x" << (j + 1);
    std::cout << "\n";
    for (long i = 0; i < A_.rows(); ++i) {
        std::cout << " ";
        for (long j = 0; j < A_.cols(); ++j) std::cout << " " << A_(i, j);
        std::cout << " | " << b_[i] << "\n";
    }
    std::cout << "  c:";
    for (long j = 0; j < c_.size(); ++j) std::cout << " " << c_[j];
    std::cout << "\n";
}

// -------------------------------------------------------------------------- Symmetrical

Symmetrical::Symmetrical(const MatrixXd& A, const VectorXd& b, const VectorXd& c, bool maximize)
    : A_(A), b_(b), c_(c), maximize_(maximize) {
    if (A_.rows() != b_.size()) throw std::invalid_argument("Symmetrical: rows(A) != size(b)");
    if (A_.cols() != c_.size()) throw std::invalid_argument("Symmetrical: cols(A) != size(c)");
}

double Symmetrical::Evaluate(const VectorXd& solution) const {
    if (solution.size() != c_.size())
        throw std::invalid_argument("Evaluate: solution size != number of variables");
    return c_.dot(solution);
}

std::unique_ptr<Symmetrical> Symmetrical::GetDual() const {
    // max c.x, Ax <= b  <->  min b.y, A^T y >= c   (and the mirror image)
    return std::make_unique<Symmetrical>(A_.transpose(), c_, b_, !maximize_);
}

std::unique_ptr<Canonical> Symmetrical::ToCanonical() const {
    const long m = A_.rows(), n = A_.cols();
    const long extra = maximize_ ? m : 2 * m;
    MatrixXd Ac(m, n + extra);
    VectorXd cc = VectorXd::Zero(n + extra);
    for (long j = 0; j < n; ++j) {
        cc[j] = c_[j];
        for (long i = 0; i < m; ++i) Ac(i, j) = A_(i, j);
    }
    std::vector<int> basis((size_t)m);
    if (maximize_) {
        for (long i = 0; i < m; ++i) {
            Ac(i, n + i) = 1.0;             // slack
            basis[(size_t)i] = (int)(n + i);
        }
    } else {
        for (long i = 0; i < m; ++i) {
            Ac(i, n + i) = -1.0;            // surplus
            Ac(i, n + m + i) = 1.0;         // artificial (zero cost, as in the reference)
            basis[(size_t)i] = (int)(n + m + i);
        }
    }
    auto can = std::make_unique<Canonical>(Ac, b_, cc, basis, !maximize_);
    can->SetOriginalVariablesCount((int)n);
    return can;
}

void Symmetrical::Print() const {
    std::cout << "Symmetric LP: " << (maximize_ ? "maximize c.x, A x <= b" : "minimize c.x, A x >= b")
              << ", x >= 0; " << A_.rows() << " rows, " << A_.cols() << " variables\n";
    for (long i = 0; i < A_.rows(); ++i) {
        std::cout << " ";
        for (long j = 0; j < A_.cols(); ++j) std::cout << " " << A_(i, j);
        std::cout << (maximize_ ? " <= " : " >= ") << b_[i] << "\n";
    }
    std::cout << "  c:";
    for (long j = 0; j < c_.size(); ++j) std::cout << " " << c_[j];
    std::cout << "\n";
}
