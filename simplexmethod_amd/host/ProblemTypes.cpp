// ProblemTypes.cpp — Canonical and Symmetrical (host-side problem containers).
#include <cmath>
#include <iostream>
#include <stdexcept>

#include "Canonical.h"
#include "Common.h"
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
    MatrixXd Ac = MatrixXd::Zero(m, n + extra);   // Eigen's sized ctor leaves memory uninitialised (Symmetrical.cpp:166-173 uses Zero/Identity blocks)
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

// ---------------------------------------------------------------------------
// Conversions of SURVEY.md §8(f) N4 (CPU modelling glue, O(mn))
// ---------------------------------------------------------------------------

std::unique_ptr<Common> Symmetrical::ToCommon() const {
    const auto row = maximize_ ? Common::ConstraintType::LessOrEqual : Common::ConstraintType::GreaterOrEqual;
    return std::make_unique<Common>(A_, b_, c_, std::vector<Common::ConstraintType>((size_t)A_.rows(), row),
                                    std::vector<Common::VariableType>((size_t)A_.cols(), Common::VariableType::NonNegative),
                                    maximize_);
}

namespace {
// the first n_keep columns of A and entries of c
void original_part(const MatrixXd& A, const VectorXd& c, long n_keep, MatrixXd& Ao, VectorXd& co) {
    Ao = MatrixXd(A.rows(), n_keep);
    co = VectorXd(n_keep);
    for (long j = 0; j < n_keep; ++j) {
        co[j] = c[j];
        for (long i = 0; i < A.rows(); ++i) Ao(i, j) = A(i, j);
    }
}
}  // namespace

std::unique_ptr<Common> Canonical::ToCommon() const {
    MatrixXd Ao;
    VectorXd co;
    original_part(A_, c_, originalVariablesCount_, Ao, co);
    return std::make_unique<Common>(Ao, b_, co,
                                    std::vector<Common::ConstraintType>((size_t)A_.rows(), Common::ConstraintType::Equal),
                                    std::vector<Common::VariableType>((size_t)originalVariablesCount_, Common::VariableType::NonNegative),
                                    !minimize_);
}

std::unique_ptr<Symmetrical> Canonical::ToSymmetrical() const {
    MatrixXd Ao;
    VectorXd co;
    original_part(A_, c_, originalVariablesCount_, Ao, co);
    const long m = A_.rows(), n = originalVariablesCount_;
    // a.x = b  ->  a.x (<=|>=) b  and  -a.x (<=|>=) -b : the same pair of rows for either sense
    MatrixXd As(2 * m, n);
    VectorXd bs(2 * m);
    for (long i = 0; i < m; ++i) {
        for (long j = 0; j < n; ++j) {
            As(2 * i, j) = Ao(i, j);
            As(2 * i + 1, j) = -Ao(i, j);
        }
        bs[2 * i] = b_[i];
        bs[2 * i + 1] = -b_[i];
    }
    return std::make_unique<Symmetrical>(As, bs, co, !minimize_);
}

std::unique_ptr<Canonical> Canonical::GetDual() const {
    const long m = A_.rows(), n = A_.cols();
    MatrixXd Ad = MatrixXd::Zero(n, 2 * m + n);
    VectorXd cd = VectorXd::Zero(2 * m + n);
    for (long i = 0; i < m; ++i) {
        cd[i] = b_[i];          // y'
        cd[m + i] = -b_[i];     // y''   (free y = y' - y'')
        for (long j = 0; j < n; ++j) {
            Ad(j, i) = A_(i, j);
            Ad(j, m + i) = -A_(i, j);
        }
    }
    std::vector<int> basis((size_t)n);
    for (long j = 0; j < n; ++j) {
        Ad(j, 2 * m + j) = 1.0;  // slack of A^T y <= c
        basis[(size_t)j] = (int)(2 * m + j);
    }
    auto dual = std::make_unique<Canonical>(Ad, c_, cd, basis, !minimize_);
    dual->SetOriginalVariablesCount((int)(2 * m));
    return dual;
}

// ---- Common ---------------------------------------------------------------

Common::Common(const MatrixXd& A, const VectorXd& b, const VectorXd& c,
               const std::vector<ConstraintType>& constraintTypes,
               const std::vector<VariableType>& variableTypes, bool maximize)
    : A_(A), b_(b), c_(c), ctypes_(constraintTypes), vtypes_(variableTypes), maximize_(maximize) {
    if (A_.rows() != b_.size()) throw std::invalid_argument("Common: rows(A) != size(b)");
    if (A_.cols() != c_.size()) throw std::invalid_argument("Common: cols(A) != size(c)");
    if ((long)ctypes_.size() != A_.rows()) throw std::invalid_argument("Common: one constraint type per row of A expected");
    if ((long)vtypes_.size() != A_.cols()) throw std::invalid_argument("Common: one variable type per column of A expected");
}

double Common::Evaluate(const VectorXd& solution) const {
    if (solution.size() != c_.size()) throw std::invalid_argument("Common::Evaluate: solution size != variable count");
    return c_.dot(solution);
}

void Common::Print() const {
    std::cout << "general form: " << (maximize_ ? "maximize" : "minimize");
    for (long j = 0; j < c_.size(); ++j) std::cout << (j ? " + " : " ") << c_[j] << "*x" << (j + 1);
    std::cout << "\n";
    for (long i = 0; i < A_.rows(); ++i) {
        for (long j = 0; j < A_.cols(); ++j) std::cout << (j ? " + " : "  ") << A_(i, j) << "*x" << (j + 1);
        const char* rel = ctypes_[(size_t)i] == ConstraintType::LessOrEqual ? " <= "
                          : ctypes_[(size_t)i] == ConstraintType::GreaterOrEqual ? " >= " : " = ";
        std::cout << rel << b_[i] << "\n";
    }
    for (size_t j = 0; j < vtypes_.size(); ++j)
        std::cout << "  x" << (j + 1)
                  << (vtypes_[j] == VariableType::Free ? " free" : vtypes_[j] == VariableType::NonNegative ? " >= 0" : " <= 0")
                  << "\n";
}

std::unique_ptr<Symmetrical> Common::ToSymmetrical() const {
    // Every output column is (source column, sign) and every output row is (source row, sign):
    // A_sym(r, k) = sign_r * sign_k * A(i, j).  Signs: x_j <= 0 enters as -x'; free x_j as the pair
    // (+x', -x''); a >= row is negated; an = row becomes the pair (+, -).  The result is always the
    // max / <= form, so a min objective is negated (Common.cpp:169-388).
    struct Part { long src; double sign; };
    std::vector<Part> cols, rows;
    for (long j = 0; j < A_.cols(); ++j) {
        switch (vtypes_[(size_t)j]) {
            case VariableType::NonNegative: cols.push_back({j, 1.0}); break;
            case VariableType::NonPositive: cols.push_back({j, -1.0}); break;
            case VariableType::Free: cols.push_back({j, 1.0}); cols.push_back({j, -1.0}); break;
        }
    }
    for (long i = 0; i < A_.rows(); ++i) {
        switch (ctypes_[(size_t)i]) {
            case ConstraintType::LessOrEqual: rows.push_back({i, 1.0}); break;
            case ConstraintType::GreaterOrEqual: rows.push_back({i, -1.0}); break;
            case ConstraintType::Equal: rows.push_back({i, 1.0}); rows.push_back({i, -1.0}); break;
        }
    }
    MatrixXd As((long)rows.size(), (long)cols.size());
    VectorXd bs((long)rows.size()), cs((long)cols.size());
    const double csign = maximize_ ? 1.0 : -1.0;
    for (size_t k = 0; k < cols.size(); ++k) {
        cs[(long)k] = csign * (cols[k].sign * c_[cols[k].src]);
        for (size_t r = 0; r < rows.size(); ++r)
            As((long)r, (long)k) = rows[r].sign * (cols[k].sign * A_(rows[r].src, cols[k].src));
    }
    for (size_t r = 0; r < rows.size(); ++r) bs[(long)r] = rows[r].sign * b_[rows[r].src];
    return std::make_unique<Symmetrical>(As, bs, cs, true);
}

std::unique_ptr<Canonical> Common::ToCanonical() const { return ToSymmetrical()->ToCanonical(); }

std::unique_ptr<Common> Common::GetDual() const {
    // max problem: row <= -> y >= 0, row >= -> y <= 0, row = -> y free; x >= 0 -> dual row >=,
    // x <= 0 -> dual row <=, x free -> dual row =.  A min problem mirrors every inequality.
    std::vector<VariableType> vt((size_t)A_.rows());
    std::vector<ConstraintType> ct((size_t)A_.cols());
    for (size_t i = 0; i < vt.size(); ++i) {
        const ConstraintType t = ctypes_[i];
        vt[i] = t == ConstraintType::Equal ? VariableType::Free
                : ((t == ConstraintType::LessOrEqual) == maximize_) ? VariableType::NonNegative
                                                                    : VariableType::NonPositive;
    }
    for (size_t j = 0; j < ct.size(); ++j) {
        const VariableType t = vtypes_[j];
        ct[j] = t == VariableType::Free ? ConstraintType::Equal
                : ((t == VariableType::NonNegative) == maximize_) ? ConstraintType::GreaterOrEqual
                                                                  : ConstraintType::LessOrEqual;
    }
    return std::make_unique<Common>(A_.transpose(), c_, b_, ct, vt, !maximize_);
}
