// Canonical.h — canonical form  opt c.x, Ax = b, x >= 0  with a starting basis.
// Same public surface as /root/reference/src/ProblemTypes/Canonical.h:10-49.
#pragma once

#include <memory>
#include <vector>

#include "IProblem.h"

class Symmetrical;
class Common;

class Canonical : public IProblem {
public:
    // Throws std::invalid_argument on A/b, A/c size mismatch, basis size != rows(A) or a basis
    // index outside [0, cols(A))  (reference: Canonical.cpp:27-46).
    Canonical(const lpla::MatrixXd& A, const lpla::VectorXd& b, const lpla::VectorXd& c,
              const std::vector<int>& basisIndices, bool minimize = true);

    double Evaluate(const lpla::VectorXd& solution) const override;   // Canonical.cpp:79-87
    void Print() const override;
    const lpla::MatrixXd& GetConstraintsMatrix() const override { return A_; }
    const lpla::VectorXd& GetRightHandSide() const override { return b_; }
    const lpla::VectorXd& GetObjectiveCoefficients() const override { return c_; }
    bool IsMaximization() const override { return !minimize_; }

    const std::vector<int>& GetBasisIndices() const { return basis_; }
    int GetOriginalVariablesCount() const { return originalVariablesCount_; }
    void SetOriginalVariablesCount(int count);                         // Canonical.cpp:156-163
    bool IsFeasibleBasis() const;                                      // Canonical.cpp:165-177
    lpla::VectorXd GetBasicSolution() const;                           // Canonical.cpp:179-197

    // conversions over the ORIGINAL variables only (slack / surplus / artificial columns dropped)
    std::unique_ptr<Common> ToCommon() const;             // all rows '=', all x >= 0 (:199-229)
    std::unique_ptr<Symmetrical> ToSymmetrical() const;   // each row -> the pair (a, b), (-a, -b) (:231-300)
    // dual of  opt c.x, Ax = b, x >= 0  in canonical form: [A^T | -A^T | I], costs [b | -b | 0],
    // rhs c, slack basis, opposite sense (:303-364)
    std::unique_ptr<Canonical> GetDual() const;

private:
    lpla::MatrixXd A_;
    lpla::VectorXd b_, c_;
    std::vector<int> basis_;
    bool minimize_;
    int originalVariablesCount_;
};
