// Canonical.h — canonical form  opt c.x, Ax = b, x >= 0  with a starting basis.
// Same public surface as /root/reference/src/ProblemTypes/Canonical.h:10-49 minus the
// conversions SURVEY.md §8 marks out of scope (ToCommon / ToSymmetrical / GetDual: N4).
#pragma once

#include <memory>
#include <vector>

#include "IProblem.h"

class Symmetrical;

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

private:
    lpla::MatrixXd A_;
    lpla::VectorXd b_, c_;
    std::vector<int> basis_;
    bool minimize_;
    int originalVariablesCount_;
};
