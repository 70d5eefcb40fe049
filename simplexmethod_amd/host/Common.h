// Common.h — general form: max/min c.x, rows of A x {<=, >=, =} b, variables {free, >= 0, <= 0}.
// Same public surface as /root/reference/src/ProblemTypes/Common.h:10-55;
// CPU-side modelling glue, O(mn) — no GPU work.  Print() uses this project's own wording.
#pragma once

#include <memory>
#include <vector>

#include "IProblem.h"

class Symmetrical;
class Canonical;

class Common : public IProblem {
public:
    enum class ConstraintType { LessOrEqual, GreaterOrEqual, Equal };
    enum class VariableType { Free, NonNegative, NonPositive };

    // Throws std::invalid_argument on A/b, A/c, constraint-type or variable-type count mismatch
    // (reference: Common.cpp:28-43).
    Common(const lpla::MatrixXd& A, const lpla::VectorXd& b, const lpla::VectorXd& c,
           const std::vector<ConstraintType>& constraintTypes,
           const std::vector<VariableType>& variableTypes, bool maximize);

    double Evaluate(const lpla::VectorXd& solution) const override;   // Common.cpp:79-85
    void Print() const override;
    const lpla::MatrixXd& GetConstraintsMatrix() const override { return A_; }
    const lpla::VectorXd& GetRightHandSide() const override { return b_; }
    const lpla::VectorXd& GetObjectiveCoefficients() const override { return c_; }
    bool IsMaximization() const override { return maximize_; }
    const std::vector<ConstraintType>& GetConstraintTypes() const { return ctypes_; }
    const std::vector<VariableType>& GetVariableTypes() const { return vtypes_; }

    // Always the max / <= form (Common.cpp:169-388): free x_j -> x' - x'', x_j <= 0 -> -x',
    // >= rows negated, = rows split into a <= pair, a min objective negated.
    std::unique_ptr<Symmetrical> ToSymmetrical() const;
    std::unique_ptr<Canonical> ToCanonical() const;    // via ToSymmetrical (Common.cpp:391-399)
    std::unique_ptr<Common> GetDual() const;            // Common.cpp:403-448

private:
    lpla::MatrixXd A_;
    lpla::VectorXd b_, c_;
    std::vector<ConstraintType> ctypes_;
    std::vector<VariableType> vtypes_;
    bool maximize_;
};
