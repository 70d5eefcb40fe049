// Symmetrical.h — symmetric form  max c.x, Ax <= b  (or min, Ax >= b),  x >= 0.
// Same public surface as /root/reference/src/ProblemTypes/Symmetrical.h:16-45.
#pragma once

#include <memory>

#include "IProblem.h"

class Canonical;
class Common;

class Symmetrical : public IProblem {
public:
    // Throws std::invalid_argument on size mismatch (reference: Symmetrical.cpp:21-28).
    Symmetrical(const lpla::MatrixXd& A, const lpla::VectorXd& b, const lpla::VectorXd& c,
                bool maximize);

    double Evaluate(const lpla::VectorXd& solution) const override;
    void Print() const override;
    const lpla::MatrixXd& GetConstraintsMatrix() const override { return A_; }
    const lpla::VectorXd& GetRightHandSide() const override { return b_; }
    const lpla::VectorXd& GetObjectiveCoefficients() const override { return c_; }
    bool IsMaximization() const override { return maximize_; }

    std::unique_ptr<Symmetrical> GetDual() const;      // Symmetrical.cpp:119-140
    // max: [A | I], slack basis, zero slack costs, Canonical(minimize=false);
    // min: [A | -I | I], artificial basis with zero cost (Symmetrical.cpp:142-223).
    std::unique_ptr<Canonical> ToCanonical() const;
    // same data, every row <= (max) or >= (min), every variable >= 0 (Symmetrical.cpp:225-273)
    std::unique_ptr<Common> ToCommon() const;

private:
    lpla::MatrixXd A_;
    lpla::VectorXd b_, c_;
    bool maximize_;
};
