// IProblem.h — abstract LP problem; mirrors /root/reference/src/ProblemTypes/IProblem.h:7-16.
#pragma once

#include <memory>
#include <string>

#include "LinAlg.h"

class IProblem {
public:
    virtual double Evaluate(const lpla::VectorXd& solution) const = 0;
    virtual void Print() const = 0;
    virtual const lpla::MatrixXd& GetConstraintsMatrix() const = 0;
    virtual const lpla::VectorXd& GetRightHandSide() const = 0;
    virtual const lpla::VectorXd& GetObjectiveCoefficients() const = 0;
    virtual bool IsMaximization() const = 0;
    virtual ~IProblem() = default;
};
