// CPU-only tests of the host problem classes; the cases and expected values are the
// reference's own (tests/test_canonical.cpp, test_symmetrical.cpp, test_parser.cpp,
// test_transformations.cpp:39-61, test_common.cpp:49-59) restated without gtest/Eigen.
#include "check.h"
#include "Canonical.h"
#include "Symmetrical.h"
#include "SymmetricalParser.h"

using lpla::MatrixXd;
using lpla::VectorXd;

static MatrixXd mat(long r, long c, std::initializer_list<double> il) {
    MatrixXd m(r, c);
    long k = 0;
    for (double x : il) { m(k / c, k % c) = x; ++k; }
    return m;
}
static VectorXd vec(std::initializer_list<double> il) {
    VectorXd v((long)il.size());
    long k = 0;
    for (double x : il) v[k++] = x;
    return v;
}

// ---- test_canonical.cpp
static MatrixXd CA() { return mat(2, 4, {1, 2, 1, 0, 3, 4, 0, 1}); }
TEST(Canonical_Creation) {                       // :31-39
    Canonical c(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {2, 3}, true);
    CHECK(!c.IsMaximization());
    CHECK(c.GetConstraintsMatrix().rows() == 2 && c.GetConstraintsMatrix().cols() == 4);
    CHECK(c.GetBasisIndices().size() == 2);
    CHECK(c.GetOriginalVariablesCount() == 4);
}
TEST(Canonical_GetBasicSolution) {               // :41-58, EXPECT_DOUBLE_EQ
    Canonical c(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {2, 3}, true);
    c.SetOriginalVariablesCount(2);
    VectorXd x = c.GetBasicSolution();
    CHECK(x.size() == 4);
    CHECK(x[0] == 0.0 && x[1] == 0.0 && x[2] == 5.0 && x[3] == 6.0);
}
TEST(Canonical_IsFeasibleBasis) {                // :60-66
    Canonical c(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {2, 3}, true);
    CHECK(c.IsFeasibleBasis());
    Canonical neg(CA(), vec({5, -6}), vec({7, 8, 0, 0}), {2, 3}, true);
    CHECK(!neg.IsFeasibleBasis());
}
TEST(Canonical_InvalidArguments) {               // :68-76 + Canonical.cpp:27-38,:156-163
    CHECK_THROWS(Canonical(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {10, 20}, true), std::invalid_argument);
    CHECK_THROWS(Canonical(CA(), vec({5, 6, 7}), vec({7, 8, 0, 0}), {2, 3}, true), std::invalid_argument);
    CHECK_THROWS(Canonical(CA(), vec({5, 6}), vec({7, 8, 0}), {2, 3}, true), std::invalid_argument);
    CHECK_THROWS(Canonical(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {2}, true), std::invalid_argument);
    CHECK_THROWS(Canonical(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {}, true), std::invalid_argument);
    Canonical c(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {2, 3}, true);
    CHECK_THROWS(c.SetOriginalVariablesCount(0), std::invalid_argument);
    CHECK_THROWS(c.SetOriginalVariablesCount(5), std::invalid_argument);
    CHECK_THROWS(c.Evaluate(vec({1, 2})), std::invalid_argument);
}
TEST(Canonical_GeneralBasisSolve) {
    // non-identity basis: columns {0,1}: [1 2;3 4] x = (5,6) -> x = (-4, 4.5)
    Canonical c(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {0, 1}, true);
    VectorXd x = c.GetBasicSolution();
    CHECK(std::fabs(x[0] + 4.0) < 1e-12 && std::fabs(x[1] - 4.5) < 1e-12 && x[2] == 0 && x[3] == 0);
    CHECK(!c.IsFeasibleBasis());
    CHECK(std::fabs(c.Evaluate(x) - (7 * -4.0 + 8 * 4.5)) < 1e-12);
}

// ---- test_symmetrical.cpp
static MatrixXd SA() { return mat(2, 2, {1, 2, 3, 4}); }
TEST(Symmetrical_Creation) {                     // :26-33
    Symmetrical s(SA(), vec({5, 6}), vec({7, 8}), true);
    CHECK(s.IsMaximization() && s.GetConstraintsMatrix().rows() == 2 && s.GetConstraintsMatrix().cols() == 2);
    CHECK_THROWS(Symmetrical(SA(), vec({5, 6, 7}), vec({7, 8}), true), std::invalid_argument);
    CHECK_THROWS(Symmetrical(SA(), vec({5, 6}), vec({7}), true), std::invalid_argument);
}
TEST(Symmetrical_Evaluate) {                     // test_common.cpp:49-59: 7*1 + 8*2 = 23
    Symmetrical s(SA(), vec({5, 6}), vec({7, 8}), true);
    CHECK(s.Evaluate(vec({1, 2})) == 23.0);
}
TEST(Symmetrical_GetDual) {                      // :36-53
    Symmetrical s(SA(), vec({5, 6}), vec({7, 8}), true);
    auto d = s.GetDual();
    CHECK(d && !d->IsMaximization());
    CHECK(d->GetConstraintsMatrix().rows() == 2 && d->GetConstraintsMatrix().cols() == 2);
    CHECK(d->GetConstraintsMatrix()(0, 1) == 3.0 && d->GetConstraintsMatrix()(1, 0) == 2.0);
    CHECK(d->GetRightHandSide()[0] == 7 && d->GetRightHandSide()[1] == 8);
    CHECK(d->GetObjectiveCoefficients()[0] == 5 && d->GetObjectiveCoefficients()[1] == 6);
}
TEST(Symmetrical_DualOfDual) {                   // test_transformations.cpp:39-61
    Symmetrical s(SA(), vec({5, 6}), vec({7, 8}), true);
    auto dd = s.GetDual()->GetDual();
    CHECK(dd->IsMaximization());
    for (int i = 0; i < 2; ++i) {
        CHECK(dd->GetRightHandSide()[i] == s.GetRightHandSide()[i]);
        CHECK(dd->GetObjectiveCoefficients()[i] == s.GetObjectiveCoefficients()[i]);
        for (int j = 0; j < 2; ++j) CHECK(dd->GetConstraintsMatrix()(i, j) == SA()(i, j));
    }
}
TEST(Symmetrical_ToCanonicalMax) {               // :55-72
    Symmetrical s(SA(), vec({5, 6}), vec({7, 8}), true);
    auto c = s.ToCanonical();
    CHECK(c && c->GetConstraintsMatrix().cols() == 4 && c->GetConstraintsMatrix().rows() == 2);
    CHECK(c->GetBasisIndices().size() == 2 && c->GetBasisIndices()[0] == 2 && c->GetBasisIndices()[1] == 3);
    CHECK(c->IsMaximization() && c->GetOriginalVariablesCount() == 2);
    CHECK(c->GetConstraintsMatrix()(0, 2) == 1 && c->GetConstraintsMatrix()(1, 3) == 1 &&
          c->GetConstraintsMatrix()(0, 3) == 0 && c->GetConstraintsMatrix()(1, 2) == 0);
    CHECK(c->GetObjectiveCoefficients()[2] == 0 && c->GetObjectiveCoefficients()[3] == 0);
    CHECK(c->IsFeasibleBasis());
}
TEST(Symmetrical_ToCanonicalMin) {               // :74-85
    Symmetrical s(SA(), vec({5, 6}), vec({7, 8}), false);
    auto c = s.ToCanonical();
    CHECK(c && c->GetConstraintsMatrix().cols() == 6 && c->GetConstraintsMatrix().rows() == 2);
    CHECK(!c->IsMaximization() && c->GetBasisIndices()[0] == 4 && c->GetBasisIndices()[1] == 5);
    CHECK(c->GetConstraintsMatrix()(0, 2) == -1 && c->GetConstraintsMatrix()(1, 5) == 1);
}

// ---- test_parser.cpp
TEST(Parser_Maximization) {                      // :4-26
    SymmetricalParser p;
    auto s = p.ParseFromString("\n  maximize\n\n  objective:\n  3 5\n\n  constraints:\n  1 2 10\n  3 4 20\n");
    CHECK(s && s->IsMaximization() && s->GetConstraintsMatrix().rows() == 2 && s->GetConstraintsMatrix().cols() == 2);
    CHECK(s->GetRightHandSide()[1] == 20 && s->GetConstraintsMatrix()(1, 0) == 3);
}
TEST(Parser_Minimization) {                      // :28-46
    SymmetricalParser p;
    auto s = p.ParseFromString("minimize\nobjective:\n7 8\nconstraints:\n1 1 5\n2 3 12\n");
    CHECK(s && !s->IsMaximization());
}
TEST(Parser_Comments) {                          // :48-69
    SymmetricalParser p;
    auto s = p.ParseFromString("# c\nmaximize\n# o\nobjective:\n1 2 3  # more\nconstraints:\n1 0 0 5 # a\n0 1 0 6\n0 0 1 7\n");
    CHECK(s && s->GetObjectiveCoefficients().size() == 3 && s->GetConstraintsMatrix().rows() == 3);
}
TEST(Parser_Invalid) {                           // :71-81 + SymmetricalParser.cpp:115-158
    SymmetricalParser p;
    CHECK(!p.ParseFromString("maximize\n# nothing else\n") && !p.GetLastError().empty());
    CHECK(!p.ParseFromString("1 2 3\n") && !p.GetLastError().empty());
    CHECK(!p.ParseFromString("max\nobjective\n1 2\nconstraints\n1 2 3 4\n"));
    CHECK(!p.ParseFromString("max\nobjective\n1 2\nsubject to\n5\n"));
    CHECK(!p.ParseFromFile("/nonexistent/file.txt") && !p.GetLastError().empty());
}
TEST(Parser_InputSymmetricFixture) {
    // the contents of /root/reference/input_symmetric.txt:1-8, CRLF line endings included
    SymmetricalParser p;
    auto s = p.ParseFromString("maximize\r\n\r\nobjective:\r\n7 8 3\r\n\r\nconstraints:\r\n1 2 3 10\r\n4 5 6 20\r\n");
    CHECK(s && s->IsMaximization());
    CHECK(s->GetConstraintsMatrix().rows() == 2 && s->GetConstraintsMatrix().cols() == 3);
    CHECK(s->GetObjectiveCoefficients()[0] == 7 && s->GetRightHandSide()[1] == 20 && s->GetConstraintsMatrix()(1, 2) == 6);
    auto c = s->ToCanonical();
    CHECK(c->GetConstraintsMatrix().cols() == 5 && c->GetBasisIndices()[0] == 3 && c->GetBasisIndices()[1] == 4);
}

int main(int argc, char** argv) { return run_all(argc > 1 ? argv[1] : nullptr); }
