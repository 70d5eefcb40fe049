// CPU-only tests of the host problem classes; the cases and expected values are the
// reference's own (tests/test_canonical.cpp, test_symmetrical.cpp, test_parser.cpp,
// test_transformations.cpp, test_common.cpp) restated without gtest/Eigen.
#include "check.h"
#include "Canonical.h"
#include "Common.h"
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

// ---- test_common.cpp (fixture :10-29: A = [1 2; 3 4], b = (5,6), c = (7,8), rows (<=, >=), x >= 0)
using CT = Common::ConstraintType;
using VT = Common::VariableType;
static Common fixture_common(bool maximize = true) {
    return Common(SA(), vec({5, 6}), vec({7, 8}), {CT::LessOrEqual, CT::GreaterOrEqual},
                  {VT::NonNegative, VT::NonNegative}, maximize);
}
TEST(Common_Creation) {                          // :38-47
    Common c = fixture_common();
    CHECK(c.IsMaximization());
    CHECK(c.GetConstraintsMatrix().rows() == 2 && c.GetConstraintsMatrix().cols() == 2);
    CHECK(c.GetRightHandSide().size() == 2 && c.GetObjectiveCoefficients().size() == 2);
}
TEST(Common_Evaluate) {                          // :49-59: 7*1 + 8*2 = 23
    CHECK(fixture_common().Evaluate(vec({1, 2})) == 23.0);
    CHECK_THROWS(fixture_common().Evaluate(vec({1, 2, 3})), std::invalid_argument);
}
TEST(Common_InvalidDimensions) {                 // :61-70 + Common.cpp:28-43
    CHECK_THROWS(Common(SA(), vec({1, 2, 3}), vec({7, 8}), {CT::LessOrEqual, CT::GreaterOrEqual},
                        {VT::NonNegative, VT::NonNegative}, true), std::invalid_argument);
    CHECK_THROWS(Common(SA(), vec({5, 6}), vec({7, 8, 9}), {CT::LessOrEqual, CT::GreaterOrEqual},
                        {VT::NonNegative, VT::NonNegative}, true), std::invalid_argument);
    CHECK_THROWS(Common(SA(), vec({5, 6}), vec({7, 8}), {CT::LessOrEqual}, {VT::NonNegative, VT::NonNegative}, true),
                 std::invalid_argument);
    CHECK_THROWS(Common(SA(), vec({5, 6}), vec({7, 8}), {CT::LessOrEqual, CT::Equal}, {VT::Free}, true),
                 std::invalid_argument);
}
TEST(Common_Copy) {                              // :72-79
    Common a = fixture_common();
    Common b2(a);
    CHECK(b2.IsMaximization() && b2.GetConstraintsMatrix().rows() == 2);
    Common c3 = fixture_common(false);
    c3 = a;
    CHECK(c3.IsMaximization());
}
TEST(Common_GetDual) {                           // :81-94 + the rules of Common.cpp:403-448
    auto d = fixture_common().GetDual();
    CHECK(d && !d->IsMaximization());
    CHECK(d->GetConstraintsMatrix().rows() == 2 && d->GetConstraintsMatrix().cols() == 2);
    CHECK(d->GetConstraintsMatrix()(0, 1) == 3 && d->GetConstraintsMatrix()(1, 0) == 2);   // A^T
    CHECK(d->GetRightHandSide()[0] == 7 && d->GetObjectiveCoefficients()[1] == 6);         // b <-> c
    // max: row <= -> y >= 0, row >= -> y <= 0; x >= 0 -> dual row >=
    CHECK(d->GetVariableTypes()[0] == VT::NonNegative && d->GetVariableTypes()[1] == VT::NonPositive);
    CHECK(d->GetConstraintTypes()[0] == CT::GreaterOrEqual && d->GetConstraintTypes()[1] == CT::GreaterOrEqual);
    // min mirrors every inequality; '=' <-> free
    Common m(SA(), vec({5, 6}), vec({7, 8}), {CT::Equal, CT::GreaterOrEqual}, {VT::Free, VT::NonPositive}, false);
    auto dm = m.GetDual();
    CHECK(dm->IsMaximization());
    CHECK(dm->GetVariableTypes()[0] == VT::Free && dm->GetVariableTypes()[1] == VT::NonNegative);
    CHECK(dm->GetConstraintTypes()[0] == CT::Equal && dm->GetConstraintTypes()[1] == CT::GreaterOrEqual);
    // the dual of the dual is the problem itself
    auto dd = dm->GetDual();
    CHECK(!dd->IsMaximization() && dd->GetConstraintTypes()[0] == CT::Equal && dd->GetConstraintTypes()[1] == CT::GreaterOrEqual);
    CHECK(dd->GetVariableTypes()[0] == VT::Free && dd->GetVariableTypes()[1] == VT::NonPositive);
    CHECK(dd->GetConstraintsMatrix() == SA());
}
// ---- test_transformations.cpp:5-37 and the conversion rules of Common.cpp:169-388
TEST(Common_ToSymmetricalToCanonical) {
    Common c(SA(), vec({5, 6}), vec({7, 8}), {CT::LessOrEqual, CT::LessOrEqual}, {VT::NonNegative, VT::NonNegative}, true);
    auto s = c.ToSymmetrical();
    CHECK(s && s->IsMaximization() && s->GetConstraintsMatrix() == SA());
    auto can = s->ToCanonical();
    CHECK(can && can->GetConstraintsMatrix().rows() == 2);
    CHECK(c.ToCanonical()->GetConstraintsMatrix() == can->GetConstraintsMatrix());
}
TEST(Common_ToSymmetricalRules) {
    // min 7x1 + 8x2 + 9x3;  x1 + 2x2 + 3x3 <= 5,  4x1 + 5x2 + 6x3 >= 6,  x1 - x2 = 1;
    // x1 >= 0, x2 <= 0, x3 free
    Common c(mat(3, 3, {1, 2, 3, 4, 5, 6, 1, -1, 0}), vec({5, 6, 1}), vec({7, 8, 9}),
             {CT::LessOrEqual, CT::GreaterOrEqual, CT::Equal}, {VT::NonNegative, VT::NonPositive, VT::Free}, false);
    auto s = c.ToSymmetrical();
    CHECK(s->IsMaximization());   // always the max / <= form
    // columns: x1, -x2', x3', -x3'';  rows: row0, -row1, row2, -row2
    const MatrixXd expect = mat(4, 4, {1, -2, 3, -3,
                                       -4, 5, -6, 6,
                                       1, 1, 0, -0.0,
                                       -1, -1, -0.0, 0});
    CHECK(s->GetConstraintsMatrix() == expect);
    CHECK(s->GetRightHandSide() == vec({5, -6, 1, -1}));
    CHECK(s->GetObjectiveCoefficients() == vec({-7, 8, -9, 9}));   // min -> max: negated
}
TEST(Symmetrical_ToCommon) {                     // Symmetrical.cpp:225-273
    auto c = Symmetrical(SA(), vec({5, 6}), vec({7, 8}), true).ToCommon();
    CHECK(c->IsMaximization() && c->GetConstraintsMatrix() == SA());
    CHECK(c->GetConstraintTypes()[0] == CT::LessOrEqual && c->GetConstraintTypes()[1] == CT::LessOrEqual);
    CHECK(c->GetVariableTypes()[0] == VT::NonNegative && c->GetVariableTypes()[1] == VT::NonNegative);
    auto m = Symmetrical(SA(), vec({5, 6}), vec({7, 8}), false).ToCommon();
    CHECK(!m->IsMaximization() && m->GetConstraintTypes()[1] == CT::GreaterOrEqual);
}
TEST(Canonical_Conversions) {                    // Canonical.cpp:199-364
    Canonical can(CA(), vec({5, 6}), vec({7, 8, 0, 0}), {2, 3}, true);
    can.SetOriginalVariablesCount(2);
    auto com = can.ToCommon();                   // original variables only, all '=', all >= 0
    CHECK(!com->IsMaximization() && com->GetConstraintsMatrix() == SA());
    CHECK(com->GetConstraintTypes()[0] == CT::Equal && com->GetVariableTypes()[1] == VT::NonNegative);
    CHECK(com->GetObjectiveCoefficients() == vec({7, 8}) && com->GetRightHandSide() == vec({5, 6}));
    auto sym = can.ToSymmetrical();              // each row -> (a, b), (-a, -b); sense kept
    CHECK(!sym->IsMaximization());
    CHECK(sym->GetConstraintsMatrix() == mat(4, 2, {1, 2, -1, -2, 3, 4, -3, -4}));
    CHECK(sym->GetRightHandSide() == vec({5, -5, 6, -6}));
    auto d = can.GetDual();                      // [A^T | -A^T | I], costs [b | -b | 0], rhs c
    CHECK(d->IsMaximization() && d->GetOriginalVariablesCount() == 4);
    CHECK(d->GetConstraintsMatrix() == mat(4, 8, {1, 3, -1, -3, 1, 0, 0, 0,
                                                  2, 4, -2, -4, 0, 1, 0, 0,
                                                  1, 0, -1, -0.0, 0, 0, 1, 0,
                                                  0, 1, -0.0, -1, 0, 0, 0, 1}));
    CHECK(d->GetObjectiveCoefficients() == vec({5, 6, -5, -6, 0, 0, 0, 0}));
    CHECK(d->GetRightHandSide() == vec({7, 8, 0, 0}));
    CHECK(d->GetBasisIndices()[0] == 4 && d->GetBasisIndices()[3] == 7);
}

int main(int argc, char** argv) { return run_all(argc > 1 ? argv[1] : nullptr); }
