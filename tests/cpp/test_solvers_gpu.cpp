// GPU tests of the drop-in C++ solver classes (Solver, EnumerationSolver) against the known
// answers of SURVEY.md §4 and each other (README.md:42: the enumeration solver cross-checks
// the simplex solver).
#include <thread>

#include "check.h"
#include "Canonical.h"
#include "Common.h"
#include "EnumerationSolver.h"
#include "SimplexSolover.h"
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

TEST(Solver_MainCppProblem) {   // /root/reference/src/main.cpp:48-57,111-113
    Symmetrical s(mat(2, 3, {1, 1, 1, 2, 1, 0}), vec({6, 8}), vec({3, 2, 4}), true);
    auto c = s.ToCanonical();
    Solver solver(*c);
    VectorXd x = solver.solve();
    CHECK(x.size() == 3 && x[0] == 0 && x[1] == 0 && x[2] == 6);
    auto r = solver.solve_ex();
    CHECK(r.objective == 24 && r.iterations == 1 && r.basis[0] == 2 && r.basis[1] == 4);
}
TEST(Solver_InputSymmetric) {   // /root/reference/input_symmetric.txt
    SymmetricalParser p;
    auto s = p.ParseFromString("maximize\r\nobjective:\r\n7 8 3\r\nconstraints:\r\n1 2 3 10\r\n4 5 6 20\r\n");
    CHECK(s);
    auto c = s->ToCanonical();
    auto r = Solver(*c).solve_ex();
    CHECK(r.x[0] == 5 && r.x[1] == 0 && r.x[2] == 0 && r.objective == 35 && r.iterations == 2);
    CHECK(r.basis[0] == 3 && r.basis[1] == 0);
    auto e = EnumerationSolver(*c).solve_ex();
    CHECK(e.rank == 2 && e.basis[0] == 0 && e.basis[1] == 3 && e.objective == 35);
    CHECK(e.feasible == 7 && e.infeasible == 3 && e.singular == 0);
    CHECK(e.x[0] == 5 && e.x[1] == 0 && e.x[2] == 0);
    VectorXd xe = EnumerationSolver(*c).solve();
    CHECK(xe[0] == r.x[0] && xe[1] == r.x[1] && xe[2] == r.x[2]);
}
TEST(Solver_Exceptions) {       // SimplexSolover.h:126, :443
    Canonical unb(mat(1, 3, {1, -1, 1}), vec({1}), vec({1, 1, 0}), {2}, false);
    CHECK_THROWS(Solver(unb).solve(), std::runtime_error);
    Canonical sing(mat(2, 4, {4, 3, 0, 1, 0, 4, 0, 4}), vec({4, 6}), vec({5, 1, 0, 0}), {0, 2}, true);
    CHECK_THROWS(Solver(sing).solve(), std::runtime_error);
    auto r = Solver(sing).solve_ex(false);
    CHECK(r.status == LP_SINGULAR);
    Canonical nofeas(mat(1, 2, {1, 1}), vec({-1}), vec({1, 1}), {0}, false);
    CHECK_THROWS(EnumerationSolver(nofeas).solve(), std::runtime_error);
}
TEST(Solvers_CrossCheck_Random) {   // README.md:42
    unsigned long long st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(st >> 11) / 9007199254740992.0; };
    for (int trial = 0; trial < 4; ++trial) {
        const int m = 5 + trial, no = 6 + trial;
        MatrixXd A(m, no);
        VectorXd b(m), c(no);
        for (int j = 0; j < no; ++j) { c[j] = rnd(); for (int i = 0; i < m; ++i) A(i, j) = rnd(); }
        for (int i = 0; i < m; ++i) b[i] = (1.0 + rnd()) * no * 0.5;
        auto can = Symmetrical(A, b, c, true).ToCanonical();
        auto s = Solver(*can).solve_ex();
        auto e = EnumerationSolver(*can).solve_ex();
        CHECK(std::fabs(s.objective - e.objective) <= 1e-10 * std::fabs(s.objective));
        for (int j = 0; j < no; ++j) CHECK(std::fabs(s.x[j] - e.x[j]) <= 1e-9 * (1 + std::fabs(s.x[j])));
        CHECK(std::fabs(can->Evaluate([&] { VectorXd f = VectorXd::Zero(m + no); for (int j = 0; j < no; ++j) f[j] = s.x[j]; return f; }()) - s.objective) < 1e-9);
    }
}
TEST(Solver_TwoPhase) {   // SURVEY 8(f) N2; flow of SimplexSolover.h:61-95,331-406 designed afresh
    // min 2x1+3x2, x1+x2 >= 4, x1+3x2 >= 6 in canonical form [A | -I]; the basis indices given to
    // Canonical are placeholders (its constructor wants m of them), twoPhaseSimplex ignores them
    Canonical can(mat(2, 4, {1, 1, -1, 0, 1, 3, 0, -1}), vec({4, 6}), vec({2, 3, 0, 0}), {0, 1}, true);
    can.SetOriginalVariablesCount(2);
    Solver s(can);
    VectorXd x = s.twoPhaseSimplex();
    // (phase II continues on the phase-I tableau, so the vertex carries phase I's rounding: 3 - 4e-16)
    CHECK(x.size() == 2 && std::fabs(x[0] - 3) <= 1e-12 && std::fabs(x[1] - 1) <= 1e-12);
    int it[3];
    auto r = s.twoPhaseSimplex_ex(true, it);
    CHECK(std::fabs(r.objective - 9) <= 1e-12 && it[0] == 2 && it[1] == 0 && r.status == LP_OPTIMAL);
    // the enumeration solver agrees (README.md:42)
    auto e = EnumerationSolver(can).solve_ex();
    CHECK(e.objective == 9 && e.x[0] == 3 && e.x[1] == 1);
    // no feasible point: x1+x2+s = 1, x1+x2-t = 3  (:352-353)
    Canonical nofeas(mat(2, 4, {1, 1, 1, 0, 1, 1, 0, -1}), vec({1, 3}), vec({1, 1, 0, 0}), {0, 1}, true);
    CHECK_THROWS(Solver(nofeas).twoPhaseSimplex(), std::runtime_error);
    CHECK(Solver(nofeas).twoPhaseSimplex_ex(false).status == LP_INFEASIBLE);
    // linearly dependent constraints (:372-380)
    Canonical dep(mat(2, 2, {1, 1, 1, 1}), vec({2, 2}), vec({1, 2}), {0, 1}, true);
    CHECK(Solver(dep).twoPhaseSimplex_ex(false).status == LP_SINGULAR);
    // negative right-hand side (make_b_nonneg, :61-68): -x1-x2+s = -2, min x1+2x2 -> (2, 0)
    Canonical neg(mat(1, 3, {-1, -1, 1}), vec({-2}), vec({1, 2, 0}), {0}, true);
    neg.SetOriginalVariablesCount(2);
    VectorXd xn = Solver(neg).twoPhaseSimplex();
    CHECK(std::fabs(xn[0] - 2) <= 1e-12 && std::fabs(xn[1]) <= 1e-12);
}
TEST(Common_EndToEnd_TwoPhase) {   // N4 + N2: a general-form problem through the whole chain
    // min 2x1 + 3x2 - x3;  x1 + x2 + x3 >= 4,  x1 + 3x2 = 6,  x1 - x3 <= 5;  x1 >= 0, x2 >= 0, x3 <= 0
    // (feasible: (3, 1, 0)); the optimum is checked against the enumeration solver, not by hand.
    using CT = Common::ConstraintType;
    using VT = Common::VariableType;
    Common com(mat(3, 3, {1, 1, 1, 1, 3, 0, 1, 0, -1}), vec({4, 6, 5}), vec({2, 3, -1}),
               {CT::GreaterOrEqual, CT::Equal, CT::LessOrEqual}, {VT::NonNegative, VT::NonNegative, VT::NonPositive}, false);
    auto can = com.ToCanonical();   // max / <= form with slacks: b has negative entries -> no feasible slack basis
    auto r = Solver(*can).twoPhaseSimplex_ex();
    auto e = EnumerationSolver(*can).solve_ex();
    CHECK(r.status == LP_OPTIMAL);
    CHECK(std::fabs(r.objective - e.objective) <= 1e-10 * (1 + std::fabs(e.objective)));
    for (long j = 0; j < r.x.size(); ++j) CHECK(std::fabs(r.x[j] - e.x[j]) <= 1e-9);
    // back in the original variables: x3 = -x3'; the symmetric form maximises -(c.x)
    const double x1 = r.x[0], x2 = r.x[1], x3 = -r.x[2];
    CHECK(x1 >= -1e-9 && x2 >= -1e-9 && x3 <= 1e-9);
    CHECK(x1 + x2 + x3 >= 4 - 1e-9 && std::fabs(x1 + 3 * x2 - 6) <= 1e-9 && x1 - x3 <= 5 + 1e-9);
    VectorXd xo(3);
    xo[0] = x1; xo[1] = x2; xo[2] = x3;
    CHECK(std::fabs(com.Evaluate(xo) + r.objective) <= 1e-9);
    // the same chain reports an infeasible general-form problem (x1 - x3 <= 2 contradicts the rest)
    Common bad(mat(3, 3, {1, 1, 1, 1, 3, 0, 1, 0, -1}), vec({4, 6, 2}), vec({2, 3, -1}),
               {CT::GreaterOrEqual, CT::Equal, CT::LessOrEqual}, {VT::NonNegative, VT::NonNegative, VT::NonPositive}, false);
    CHECK(Solver(*bad.ToCanonical()).twoPhaseSimplex_ex(false).status == LP_INFEASIBLE);
    CHECK_THROWS(EnumerationSolver(*bad.ToCanonical()).solve(), std::runtime_error);
}
TEST(Enumeration_MultiGpuShardsOnOneDevice) {
    // The sharded path of EnumerationSolver::solve_ex(n_gpus): one host thread, one context and one
    // replica of the problem per shard, shard g on device g % lp_device_count(), the incumbent
    // exchanged by lp_enum_solve_sharded.  On a one-GPU box the shards share the device (exchange
    // through host memory); the answer must not depend on the number of shards (tie rule).
    auto small = Symmetrical(mat(2, 3, {1, 2, 3, 4, 5, 6}), vec({10, 20}), vec({7, 8, 3}), true).ToCanonical();
    auto a = EnumerationSolver(*small).solve_ex(1);
    for (int shards : {2, 3, 8, 16}) {   // 16 > C(5,2) = 10: some shards are empty
        auto s = EnumerationSolver(*small).solve_ex(shards);
        CHECK(s.status == LP_OPTIMAL && s.rank == a.rank && s.objective == a.objective);
        CHECK(s.feasible == a.feasible && s.infeasible == a.infeasible && s.singular == a.singular);
        CHECK(s.basis == a.basis && s.x[0] == a.x[0] && s.x[1] == a.x[1] && s.x[2] == a.x[2]);
    }
    // a problem large enough for the shared-prefix kernels (C(22,10) = 646,646 per solve; shards of
    // 2^20+ subsets would take them on bigger shapes — here the direct kernel), ties included
    unsigned long long st = 777;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(st >> 11) / 9007199254740992.0; };
    const int m = 10, no = 12;
    MatrixXd A(m, no);
    VectorXd b(m), c(no);
    for (int j = 0; j < no; ++j) { c[j] = 1.0 + (j % 3); for (int i = 0; i < m; ++i) A(i, j) = rnd(); }   // repeated costs: near-ties
    for (int i = 0; i < m; ++i) b[i] = (1.0 + rnd()) * no * 0.5;
    auto can = Symmetrical(A, b, c, true).ToCanonical();
    auto one = EnumerationSolver(*can).solve_ex(1);
    auto simplex = Solver(*can).solve_ex();
    CHECK(std::fabs(one.objective - simplex.objective) <= 1e-10 * std::fabs(simplex.objective));
    for (int shards : {2, 3, 8}) {
        auto s = EnumerationSolver(*can).solve_ex(shards);
        CHECK(s.status == LP_OPTIMAL && s.rank == one.rank && s.objective == one.objective);
        CHECK(s.feasible == one.feasible && s.infeasible == one.infeasible && s.singular == one.singular);
        CHECK(s.basis == one.basis);
        for (int j = 0; j < no; ++j) CHECK(s.x[j] == one.x[j]);
    }
    // no feasible basis in any shard: every shard reports it, the wrapper throws like the single-GPU path
    Canonical nofeas(mat(1, 2, {1, 1}), vec({-1}), vec({1, 1}), {0}, false);
    CHECK(EnumerationSolver(nofeas).solve_ex(2, false).status == LP_INFEASIBLE);
    CHECK_THROWS(EnumerationSolver(nofeas).solve_ex(3), std::runtime_error);
}
TEST(Enumeration_FailingShardStillJoinsTheExchange) {
    // One shard cannot enumerate (as if its upload had failed).  It must still contribute its record to the
    // exchange — otherwise the other shards' threads wait forever — and every shard comes back with its
    // status.  First, middle and last shard; exchange through host memory (shards share the device).
    auto can = Symmetrical(mat(2, 3, {1, 2, 3, 4, 5, 6}), vec({10, 20}), vec({7, 8, 3}), true).ToCanonical();
    for (int bad : {0, 1, 3}) {
        EnumerationSolver es(*can);
        es.debug_fail_shard(bad);
        auto s = es.solve_ex(4, false, EnumerationSolver::EXCHANGE_LOCAL);
        CHECK(s.status == LP_BAD_ARG);
        EnumerationSolver es2(*can);
        es2.debug_fail_shard(bad);
        CHECK_THROWS(es2.solve_ex(4, true, EnumerationSolver::EXCHANGE_LOCAL), std::invalid_argument);
    }
    // A shard whose DEVICE cannot be opened ends the whole call during the set-up: no thread has been started,
    // nobody has entered a collective (with RCCL: ncclCommInitRank), so nobody can be left waiting for it.
    for (auto ex : {EnumerationSolver::EXCHANGE_LOCAL, EnumerationSolver::EXCHANGE_RCCL}) {
        const int shards = ex == EnumerationSolver::EXCHANGE_RCCL ? 1 : 3;   // (RCCL needs a device per shard)
        EnumerationSolver es3(*can);
        es3.debug_shard_device(shards - 1, 4096);
        CHECK(es3.solve_ex(shards, false, ex).status == LP_BAD_ARG);
        EnumerationSolver es4(*can);
        es4.debug_shard_device(shards - 1, 4096);
        CHECK_THROWS(es4.solve_ex(shards, true, ex), std::invalid_argument);
        CHECK(EnumerationSolver(*can).solve_ex(shards, true, ex).status == LP_OPTIMAL);   // and the next call works
    }
    // the C entry point with no problem takes part as a failed participant too
    lp_comm* comms[2] = {nullptr, nullptr};
    CHECK(lp_comm_create_local(2, comms) == LP_OPTIMAL);
    int rc0 = 0, rc1 = 0;
    std::thread t0([&] { rc0 = lp_enum_solve_sharded(comms[0], nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr); });
    std::thread t1([&] { rc1 = lp_enum_shard_abstain(comms[1], LP_SINGULAR); });
    t0.join();
    t1.join();
    CHECK(rc0 == LP_BAD_ARG && rc1 == LP_SINGULAR);
    lp_comm_destroy(comms[0]);
    lp_comm_destroy(comms[1]);
}
TEST(Enumeration_WideShape) {
    // 18 rows (more than the 16 of the tuned leaf kernels): C(28,18) = 13.1 M bases on 32-row records and
    // the general leaf kernel; the simplex solver is the cross-check (README.md:42), and the answer
    // does not depend on the number of shards.
    unsigned long long st = 4242;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(st >> 11) / 9007199254740992.0; };
    const int m = 18, no = 10;
    MatrixXd A(m, no);
    VectorXd b(m), c(no);
    for (int j = 0; j < no; ++j) { c[j] = rnd(); for (int i = 0; i < m; ++i) A(i, j) = rnd(); }
    for (int i = 0; i < m; ++i) b[i] = (1.0 + rnd()) * no * 0.5;
    auto can = Symmetrical(A, b, c, true).ToCanonical();
    auto simplex = Solver(*can).solve_ex();
    auto one = EnumerationSolver(*can).solve_ex(1);
    CHECK(one.status == LP_OPTIMAL);
    CHECK(one.feasible + one.infeasible + one.singular == lp_binom(m + no, m));
    CHECK(std::fabs(one.objective - simplex.objective) <= 1e-10 * std::fabs(simplex.objective));
    for (int j = 0; j < no; ++j) CHECK(std::fabs(one.x[j] - simplex.x[j]) <= 1e-9 * (1 + std::fabs(simplex.x[j])));
    auto three = EnumerationSolver(*can).solve_ex(3);
    CHECK(three.status == LP_OPTIMAL && three.rank == one.rank && three.objective == one.objective);
    CHECK(three.feasible == one.feasible && three.infeasible == one.infeasible && three.singular == one.singular);
    CHECK(three.basis == one.basis);
}
TEST(Enumeration_RcclExchange) {
    // The RCCL communicator of the C ABI (ncclCommInitRank + ONE ncclAllGather of the 48-byte
    // record) with as many shards as this box has devices — one on a one-GPU box, where it still
    // proves that librccl loads, the communicator initialises and the collective runs on the
    // library's stream; the driver's 8-GPU node runs the same entry point with world = 8.
    auto can = Symmetrical(mat(2, 3, {1, 2, 3, 4, 5, 6}), vec({10, 20}), vec({7, 8, 3}), true).ToCanonical();
    auto a = EnumerationSolver(*can).solve_ex(1);
    const int ndev = lp_device_count();
    auto s = EnumerationSolver(*can).solve_ex(ndev, true, EnumerationSolver::EXCHANGE_RCCL);
    CHECK(s.status == LP_OPTIMAL && s.rank == a.rank && s.objective == a.objective && s.feasible == a.feasible);
}

int main(int argc, char** argv) { return run_all(argc > 1 ? argv[1] : nullptr); }
