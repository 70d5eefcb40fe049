"""CPU-only: pins the oracle (oracle/lp_oracle.c) against everything the reference holds for the
hot path, and checks its two simplex restatements against each other and against scipy."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as o
from tests import lpcases

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_reference_fixture_basic_solution():
    # /root/reference/tests/test_canonical.cpp:41-66: basis {2,3} -> x = (0,0,5,6), feasible
    A, b, c, basis = lpcases.test_canonical_fixture()
    st, x = o.basic_solution(A, b, basis)
    assert st == o.OPTIMAL
    assert x.tolist() == [0.0, 0.0, 5.0, 6.0]  # EXPECT_DOUBLE_EQ in the reference
    assert o.is_feasible_basis(A, b, basis)
    assert o.evaluate(c, x) == 0.0


def test_reference_fixture_evaluate():
    # /root/reference/tests/test_common.cpp:49-59: c.x = 7*1 + 8*2 = 23
    assert o.evaluate([7, 8.0], [1, 2.0]) == 23.0


def test_basic_solution_general():
    rng = np.random.default_rng(3)
    A = rng.normal(size=(6, 11))
    b = rng.normal(size=6)
    basis = np.array([9, 1, 4, 0, 7, 3], dtype=np.int32)
    st, x = o.basic_solution(A, b, basis)
    assert st == o.OPTIMAL
    np.testing.assert_allclose(A @ x, b, rtol=1e-12, atol=1e-12)
    assert np.count_nonzero(x) <= 6


@pytest.mark.parametrize("fn", [o.simplex_reference, o.simplex_tableau])
def test_known_answers_survey_s4(fn):
    # hand-traced pivot sequences of SimplexSolover.h:135-209
    A, b, c, basis, no = lpcases.main_cpp_lp()
    r = fn(A, b, c, basis, True, no, trace_cap=8)
    assert r["status"] == o.OPTIMAL and r["trace"] == [(2, 0)]
    assert r["basis"].tolist() == [2, 4] and r["x"].tolist() == [0, 0, 6] and r["obj"] == 24
    A, b, c, basis, no = lpcases.input_symmetric_lp()
    r = fn(A, b, c, basis, True, no, trace_cap=8)
    assert r["status"] == o.OPTIMAL and r["trace"] == [(1, 1), (0, 1)]
    assert r["basis"].tolist() == [3, 0] and r["x"].tolist() == [5, 0, 0] and r["obj"] == 35


def test_enumeration_table_survey_s4():
    # C(5,2) = 10 subsets of input_symmetric.txt
    A, b, c, _, no = lpcases.input_symmetric_lp()
    verdicts = []
    for k in range(10):
        s = o.unrank(5, 2, k)
        assert o.rank_of(5, s) == k
        st, xB, z = o.enum_subset(A, b, c, s)
        verdicts.append(st)
    assert [k for k, v in enumerate(verdicts) if v == o.SUBSET_INFEASIBLE] == [0, 3, 6]
    assert verdicts.count(o.SUBSET_SINGULAR) == 0
    r = o.enum_solve(A, b, c, True, no)
    assert r["status"] == o.OPTIMAL and r["rank"] == 2 and r["basis"].tolist() == [0, 3]
    assert r["x"].tolist() == [5, 0, 0] and r["obj"] == 35 and r["counts"] == [7, 3, 0]


def test_chain_select_is_order_dependent():
    # the EPS-hysteresis scan of SimplexSolover.h:153-161 is not an arg-max
    v = [1.0, 1.0 + 0.6e-9, 1.0 + 1.2e-9]
    assert o.chain_select(v, want_max=True)[0] == 2      # 1.2e-9 above the first accepted value
    v = [1.0, 1.0 + 0.6e-9, 1.0 + 0.9e-9]
    assert o.chain_select(v, want_max=True)[0] == 0      # nothing exceeds best + eps
    assert o.chain_select([3.0, 1.0, 1.0 - 0.5e-9, 0.5], want_max=False) == (3, 0.5)
    assert o.chain_select([0.0, 0.0, 0.0], want_max=False)[0] == 0   # degenerate ties: first
    assert o.chain_select([5.0, 7.0], mask=[0, 0], want_max=True)[0] == -1


def test_binomials_and_ranking():
    assert o.binom(28, 14) == 40116600 and o.binom(32, 16) == 601080390
    assert o.binom(64, 32) == 1832624140942590534 and o.binom(5, 7) == 0
    n, m = 9, 4
    s = o.unrank(n, m, 0)
    for k in range(o.binom(n, m)):
        assert o.rank_of(n, s) == k
        assert np.array_equal(o.unrank(n, m, k), s)
        o.lib().orc_next_subset(n, m, s.ctypes.data_as(o._ip))


@pytest.mark.parametrize("seed,m,n", [(0, 8, 16), (1, 16, 32), (2, 32, 64), (3, 64, 128), (4, 24, 80)])
def test_tableau_matches_reference_shaped(seed, m, n):
    """Same pivot rules => same basis sequence; vertex within 1e-10 relative (north star)."""
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r1 = o.simplex_reference(A, b, c, basis, True, n - m, trace_cap=4096)
    r2 = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=4096)
    assert r1["status"] == r2["status"] == o.OPTIMAL
    assert r1["trace"] == r2["trace"] and r1["iters"] == r2["iters"] > 0
    assert np.array_equal(r1["basis"], r2["basis"])
    np.testing.assert_allclose(r2["x"], r1["x"], rtol=1e-10, atol=1e-12)
    assert abs(r2["obj"] - r1["obj"]) <= 1e-10 * abs(r1["obj"])


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_general_basis_and_minimise(seed):
    A, b, c, basis = lpcases.general_lp(seed, 7, 15)
    for maximize in (True, False):
        r1 = o.simplex_reference(A, b, c, basis, maximize, A.shape[1], trace_cap=512)
        r2 = o.simplex_tableau(A, b, c, basis, maximize, A.shape[1], trace_cap=512)
        assert r1["status"] == r2["status"] == o.OPTIMAL
        assert r1["trace"] == r2["trace"]
        np.testing.assert_allclose(r2["x"], r1["x"], rtol=1e-9, atol=1e-10)


def test_status_codes():
    # unbounded: max x1 with x1 - x2 <= 1
    A = np.array([[1.0, -1.0, 1.0]])
    for fn in (o.simplex_reference, o.simplex_tableau):
        assert fn(A, [1.0], [1.0, 1.0, 0.0], [2], True, 2)["status"] == o.UNBOUNDED
    # singular initial basis (column 2 of the commented-out LP in main.cpp:24-34 is zero)
    A = np.array([[4, 3, 0, 1], [0, 4, 0, 4.0]])
    for fn in (o.simplex_reference, o.simplex_tableau):
        assert fn(A, [4, 6.0], [5, 1, 0, 0.0], [0, 2], False, 4)["status"] == o.SINGULAR
    # iteration limit (SimplexSolover.h:450) and bad basis index (Canonical.cpp:40-46)
    A, b, c, basis = lpcases.random_lp(5, 16, 32)
    for fn in (o.simplex_reference, o.simplex_tableau):
        assert fn(A, b, c, basis, True, 16, max_iter=2)["status"] == o.ITER_LIMIT
        assert fn(A, b, c, [0] * 15 + [99], True, 16)["status"] == o.BAD_ARG


def test_against_scipy_linprog():
    """scipy (HiGHS) is an independent checker for the optimum only; never shipped."""
    from scipy.optimize import linprog
    for seed, m, n in [(21, 10, 20), (22, 20, 50), (23, 40, 80)]:
        A, b, c, basis = lpcases.random_lp(seed, m, n)
        r = o.simplex_tableau(A, b, c, basis, True, n - m)
        ref = linprog(-c[:n - m], A_ub=A[:, :n - m], b_ub=b, bounds=(0, None), method="highs")
        assert ref.status == 0
        assert abs(r["obj"] + ref.fun) <= 1e-9 * abs(ref.fun)


def test_enumeration_agrees_with_simplex():
    """README.md:42: the enumeration solver cross-checks the simplex solver."""
    for seed, m, n in [(31, 4, 9), (32, 5, 11), (33, 6, 12)]:
        A, b, c, basis = lpcases.random_lp(seed, m, n)
        s = o.simplex_tableau(A, b, c, basis, True, n - m)
        e = o.enum_solve(A, b, c, True, n - m)
        assert e["status"] == o.OPTIMAL and sum(e["counts"]) == o.binom(n, m)
        assert abs(e["obj"] - s["obj"]) <= 1e-10 * abs(s["obj"])
        np.testing.assert_allclose(e["x"], s["x"], rtol=1e-9, atol=1e-10)
        assert sorted(s["basis"].tolist()) == e["basis"].tolist()


def test_enum_range_sharding_is_exact():
    A, b, c, _ = lpcases.random_lp(41, 5, 12)
    total = o.binom(12, 5)
    st, z, counts = o.enum_range(A, b, c, True, 0, total)
    for parts in (2, 3, 8):
        cuts = [total * k // parts for k in range(parts + 1)]
        zs, cs = [], np.zeros(3, dtype=np.int64)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            _, zz, cc = o.enum_range(A, b, c, True, lo, hi)
            zs.append(zz)
            cs += cc
        assert max(zs) == z and cs.tolist() == counts
        firsts = [o.enum_first_within(A, b, c, True, lo, hi, z) for lo, hi in zip(cuts[:-1], cuts[1:])]
        assert min(firsts) == o.enum_first_within(A, b, c, True, 0, total, z)


def test_golden_vectors():
    """tests/golden/*.json were written by tests/golden/make_golden.py from this oracle
    (restatement-derived: the reference has no recorded solver outputs)."""
    path = os.path.join(GOLDEN, "simplex_cases.json")
    if not os.path.exists(path):
        pytest.skip("golden vectors not generated yet")
    for case in json.load(open(path)):
        A, b, c, basis = lpcases.random_lp(case["seed"], case["m"], case["n"])
        r = o.simplex_tableau(A, b, c, basis, True, case["n"] - case["m"], trace_cap=1 << 14)
        assert r["iters"] == case["iters"] and r["basis"].tolist() == case["basis"]
        assert r["obj"] == case["obj"]
        assert [list(t) for t in r["trace"][:16]] == case["trace_head"]
        for j, v in case["x_nonzero"].items():
            assert r["x"][int(j)] == v
        # the tableau form (what the GPU executes) against the committed outputs of the
        # reference-shaped form (SimplexSolover.h:429-447), including 512 x 1024: same pivot
        # trace and basis, vertex / objective within 1e-10 relative
        ref = case["reference_shaped"]
        assert [list(t) for t in r["trace"]] == ref["trace"] and r["basis"].tolist() == ref["basis"]
        assert abs(r["obj"] - ref["obj"]) <= 1e-10 * abs(ref["obj"])
        for j, v in ref["x_nonzero"].items():
            assert abs(r["x"][int(j)] - v) <= 1e-10 * abs(v) + 1e-13
    enum = json.load(open(os.path.join(GOLDEN, "enum_cases.json")))
    for case in enum["random"]:
        A, b, c, _ = lpcases.random_lp(case["seed"], case["m"], case["n"])
        e = o.enum_solve(A, b, c, True, case["n"] - case["m"])
        assert e["rank"] == case["rank"] and e["basis"].tolist() == case["basis"]
        assert e["obj"] == case["obj"] and e["counts"] == case["counts"] and e["x"].tolist() == case["x"]
    A, b, c, _, _ = lpcases.input_symmetric_lp()
    for row in enum["input_symmetric"]:
        st, xB, z = o.enum_subset(A, b, c, row["subset"])
        assert st == row["verdict"] and xB.tolist() == row["xB"] and z == row["z"]
