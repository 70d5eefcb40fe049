"""Two-phase simplex (build-defined, parity unpinned by the reference).
CPU part: the oracle's restatement against brute-force enumeration (a different algorithm) and
hand-checked cases.  GPU part: lp_simplex_two_phase through the C ABI, bit-exact against the
oracle, and cross-checked against the GPU enumeration."""
import numpy as np
import pytest

from oracle import pyoracle as o
from tests import lpcases


def test_oracle_textbook_min():
    # min 2x1+3x2, x1+x2 >= 4, x1+3x2 >= 6: vertex (3,1), objective 9
    A = np.array([[1, 1, -1, 0], [1, 3, 0, -1.0]])
    r = o.two_phase(A, [4, 6.0], [2, 3, 0, 0.0], maximize=False, n_orig=2)
    # (phase II continues on the phase-I tableau, so the vertex carries phase I's rounding: 3 - 4e-16)
    assert r["status"] == o.OPTIMAL and np.allclose(r["x"], [3, 1], rtol=0, atol=1e-12) and abs(r["obj"] - 9) <= 1e-12
    assert sorted(r["basis"].tolist()) == [0, 1]


def test_oracle_infeasible_dependent_negative_b():
    # x1+x2+s = 1 and x1+x2-t = 3 cannot both hold
    A = np.array([[1, 1, 1, 0], [1, 1, 0, -1.0]])
    assert o.two_phase(A, [1, 3.0], [1, 1, 0, 0.0], maximize=False, n_orig=2)["status"] == o.INFEASIBLE
    # the same row twice: the second artificial cannot leave (SimplexSolover.h:372-380)
    A = np.array([[1, 1.0], [1, 1.0]])
    assert o.two_phase(A, [2, 2.0], [1, 2.0], maximize=False, n_orig=2)["status"] == o.SINGULAR
    # -x1-x2+s = -2 (make_b_nonneg, :61-68): min x1+2x2 -> (2,0)
    r = o.two_phase(np.array([[-1, -1, 1.0]]), [-2.0], [1, 2, 0.0], maximize=False, n_orig=2)
    assert r["status"] == o.OPTIMAL and np.allclose(r["x"], [2, 0], rtol=0, atol=1e-12) and abs(r["obj"] - 2) <= 1e-12


@pytest.mark.parametrize("seed", range(12))
def test_oracle_matches_enumeration(seed):
    m, k = 5 + seed % 3, 4 + seed % 4
    A, b, c, no = lpcases.min_lp(seed, m, k, equalities=seed % 2, negative_rows=seed % 3,
                                 zero_rhs=(seed // 3) % 2)
    r = o.two_phase(A, b, c, maximize=False, n_orig=no)
    e = o.enum_solve(A, b, c, maximize=False, n_orig=no)
    assert r["status"] == o.OPTIMAL and e["status"] == o.OPTIMAL
    assert abs(r["obj"] - e["obj"]) <= 1e-9 * max(1.0, abs(e["obj"]))
    np.testing.assert_allclose(r["x"], e["x"], rtol=0, atol=1e-8)


def test_oracle_drive_out_happens():
    """At least one of the degenerate cases must exercise replaceArtificialColumns."""
    total = 0
    for seed in range(60):
        A, b, c, no = lpcases.degenerate_eq_lp(seed)
        r = o.two_phase(A, b, c, maximize=False, n_orig=no)
        assert r["status"] == o.OPTIMAL
        e = o.enum_solve(A, b, c, maximize=False, n_orig=no)
        assert abs(r["obj"] - e["obj"]) <= 1e-9 * max(1.0, abs(e["obj"]))
        total += r["iters"][1]
    assert total > 0


def _golden():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "two_phase_cases.json")))


def _golden_case(g):
    if g["kind"] == "min":
        a = g["args"]
        return lpcases.min_lp(g["seed"], a[0], a[1], equalities=a[2], negative_rows=a[3], zero_rhs=a[4])
    return lpcases.degenerate_eq_lp(g["seed"])


def test_oracle_golden():
    """tests/golden/two_phase_cases.json (optimum cross-checked against scipy HiGHS when written)."""
    for g in _golden():
        A, b, c, no = _golden_case(g)
        r = o.two_phase(A, b, c, maximize=False, n_orig=no)
        assert r["status"] == o.OPTIMAL and r["iters"] == g["iters"]
        assert r["basis"].tolist() == g["basis"] and r["obj"] == g["obj"] and r["x"].tolist() == g["x"]


# ------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


def _same(g, r):
    assert g["status"] == r["status"]
    assert g["iters"] == r["iters"]
    if r["status"] == o.OPTIMAL:
        assert np.array_equal(g["basis"], r["basis"])
        assert np.array_equal(g["x"], r["x"])  # bit for bit
        assert g["obj"] == r["obj"]


@gpu
def test_gpu_golden(ctx):
    for g in _golden():
        A, b, c, no = _golden_case(g)
        r = ctx.two_phase(A, b, c, maximize=False, n_orig=no)
        assert r["status"] == 0 and r["iters"] == g["iters"] and r["basis"].tolist() == g["basis"]
        assert r["obj"] == g["obj"] and r["x"].tolist() == g["x"]


@gpu
def test_gpu_known_cases(ctx):
    A = np.array([[1, 1, -1, 0], [1, 3, 0, -1.0]])
    g = ctx.two_phase(A, [4, 6.0], [2, 3, 0, 0.0], maximize=False, n_orig=2)
    # (phase II continues on the phase-I tableau: the vertex carries phase I's rounding, 3 - 4e-16;
    # bit-exactness is against the oracle, below)
    assert g["status"] == 0 and np.allclose(g["x"], [3, 1], rtol=0, atol=1e-12) and abs(g["obj"] - 9) <= 1e-12
    r = o.two_phase(A, [4, 6.0], [2, 3, 0, 0.0], maximize=False, n_orig=2)
    assert g["x"].tolist() == r["x"].tolist() and g["obj"] == r["obj"]
    A = np.array([[1, 1, 1, 0], [1, 1, 0, -1.0]])
    assert ctx.two_phase(A, [1, 3.0], [1, 1, 0, 0.0], maximize=False, n_orig=2)["status"] == o.INFEASIBLE
    A = np.array([[1, 1.0], [1, 1.0]])
    assert ctx.two_phase(A, [2, 2.0], [1, 2.0], maximize=False, n_orig=2)["status"] == o.SINGULAR
    g = ctx.two_phase(np.array([[-1, -1, 1.0]]), [-2.0], [1, 2, 0.0], maximize=False, n_orig=2)
    assert g["status"] == 0 and np.allclose(g["x"], [2, 0], rtol=0, atol=1e-12)


@gpu
@pytest.mark.parametrize("seed,m,k,eq,neg,zr", [(0, 5, 4, 0, 0, 0), (1, 6, 5, 1, 2, 0), (2, 7, 6, 2, 0, 2),
                                                 (3, 16, 12, 3, 4, 2), (4, 33, 40, 0, 5, 0),
                                                 (5, 64, 64, 4, 8, 3), (6, 128, 96, 0, 0, 0)])
def test_gpu_bit_exact_vs_oracle(ctx, seed, m, k, eq, neg, zr):
    A, b, c, no = lpcases.min_lp(seed, m, k, equalities=eq, negative_rows=neg, zero_rhs=zr)
    r = o.two_phase(A, b, c, maximize=False, n_orig=no)
    g = ctx.two_phase(A, b, c, maximize=False, n_orig=no)
    assert r["status"] == o.OPTIMAL
    _same(g, r)


@gpu
def test_gpu_corner_shapes(ctx):
    """One row; every row negated; a maximisation (bounded: negative costs); equality rows only."""
    cases = []
    A, b, c, no = lpcases.min_lp(11, 1, 3)
    cases.append((A, b, c, no, False))
    A, b, c, no = lpcases.min_lp(12, 6, 5, negative_rows=6)
    cases.append((A, b, c, no, False))
    A, b, c, no = lpcases.min_lp(13, 7, 6, equalities=2, negative_rows=3)
    cases.append((A, b, -c, no, True))            # max -c.x  ==  min c.x
    A, b, c, no = lpcases.degenerate_eq_lp(14, m=5, k=9, zero_rows=1)
    cases.append((A, b, c, no, False))
    for A, b, c, no, mx in cases:
        r = o.two_phase(A, b, c, maximize=mx, n_orig=no)
        g = ctx.two_phase(A, b, c, maximize=mx, n_orig=no)
        assert r["status"] == o.OPTIMAL
        _same(g, r)
    # an unbounded phase II: min -x1 with x1 - x2 = 1 (x1 can grow with x2)
    A = np.array([[1.0, -1.0]])
    r = o.two_phase(A, [1.0], [-1.0, 0.0], maximize=False, n_orig=2)
    g = ctx.two_phase(A, [1.0], [-1.0, 0.0], maximize=False, n_orig=2)
    assert r["status"] == o.UNBOUNDED and g["status"] == o.UNBOUNDED


@gpu
def test_gpu_drive_out_bit_exact(ctx):
    hits = 0
    for seed in range(60):
        A, b, c, no = lpcases.degenerate_eq_lp(seed)
        r = o.two_phase(A, b, c, maximize=False, n_orig=no)
        if r["status"] == o.OPTIMAL and r["iters"][1] > 0:
            _same(ctx.two_phase(A, b, c, maximize=False, n_orig=no), r)
            hits += 1
            if hits == 4:
                break
    assert hits > 0


@gpu
@pytest.mark.parametrize("seed", range(4))
def test_gpu_two_phase_vs_gpu_enumeration(ctx, seed):
    """README.md:42's cross-check for min problems: both GPU solvers, different algorithms."""
    A, b, c, no = lpcases.min_lp(seed, 8, 8, negative_rows=seed)
    g = ctx.two_phase(A, b, c, maximize=False, n_orig=no)
    e = ctx.enum_solve(A, b, c, maximize=False, n_orig=no)
    assert g["status"] == 0 and e["status"] == 0
    assert abs(g["obj"] - e["obj"]) <= 1e-10 * max(1.0, abs(e["obj"]))
    np.testing.assert_allclose(g["x"], e["x"], rtol=0, atol=1e-9)


@gpu
def test_gpu_force_pivot_and_row(ctx):
    """lp_simplex_row / lp_simplex_force_pivot against the oracle's tableau after one pivot."""
    A, b, c, basis = lpcases.random_lp(3, 12, 30)
    p = ctx.simplex_problem(A, b, c, basis, True, 18)
    T0 = np.vstack([np.hstack([A, b[:, None]]), np.append(c, 0.0)[None, :]])
    np.testing.assert_array_equal(p.row(12), T0[12])
    p.force_pivot(4, 7)
    T = T0.copy()
    import math
    ur = T[4, 7]
    for i in range(13):
        if i != 4:
            l = -T[i, 7] / ur
            T[i] = [math.fma(l, T[4, j], T[i, j]) for j in range(31)] if hasattr(math, "fma") else T[i] + l * T[4]
            T[i, 7] = 0.0
    T[4] = T[4] * (1.0 / ur)
    T[4, 7] = 1.0
    got = np.vstack([p.row(i) for i in range(13)])
    if hasattr(math, "fma"):
        np.testing.assert_array_equal(got, T)
    else:
        np.testing.assert_allclose(got, T, rtol=1e-13, atol=1e-13)
    assert p.download()["basis"][4] == 7
    p.free()
