"""GPU parity: the HIP simplex path (through the C ABI) against the oracle's tableau
restatement — bit-exact pivots, basis, tableau and vertex — and against the reference-shaped
restatement within the north star's 1e-10."""
import numpy as np
import pytest

from oracle import pyoracle as o
from simplexmethod_amd import capi
from tests import lpcases

pytestmark = pytest.mark.gpu

ALGOS = [capi.SIMPLEX_LAUNCH, capi.SIMPLEX_LOOKAHEAD, capi.SIMPLEX_RESIDENT, capi.SIMPLEX_OVERLAP]


def _run(ctx, A, b, c, basis, maximize, n_orig, trace_cap=1 << 14, max_iter=capi.MAX_ITER,
         algo=capi.SIMPLEX_AUTO):
    p = ctx.simplex_problem(A, b, c, basis, maximize, n_orig)
    rc, st = p.run(max_iter=max_iter, algo=algo)
    out = p.download(trace_cap=min(trace_cap, max(st.pivots, 1)), want_tableau=True)
    p.free()
    # the algorithm that was asked for is the one that answered (no silent fallback)
    assert st.fell_back == 0
    if algo != capi.SIMPLEX_AUTO and rc != capi.SINGULAR:
        assert st.algo_used == algo, (st.algo_used, algo)
    out.update(status=rc, iters=st.pivots, algo_used=st.algo_used)
    return out


def _assert_bit_exact(g, r):
    assert g["status"] == r["status"]
    assert g["iters"] == r["iters"]
    k = r["iters"]
    assert list(zip(g["trace_enter"][:k].tolist(), g["trace_leave"][:k].tolist())) == r["trace"][:k]
    assert np.array_equal(g["basis"], r["basis"])
    if r["status"] == o.OPTIMAL:
        assert np.array_equal(g["x"], r["x"])        # bit for bit
        assert g["obj"] == r["obj"]
    if r["tableau"] is not None and r["status"] in (o.OPTIMAL, o.ITER_LIMIT, o.UNBOUNDED):
        assert np.array_equal(g["tableau"], r["tableau"])


def test_known_answers(ctx):
    A, b, c, basis, no = lpcases.main_cpp_lp()
    r = ctx.simplex_solve(A, b, c, basis, True, no)
    assert r["status"] == 0 and r["x"].tolist() == [0, 0, 6] and r["obj"] == 24
    assert r["basis"].tolist() == [2, 4] and r["iters"] == 1
    A, b, c, basis, no = lpcases.input_symmetric_lp()
    r = ctx.simplex_solve(A, b, c, basis, True, no)
    assert r["status"] == 0 and r["x"].tolist() == [5, 0, 0] and r["obj"] == 35
    assert r["basis"].tolist() == [3, 0] and r["iters"] == 2


@pytest.mark.parametrize("seed,m,n", [(0, 2, 5), (1, 8, 16), (2, 16, 32), (3, 33, 71),
                                       (4, 64, 128), (5, 128, 256), (6, 100, 1500)])
@pytest.mark.parametrize("algo", ALGOS)
def test_random_lp_bit_exact(ctx, seed, m, n, algo):
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True)
    g = _run(ctx, A, b, c, basis, True, n - m, algo=algo)
    assert r["status"] == o.OPTIMAL and r["iters"] > 0
    _assert_bit_exact(g, r)


@pytest.mark.parametrize("algo", ALGOS)
def test_baseline_config_512x1024(ctx, algo):
    """BASELINE.json configs[1]: m=512, n=1024, seed 0."""
    m, n = 512, 1024
    A, b, c, basis = lpcases.random_lp(0, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True)
    g = _run(ctx, A, b, c, basis, True, n - m, algo=algo)
    _assert_bit_exact(g, r)
    # size-independent properties of the final tableau: basic columns are exact unit
    # vectors, reduced costs of an optimum are <= eps, xB >= 0 up to rounding
    T = g["tableau"]
    for t, j in enumerate(g["basis"]):
        col = T[:m, j]
        assert col[t] == 1.0 and np.count_nonzero(col) == 1 and T[m, j] == 0.0
    assert T[m, :n].max() <= 1e-9
    assert T[:m, n].min() >= -1e-9
    xfull = np.zeros(n)
    xfull[g["basis"]] = T[:m, n]
    np.testing.assert_allclose(A @ xfull, b, rtol=1e-9)


@pytest.mark.parametrize("mode", ["LP_RESIDENT_SPREAD", "LP_RESIDENT_FORCE_SC1"])
def test_baseline_config_512x1024_resident_modes(ctx, monkeypatch, mode):
    """The chip-resident kernel's placement-dependent forms, forced: participants spread over all
    XCDs (write-through stores, no shared L2) and write-through stores on one XCD.  Same bits as the oracle."""
    monkeypatch.setenv(mode, "1")
    m, n = 512, 1024
    A, b, c, basis = lpcases.random_lp(0, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True)
    g = _run(ctx, A, b, c, basis, True, n - m, algo=capi.SIMPLEX_RESIDENT)
    _assert_bit_exact(g, r)


@pytest.mark.parametrize("seed,m,n", [(41, 768, 1536), (42, 600, 1300), (43, 960, 1920), (44, 513, 700)])
def test_resident_rows_beyond_512(ctx, seed, m, n):
    """512 < m <= 960: 16 columns per workgroup, up to 960 row threads (one row per thread) plus the
    communication wave, workgroups on several XCDs.  AUTO picks the chip-resident algorithm for these
    shapes too."""
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True)
    assert r["status"] == o.OPTIMAL and r["iters"] > 50
    for algo in (capi.SIMPLEX_RESIDENT, capi.SIMPLEX_AUTO):
        g = _run(ctx, A, b, c, basis, True, n - m, algo=algo)
        assert g["algo_used"] == capi.SIMPLEX_RESIDENT
        _assert_bit_exact(g, r)


@pytest.mark.parametrize("seed,m,n", [(0, 512, 1024), (43, 960, 1920), (6, 100, 1500)])
@pytest.mark.parametrize("max_iter", [1, 2, 3, 30, 31])
def test_resident_iteration_limit_bit_exact(ctx, seed, m, n, max_iter):
    """The pivot that reaches the iteration limit is applied and nothing is chosen behind it
    (SimplexSolover.h:450): on that last pivot the kernel publishes no candidate, and the rank-1 update still
    has to wait, behind the row barrier, until every wave has read the pivot row.  Tableau, reduced costs, xB,
    basis and trace bit-exact at both register layouts (32 and 16 columns per workgroup) and on a wide shape
    (47 workgroups)."""
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=64, want_tableau=True, max_iter=max_iter)
    assert r["status"] == o.ITER_LIMIT and r["iters"] == max_iter
    g = _run(ctx, A, b, c, basis, True, n - m, trace_cap=64, max_iter=max_iter, algo=capi.SIMPLEX_RESIDENT)
    _assert_bit_exact(g, r)


def test_resident_shape_limit(ctx):
    """m > 960 does not fit (one row per thread + the communication wave <= 1024 threads): an explicit
    request says so, AUTO solves it on the look-ahead path, same bits as the oracle."""
    m, n = 1024, 1200
    A, b, c, basis = lpcases.random_lp(46, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    with pytest.raises(capi.LPError):
        p.run(algo=capi.SIMPLEX_RESIDENT)
    p.free()
    g = _run(ctx, A, b, c, basis, True, n - m, algo=capi.SIMPLEX_AUTO)
    assert g["algo_used"] == capi.SIMPLEX_LOOKAHEAD
    _assert_bit_exact(g, r)


@pytest.mark.parametrize("max_iter", [1, 2, 37, 38])
def test_overlap_large_shape(ctx, max_iter):
    """2048 x 4096 (beyond the chip-resident shapes, look-ahead depth 1): AUTO takes the one-launch-per-pivot
    path whose update of pivot k overlaps the selection of pivot k+1 (simplex_overlap.hip).  Stopped after an
    odd and an even number of pivots (the tableau alternates between two buffers): same pivots and the same
    tableau, bit for bit, as the oracle."""
    m, n = 2048, 4096
    A, b, c, basis = lpcases.random_lp(47, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=64, want_tableau=True, max_iter=max_iter)
    assert r["status"] == o.ITER_LIMIT and r["iters"] == max_iter
    g = _run(ctx, A, b, c, basis, True, n - m, trace_cap=64, max_iter=max_iter, algo=capi.SIMPLEX_AUTO)
    assert g["algo_used"] == capi.SIMPLEX_OVERLAP
    _assert_bit_exact(g, r)


@pytest.mark.parametrize("max_iter", [2, 3])
def test_overlap_general_selector_and_tiled_update(ctx, max_iter):
    """1100 x 11000 (97 MB): the one-launch-per-pivot path in its OTHER forms — the priced cost row does not fit LDS
    beside the selector's vectors, so the general selector runs (staging in global memory), and 1548 tiles are more
    than three rounds of the resident workgroups, so the update takes one tile per workgroup instead of persistent
    linear shares.  Same pivots and the same tableau, bit for bit, as the oracle, after an even and an odd number of
    pivots (the tableau alternates between two buffers)."""
    m, n = 1100, 11000
    A, b, c, basis = lpcases.random_lp(48, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=64, want_tableau=True, max_iter=max_iter)
    assert r["status"] == o.ITER_LIMIT and r["iters"] == max_iter
    g = _run(ctx, A, b, c, basis, True, n - m, trace_cap=64, max_iter=max_iter, algo=capi.SIMPLEX_OVERLAP)
    _assert_bit_exact(g, r)


def test_resident_fallback_is_visible(ctx, monkeypatch):
    """A hand-off failure (injected: the placement census reports one) must be impossible to miss: with
    LP_RESIDENT_STRICT (the whole GPU session) the run returns an error; without it the solve is re-run on
    the look-ahead path, the answer is the same bit for bit, and lp_simplex_stats says which algorithm
    answered and that it was a fallback.  Nothing of the failed launch reaches the tableau."""
    A, b, c, basis = lpcases.random_lp(45, 64, 160)
    r = o.simplex_tableau(A, b, c, basis, True, 96, trace_cap=4096, want_tableau=True)
    monkeypatch.setenv("LP_RESIDENT_INJECT_FAILURE", "1")
    p = ctx.simplex_problem(A, b, c, basis, True, 96)
    with pytest.raises(capi.LPError):
        p.run(algo=capi.SIMPLEX_RESIDENT)
    monkeypatch.delenv("LP_RESIDENT_STRICT")
    for algo in (capi.SIMPLEX_RESIDENT, capi.SIMPLEX_AUTO):
        p.reset()
        rc, st = p.run(algo=algo)
        assert rc == 0 and st.fell_back == 1 and st.algo_used == capi.SIMPLEX_LOOKAHEAD
        g = p.download(trace_cap=st.pivots, want_tableau=True)
        g.update(status=rc, iters=st.pivots)
        _assert_bit_exact(g, r)
    monkeypatch.delenv("LP_RESIDENT_INJECT_FAILURE")
    p.reset()
    rc, st = p.run(algo=capi.SIMPLEX_AUTO)
    assert rc == 0 and st.fell_back == 0 and st.algo_used == capi.SIMPLEX_RESIDENT and st.launches == 2
    p.free()


@pytest.mark.parametrize("seed,m,n", [(7, 16, 32), (8, 48, 96)])
def test_against_reference_shaped_oracle(ctx, seed, m, n):
    """North star: basis indices exact, objective / vertex within 1e-10 relative of the path
    that recomputes Binv by full-pivot LU every iteration (SimplexSolover.h:446)."""
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r = o.simplex_reference(A, b, c, basis, True, n - m, trace_cap=4096)
    g = ctx.simplex_solve(A, b, c, basis, True, n - m)
    assert g["status"] == r["status"] == 0 and g["iters"] == r["iters"]
    assert np.array_equal(g["basis"], r["basis"])
    np.testing.assert_allclose(g["x"], r["x"], rtol=1e-10, atol=1e-12)
    assert abs(g["obj"] - r["obj"]) <= 1e-10 * abs(r["obj"])


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
@pytest.mark.parametrize("maximize", [True, False])
@pytest.mark.parametrize("algo", ALGOS)
def test_general_basis_crash_and_minimise(ctx, seed, maximize, algo):
    """Non-slack initial basis (computeBFS, SimplexSolover.h:423) and the minimise rules
    (:163-174)."""
    A, b, c, basis = lpcases.general_lp(seed, 9, 20)
    n = A.shape[1]
    r = o.simplex_tableau(A, b, c, basis, maximize, n, trace_cap=4096, want_tableau=True)
    g = _run(ctx, A, b, c, basis, maximize, n, algo=algo)
    assert r["status"] == o.OPTIMAL
    _assert_bit_exact(g, r)


def test_status_codes(ctx):
    A = np.array([[1.0, -1.0, 1.0]])
    assert ctx.simplex_solve(A, [1.0], [1.0, 1.0, 0.0], [2], True, 2)["status"] == capi.UNBOUNDED
    A = np.array([[4, 3, 0, 1], [0, 4, 0, 4.0]])   # main.cpp:24-34 (commented-out LP), basis {0,2}
    assert ctx.simplex_solve(A, [4, 6.0], [5, 1, 0, 0.0], [0, 2], False, 4)["status"] == capi.SINGULAR
    A, b, c, basis = lpcases.random_lp(5, 16, 32)
    for algo in ALGOS:
        for lim in (1, 3, 16, 17):
            r = o.simplex_tableau(A, b, c, basis, True, 16, max_iter=lim, trace_cap=64, want_tableau=True)
            g = _run(ctx, A, b, c, basis, True, 16, max_iter=lim, algo=algo)
            _assert_bit_exact(g, r)
        A1 = np.array([[1.0, -1.0, 1.0]])
        p = ctx.simplex_problem(A1, [1.0], [1.0, 1.0, 0.0], [2], True, 2)
        assert p.run(algo=algo)[0] == capi.UNBOUNDED
        p.free()
    with pytest.raises(capi.LPError):
        ctx.simplex_solve(A, b, c, [0] * 15 + [99], True, 16)
    with pytest.raises(capi.LPError):
        ctx.simplex_solve(A, b, c, basis, True, 0)


def test_degenerate_ties(ctx):
    """Degenerate vertex: several ratios tie at 0 — the scan must keep the FIRST
    (SimplexSolover.h:187 uses r < theta - EPS)."""
    A = np.array([[1, 1, 1, 0, 0], [1, 2, 0, 1, 0], [2, 1, 0, 0, 1.0]])
    b = np.array([0.0, 0.0, 4.0])
    c = np.array([3, 2, 0, 0, 0.0])
    basis = np.array([2, 3, 4], dtype=np.int32)
    r = o.simplex_tableau(A, b, c, basis, True, 2, trace_cap=64, want_tableau=True)
    for algo in ALGOS:
        g = _run(ctx, A, b, c, basis, True, 2, algo=algo)
        _assert_bit_exact(g, r)


def test_reset_and_repeat(ctx):
    A, b, c, basis = lpcases.random_lp(9, 40, 90)
    p = ctx.simplex_problem(A, b, c, basis, True, 50)
    rc1, s1 = p.run()
    d1 = p.download()
    p.reset()
    rc2, s2 = p.run()
    d2 = p.download()
    assert rc1 == rc2 == 0 and s1.pivots == s2.pivots
    assert np.array_equal(d1["x"], d2["x"]) and np.array_equal(d1["basis"], d2["basis"])
    p.reset()
    ms = p.bench_update(0, 0, 10)   # A[0,0] ~ U(0,1): a valid pivot element
    assert ms > 0
    p.free()


def test_golden_vectors_gpu(ctx):
    """tests/golden/simplex_cases.json (restatement-derived, scipy-checked; see make_golden.py)."""
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "simplex_cases.json")
    for case in json.load(open(path)):
        A, b, c, basis = lpcases.random_lp(case["seed"], case["m"], case["n"])
        g = ctx.simplex_solve(A, b, c, basis, True, case["n"] - case["m"])
        assert g["status"] == 0 and g["iters"] == case["iters"]
        assert g["basis"].tolist() == case["basis"] and g["obj"] == case["obj"]
        for j, v in case["x_nonzero"].items():
            assert g["x"][int(j)] == v


def test_config1_against_reference_shaped_golden(ctx):
    """BASELINE configs[1] (m=512, n=1024, seed 0) against the committed outputs of the
    reference-SHAPED restatement (Binv recomputed by full-pivot LU every iteration,
    /root/reference/src/SimplexSolover.h:429-447; generated once by tests/golden/make_golden.py,
    ~20 s of CPU): pivot trace and basis exactly, vertex and objective within the north star's
    1e-10 relative — on every device algorithm."""
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "simplex_cases.json")
    case = [c for c in json.load(open(path)) if (c["m"], c["n"]) == (512, 1024)][0]
    ref = case["reference_shaped"]
    A, b, c, basis = lpcases.random_lp(case["seed"], 512, 1024)
    xref = np.zeros(512)
    for j, v in ref["x_nonzero"].items():
        xref[int(j)] = v
    for algo in ALGOS + [capi.SIMPLEX_AUTO]:
        g = _run(ctx, A, b, c, basis, True, 512, algo=algo)
        assert g["status"] == 0 and g["iters"] == ref["iters"] == 345
        trace = [list(t) for t in zip(g["trace_enter"][:345].tolist(), g["trace_leave"][:345].tolist())]
        assert trace == ref["trace"]
        assert g["basis"].tolist() == ref["basis"]
        assert abs(g["obj"] - ref["obj"]) <= 1e-10 * abs(ref["obj"])
        scale = np.abs(xref).max()
        assert np.abs(g["x"] - xref).max() <= 1e-10 * scale
        nz = xref != 0
        assert (np.abs(g["x"][nz] - xref[nz]) <= 1e-10 * np.abs(xref[nz]) + 1e-14 * scale).all()


def test_resident_solves_under_contention(ctx):
    """The chip-resident kernel's workgroups wait for each other inside one launch, so they must all
    become resident and every hand-off must survive an uneven, busy chip: four host threads, each
    with a context (stream) of its own, solve different LPs at the same time on the one device —
    their launches compete for the same XCD's CUs and L2 — while the main thread runs enumeration
    passes underneath.  Every solve must be the chip-resident one (no silent fallback: the launch
    count says so) and bit-exact."""
    import threading
    cases = [(31, 256, 512), (32, 200, 700), (33, 512, 1024), (34, 96, 2000)]
    refs, outs = {}, {}
    for seed, m, n in cases:
        A, b, c, basis = lpcases.random_lp(seed, m, n)
        refs[seed] = (A, b, c, basis, o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14))

    def run(seed, m, n):
        cx = capi.Context(0)
        A, b, c, basis, _ = refs[seed]
        res = []
        for _ in range(6):
            p = cx.simplex_problem(A, b, c, basis, True, n - m)
            rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
            d = p.download(trace_cap=max(st.pivots, 1))
            p.free()
            res.append((rc, st.pivots, st.launches, d))
        cx.close()
        outs[seed] = res

    th = [threading.Thread(target=run, args=cs) for cs in cases]
    for t in th:
        t.start()
    Ae, be, ce, _ = lpcases.random_lp(0, 14, 28)
    ep = ctx.enum_problem(Ae, be, ce, True)
    for _ in range(20):
        ep.range(0, ep.total)
    ep.free()
    for t in th:
        t.join()
    for seed, m, n in cases:
        r = refs[seed][4]
        for rc, pivots, launches, d in outs[seed]:
            assert rc == r["status"] == 0 and pivots == r["iters"]
            assert launches == 2, "the solve fell back to the launch-based path"
            assert np.array_equal(d["basis"], r["basis"]) and np.array_equal(d["x"], r["x"]) and d["obj"] == r["obj"]
            k = r["iters"]
            assert list(zip(d["trace_enter"][:k].tolist(), d["trace_leave"][:k].tolist())) == r["trace"][:k]


def test_midrange_division_matches_division(ctx):
    """The chip-resident kernel's quotients by the pivot element (lpdev::mid_div: the division's own instruction
    sequence without the range scaling, the reciprocal refined once per denominator) against the device's plain
    division, which is IEEE's: 4 M operand pairs — random mantissas and exponents over the whole range the fast
    sequence is applied to, edge mantissas, quotients of equal operands, signed zero numerators, and operands
    outside the range (where the kernel's rule takes the plain division)."""
    rng = np.random.default_rng(2027)
    n = 1 << 22

    def operands(lo, hi):
        mant = rng.integers(0, 1 << 52, size=n, dtype=np.uint64)
        edge = np.array([0, 1, 2, 3, (1 << 52) - 1, (1 << 52) - 2, 1 << 51, (1 << 51) + 1, (1 << 51) - 1,
                         0x5555555555555, 0xAAAAAAAAAAAAA, 0x6A09E667F3BCD, 0x6A09E667F3BCC], dtype=np.uint64)
        mant[: 64 * len(edge)] = np.tile(edge, 64)
        expo = rng.integers(lo, hi + 1, size=n).astype(np.int64)
        v = ((expo + 1023).astype(np.uint64) << np.uint64(52) | mant).view(np.float64).copy()
        v[rng.random(n) < 0.5] *= -1.0
        return v
    num, den = operands(-500, 500), operands(-500, 500)
    rng.shuffle(den)
    num[:4096:2] = den[:4096:2]                 # equal operands: the quotient is exactly 1
    num[4096:8192:4] = 0.0                      # zero numerators of both signs
    num[4097:8192:4] = -0.0
    num[8192:9216] = operands(-1022, 1023)[:1024]     # outside the range: the plain division answers
    den[9216:10240] = operands(-1022, 1023)[:1024]
    num[10240:10250] = [np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1.7e308, 2.0 ** 501, 2.0 ** -501, 2.0 ** 500, 2.0 ** -500]
    fast, plain = ctx.debug_division(num, den)
    with np.errstate(all="ignore"):
        host = num / den
    same = (plain.view(np.uint64) == host.view(np.uint64)) | (np.isnan(plain) & np.isnan(host))
    assert same.all()                                 # the device's division is IEEE's
    bad = np.flatnonzero((fast.view(np.uint64) != plain.view(np.uint64)) & ~(np.isnan(fast) & np.isnan(plain)))
    assert bad.size == 0, (num[bad[:4]], den[bad[:4]], fast[bad[:4]], plain[bad[:4]])


def test_update_microbenchmarks_and_profiling(ctx):
    """The measurement hooks bench.py uses leave the problem intact."""
    A, b, c, basis = lpcases.random_lp(3, 64, 160)
    r = o.simplex_tableau(A, b, c, basis, True, 96)
    p = ctx.simplex_problem(A, b, c, basis, True, 96)
    ms1 = p.bench_update(0, 0, 20)
    msj, j = p.bench_update_rankj(20)
    assert ms1 > 0 and msj > 0 and 1 <= j <= 16
    p.reset()
    p.profile(True)
    rc, st = p.run(algo=capi.SIMPLEX_LOOKAHEAD)
    assert rc == 0 and st.update_launches > 0 and st.update_ms > 0
    p.profile(False)
    d = p.download()
    assert st.pivots == r["iters"] and np.array_equal(d["x"], r["x"]) and np.array_equal(d["basis"], r["basis"])
    p.reset()
    rc, st = p.run(algo=capi.SIMPLEX_LOOKAHEAD)
    assert st.update_launches == 0 and st.pivots == r["iters"]
    p.free()


def test_fuzz_small_problems(ctx):
    """200 small random LPs — signed data, both senses, slack and general starting bases, tight
    iteration limits: status, pivot sequence, basis and vertex bit-exact against the oracle on
    both device algorithms."""
    rng = np.random.default_rng(2024)
    for trial in range(200):
        m = int(rng.integers(1, 7))
        extra = int(rng.integers(0, 7))
        n = m + extra
        maximize = bool(rng.integers(0, 2))
        max_iter = int(rng.choice([1, 2, 5, 10000]))
        if trial % 3 == 0 and extra > 0:
            A, b, c, basis = lpcases.general_lp(1000 + trial, m, n, signed=True)   # (m+1) x (n+1)
            n = A.shape[1]
        else:
            A = np.hstack([rng.uniform(-1, 1, (m, extra)), np.eye(m)])
            b = rng.uniform(0.1, 2.0, m)
            c = np.concatenate([rng.uniform(-1, 1, extra), np.zeros(m)])
            basis = np.arange(extra, n, dtype=np.int32)
        r = o.simplex_tableau(A, b, c, basis, maximize, n, max_iter=max_iter, trace_cap=64, want_tableau=True)
        for algo in ALGOS:
            g = _run(ctx, A, b, c, basis, maximize, n, trace_cap=64, max_iter=max_iter, algo=algo)
            _assert_bit_exact(g, r)
