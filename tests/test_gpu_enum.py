"""GPU parity: vertex enumeration (through the C ABI) against the oracle."""
import numpy as np
import pytest

from oracle import pyoracle as o
from simplexmethod_amd import capi
from tests import lpcases

pytestmark = pytest.mark.gpu

ALGOS = [capi.ENUM_DIRECT]


def test_input_symmetric_table(ctx):
    A, b, c, _, no = lpcases.input_symmetric_lp()
    r = ctx.enum_solve(A, b, c, True, no)
    assert r["status"] == 0 and r["rank"] == 2 and r["basis"].tolist() == [0, 3]
    assert r["x"].tolist() == [5, 0, 0] and r["obj"] == 35 and r["counts"] == [7, 3, 0]
    p = ctx.enum_problem(A, b, c, True)
    for k in range(10):
        st, xB, z = o.enum_subset(A, b, c, o.unrank(5, 2, k))
        v = p.vertex(k)
        assert v["verdict"] == st and v["basis"].tolist() == o.unrank(5, 2, k).tolist()
        if st != o.SUBSET_SINGULAR:
            assert np.array_equal(v["x"][v["basis"]], xB) and v["obj"] == z
    p.free()


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("seed,m,n", [(1, 1, 4), (2, 3, 7), (3, 5, 12), (4, 8, 16), (5, 10, 20),
                                       (6, 14, 18), (7, 16, 19), (8, 17, 20), (9, 6, 6)])
def test_enum_matches_oracle(ctx, algo, seed, m, n):
    A, b, c, _ = lpcases.random_lp(seed, m, n) if n > m else (np.eye(m) * 2, np.ones(m), np.ones(m), None)
    total = o.binom(n, m)
    st, z, counts = o.enum_range(A, b, c, True, 0, total)
    p = ctx.enum_problem(A, b, c, True)
    rc, gz, gcounts, stats = p.range(0, total, algo)
    assert rc == st and gcounts == counts and sum(gcounts) == total
    if st == o.OPTIMAL:
        assert gz == z                                       # bit for bit
        k = o.enum_first_within(A, b, c, True, 0, total, z)
        assert p.first_within(0, total, z) == k
        v = p.vertex(k)
        _, xB, zz = o.enum_subset(A, b, c, o.unrank(n, m, k))
        assert v["obj"] == zz and np.array_equal(v["x"][v["basis"]], xB)
    p.free()


@pytest.mark.parametrize("algo", ALGOS)
def test_minimise_and_signed_data(ctx, algo):
    rng = np.random.default_rng(5)
    m, n = 6, 14
    A = rng.normal(size=(m, n))
    b = rng.normal(size=m)
    c = rng.normal(size=n)
    total = o.binom(n, m)
    for maximize in (True, False):
        st, z, counts = o.enum_range(A, b, c, maximize, 0, total)
        p = ctx.enum_problem(A, b, c, maximize)
        rc, gz, gcounts, _ = p.range(0, total, algo)
        assert rc == st and gcounts == counts and gz == z
        assert p.first_within(0, total, z) == o.enum_first_within(A, b, c, maximize, 0, total, z)
        p.free()


@pytest.mark.parametrize("algo", ALGOS)
def test_singular_subsets_counted(ctx, algo):
    # duplicate and zero columns -> many singular bases
    rng = np.random.default_rng(9)
    m, n = 4, 10
    A = rng.uniform(size=(m, n))
    A[:, 3] = A[:, 1]
    A[:, 7] = 0.0
    b = rng.uniform(1, 2, size=m)
    c = rng.uniform(size=n)
    total = o.binom(n, m)
    st, z, counts = o.enum_range(A, b, c, True, 0, total)
    assert counts[2] > 0
    p = ctx.enum_problem(A, b, c, True)
    rc, gz, gcounts, _ = p.range(0, total, algo)
    assert gcounts == counts and gz == z
    p.free()


@pytest.mark.parametrize("algo", ALGOS)
def test_shards_compose(ctx, algo):
    """The rank space shards across GPUs: any split gives the same answer."""
    A, b, c, _ = lpcases.random_lp(12, 7, 16)
    total = o.binom(16, 7)
    p = ctx.enum_problem(A, b, c, True)
    _, z, counts, _ = p.range(0, total, algo)
    for parts in (2, 3, 8):
        cuts = [total * k // parts for k in range(parts + 1)]
        zs, cs, firsts = [], np.zeros(3, dtype=np.int64), []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            _, zz, cc, _ = p.range(lo, hi, algo)
            zs.append(zz)
            cs += cc
        assert max(zs) == z and cs.tolist() == counts
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            p.range(lo, hi, algo)
            firsts.append(p.first_within(lo, hi, z))
        assert min(firsts) == o.enum_first_within(A, b, c, True, 0, total, z)
    # empty shard
    rc, zz, cc, _ = p.range(5, 5, algo)
    assert rc == capi.INFEASIBLE and cc == [0, 0, 0] and zz == -np.inf
    p.free()


def test_enumeration_agrees_with_simplex_on_gpu(ctx):
    """README.md:42 — cross-check of the two solvers, both on the GPU."""
    for seed, m, n in [(31, 4, 9), (32, 6, 13), (33, 8, 17)]:
        A, b, c, basis = lpcases.random_lp(seed, m, n)
        s = ctx.simplex_solve(A, b, c, basis, True, n - m)
        e = ctx.enum_solve(A, b, c, True, n - m)
        assert s["status"] == e["status"] == 0
        assert abs(e["obj"] - s["obj"]) <= 1e-10 * abs(s["obj"])
        np.testing.assert_allclose(e["x"], s["x"], rtol=1e-9, atol=1e-10)
        assert sorted(s["basis"].tolist()) == e["basis"].tolist()


def test_bad_arguments(ctx):
    A, b, c, _ = lpcases.random_lp(1, 3, 7)
    p = ctx.enum_problem(A, b, c, True)
    with pytest.raises(capi.LPError):
        p.range(0, p.total + 1)
    with pytest.raises(capi.LPError):
        p.vertex(p.total)
    p.free()
    with pytest.raises(capi.LPError):
        ctx.enum_problem(np.ones((40, 70)), np.ones(40), np.ones(70))


def test_golden_vectors_gpu(ctx):
    """tests/golden/enum_cases.json (restatement-derived; see make_golden.py)."""
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "enum_cases.json")
    enum = json.load(open(path))
    for case in enum["random"]:
        A, b, c, _ = lpcases.random_lp(case["seed"], case["m"], case["n"])
        e = ctx.enum_solve(A, b, c, True, case["n"] - case["m"])
        assert e["rank"] == case["rank"] and e["basis"].tolist() == case["basis"]
        assert e["obj"] == case["obj"] and e["counts"] == case["counts"] and e["x"].tolist() == case["x"]


# ---- shared-prefix path (enum_prefix.hip): same answers as the oracle, bit for bit ------------

PREFIX_SHAPES = [(6, 12, 51), (7, 16, 52), (8, 16, 53), (10, 20, 54), (6, 20, 55), (8, 24, 56),
                 (12, 20, 57), (16, 20, 58), (9, 25, 59), (11, 23, 60), (15, 22, 61), (7, 9, 62)]


@pytest.mark.parametrize("m,n,seed", PREFIX_SHAPES)
def test_prefix_matches_oracle(ctx, m, n, seed):
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    total = o.binom(n, m)
    st, z, counts = o.enum_range(A, b, c, True, 0, total)
    p = ctx.enum_problem(A, b, c, True)
    rc, gz, gcounts, stats = p.range(0, total, capi.ENUM_PREFIX)
    assert gcounts == counts and sum(gcounts) == total
    assert rc == st and gz == z
    k = o.enum_first_within(A, b, c, True, 0, total, z)
    assert p.first_within(0, total, z) == k
    # a wider tolerance does not go through the tie rule cached by the range pass
    assert p.first_within(0, total, z, 1e-2 * abs(z)) == o.enum_first_within(A, b, c, True, 0, total, z, 1e-2 * abs(z))
    v = p.vertex(k)
    _, xB, zz = o.enum_subset(A, b, c, o.unrank(n, m, k))
    assert v["obj"] == zz and np.array_equal(v["x"][v["basis"]], xB)
    # and the direct kernel agrees with the prefix kernel
    rc2, dz, dcounts, _ = p.range(0, total, capi.ENUM_DIRECT)
    assert (rc2, dz, dcounts) == (rc, gz, gcounts)
    p.free()


# ---- wide shapes: 32-row records (m > 16) and more than 16 selectable columns (n - m > 16) run the
# general leaf kernel over the same depth m-7 records

WIDE_SHAPES = [(17, 23, 81), (18, 24, 82), (20, 25, 83), (8, 26, 84), (7, 30, 85), (9, 27, 86),
               (24, 28, 87), (17, 20, 88), (7, 39, 89), (7, 46, 92)]


@pytest.mark.parametrize("m,n,seed", WIDE_SHAPES)
def test_wide_prefix_matches_oracle(ctx, m, n, seed):
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    total = o.binom(n, m)
    p = ctx.enum_problem(A, b, c, True)
    # windows of the rank space against the oracle (the whole space when it is small)
    windows = [(0, total)] if total <= 400_000 else [(0, 60_000), (total // 3, total // 3 + 60_000), (total - 60_000, total)]
    for lo, hi in windows:
        st, z, counts = o.enum_range(A, b, c, True, lo, hi)
        rc, gz, gcounts, _ = p.range(lo, hi, capi.ENUM_PREFIX)
        assert (rc, gz, gcounts) == (st, z, counts), (m, n, lo, hi)
        if st == 0:
            assert p.first_within(lo, hi, z) == o.enum_first_within(A, b, c, True, lo, hi, z)
    # the whole space: shared-prefix path against the direct kernel (itself oracle-checked above and in
    # test_direct_*), optimum, counts, tie rule and the vertex
    rc, gz, gcounts, _ = p.range(0, total, capi.ENUM_PREFIX)
    assert sum(gcounts) == total
    k = p.first_within(0, total, gz) if rc == 0 else None
    rc2, dz, dcounts, _ = p.range(0, total, capi.ENUM_DIRECT)
    assert (rc2, dz, dcounts) == (rc, gz, gcounts)
    if rc == 0:
        assert p.first_within(0, total, dz) == k
        v = p.vertex(k)
        _, xB, zz = o.enum_subset(A, b, c, o.unrank(n, m, k))
        assert v["obj"] == zz == gz and np.array_equal(v["x"][v["basis"]], xB)
    p.free()


@pytest.mark.parametrize("m,n,seed", [(9, 64, 93), (16, 64, 94), (32, 64, 95), (20, 52, 96)])
def test_widest_shapes_windows(ctx, m, n, seed):
    """The largest shapes the library takes (n = 64: up to 57 selectable columns on 16-row records; m = 32
    rows; C(64,32) = 1.8e18 ranks): windows of the rank space against the oracle, on the shared-prefix path
    and on the direct kernel."""
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    total = o.binom(n, m)
    p = ctx.enum_problem(A, b, c, True)
    w = 20_000 if m >= 20 else 50_000
    for lo in (0, total // 3, total - w):
        ref = o.enum_range(A, b, c, True, lo, lo + w)
        assert p.range(lo, lo + w, capi.ENUM_PREFIX)[:3] == ref, (m, n, lo)
        assert p.range(lo, lo + w, capi.ENUM_DIRECT)[:3] == ref, (m, n, lo)
        if ref[0] == 0:
            assert p.first_within(lo, lo + w, ref[1]) == o.enum_first_within(A, b, c, True, lo, lo + w, ref[1])
    p.free()


def test_wide_range_is_split_when_the_level_buffers_are_small(ctx, monkeypatch):
    """Large shapes have more depth m-7 nodes than memory holds (C(n-7, m-7) records): the range is then
    enumerated in sub-ranges whose nodes fit, each with its own list, and pass 2 re-runs only the
    sub-range that holds the winner.  Forced here with a 64 KB budget for the level buffers (8-13 records)."""
    monkeypatch.setenv("LP_ENUM_LEVEL_BUDGET_KB", "64")
    for m, n, seed in [(18, 24, 82), (8, 26, 84), (11, 21, 91)]:   # 32-row records, general kernel, tuned kernels
        A, b, c, _ = lpcases.random_lp(seed, m, n)
        total = o.binom(n, m)
        lo, hi = 0, total
        ref = o.enum_range(A, b, c, True, lo, hi)
        p = ctx.enum_problem(A, b, c, True)
        got = p.range(lo, hi, capi.ENUM_PREFIX)
        assert got[:3] == ref, (m, n)
        assert got[3].launches > 30          # several sub-ranges ran
        assert p.first_within(lo, hi, ref[1]) == o.enum_first_within(A, b, c, True, lo, hi, ref[1])
        assert p.first_within(lo, hi, ref[1], 1e-3) == o.enum_first_within(A, b, c, True, lo, hi, ref[1], 1e-3)
        p.free()


def test_prefix_shards_and_minimise(ctx):
    rng = np.random.default_rng(77)
    m, n = 7, 17
    A = rng.normal(size=(m, n))
    b = rng.normal(size=m)
    c = rng.normal(size=n)
    total = o.binom(n, m)
    for maximize in (True, False):
        st, z, counts = o.enum_range(A, b, c, maximize, 0, total)
        p = ctx.enum_problem(A, b, c, maximize)
        rc, gz, gcounts, _ = p.range(0, total, capi.ENUM_PREFIX)
        assert (rc, gz, gcounts) == (st, z, counts)
        for parts in (2, 3, 8):
            cuts = [total * k // parts for k in range(parts + 1)]
            zs, cs, firsts = [], np.zeros(3, dtype=np.int64), []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                r1, zz, cc, _ = p.range(lo, hi, capi.ENUM_PREFIX)
                ro, zo, co = o.enum_range(A, b, c, maximize, lo, hi)
                assert (r1, zz, cc) == (ro, zo, co)
                zs.append(zz)
                cs += cc
                firsts.append(p.first_within(lo, hi, z))
            assert cs.tolist() == counts
            assert min(firsts) == o.enum_first_within(A, b, c, maximize, 0, total, z)
        p.free()


def test_prefix_singular_subtrees(ctx):
    rng = np.random.default_rng(9)
    m, n = 6, 14
    A = rng.uniform(size=(m, n))
    A[:, 3] = A[:, 1]          # duplicate column: every subset holding both is singular
    A[:, 7] = 0.0              # zero column
    A[:, 12] = 2.0 * A[:, 10]  # and late in the subset, to hit the 2x2 block
    b = rng.uniform(1, 2, size=m)
    c = rng.uniform(size=n)
    total = o.binom(n, m)
    st, z, counts = o.enum_range(A, b, c, True, 0, total)
    assert counts[2] > 0
    p = ctx.enum_problem(A, b, c, True)
    rc, gz, gcounts, _ = p.range(0, total, capi.ENUM_PREFIX)
    assert (rc, gz, gcounts) == (st, z, counts)
    p.free()


@pytest.mark.parametrize("m,n", [(8, 24), (9, 22), (7, 23)])
def test_prefix_singular_pivots_at_every_stage(ctx, m, n):
    """Duplicate / zero / proportional columns placed so that a singular pivot turns up in the
    breadth-first levels (holes), in the leaf kernels' in-LDS pivots (child and group), in the
    lanes' own steps, in the 2x2 block and in the thin kernel — whole ranges and shards."""
    A, b, c, _ = lpcases.random_lp(100 + n, m, n)   # [A0 | I]: feasible bases exist
    A[:, 1] = A[:, 0]                 # pruned at depth 2: a hole high in the tree
    A[:, m - 3] = 3.0 * A[:, m - 5]   # singular around the depth m-7 .. m-5 pivots
    A[:, n // 2] = 0.0                # zero column in the middle
    A[:, n - 2] = A[:, n - 5]         # late: the lanes' own steps / the thin kernel's tails
    A[:, n - 1] = 0.5 * A[:, n - 3]   # and the 2x2 block
    total = o.binom(n, m)
    st, z, counts = o.enum_range(A, b, c, True, 0, total)
    assert counts[2] > total // 4 and counts[0] > 0
    p = ctx.enum_problem(A, b, c, True)
    rc, gz, gcounts, _ = p.range(0, total, capi.ENUM_PREFIX)
    assert (rc, gz, gcounts) == (st, z, counts)
    assert p.first_within(0, total, z) == o.enum_first_within(A, b, c, True, 0, total, z)
    cuts = [0, total // 7, total // 3, total // 2 + 11, total - 1000, total]
    acc = np.zeros(3, dtype=np.int64)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        r1, zz, cc, _ = p.range(lo, hi, capi.ENUM_PREFIX)
        assert (r1, zz, cc) == o.enum_range(A, b, c, True, lo, hi)
        acc += cc
    assert acc.tolist() == counts
    p.free()


def test_prefix_all_singular(ctx):
    """Two equal rows: (nearly) no subset is regular — whole subtrees are cut as soon as a level sees
    it; the few subsets whose cancellation leaves a pivot above the threshold are classified by
    the same arithmetic on both sides."""
    rng = np.random.default_rng(5)
    m, n = 8, 22
    A = rng.uniform(size=(m, n))
    A[5] = A[2]
    b = rng.uniform(1, 2, size=m)
    c = rng.uniform(size=n)
    total = o.binom(n, m)
    p = ctx.enum_problem(A, b, c, True)
    rc, gz, gcounts, _ = p.range(0, total, capi.ENUM_PREFIX)
    ref = o.enum_range(A, b, c, True, 0, total)
    assert (rc, gz, gcounts) == ref and gcounts[2] > 0.99 * total
    p.free()


@pytest.mark.parametrize("m,n", [(8, 20), (9, 24), (17, 21), (7, 26)])   # the last two: general leaf kernel
def test_prefix_nan_and_inf_entries(ctx, m, n):
    """NaN / inf in the data: a NaN is never a pivot maximum, and a subset whose solution holds a
    NaN counts as infeasible (Canonical.cpp:171 compares x >= -1e-9) — same verdicts, subset by
    subset, as the oracle, on both kernels."""
    A, b, c, _ = lpcases.random_lp(300 + n, m, n)
    A[2, 1] = np.nan
    A[5, n // 2] = np.inf
    A[m - 1, n - 3] = -np.inf
    A[0, 3] = np.nan
    b[m - 2] = np.nan
    total = o.binom(n, m)
    st, z, counts = o.enum_range(A, b, c, True, 0, total)
    p = ctx.enum_problem(A, b, c, True)
    for algo in (capi.ENUM_PREFIX, capi.ENUM_DIRECT):
        rc, gz, gcounts, _ = p.range(0, total, algo)
        assert gcounts == counts and rc == st
        assert gz == z or (np.isnan(gz) and np.isnan(z))
    p.free()


@pytest.mark.parametrize("m,n", [(10, 12), (12, 15), (16, 18)])
def test_prefix_narrowest_trees_and_tiny_ranges(ctx, m, n):
    """n - m = 2, 3: every node has two or three children; plus empty and one-subset ranges."""
    A, b, c, _ = lpcases.random_lp(400 + n, m, n)
    total = o.binom(n, m)
    p = ctx.enum_problem(A, b, c, True)
    ref = o.enum_range(A, b, c, True, 0, total)
    assert p.range(0, total, capi.ENUM_PREFIX)[:3] == ref
    assert p.range(0, total, capi.ENUM_DIRECT)[:3] == ref
    for lo, hi in [(0, 1), (total - 1, total), (total // 2, total // 2 + 1), (3, 3), (total // 3, total // 3 + 7)]:
        got = p.range(lo, hi, capi.ENUM_PREFIX)[:3]
        want = o.enum_range(A, b, c, True, lo, hi) if hi > lo else (o.INFEASIBLE, -np.inf, [0, 0, 0])
        assert got == want, (lo, hi)
        assert p.range(lo, hi, capi.ENUM_DIRECT)[:3] == want, (lo, hi)
    p.free()


def test_prefix_rejects_unsupported_shapes(ctx):
    A, b, c, _ = lpcases.random_lp(1, 3, 7)
    p = ctx.enum_problem(A, b, c, True)
    with pytest.raises(capi.LPError):
        p.range(0, p.total, capi.ENUM_PREFIX)
    p.free()


@pytest.mark.parametrize("m,n", [(18, 30), (12, 32), (20, 30)])
def test_wide_full_size_properties(ctx, m, n):
    """Wide shapes at sizes the oracle cannot walk (86 M, 226 M and 30 M subsets): counts add up to C(n,m);
    the optimum is the simplex optimum (a different algorithm on the same GPU); uneven shards compose to
    the whole; the sharded entry point gives the same winner; the oracle agrees on a window around the
    optimum."""
    A, b, c, basis = lpcases.random_lp(3, m, n)
    p = ctx.enum_problem(A, b, c, True)
    total = p.total
    rc, z, counts, _ = p.range(0, total)
    assert rc == 0 and sum(counts) == total and counts[0] > 0
    k = p.first_within(0, total, z)
    v = p.vertex(k, n - m)
    assert v["obj"] == z
    s = ctx.simplex_solve(A, b, c, basis, True, n - m)
    assert s["status"] == 0 and abs(s["obj"] - z) <= 1e-10 * abs(z)
    np.testing.assert_allclose(v["x"], s["x"], rtol=0, atol=1e-9)
    cuts = [0, total // 11, total // 3 + 7, total - total // 5, total]
    acc, bests, firsts = np.zeros(3, dtype=np.int64), [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        r1, zz, cc, _ = p.range(lo, hi)
        acc += cc
        bests.append(zz if r1 == 0 else -np.inf)
        firsts.append(p.first_within(lo, hi, z) if r1 == 0 and zz >= z - 1e-9 else 2 ** 64 - 1)
    assert acc.tolist() == counts and max(bests) == z and min(firsts) == k
    g = p.solve_sharded(None, n - m)
    assert g["rank"] == k and g["obj"] == z and g["counts"] == counts
    lo = max(0, k - 20_000)
    hi = min(total, k + 20_000)
    ref = o.enum_range(A, b, c, True, lo, hi)
    assert p.range(lo, hi, capi.ENUM_PREFIX)[:3] == ref and ref[1] == z
    p.free()


# ---- BASELINE.json's full sizes: size-independent properties -----------------------------------

@pytest.mark.parametrize("m,n", [(14, 28), (16, 32)])   # configs[2], configs[3]
def test_full_size_properties(ctx, m, n):
    """C(28,14) = 40,116,600 and C(32,16) = 601,080,390 subsets are too many for the oracle, so:
    the three counts add up to C(n,m); the optimum is the simplex optimum (a different algorithm
    on the same GPU); uneven shards compose to the whole (counts add, best = max, tie rule = min);
    the direct kernel agrees with the shared-prefix kernels on a 20 M window; the oracle agrees on
    a 200 k window around the optimum."""
    A, b, c, basis = lpcases.random_lp(0, m, n)
    p = ctx.enum_problem(A, b, c, True)
    total = p.total
    assert total == o.binom(n, m)
    rc, z, counts, _ = p.range(0, total)
    assert rc == 0 and sum(counts) == total and counts[0] > 0
    k = p.first_within(0, total, z)
    v = p.vertex(k, n - m)
    assert v["obj"] == z
    s = ctx.simplex_solve(A, b, c, basis, True, n - m)
    assert s["status"] == 0 and abs(s["obj"] - z) <= 1e-10 * abs(z)
    np.testing.assert_allclose(v["x"], s["x"], rtol=0, atol=1e-9)
    # uneven shards
    cuts = [0, total // 11, total // 3 + 7, total // 2, total - total // 5, total - 12345, total]
    acc, bests, firsts = np.zeros(3, dtype=np.int64), [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        r1, zz, cc, _ = p.range(lo, hi)
        acc += cc
        bests.append(zz if r1 == 0 else -np.inf)
        firsts.append(p.first_within(lo, hi, z) if r1 == 0 and zz >= z - 1e-9 else 2 ** 64 - 1)
    assert acc.tolist() == counts and max(bests) == z and min(firsts) == k
    # direct kernel on a window
    lo = total // 2
    hi = min(total, lo + 20_000_000)
    assert p.range(lo, hi, capi.ENUM_DIRECT)[:3] == p.range(lo, hi, capi.ENUM_PREFIX)[:3]
    # the oracle on a window around the optimum
    lo = max(0, k - 100_000)
    hi = min(total, k + 100_000)
    st, oz, ocounts = o.enum_range(A, b, c, True, lo, hi)
    r1, gz, gcounts, _ = p.range(lo, hi)
    assert (r1, gz, gcounts) == (st, oz, ocounts) and gz == z
    p.free()


@pytest.mark.parametrize("m,n", [(14, 28), (16, 32)])   # configs[2], configs[3]
def test_vertices_against_the_reference_pinned_qr_solve(ctx, m, n):
    """The GPU's per-subset arithmetic (Gauss-Jordan + 2x2 block, build-defined) against the ONE
    per-basis solve the reference pins with a fixture of its own: Canonical::GetBasicSolution's
    ColPivHouseholderQR (/root/reference/src/ProblemTypes/Canonical.cpp:179-197, fixture
    /root/reference/tests/test_canonical.cpp:41-66 -> oracle orc_basic_solution, checked in
    tests/test_oracle.py::test_reference_fixture_basic_solution).  The winning vertex and ~1000
    feasible ranks spread over the whole rank space (first feasible rank of each of 1000 equal
    windows): every coordinate within the north star's 1e-10 relative, objective likewise, and the
    reference's feasibility test (IsFeasibleBasis, Canonical.cpp:165-177) agrees."""
    A, b, c, _ = lpcases.random_lp(0, m, n)
    p = ctx.enum_problem(A, b, c, True)
    total = p.total
    rc, z, counts, _ = p.range(0, total)
    assert rc == 0
    ranks = [p.first_within(0, total, z)]
    windows = 1000
    W = total // windows
    none = 2 ** 64 - 1
    for i in range(windows):
        r = p.first_within(i * W, (i + 1) * W if i + 1 < windows else total, -np.inf)
        if r != none:
            ranks.append(r)
    assert len(ranks) > min(windows, counts[0]) // 3      # feasible bases are spread out
    worst = 0.0
    for r in ranks:
        v = p.vertex(r, n)
        assert v["verdict"] == capi.SUBSET_FEASIBLE
        basis = v["basis"]
        assert o.rank_of(n, basis) == r
        st, xq = o.basic_solution(A, b, basis)
        assert st == 0 and o.is_feasible_basis(A, b, basis)
        scale = np.abs(xq).max()
        err = np.abs(v["x"] - xq).max() / scale
        worst = max(worst, err)
        assert err <= 1e-10, (r, err)
        zq = o.evaluate(c, xq)
        assert abs(v["obj"] - zq) <= 1e-10 * abs(zq), (r, v["obj"], zq)
    # the winner is the best of the sampled vertices under the QR path too
    zs = [o.evaluate(c, o.basic_solution(A, b, p.vertex(r, n)["basis"])[1]) for r in ranks[:50]]
    assert max(zs) <= zs[0] + 1e-9
    p.free()


def _solve_sharded_threads(A, b, c, maximize, n_orig, world, comms):
    """One host thread per shard, each with a context and a replica of the problem of its own (they
    share the one device of this box), all inside lp_enum_solve_sharded at the same time."""
    import threading
    out = [None] * world

    def run(r):
        cx = capi.Context(0)
        p = cx.enum_problem(A, b, c, maximize)
        try:
            out[r] = p.solve_sharded(comms[r], n_orig)
        except capi.LPError as e:
            out[r] = e
        p.free()
        cx.close()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out


@pytest.mark.parametrize("world", [2, 3, 6])
def test_c_abi_sharded_solve_matches_oracle(ctx, world):
    """lp_enum_solve_sharded (the C entry point a C++ host or bench.py --gpus N calls): shards of the
    rank space, ONE all-gather of the incumbent record, identical answers on every participant and
    for every world size — against the oracle's single-range solve.  Includes a problem with
    duplicated columns (distinct bases, equal objective: the cross-shard tie rule) and one with no
    feasible basis."""
    cases = []
    A, b, c, _ = lpcases.random_lp(21, 6, 14)
    cases.append((A, b, c, True))
    A2 = A.copy()
    A2[:, 5] = A2[:, 1]                      # a duplicated column: tied vertices with different ranks
    c2 = c.copy()
    c2[5] = c2[1]
    cases.append((A2, b, c2, True))
    cases.append((A, b, c, False))
    for A_, b_, c_, mx in cases:
        m, n = A_.shape
        ref = o.enum_solve(A_, b_, c_, mx, n - m)
        comms = capi.Comm.local(world)
        got = _solve_sharded_threads(A_, b_, c_, mx, n - m, world, comms)
        for cm in comms:
            cm.destroy()
        for g in got:
            assert not isinstance(g, Exception), g
            assert g["status"] == ref["status"] == 0
            assert g["rank"] == ref["rank"] and g["obj"] == ref["obj"] and g["counts"] == ref["counts"]
            assert np.array_equal(g["x"], ref["x"]) and np.array_equal(g["basis"], ref["basis"])
    # no feasible basis anywhere: every participant reports LP_INFEASIBLE
    An = np.array([[1.0, 1.0, 1.0]])
    comms = capi.Comm.local(world)
    got = _solve_sharded_threads(An, [-1.0], [1.0, 1.0, 0.0], True, 2, world, comms)
    for cm in comms:
        cm.destroy()
    assert all(g["status"] == capi.INFEASIBLE for g in got)


def test_c_abi_sharded_solve_full_size_two_shards(ctx):
    """C(28,14) in two cost-balanced shards through lp_enum_solve_sharded (shared-prefix kernels on
    both shards at once on one device) against the single-range solve."""
    A, b, c, _ = lpcases.random_lp(0, 14, 28)
    one = ctx.enum_solve(A, b, c, True, 14)
    comms = capi.Comm.local(2)
    got = _solve_sharded_threads(A, b, c, True, 14, 2, comms)
    for cm in comms:
        cm.destroy()
    for g in got:
        assert g["rank"] == one["rank"] and g["obj"] == one["obj"] and g["counts"] == one["counts"]
        assert np.array_equal(g["x"], one["x"])


def test_c_abi_rccl_communicator(ctx):
    """The RCCL backend of lp_comm with world = number of devices here (1 on the test box): librccl
    loads (dlopen), ncclCommInitRank and the 48-byte ncclAllGather run on the library's stream."""
    A, b, c, _ = lpcases.random_lp(22, 6, 13)
    ref = o.enum_solve(A, b, c, True, 7)
    comm = capi.Comm.rccl(ctx, 0, 1, capi.Comm.unique_id())
    assert comm.rank() == 0 and comm.world() == 1
    p = ctx.enum_problem(A, b, c, True)
    g = p.solve_sharded(comm, 7)
    p.free()
    comm.destroy()
    assert g["rank"] == ref["rank"] and g["obj"] == ref["obj"] and g["counts"] == ref["counts"]


def test_feasible_list_spills_into_subranges(ctx, monkeypatch):
    """Degenerate LPs (b = 0: every non-singular basis is feasible) overflow the feasible list of the
    shared-prefix path; the range is then enumerated in sub-ranges, one list at a time, and pass 2
    re-runs only the sub-range that holds the winner — never the direct kernel.  The list is shrunk
    to 300 entries here so that small problems (oracle-checkable) take that path, including a
    second-level split."""
    monkeypatch.setenv("LP_ENUM_LIST_CAP", "300")
    for seed, m, n, zero_rows in [(71, 8, 18, 8), (72, 8, 18, 5), (73, 10, 20, 10), (74, 7, 17, 0)]:
        A, b, c, _ = lpcases.random_lp(seed, m, n)
        b = b.copy()
        b[:zero_rows] = 0.0
        total = o.binom(n, m)
        ref = o.enum_range(A, b, c, True, 0, total)
        p = ctx.enum_problem(A, b, c, True)
        got = p.range(0, total, capi.ENUM_PREFIX)[:3]
        assert got == ref, (seed, m, n)
        assert ref[2][0] > 300 or zero_rows == 0
        k = p.first_within(0, total, ref[1])
        assert k == o.enum_first_within(A, b, c, True, 0, total, ref[1])
        # a sub-range of the rank space, and a second tolerance (served from the per-sub-range bests)
        lo, hi = total // 7, total - total // 5
        ref2 = o.enum_range(A, b, c, True, lo, hi)
        assert p.range(lo, hi, capi.ENUM_PREFIX)[:3] == ref2
        assert p.first_within(lo, hi, ref2[1], 1e-3) == o.enum_first_within(A, b, c, True, lo, hi, ref2[1], 1e-3)
        p.free()


def test_feasible_list_grows_and_is_evaluated_from_records(ctx, monkeypatch):
    """The same degenerate LPs on the default route: the pass that overflows the list reports the number
    of feasible subsets.  More than a third of the range (or a list 16x over capacity): the dense form —
    no list, the general leaf kernel writes every subset's score by rank (all rows zero below).  Fewer:
    the list is re-allocated to hold them and the pass runs once more; every entry's objective then
    comes from the depth m-7 record it was found under (k_enum_eval_records).  Counts, optimum and the
    tie rule's rank against the oracle, and bit-identical to the from-scratch evaluation of the direct
    kernel; a second pass over a sub-range starts in the form the first one ended in."""
    monkeypatch.setenv("LP_ENUM_LIST_START", "300")
    for seed, m, n, zero_rows in [(71, 8, 18, 8), (72, 8, 18, 5), (73, 10, 20, 10), (74, 7, 17, 7), (75, 12, 22, 9)]:
        A, b, c, _ = lpcases.random_lp(seed, m, n)
        b = b.copy()
        b[:zero_rows] = 0.0
        total = o.binom(n, m)
        ref = o.enum_range(A, b, c, True, 0, total)
        assert ref[2][0] > 300
        p = ctx.enum_problem(A, b, c, True)
        got = p.range(0, total, capi.ENUM_PREFIX)
        assert got[:3] == ref, (seed, m, n)
        assert p.first_within(0, total, ref[1]) == o.enum_first_within(A, b, c, True, 0, total, ref[1])
        assert p.range(0, total, capi.ENUM_DIRECT)[:3] == ref
        lo, hi = total // 7, total - total // 5
        ref2 = o.enum_range(A, b, c, True, lo, hi)
        assert p.range(lo, hi, capi.ENUM_PREFIX)[:3] == ref2
        assert p.first_within(lo, hi, ref2[1], 1e-3) == o.enum_first_within(A, b, c, True, lo, hi, ref2[1], 1e-3)
        p.free()


def test_dense_form_with_pruned_subtrees(ctx, monkeypatch):
    """Dense form (rank-indexed scores) on LPs whose breadth-first levels prune whole subtrees as singular
    (a duplicated column, a zero column): the pruned subsets never get a score written, and with b = 0
    every feasible score is 0, so any stale 0.0 (or an old score of a previous pass at a shifted offset)
    in the pooled score buffer would pass the tie rule's `>= star - tol` test.  Full range first, then a
    sub-range that reuses the same buffer at other offsets; counts, optimum and the tie rule's rank
    against the oracle."""
    monkeypatch.setenv("LP_ENUM_LIST_START", "300")
    for seed, m, n in [(81, 8, 18), (82, 9, 19), (83, 10, 20)]:
        A, b, c, _ = lpcases.random_lp(seed, m, n)
        A = A.copy()
        A[:, 1] = A[:, 0]          # duplicate of the FIRST column: every subset with both is singular,
        A[:, 3] = 0.0              # and a zero column: ranks 0.. are all pruned at depth 1
        b = np.zeros_like(b)
        total = o.binom(n, m)
        ref = o.enum_range(A, b, c, True, 0, total)
        assert ref[2][0] > total // 3 and ref[2][2] > 0   # dense territory, with singular subsets
        p = ctx.enum_problem(A, b, c, True)
        # a previous pass of ANOTHER shape leaves its scores in the pooled buffer
        got = p.range(0, total, capi.ENUM_PREFIX)
        assert got[:3] == ref, (seed, m, n)
        want = o.enum_first_within(A, b, c, True, 0, total, ref[1])
        assert p.first_within(0, total, ref[1]) == want
        for lo, hi in [(total // 7, total - total // 5), (0, total // 3), (total // 2, total)]:
            ref2 = o.enum_range(A, b, c, True, lo, hi)
            assert p.range(lo, hi, capi.ENUM_PREFIX)[:3] == ref2, (seed, lo, hi)
            assert p.first_within(lo, hi, ref2[1]) == o.enum_first_within(A, b, c, True, lo, hi, ref2[1])
        p.free()


def test_fuzz_structured_problems_wide(ctx):
    """Structured data (integers in {-1, 0, 1, 2}: ties, zero pivots, duplicate and zero columns, singular
    prefixes pruning whole subtrees) on the general leaf kernel's shapes — 32-row records and more than 16
    selectable columns — both senses, degenerate right-hand sides included: counts, optimum and the tie
    rule's rank against the oracle, on both kernels."""
    rng = np.random.default_rng(78)
    shapes = [(17, 20), (18, 21), (20, 23), (7, 24), (8, 25), (7, 27), (19, 22), (24, 27)]
    for trial in range(24):
        m, n = shapes[trial % len(shapes)]
        A = rng.integers(-1, 3, size=(m, n)).astype(np.float64)
        if trial % 3 == 0:
            A[:, n - m:] += np.eye(m)          # fewer singular bases
        b = rng.integers(0, 4, size=m).astype(np.float64)
        c = rng.integers(-2, 3, size=n).astype(np.float64)
        maximize = bool(rng.integers(0, 2))
        total = o.binom(n, m)
        ref = o.enum_range(A, b, c, maximize, 0, total)
        p = ctx.enum_problem(A, b, c, maximize)
        for algo in (capi.ENUM_DIRECT, capi.ENUM_PREFIX):
            got = p.range(0, total, algo)[:3]
            assert got == ref, (trial, m, n, algo)
            if ref[0] == 0:
                assert p.first_within(0, total, ref[1]) == o.enum_first_within(A, b, c, maximize, 0, total, ref[1])
        p.free()


def test_fuzz_small_structured_problems(ctx):
    """150 small problems with integer data in {-1, 0, 1, 2} (ties, zero pivots, duplicate columns
    everywhere), both senses: counts, optimum and the tie rule's rank against the oracle — on the
    direct kernel, and on the shared-prefix kernels where the shape allows."""
    rng = np.random.default_rng(77)
    for trial in range(150):
        m = int(rng.integers(1, 9))
        n = m + int(rng.integers(1, 7))
        A = rng.integers(-1, 3, size=(m, n)).astype(np.float64)
        b = rng.integers(0, 4, size=m).astype(np.float64)
        c = rng.integers(-2, 3, size=n).astype(np.float64)
        maximize = bool(rng.integers(0, 2))
        total = o.binom(n, m)
        ref = o.enum_range(A, b, c, maximize, 0, total)
        p = ctx.enum_problem(A, b, c, maximize)
        algos = [capi.ENUM_DIRECT] + ([capi.ENUM_PREFIX] if 6 <= m <= 16 and 2 <= n - m else [])
        for algo in algos:
            got = p.range(0, total, algo)[:3]
            assert got == ref, (trial, m, n, algo)
            if ref[0] == 0:
                assert p.first_within(0, total, ref[1]) == o.enum_first_within(A, b, c, maximize, 0, total, ref[1])
        p.free()


# ---- the leaf kernels' fast reciprocal (enum_leaf.hip: recip_midrange) and its way out

def test_leaf_reciprocal_matches_division(ctx):
    """recip_midrange(x) == 1.0 / x bit for bit on its whole range [2^-500, 2^500] (both computed on
    the device): random mantissas over every exponent, the mantissas whose quotients sit next to a
    rounding boundary, and both ends of the range."""
    rng = np.random.default_rng(2026)
    n = 1 << 21
    mant = rng.integers(0, 1 << 52, size=n, dtype=np.uint64)
    edge = np.array([0, 1, 2, 3, (1 << 52) - 1, (1 << 52) - 2, 1 << 51, (1 << 51) + 1, (1 << 51) - 1,
                     0x5555555555555, 0xAAAAAAAAAAAAA, 0x6A09E667F3BCD, 0x6A09E667F3BCC], dtype=np.uint64)
    mant[: 64 * len(edge)] = np.tile(edge, 64)
    expo = rng.integers(-500, 500, size=n).astype(np.int64)
    expo[:1001] = np.arange(-500, 501)
    mant[:1001:7] = 0
    x = ((expo + 1023).astype(np.uint64) << np.uint64(52) | mant).view(np.float64)
    x[expo == 500] = 2.0 ** 500                      # (the range is closed: 2^500 itself, not beyond)
    x[1::2] *= -1.0
    assert np.all(np.abs(x) >= 2.0 ** -500) and np.all(np.abs(x) <= 2.0 ** 500)
    fast, plain = ctx.debug_reciprocal(x)
    assert np.array_equal(plain, 1.0 / x)            # the device's division is IEEE's
    bad = np.flatnonzero(fast.view(np.uint64) != plain.view(np.uint64))
    assert bad.size == 0, (x[bad[:4]], fast[bad[:4]], plain[bad[:4]])


@pytest.mark.parametrize("m,n,seed", [(8, 18, 3), (7, 15, 4), (6, 13, 5), (17, 22, 6)])
@pytest.mark.parametrize("scale", [2.0 ** -510, 2.0 ** 505])
def test_prefix_pivots_outside_fast_reciprocal_range(ctx, m, n, seed, scale):
    """A problem scaled so that every pivot lies outside the fast reciprocal's range: the pass reports
    it, is repeated with plain divisions (visible through lp_enum_exact_division) and matches the oracle."""
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    A, b = A * scale, b * scale
    total = o.binom(n, m)
    st, z, counts = o.enum_range(A, b, c, True, 0, total)
    assert counts[0] > 0
    p = ctx.enum_problem(A, b, c, True)
    assert not p.exact_division
    assert p.range(0, total, capi.ENUM_PREFIX)[:3] == (st, z, counts)
    assert p.exact_division
    assert p.range(total // 3, total, capi.ENUM_PREFIX)[:3] == o.enum_range(A, b, c, True, total // 3, total)
    assert p.range(0, total, capi.ENUM_DIRECT)[:3] == (st, z, counts)
    p.free()


@pytest.mark.parametrize("m,n,seed", [(7, 16, 31), (8, 15, 32), (6, 18, 33)])
def test_fast_reciprocal_flag_with_singular_subsets(ctx, m, n, seed):
    """Small-integer data: thousands of subsets are singular by an exactly-zero pivot.  Such a pivot's verdict
    is final (nothing was divided by anything out of range before it), so the fast leaf kernels must NOT switch
    to plain divisions on it — and the same data scaled to 2^-510 / 2^505, where the pivots in front of a zero
    one are out of range too (every later value then rests on the fast reciprocal's unspecified quotients, also
    the ones that make a subset LOOK singular), must switch whatever the subsets' verdicts are.  Counts,
    optimum and tie-rule rank equal the oracle's every time."""
    rng = np.random.default_rng(seed)
    A = np.hstack([rng.integers(-1, 3, (m, n - m)).astype(float), np.eye(m)])
    b = rng.integers(0, 4, m).astype(float)
    c = np.concatenate([rng.integers(-2, 4, n - m).astype(float), np.zeros(m)])
    total = o.binom(n, m)
    for scale in (1.0, 2.0 ** -510, 2.0 ** 505):
        As, bs = A * scale, b * scale
        st, z, counts = o.enum_range(As, bs, c, True, 0, total)
        assert counts[2] > 100 and counts[0] > 0
        p = ctx.enum_problem(As, bs, c, True)
        assert p.range(0, total, capi.ENUM_PREFIX)[:3] == (st, z, counts)
        assert p.exact_division == (scale != 1.0)
        assert p.range(0, total, capi.ENUM_DIRECT)[:3] == (st, z, counts)
        p.free()


@pytest.mark.parametrize("m,n,seed", [(8, 20, 11), (16, 22, 12), (7, 30, 13), (18, 24, 14), (6, 14, 15)])
def test_prefix_fast_and_plain_division_agree(ctx, m, n, seed, monkeypatch):
    """The default leaf kernels (fast reciprocal) and their plain-division instantiations
    (LP_ENUM_EXACT_DIV=1) return the same bits; the default ones do not switch on ordinary data."""
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    total = o.binom(n, m)
    p = ctx.enum_problem(A, b, c, True)
    ref = p.range(0, total, capi.ENUM_PREFIX)[:3]
    k = p.first_within(0, total, ref[1])
    assert not p.exact_division
    p.free()
    monkeypatch.setenv("LP_ENUM_EXACT_DIV", "1")
    q = ctx.enum_problem(A, b, c, True)
    assert q.range(0, total, capi.ENUM_PREFIX)[:3] == ref and q.exact_division
    assert q.first_within(0, total, ref[1]) == k
    q.free()


@pytest.mark.parametrize("m,n,seed", [(16, 24, 21), (12, 26, 22), (9, 30, 23), (16, 32, 24)])
def test_prefix_level_kernels_agree(ctx, m, n, seed, monkeypatch):
    """The breadth-first levels in their three forms — parent staged in LDS (default for wide levels of 16-row
    records), operands fetched from the record (LP_ENUM_EXPAND_UNSTAGED=1), one wave per child (every level
    narrow: LP_ENUM_NARROW_MULT large) — build the same records: counts, optimum and tie-rule rank of a range
    are identical, and equal to the direct kernel's."""
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    total = o.binom(n, m)
    lo, hi = total // 7, min(total, total // 7 + (1 << 22))
    p = ctx.enum_problem(A, b, c, True)
    ref = p.range(lo, hi, capi.ENUM_PREFIX)[:3]
    k = p.first_within(lo, hi, ref[1]) if ref[0] == 0 else None
    assert p.range(lo, hi, capi.ENUM_DIRECT)[:3] == ref
    for var, val in (("LP_ENUM_EXPAND_UNSTAGED", "1"), ("LP_ENUM_NARROW_MULT", "100000000")):
        monkeypatch.setenv(var, val)
        assert p.range(lo, hi, capi.ENUM_PREFIX)[:3] == ref, var
        if k is not None:
            assert p.first_within(lo, hi, ref[1]) == k
        monkeypatch.delenv(var)
    p.free()


@pytest.mark.parametrize("m,n,seed,window", [(12, 26, 22, 1 << 22), (14, 28, 0, None), (16, 32, 0, 1 << 26), (18, 30, 5, 1 << 24)])
def test_leaf_item_dealing_loses_nothing_under_repetition(ctx, m, n, seed, window):
    """The leaf kernels deal their work items in runs that live in LDS and are emptied by compare-and-swap, and a
    wave that finds the table dealt out takes what the other waves of its workgroup still hold (round 4).  A wave that
    was told "nothing left" must stay out: a later, luckier draw used to lose the item behind it — two items in one
    pass out of hundreds (found by test_prefix_level_kernels_agree[12-26-22]).  The same range many times over: the
    three counts must add up to the range and never change, the optimum neither; once against the direct kernel."""
    A, b, c, _ = lpcases.random_lp(seed, m, n)
    total = o.binom(n, m)
    lo = total // 7 if window else 0
    hi = min(total, lo + window) if window else total
    p = ctx.enum_problem(A, b, c, True)
    ref = p.range(lo, hi, capi.ENUM_PREFIX)[:3]
    assert sum(ref[2]) == hi - lo
    if hi - lo <= (1 << 24):
        assert p.range(lo, hi, capi.ENUM_DIRECT)[:3] == ref
    for _ in range(40):
        assert p.range(lo, hi, capi.ENUM_PREFIX)[:3] == ref
    p.free()
