"""GPU parity: batched simplex (one LP per workgroup, BASELINE.json configs[4]) against the
oracle's tableau restatement, LP by LP, bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle as o
from simplexmethod_amd import capi
from tests import lpcases

pytestmark = pytest.mark.gpu


def _batch(seeds, m, n):
    As, bs, cs, bas = zip(*[lpcases.random_lp(s, m, n) for s in seeds])
    return np.stack(As), np.stack(bs), np.stack(cs), np.stack(bas)


def _check(g, A, b, c, basis, maximize, n_orig, max_iter=capi.MAX_ITER):
    for k in range(A.shape[0]):
        r = o.simplex_tableau(A[k], b[k], c[k], basis[k], maximize, n_orig, max_iter=max_iter)
        assert g["status"][k] == r["status"], k
        assert g["iters"][k] == r["iters"], k
        assert np.array_equal(g["basis"][k], r["basis"]), k
        if r["status"] == o.OPTIMAL:
            assert np.array_equal(g["x"][k], r["x"]), k
            assert g["obj"][k] == r["obj"], k


@pytest.mark.parametrize("m,n,count", [(2, 5, 7), (8, 16, 33), (16, 40, 20), (32, 64, 40),
                                        (64, 128, 24), (128, 256, 48), (12, 1012, 6)])
def test_batched_matches_oracle(ctx, m, n, count):
    A, b, c, basis = _batch(range(100, 100 + count), m, n)
    g = ctx.simplex_solve_batched(A, b, c, basis, True, n - m)
    assert (g["status"] == 0).all()
    _check(g, A, b, c, basis, True, n - m)


def test_full_baseline_batch_4096(ctx):
    """BASELINE.json configs[4] at its full size: 4096 LPs of m=128, n=256 (seeds 0..4095, what
    bench.py times) — status, pivot count, basis, vertex and objective of EVERY LP bit-exact against
    the oracle's tableau restatement (197,443 pivots; ~6 s of CPU)."""
    batch, m, n = 4096, 128, 256
    A = np.empty((batch, m, n)); b = np.empty((batch, m)); c = np.empty((batch, n))
    basis = np.empty((batch, m), dtype=np.int32)
    for k in range(batch):
        A[k], b[k], c[k], basis[k] = capi.gen_lp(k, m, n)
    g = ctx.simplex_solve_batched(A, b, c, basis, True, n - m)
    assert (g["status"] == 0).all() and int(g["iters"].sum()) == 197443
    _check(g, A, b, c, basis, True, n - m)


def test_batched_minimise_unbounded_and_limit(ctx):
    A, b, c, basis = _batch(range(5), 12, 30)
    # minimise: the slack vertex is already optimal for c >= 0 (0 pivots) — and with negated
    # costs the minimise rules (:163-174) do real work
    g = ctx.simplex_solve_batched(A, b, -c, basis, False, 18)
    _check(g, A, b, -c, basis, False, 18)
    g = ctx.simplex_solve_batched(A, b, c, basis, True, 18, max_iter=3)
    assert (g["status"] == capi.ITER_LIMIT).all()
    _check(g, A, b, c, basis, True, 18, max_iter=3)
    Au = A.copy()
    Au[:, :, 0] = -1.0          # column 0 never limits the ratio test -> unbounded
    cu = c.copy()
    cu[:, 0] = 10.0
    g = ctx.simplex_solve_batched(Au, b, cu, basis, True, 18)
    assert (g["status"] == capi.UNBOUNDED).all()
    _check(g, Au, b, cu, basis, True, 18)


def test_batched_degenerate_ties(ctx):
    """Every LP has b = 0 rows: ratios tie at 0 and the keyed scan must keep the first position."""
    A, b, c, basis = _batch(range(40, 52), 10, 24)
    b[:, ::2] = 0.0
    g = ctx.simplex_solve_batched(A, b, c, basis, True, 14)
    _check(g, A, b, c, basis, True, 14)


def test_batched_fallback_general_basis(ctx):
    """Non-slack starting bases cannot use the LDS-resident kernel; the fallback must agree too."""
    probs = [lpcases.general_lp(s, 6, 14) for s in (1, 2, 3)]
    A = np.stack([p[0] for p in probs]); b = np.stack([p[1] for p in probs])
    c = np.stack([p[2] for p in probs]); basis = np.stack([p[3] for p in probs])
    n = A.shape[2]
    g = ctx.simplex_solve_batched(A, b, c, basis, True, n)
    _check(g, A, b, c, basis, True, n)


def test_batched_problem_object_and_timing(ctx):
    A, b, c, basis = _batch(range(7, 7 + 64), 32, 64)
    p = ctx.batched_problem(A, b, c, basis, True, 32)
    ms1 = p.run()
    d1 = p.download()
    ms2 = p.run()
    d2 = p.download()
    assert ms1 > 0 and ms2 > 0
    assert np.array_equal(d1["x"], d2["x"]) and np.array_equal(d1["iters"], d2["iters"])
    _check(d2, A, b, c, basis, True, 32)
    p.free()


def test_fuzz_shapes_and_signed_data(ctx):
    """40 shapes x 12 LPs with signed constraint data (unbounded and degenerate ones among them),
    both senses, tight iteration limits: the LDS-resident kernel against the oracle, LP by LP."""
    rng = np.random.default_rng(4242)
    for trial in range(40):
        m = int(rng.integers(1, 20))
        extra = int(rng.integers(1, 24))
        n = m + extra
        count = 12
        A = np.empty((count, m, n)); b = np.empty((count, m)); c = np.empty((count, n))
        basis = np.tile(np.arange(extra, n, dtype=np.int32), (count, 1))
        for k in range(count):
            A[k] = np.hstack([np.round(rng.uniform(-1, 1.5, (m, extra)), 1), np.eye(m)])
            b[k] = np.round(rng.uniform(0.0, 2.0, m), 1)          # zeros: degenerate vertices
            c[k] = np.concatenate([np.round(rng.uniform(-1, 1, extra), 1), np.zeros(m)])
        maximize = bool(rng.integers(0, 2))
        max_iter = int(rng.choice([1, 3, 10000]))
        g = ctx.simplex_solve_batched(A, b, c, basis, maximize, n, max_iter=max_iter)
        _check(g, A, b, c, basis, maximize, n, max_iter=max_iter)
