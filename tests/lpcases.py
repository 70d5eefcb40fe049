"""Shared problem fixtures for the parity tests (inputs only; expected values live in
tests/golden/*.json or come from the oracle at test time)."""
import numpy as np

from simplexmethod_amd import capi


def main_cpp_lp():
    """The LP hard-coded in /root/reference/src/main.cpp:48-57, through
    Symmetrical::ToCanonical (Symmetrical.cpp:169-188): max 3x1+2x2+4x3."""
    A = np.array([[1, 1, 1, 1, 0], [2, 1, 0, 0, 1.0]])
    return A, np.array([6, 8.0]), np.array([3, 2, 4, 0, 0.0]), np.array([3, 4], dtype=np.int32), 3


def input_symmetric_lp():
    """/root/reference/input_symmetric.txt:1-8 in canonical form: max 7x1+8x2+3x3."""
    A = np.array([[1, 2, 3, 1, 0], [4, 5, 6, 0, 1.0]])
    return A, np.array([10, 20.0]), np.array([7, 8, 3, 0, 0.0]), np.array([3, 4], dtype=np.int32), 3


def test_canonical_fixture():
    """/root/reference/tests/test_canonical.cpp:12-22."""
    A = np.array([[1, 2, 1, 0], [3, 4, 0, 1.0]])
    return A, np.array([5, 6.0]), np.array([7, 8, 0, 0.0]), np.array([2, 3], dtype=np.int32)


test_canonical_fixture.__test__ = False


def random_lp(seed, m, n):
    return capi.gen_lp(seed, m, n)


def general_lp(seed, m, n, signed=True):
    """Random LP with a NON-slack feasible starting basis: take a random m-subset as basis,
    pick xB > 0 and set b = B xB so that the basis is feasible; costs on every column."""
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1.0 if signed else 0.0, 1.0, size=(m, n))
    basis = np.sort(rng.choice(n, size=m, replace=False)).astype(np.int32)
    rng.shuffle(basis)
    xB = rng.uniform(0.5, 2.0, size=m)
    b = A[:, basis] @ xB
    c = rng.uniform(-1.0, 1.0, size=n)
    # bound the feasible region: add a positive row  sum(x) + s = big  (slack column appended)
    A2 = np.zeros((m + 1, n + 1))
    A2[:m, :n] = A
    A2[m, :n] = 1.0
    A2[m, n] = 1.0
    b2 = np.concatenate([b, [4.0 * n]])
    c2 = np.concatenate([c, [0.0]])
    basis2 = np.concatenate([basis, [n]]).astype(np.int32)
    return A2, b2, c2, basis2


def min_lp(seed, m, k, equalities=0, negative_rows=0, zero_rhs=0):
    """Random Symmetrical-style MIN problem in canonical form WITHOUT a starting basis
    (SURVEY 8(f) N2): min c.x, A0 x >= b  ->  [A0 | -I], surplus columns, n = k + m.
    equalities: the first rows get no surplus column weight (their -1 is zeroed: an equality
    row); negative_rows: that many rows are multiplied by -1 (b < 0: exercises make_b_nonneg);
    zero_rhs: that many rows get b = 0 (degenerate: artificials may stay basic at level 0)."""
    rng = np.random.default_rng(1000 + seed)
    A0 = rng.uniform(0.0, 1.0, size=(m, k))
    b = rng.uniform(1.0, 2.0, size=m)
    c = np.concatenate([rng.uniform(0.1, 1.0, size=k), np.zeros(m)])
    S = -np.eye(m)
    for i in range(equalities):
        S[i, i] = 0.0
    if equalities:  # keep the problem feasible: equality rows hold at a positive point
        x0 = rng.uniform(0.5, 1.5, size=k)
        b[:equalities] = A0[:equalities] @ x0
        b[equalities:] = np.minimum(b[equalities:], A0[equalities:] @ x0)
    for i in range(zero_rhs):
        b[m - 1 - i] = 0.0
    A = np.hstack([A0, S])
    for i in range(negative_rows):
        A[i] = -A[i]
        b[i] = -b[i]
    return A, b, c, k


def degenerate_eq_lp(seed, m=4, k=7, zero_rows=2):
    """Equality-form min problem A x = b, x >= 0 with `zero_rows` rows of rhs exactly 0 that the
    feasible point x0 satisfies: phase I of the two-phase simplex often ends with an artificial
    still basic at level 0, which exercises replaceArtificialColumns (SimplexSolover.h:331-381)."""
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (m, k))
    x0 = np.concatenate([rng.uniform(0.5, 1.5, 3), np.zeros(k - 3)])
    b = A @ x0
    for i in range(m - zero_rows, m):
        a = rng.uniform(-1, 1, k)
        a[:3] -= (a[:3] @ x0[:3]) / (x0[:3] @ x0[:3]) * x0[:3]
        A[i] = a
        b[i] = 0.0
    c = rng.uniform(0.1, 1, k)
    return A, b, c, k
