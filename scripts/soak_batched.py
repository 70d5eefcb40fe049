"""Randomised cross-check of the batched simplex kernel (every register/LDS instantiation, chosen by shape)
against the single-LP path: random shapes m in [2, 200], n - m in [1, 300], batches of 64 LPs with ties and
degenerate right-hand sides — status, pivot count, basis and vertex bit-identical.

    python scripts/soak_batched.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from simplexmethod_amd import capi  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = capi.Context(0)
    t_end = time.time() + budget
    trials, lps = 0, 0
    while time.time() < t_end:
        m = int(rng.integers(2, 200))
        no = int(rng.integers(1, 300))
        n = m + no
        if (m + 1) * (no + 1) * 8 > 150_000 and rng.integers(0, 2):   # (mostly shapes that fit the batched kernels)
            continue
        batch = 32
        kind = int(rng.integers(0, 3))
        A = np.empty((batch, m, n)); b = np.empty((batch, m)); c = np.empty((batch, n))
        basis = np.tile(np.arange(no, n, dtype=np.int32), (batch, 1))
        for k in range(batch):
            if kind == 0:
                A0 = rng.uniform(0, 1, size=(m, no)); b[k] = rng.uniform(1, 2, size=m) * no / 2; c0 = rng.uniform(0, 1, size=no)
            elif kind == 1:
                A0 = rng.integers(-1, 4, size=(m, no)).astype(float); b[k] = rng.integers(0, 5, size=m); c0 = rng.integers(-1, 4, size=no).astype(float)
            else:
                A0 = rng.normal(size=(m, no)); bb = np.abs(rng.normal(size=m)); bb[: int(rng.integers(0, m + 1))] = 0.0
                b[k] = bb; c0 = rng.normal(size=no)
            A[k] = np.hstack([A0, np.eye(m)]); c[k] = np.concatenate([c0, np.zeros(m)])
        maximize = bool(rng.integers(0, 2))
        max_iter = int(rng.choice([40, 2000]))
        got = ctx.simplex_solve_batched(A, b, c, basis, maximize, no, max_iter=max_iter)
        for k in range(0, batch, 5):
            p = ctx.simplex_problem(A[k], b[k], c[k], basis[k], maximize, no)
            rc, st = p.run(max_iter=max_iter)
            d = p.download()
            p.free()
            same = (rc == got["status"][k] and st.pivots == got["iters"][k] and np.array_equal(d["basis"], got["basis"][k])
                    and (rc != 0 or np.array_equal(d["x"], got["x"][k], equal_nan=True)))   # (x is returned for optimal LPs only)
            if not same:
                print("MISMATCH", m, n, kind, maximize, max_iter, k, rc, got["status"][k], st.pivots, got["iters"][k],
                      "basis equal", np.array_equal(d["basis"], got["basis"][k]), "x equal", np.array_equal(d["x"], got["x"][k], equal_nan=True))
                print(" single basis", d["basis"].tolist(), "x", d["x"].tolist())
                print(" batched basis", got["basis"][k].tolist(), "x", got["x"][k].tolist())
                return 1
        trials += 1
        lps += batch
    print("ok:", trials, "batches,", lps, "LPs")
    return 0


if __name__ == "__main__":
    sys.exit(main())
