"""Randomised cross-check of the simplex kernels on the GPU: random shapes and data (uniform, small
integers with ties everywhere, degenerate right-hand sides), both senses — the chip-resident kernel
against the launch-per-pivot kernel and the look-ahead kernel: status, pivot count, pivot trace, basis, every
tableau element and the vertex must be bit-identical.

    python scripts/soak_simplex.py [seconds] [seed]
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
    trials, pivots, statuses = 0, 0, {}
    while time.time() < t_end:
        m = int(rng.integers(1, 160)) if rng.integers(0, 4) else int(rng.integers(160, 513))
        if rng.integers(0, 12) == 0:
            m = int(rng.integers(513, 961))   # 16 columns per workgroup, the LDS mirror of the slab (simplex_resident.hip: MIRROR)
        no = int(rng.integers(1, 2 * m + 2))
        n = no + m
        kind = int(rng.integers(0, 3))
        if kind == 0:
            A0 = rng.uniform(0, 1, size=(m, no)); b = rng.uniform(1, 2, size=m) * no / 2; c0 = rng.uniform(0, 1, size=no)
        elif kind == 1:
            A0 = rng.integers(-1, 4, size=(m, no)).astype(float); b = rng.integers(0, 5, size=m).astype(float)
            c0 = rng.integers(-1, 4, size=no).astype(float)
        else:
            A0 = rng.normal(size=(m, no)); b = np.abs(rng.normal(size=m)); b[: int(rng.integers(0, m + 1))] = 0.0
            c0 = rng.normal(size=no)
        A = np.hstack([A0, np.eye(m)])
        c = np.concatenate([c0, np.zeros(m)])
        basis = np.arange(no, n, dtype=np.int32)
        maximize = bool(rng.integers(0, 2))
        max_iter = int(rng.choice([50, 400, 5000]))
        got = []
        for algo in (capi.SIMPLEX_RESIDENT, capi.SIMPLEX_LAUNCH, capi.SIMPLEX_LOOKAHEAD, capi.SIMPLEX_OVERLAP):
            p = ctx.simplex_problem(A, b, c, basis, maximize, no)
            rc, st = p.run(max_iter=max_iter, algo=algo)
            d = p.download(trace_cap=min(st.pivots, 5000), want_tableau=True)
            got.append((rc, st.pivots, d))
            p.free()
        ref = got[0]
        for other in got[1:]:
            same = (ref[0] == other[0] and ref[1] == other[1]
                    and np.array_equal(ref[2]["basis"], other[2]["basis"])
                    and np.array_equal(ref[2]["trace_enter"], other[2]["trace_enter"])
                    and np.array_equal(ref[2]["trace_leave"], other[2]["trace_leave"])
                    and np.array_equal(ref[2]["tableau"], other[2]["tableau"], equal_nan=True)
                    and np.array_equal(ref[2]["x"], other[2]["x"], equal_nan=True))
            if not same:
                print("MISMATCH", m, n, kind, maximize, max_iter, ref[0], ref[1], other[0], other[1])
                return 1
        trials += 1
        pivots += ref[1]
        statuses[ref[0]] = statuses.get(ref[0], 0) + 1
    print("ok:", trials, "problems,", pivots, "pivots; statuses", statuses)
    return 0


if __name__ == "__main__":
    sys.exit(main())
