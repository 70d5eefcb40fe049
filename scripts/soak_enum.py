"""Randomised cross-check of the two enumeration kernels on the GPU: random shapes (tuned box, 32-row
records, wide column ranges), random data (uniform, small integers, zero right-hand sides), random rank
ranges, random list capacities — shared-prefix path against the direct kernel: status, optimum, counts and
the tie rule's rank must be identical.

    python scripts/soak_enum.py [seconds] [seed]
"""
import math
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
    trials = 0
    kinds = {}
    while time.time() < t_end:
        m = int(rng.integers(6, 27))
        nm = int(rng.integers(2, 27))
        n = m + nm
        total = math.comb(n, m)
        if total > 6_000_000 or n > 64 or (m > 16 and nm > 32) or (m == 6 and nm > 16):
            continue
        kind = int(rng.integers(0, 4))
        if kind == 0:
            A = rng.uniform(0, 1, size=(m, n)); b = rng.uniform(1, 2, size=m) * nm / 2; c = rng.uniform(0, 1, size=n)
        elif kind == 1:
            A = rng.integers(-1, 3, size=(m, n)).astype(float); b = rng.integers(0, 4, size=m).astype(float)
            c = rng.integers(-2, 3, size=n).astype(float)
        elif kind == 2:
            A = rng.normal(size=(m, n)); b = np.abs(rng.normal(size=m)); c = rng.normal(size=n)
            b[: int(rng.integers(0, m + 1))] = 0.0
        else:
            A = rng.uniform(-1, 1, size=(m, n)); A[:, n - m:] += np.eye(m); b = rng.uniform(0, 1, size=m); c = rng.normal(size=n)
        maximize = bool(rng.integers(0, 2))
        if rng.integers(0, 3) == 0:
            os.environ["LP_ENUM_LIST_START"] = str(int(rng.integers(64, 4000)))
        else:
            os.environ.pop("LP_ENUM_LIST_START", None)
        p = ctx.enum_problem(A, b, c, maximize)
        ranges = [(0, total)]
        for _ in range(2):
            lo = int(rng.integers(0, total))
            hi = int(rng.integers(lo, total + 1))
            ranges.append((lo, hi))
        for lo, hi in ranges:
            if hi == lo:
                continue
            rp = p.range(lo, hi, capi.ENUM_PREFIX)[:3]
            rd = p.range(lo, hi, capi.ENUM_DIRECT)[:3]
            same = rp[0] == rd[0] and rp[2] == rd[2] and (rp[1] == rd[1] or (rp[1] != rp[1] and rd[1] != rd[1]))
            if not same:
                print("MISMATCH", m, n, kind, maximize, lo, hi, rp, rd, os.environ.get("LP_ENUM_LIST_START"))
                return 1
            if rp[0] == 0:
                rp2 = p.range(lo, hi, capi.ENUM_PREFIX)   # (first_within is served by the last range pass)
                k1 = p.first_within(lo, hi, rp2[1])
                p.range(lo, hi, capi.ENUM_DIRECT)
                k2 = p.first_within(lo, hi, rd[1])
                if k1 != k2:
                    print("TIE RULE MISMATCH", m, n, kind, maximize, lo, hi, k1, k2)
                    return 1
        p.free()
        trials += 1
        kinds[(m > 16, nm > 16)] = kinds.get((m > 16, nm > 16), 0) + 1
    print("ok:", trials, "problems;", "(m > 16, n - m > 16) ->", kinds)
    return 0


if __name__ == "__main__":
    sys.exit(main())
