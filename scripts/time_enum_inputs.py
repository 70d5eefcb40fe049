"""Enumeration C(n,m) on several inputs (three seeds, the fully degenerate b = 0 LP, a half-degenerate
one): wall time of pass 1 + tie rule, kernel time of pass 1, counts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from simplexmethod_amd import capi  # noqa: E402


def main():
    m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (16, 32)
    ctx = capi.Context(0)
    cases = [("seed 1", 1, None), ("seed 2", 2, None), ("b = 0", 0, 0), ("b = 0 on half of the rows", 0, m // 2)]
    for label, seed, zero_rows in cases:
        A, b, c, _ = capi.gen_lp(seed, m, n)
        if zero_rows is not None:
            b = b.copy()
            b[zero_rows:] = 0.0
        p = ctx.enum_problem(A, b, c, True)
        for rep in range(2):
            t0 = time.perf_counter()
            rc, z, counts, st = p.range(0, p.total, capi.ENUM_AUTO)
            t1 = time.perf_counter()
            k = p.first_within(0, p.total, z) if rc == 0 else None
            t2 = time.perf_counter()
            print(f"{label:28s} rep {rep}: range {1e3 * (t1 - t0):9.3f} ms (kernels {st.kernel_ms:9.3f} ms, "
                  f"{st.launches} launches)  tie rule {1e3 * (t2 - t1):8.3f} ms  rc={rc} z={z!r} rank={k} "
                  f"counts={counts}  {p.total / (t2 - t0) / 1e9:.2f} G subsets/s", flush=True)
        p.free()


if __name__ == "__main__":
    main()
