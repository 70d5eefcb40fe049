"""Chip-resident simplex, quick look (GPU box): parity against the oracle's tableau form on ten shapes / iteration
limits and best-of-6 kernel time per pivot for each (LP_RESIDENT_STRICT: a fallback is an error)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LP_RESIDENT_STRICT", "1")
import numpy as np
from oracle import pyoracle as o
from simplexmethod_amd import capi
from tests import lpcases
ctx = capi.Context(0)


def run(seed, m, n, max_iter=capi.MAX_ITER):
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True, max_iter=max_iter)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    rc, st = p.run(max_iter=max_iter, algo=capi.SIMPLEX_RESIDENT)
    g = p.download(trace_cap=max(st.pivots, 1), want_tableau=True)
    ok = (rc == r["status"] and st.pivots == r["iters"] and np.array_equal(g["basis"], r["basis"]) and np.array_equal(g["tableau"], r["tableau"]))
    best = 1e9
    for _ in range(6):
        p.reset()
        rc, st = p.run(max_iter=max_iter, algo=capi.SIMPLEX_RESIDENT)
        best = min(best, st.update_ms)
    print("%3d %4d x %4d limit %5d: rc %d pivots %4d %s  best kernel %.4f ms = %.3f us/pivot" %
          (seed, m, n, max_iter, rc, st.pivots, "OK" if ok else "MISMATCH", best, 1e3 * best / max(st.pivots, 1)), flush=True)
    p.free()
    return ok


allok = True
for args in [(0, 2, 5), (1, 8, 16), (3, 33, 71), (5, 128, 256), (6, 100, 1500), (0, 512, 1024), (41, 768, 1536), (43, 960, 1920),
             (0, 512, 1024, 1), (0, 512, 1024, 7)]:
    allok &= run(*args)
print("ALL OK" if allok else "FAILURES")
