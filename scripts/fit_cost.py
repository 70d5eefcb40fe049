"""Calibrates dist.balanced_shard_bounds: times pass 1 over equal slices of the C(32,16) rank space
on one GPU and fits  ms = k0 + k1*subsets + k2*N7 + k3*N6  (N7 = depth m-7 prefixes inside the
slice, N6 = depth m-6 prefixes with at least 9 selectable columns).  Prints the coefficients in
subset-equivalents (k2/k1, k3/k1, k0/k1)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simplexmethod_amd import capi, dist as D

m, n = 16, 32
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, m, n)
p = ctx.enum_problem(A, b, c, True)
total = p.total
p.range(0, total)

def n7(x):
    return D._binom(n - 7, m - 7) if x >= total else D._lexrank(n - 7, D._unrank(n, m, x)[:m - 7])

def n6(x):
    return D._binom(n - 9, m - 6) if x >= total else D._lexrank(n - 9, D._unrank(n, m, x)[:m - 6])

rows = []
w = [1 + (k % 3) for k in range(parts)]          # slices of 1, 2, 3 units: separates the fixed cost
cum = [0]
for x in w:
    cum.append(cum[-1] + x)
for k in range(parts):
    lo, hi = total * cum[k] // cum[-1], total * cum[k + 1] // cum[-1]
    best = 1e9
    for _ in range(3):
        rc, z, counts, st = p.range(lo, hi)
        best = min(best, st.kernel_ms)
    rows.append((best, hi - lo, n7(hi) - n7(lo), n6(hi) - n6(lo)))
    print(k, rows[-1], flush=True)
R = np.array(rows, dtype=float)
X = np.column_stack([np.ones(len(R)), R[:, 1], R[:, 2], R[:, 3]])
coef, res, *_ = np.linalg.lstsq(X, R[:, 0], rcond=None)
pred = X @ coef
print("coef ms:", coef.tolist())
print("subset-equivalents: fixed %.0f  per N7 %.1f  per N6 %.1f" % (coef[0] / coef[1], coef[2] / coef[1], coef[3] / coef[1]))
print("max rel err %.3f" % np.max(np.abs(pred - R[:, 0]) / R[:, 0]))
