"""Where does the batched kernel's time go: the initial condensed tableaus (max_iter = 0: every LP builds its
tableau and writes its outputs, no pivot) against capped and full solves."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
batch, m, n = 4096, 128, 256
A = np.empty((batch, m, n)); b = np.empty((batch, m)); c = np.empty((batch, n))
basis = np.empty((batch, m), dtype=np.int32)
for k in range(batch):
    A[k], b[k], c[k], basis[k] = capi.gen_lp(k, m, n)
p = ctx.batched_problem(A, b, c, basis, True, n - m)
p.run()
for cap in (0, 1, 8, 16, 32, 48, 64, 10000):
    ms = min(p.run(max_iter=cap) for _ in range(4))
    d = p.download()
    print("max_iter %5d: %.3f ms, pivots executed %d, max per LP %d" % (cap, ms, int(d["iters"].sum()), int(d["iters"].max())), flush=True)
it = d["iters"]
print("pivots per LP: mean %.1f, p50 %d, p90 %d, p99 %d, max %d" % (it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), it.max()))
p.free()
