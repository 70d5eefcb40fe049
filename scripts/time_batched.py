"""Timing of the batched simplex kernel (BASELINE configs[4]: 4096 LPs of 128 x 256), best of 5: the form
lp_batched_launch selects (the 512-thread register form, two LPs per CU; the A/B knobs of rounds 2-3 are gone)."""
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
ms = min(p.run() for _ in range(5))
d = p.download()
print("512-thread register form, 2 LPs per CU: %.3f ms, %d pivots, all optimal %s" % (ms, int(d["iters"].sum()), bool((d["status"] == 0).all())), flush=True)
p.free()
