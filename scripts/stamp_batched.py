"""Per-phase cycle stamps of the batched simplex kernel (BASELINE configs[4]: 4096 LPs of 128 x 256):
LP_BATCHED_STAMPS=1 selects the instrumented instantiation, which prints workgroup 0's sums.
  LP_BATCHED_STAMPS=1 python3 scripts/stamp_batched.py 2> profiles/r02_batched_stamps.txt"""
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
for _ in range(3):
    ms = p.run()
d = p.download()
piv = int(d["iters"].sum())
print("batched: %.3f ms, %d pivots, %.3f us per pivot per CU (256 CUs), LDS floor 1.7 us" %
      (ms, piv, ms * 1e3 / (piv / 256.0)), file=sys.stderr)
p.free()
