"""Workload for the PMC passes on the batched simplex kernel (BASELINE configs[4]: 4096 LPs of 128 x 256, seeds 0..4095):
three runs of k_batched_simplex_reg<512,44,true>.  Run under rocprofv3 --pmc ... (scripts/refresh_profiles.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
batch, m, n = 4096, 128, 256
A = np.empty((batch, m, n)); b = np.empty((batch, m)); c = np.empty((batch, n)); basis = np.empty((batch, m), dtype=np.int32)
for k in range(batch):
    A[k], b[k], c[k], basis[k] = capi.gen_lp(k, m, n)
p = ctx.batched_problem(A, b, c, basis, True, n - m)
for _ in range(3):
    print("batched ms", p.run())
d = p.download()
print("pivots", int(d["iters"].sum()))
p.free()
