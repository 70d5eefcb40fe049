"""A/B timing of the batched simplex kernels (BASELINE configs[4]: 4096 LPs of 128 x 256).
LP_BATCHED_LDS=1: the LDS-resident form; LP_BATCHED_1024=1: the 1024-thread register form;
default: the 512-thread register form (two LPs per CU)."""
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
ref = None
for env in [{"LP_BATCHED_LDS": "1"}, {"LP_BATCHED_1024": "1"}, {}]:
    for k in ("LP_BATCHED_LDS", "LP_BATCHED_1024"):
        os.environ.pop(k, None)
    os.environ.update(env)
    p.run()
    ms = min(p.run() for _ in range(5))
    d = p.download()
    piv = int(d["iters"].sum())
    same = True
    if ref is None:
        ref = d
    else:
        same = all(np.array_equal(d[k], ref[k]) for k in ("x", "basis", "iters", "status", "obj"))
    print(env or "default (512-thread register form, 2 LPs per CU)", "%.3f ms, %d pivots, all optimal %s, identical to the LDS form %s" %
          (ms, piv, bool((d["status"] == 0).all()), same), flush=True)
p.free()
