import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from simplexmethod_amd import capi
ctx = capi.Context(0)
batch, m, n = 4096, 128, 256
A = np.empty((batch, m, n)); b = np.empty((batch, m)); c = np.empty((batch, n))
basis = np.empty((batch, m), dtype=np.int32)
for k in range(batch):
    A[k], b[k], c[k], basis[k] = capi.gen_lp(k, m, n)
p = ctx.batched_problem(A, b, c, basis, True, n - m)
p.run()
for kb in (0, 84):
    if kb: os.environ["LP_BATCHED_MIN_LDS_KB"] = str(kb)
    ms = min(p.run() for _ in range(5))
    print("min LDS %3d KB per workgroup: %.3f ms" % (kb, ms), flush=True)
