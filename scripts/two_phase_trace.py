import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from simplexmethod_amd import capi
ctx = capi.Context(0)
m = k = 256
rng = np.random.default_rng(0)
A = np.hstack([rng.uniform(0.0, 1.0, size=(m, k)), -np.eye(m)])
b = rng.uniform(1.0, 2.0, size=m)
c = np.concatenate([rng.uniform(0.1, 1.0, size=k), np.zeros(m)])
for rep in range(4):
    t0 = time.perf_counter()
    r = ctx.two_phase(A, b, c, maximize=False, n_orig=k)
    print("rep", rep, "ms", 1e3 * (time.perf_counter() - t0), r["status"], r["iters"], file=sys.stderr)
