"""Times BASELINE.json configs[4]: 4096 random LPs, m=128, n=256 (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simplexmethod_amd import capi

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m, n = 128, 256
t0 = time.time()
A = np.empty((batch, m, n)); b = np.empty((batch, m)); c = np.empty((batch, n)); basis = np.empty((batch, m), dtype=np.int32)
for k in range(batch):
    A[k], b[k], c[k], basis[k] = capi.gen_lp(k, m, n)
print("gen %.1fs" % (time.time() - t0))
ctx = capi.Context(0)
p = ctx.batched_problem(A, b, c, basis, True, n - m)
for it in range(3):
    ms = p.run()
    d = p.download()
    piv = int(d["iters"].sum())
    print(f"run {it}: {ms:.2f} ms, {batch / ms * 1e3:.0f} LPs/s, pivots {piv} (mean {piv / batch:.1f}), "
          f"{ms * 1e3 / piv * 256:.2f} us per pivot per CU-slot, status ok {(d['status'] == 0).all()}, "
          f"equiv tableau GB/s {16.0 * m * (n + 1) * piv / (ms * 1e-3) / 1e9:.0f}")
