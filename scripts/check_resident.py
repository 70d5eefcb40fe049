"""Quick GPU check of the chip-resident simplex: parity against the oracle on a few shapes,
timing at 512 x 1024 and (optionally) the per-phase cycle stamps of workgroup 0."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as o          # noqa: E402
from simplexmethod_amd import capi        # noqa: E402
from tests import lpcases                 # noqa: E402

ctx = capi.Context(0)
ok = True
for seed, m, n in [(0, 2, 5), (1, 8, 16), (2, 16, 32), (3, 33, 71), (4, 64, 128), (5, 128, 256), (6, 100, 1500),
                   (0, 512, 1024)]:
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    t0 = time.time()
    rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
    dt = time.time() - t0
    g = p.download(trace_cap=max(st.pivots, 1), want_tableau=True)
    same = (rc == r["status"] and st.pivots == r["iters"] and np.array_equal(g["basis"], r["basis"])
            and np.array_equal(g["tableau"], r["tableau"]) and np.array_equal(g["x"], r["x"]))
    k = r["iters"]
    tr = list(zip(g["trace_enter"][:k].tolist(), g["trace_leave"][:k].tolist())) == r["trace"][:k]
    print("seed %d %dx%d: rc=%d pivots=%d (oracle %d) launches=%d solve_ms=%.3f wall=%.3fs err=%r bit-exact=%s trace=%s" %
          (seed, m, n, rc, st.pivots, r["iters"], st.launches, st.solve_ms, dt, ctx.error(), same, tr), flush=True)
    ok &= same and tr
    p.free()

A, b, c, basis = lpcases.random_lp(0, 512, 1024)
p = ctx.simplex_problem(A, b, c, basis, True, 512)
for algo, name in [(capi.SIMPLEX_RESIDENT, "resident"), (capi.SIMPLEX_LOOKAHEAD, "lookahead")]:
    best = 1e9
    for rep in range(10):
        p.reset()
        rc, st = p.run(algo=algo)
        best = min(best, st.solve_ms)
    print("%s: 512x1024 %d pivots best solve_ms=%.3f -> %.3f us/pivot" % (name, st.pivots, best, 1e3 * best / st.pivots),
          flush=True)
# stamps
ctx.lib.lp_debug_simplex_stamps(p.h, 400, None)
p.reset()
rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
G = 32
buf = (C.c_ulonglong * (16 * G))()
ctx.lib.lp_debug_simplex_stamps(p.h, G, buf)
acc = np.array(buf[:16 * G], dtype=np.float64).reshape(G, 16) / st.pivots
names = ["loop", "poll record A", "decide (+ record B)", "decision barrier",
         "read decision, pivot row -> LDS, request column, 2 quotients", "pivot-row barrier",
         "reduced costs + next pricing", "column wait, eta entry", "candidate, publish, ratio stage 1",
         "ratio barrier", "(wave W2: stage 2 + record B), prefetch", "rank-1 update of 32 columns"]
print("stamped run: %.3f ms; cycles per pivot (mean over %d pivots), workgroup 0: total %d" % (st.solve_ms, st.pivots, acc[0, :12].sum()))
for i in range(12):
    print("  %-62s %6d   (min %5d  max %5d over workgroups)" % (names[i], acc[0, i], acc[:, i].min(), acc[:, i].max()))
print("poll-A wait per workgroup:", " ".join("%d" % v for v in acc[:, 1]))
print("decide per workgroup:     ", " ".join("%d" % v for v in acc[:, 2]))
p.free()
ctx.close()
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
