"""Quick GPU check of the chip-resident simplex: parity against the oracle on a few shapes,
timing at 512 x 1024 (and of the forms kept for A/B: LP_RESIDENT_PUBL, the round-2 kernel) and the
per-phase cycle stamps of workgroup 0."""
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

os.environ["LP_RESIDENT_STRICT"] = "1"
ctx = capi.Context(0)
ok = True
shapes = [(0, 2, 5), (1, 8, 16), (2, 16, 32), (3, 33, 71), (4, 64, 128), (5, 128, 256), (6, 100, 1500),
          (0, 512, 1024), (41, 768, 1536), (44, 513, 700)]
if "--quick" in sys.argv:
    shapes = [(2, 16, 32), (0, 512, 1024), (41, 768, 1536)]
for seed, m, n in shapes:
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    t0 = time.time()
    try:
        rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
    except capi.LPError as e:
        print("seed %d %dx%d: FAILED %s" % (seed, m, n, e), flush=True)
        ok = False
        p.free()
        continue
    dt = time.time() - t0
    g = p.download(trace_cap=max(st.pivots, 1), want_tableau=True)
    same = (rc == r["status"] and st.pivots == r["iters"] and np.array_equal(g["basis"], r["basis"])
            and np.array_equal(g["tableau"], r["tableau"]) and np.array_equal(g["x"], r["x"]))
    k = r["iters"]
    tr = list(zip(g["trace_enter"][:k].tolist(), g["trace_leave"][:k].tolist())) == r["trace"][:k]
    print("seed %d %dx%d: rc=%d pivots=%d (oracle %d) algo_used=%d solve_ms=%.3f (%.3f us/pivot) wall=%.3fs bit-exact=%s trace=%s" %
          (seed, m, n, rc, st.pivots, r["iters"], st.algo_used, st.solve_ms, 1e3 * st.solve_ms / max(st.pivots, 1), dt, same, tr),
          flush=True)
    ok &= same and tr
    p.free()


def best_of(p, algo, reps=20):
    best = 1e9
    for _ in range(reps):
        p.reset()
        rc, st = p.run(algo=algo)
        best = min(best, st.solve_ms)
    return best, st


A, b, c, basis = lpcases.random_lp(0, 512, 1024)
p = ctx.simplex_problem(A, b, c, basis, True, 512)
for rnd in range(2):
    best, st = best_of(p, capi.SIMPLEX_RESIDENT)
    print("resident (column u published, default): 512x1024 %d pivots best solve_ms=%.4f -> %.3f us/pivot" % (st.pivots, best, 1e3 * best / st.pivots), flush=True)
    os.environ["LP_RESIDENT_PUBL"] = "1"
    best, st = best_of(p, capi.SIMPLEX_RESIDENT)
    del os.environ["LP_RESIDENT_PUBL"]
    print("resident (eta column published)        : best solve_ms=%.4f -> %.3f us/pivot" % (best, 1e3 * best / st.pivots), flush=True)
best, st = best_of(p, capi.SIMPLEX_LOOKAHEAD, 5)
print("lookahead: best solve_ms=%.3f -> %.3f us/pivot" % (best, 1e3 * best / st.pivots), flush=True)
A2, b2, c2, basis2 = lpcases.random_lp(41, 768, 1536)
p2 = ctx.simplex_problem(A2, b2, c2, basis2, True, 768)
for algo, name in [(capi.SIMPLEX_RESIDENT, "resident"), (capi.SIMPLEX_LOOKAHEAD, "lookahead")]:
    best, st = best_of(p2, algo, 5)
    print("768x1536 %s: %d pivots best solve_ms=%.3f -> %.3f us/pivot" % (name, st.pivots, best, 1e3 * best / st.pivots), flush=True)
p2.free()
# stamps
ctx.lib.lp_debug_simplex_stamps(p.h, 400, None)
p.reset()
rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
G = 32
buf = (C.c_ulonglong * (16 * G))()
ctx.lib.lp_debug_simplex_stamps(p.h, G, buf)
acc = np.array(buf[:16 * G], dtype=np.float64).reshape(G, 16) / st.pivots
names = capi.RESIDENT_STAMP_NAMES
print("stamped run: %.3f ms; cycles per pivot (mean over %d pivots), workgroup 0: communication wave %d, row wave 0 %d" %
      (st.solve_ms, st.pivots, acc[0, :5].sum(), acc[0, 6:15].sum()))
for i in range(16):
    print("  %-80s %6d   (min %5d  max %5d over workgroups)" % (names[i], acc[0, i], acc[:, i].min(), acc[:, i].max()))
print("record wait per workgroup:", " ".join("%d" % v for v in acc[:, 0]))
p.free()
ctx.close()
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
