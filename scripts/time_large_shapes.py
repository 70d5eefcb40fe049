"""Shapes beyond the chip-resident path (m > 960): us per pivot of the look-ahead and the one-launch-pair-per-pivot
paths, and the rate of the rank-1 update's algorithmic bytes, 16 * m * (n + 1) per pivot (read + write of the tableau)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
shapes = [(512, 1024), (1024, 2048), (1536, 3072), (2048, 4096), (3072, 6144), (4096, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for m, n in shapes:
    A, b, c, basis = capi.gen_lp(0, m, n)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    for name, algo in (("lookahead", capi.SIMPLEX_LOOKAHEAD), ("launch", capi.SIMPLEX_LAUNCH)):
        best = 1e9
        for _ in range(3):
            p.reset()
            try:
                rc, st = p.run(algo=algo, max_iter=400)
            except capi.LPError as e:
                print("   %s: %s" % (name, e))
                break
            best = min(best, st.solve_ms)
        us = 1e3 * best / max(st.pivots, 1)
        print("%5d x %5d %-9s rc=%d pivots=%4d  %8.3f us/pivot  %6.2f TB/s of 16*m*(n+1) bytes" %
              (m, n, name, rc, st.pivots, us, 16.0 * m * (n + 1) / us / 1e6), flush=True)
    ms = p.bench_update(m // 3, n // 5, 50)
    print("%5d x %5d rank-1 update kernel alone %8.3f us/launch  %6.2f TB/s" % (m, n, ms * 1e3, 16.0 * m * (n + 1) / (ms * 1e3) / 1e6), flush=True)
    try:
        msj, piv = p.bench_update_rankj(50)
        print("%5d x %5d rank-J update kernel alone (J=%d) %8.3f us/launch  %6.2f TB/s" % (m, n, piv, msj * 1e3, 16.0 * m * (n + 1) / (msj * 1e3) / 1e6), flush=True)
    except capi.LPError as e:
        print("   rank-J:", e)
    p.free()
