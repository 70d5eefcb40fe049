"""One-launch-per-pivot path with the selection overlapped (LP_SIMPLEX_ALGO_OVERLAP): us per pivot and the rate of the
algorithmic bytes 16 * m * (n + 1) against the launch-pair and the look-ahead path, first 300 pivots, best of 3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
shapes = [(1024, 2048), (1536, 3072), (2048, 4096), (2048, 8192), (3072, 6144), (4096, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for m, n in shapes:
    A, b, c, basis = capi.gen_lp(0, m, n)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    for name, algo in (("overlap", capi.SIMPLEX_OVERLAP), ("launch", capi.SIMPLEX_LAUNCH), ("lookahead", capi.SIMPLEX_LOOKAHEAD)):
        best = 1e9
        for _ in range(3):
            p.reset()
            try:
                rc, st = p.run(algo=algo, max_iter=300)
            except capi.LPError as ex:
                print("   %s: %s" % (name, ex))
                break
            best = min(best, st.solve_ms)
        if best == 1e9:
            continue
        us = 1e3 * best / max(st.pivots, 1)
        print("%5d x %5d %-8s rc=%d pivots=%4d  %8.3f us/pivot  %6.2f TB/s (%.3f of 8 TB/s)" %
              (m, n, name, rc, st.pivots, us, 16.0 * m * (n + 1) / us / 1e6, 16.0 * m * (n + 1) / us / 8e6), flush=True)
    ms = min(p.bench_update(m // 3, n // 5, 50) for _ in range(3))
    print("%5d x %5d rank-1 update kernel alone %8.3f us/launch  %6.2f TB/s" % (m, n, ms * 1e3, 16.0 * m * (n + 1) / (ms * 1e3) / 1e6), flush=True)
    p.free()
