"""Per-pivot time of the chip-resident kernel against the number of row waves per workgroup (G = 32 workgroups
throughout: n = 1024): is the wave-uniform work of a pivot slowed by two row waves sharing a SIMD?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LP_RESIDENT_STRICT"] = "1"
from simplexmethod_amd import capi
from tests import lpcases
ctx = capi.Context(0)
for m in (64, 128, 192, 256, 320, 384, 448, 512):
    n = 1024
    A, b, c, basis = lpcases.random_lp(100 + m, m, n)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    best = 1e9
    for _ in range(8):
        p.reset()
        rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
        best = min(best, st.update_ms)
    # the same launch with the iteration limit at half the pivots: the difference is pure pivots
    half = st.pivots // 2
    bh = 1e9
    for _ in range(8):
        p.reset()
        try:
            rc2, st2 = p.run(algo=capi.SIMPLEX_RESIDENT, max_iter=half)
        except capi.LPError:
            pass
        bh = min(bh, ctx_stats.update_ms) if False else bh
    print("m=%3d (%d row waves): %4d pivots, kernel %.4f ms -> %.3f us per pivot (incl. load/store of the tableau)" %
          (m, (m + 63) // 64, st.pivots, best, 1e3 * best / st.pivots), flush=True)
    p.free()
