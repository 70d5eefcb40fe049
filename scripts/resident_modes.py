"""The chip-resident simplex under its placement-independent modes: default (one XCD, plain stores),
write-through stores forced, participants spread over all XCDs, and both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simplexmethod_amd import capi
from tests import lpcases
ctx = capi.Context(0)
cases = [(5, 128, 256), (4, 64, 128), (0, 512, 1024), (6, 100, 1500)]
for env in [{}, {"LP_RESIDENT_FORCE_SC1": "1"}, {"LP_RESIDENT_SPREAD": "1"}, {"LP_RESIDENT_SPREAD": "1", "LP_RESIDENT_FORCE_SC1": "1"}]:
    for k in ("LP_RESIDENT_FORCE_SC1", "LP_RESIDENT_SPREAD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for seed, m, n in cases:
        A, b, c, basis = lpcases.random_lp(seed, m, n)
        p = ctx.simplex_problem(A, b, c, basis, True, n - m)
        rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
        print(env, seed, m, n, "rc", rc, "pivots", st.pivots, "launches", st.launches, "ms %.3f" % st.solve_ms,
              ctx.error() if st.launches != 2 else "", flush=True)
        p.free()
