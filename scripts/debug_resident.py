"""Diagnostic: repeated chip-resident solves in the placement-dependent forms (participants spread over all XCDs,
write-through stores forced); counts hand-off timeouts (LP_RESIDENT_DEBUG=1: the library dumps the pricing-record area)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LP_RESIDENT_STRICT"] = "1"
from simplexmethod_amd import capi
from tests import lpcases
ctx = capi.Context(0)
probs = []
for seed, m, n in [(0, 512, 1024)]:
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    probs.append((m, n, ctx.simplex_problem(A, b, c, basis, True, n - m)))
for mode in ([], ["LP_RESIDENT_SPREAD"], ["LP_RESIDENT_FORCE_SC1"], ["LP_RESIDENT_SPREAD", "LP_RESIDENT_FORCE_SC1"]):
    for v in mode:
        os.environ[v] = "1"
    for m, n, p in probs:
        fails, best, msgs = 0, 1e9, set()
        for rep in range(12):
            p.reset()
            try:
                rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
                best = min(best, st.solve_ms)
            except capi.LPError as e:
                fails += 1
                msgs.add(str(e)[58:110])
        print("%-50s %dx%d: %2d/12 timeouts, best %.4f ms  %s" % ("+".join(mode) or "default", m, n, fails, best, sorted(msgs)[:3]), flush=True)
    for v in mode:
        del os.environ[v]
