"""Scan dist.RECORD_COST: max / mean shard time of the 8-way cut of C(32,16) for several values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi, dist as lpdist

m, n = 16, 32
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, m, n)
p = ctx.enum_problem(A, b, c, True)
p.range(0, p.total)
for rc_cost in [int(x) for x in sys.argv[1:]] or [60, 100, 140, 180, 230]:
    for parts in (2, 4, 8):
        times = []
        for r in range(parts):
            lo, hi = lpdist.balanced_shard_bounds(n, m, r, parts, record_cost=rc_cost)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                rc, z, counts, st = p.range(lo, hi)
                if rc == 0:
                    p.first_within(lo, hi, z)
                best = min(best, time.perf_counter() - t0)
            times.append(best * 1e3)
        print(f"cost {rc_cost} N={parts}: {[round(t, 2) for t in times]} max {max(times):.2f} mean {sum(times)/parts:.2f}", flush=True)
