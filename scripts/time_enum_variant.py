"""C(32,16) seed 0 full pass, the slowest of its 8 cost-balanced shards and C(28,14) on the library named by
LP_LIB_PATH: best and mean wall ms per pass, counts and optimum for comparison between builds (scripts/README.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
from simplexmethod_amd import dist as lpdist
ctx = capi.Context(0)


def timed(p, lo, hi, reps):
    p.range(lo, hi)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = p.range(lo, hi)
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, sum(ts) / len(ts) * 1e3, r


name = os.environ.get("LP_LIB_PATH", "default")
A, b, c, _ = capi.gen_lp(0, 16, 32)
p = ctx.enum_problem(A, b, c, True)
for _ in range(2):
    p.range(0, p.total)
best, mean, r = timed(p, 0, p.total, 12)
print("%s: C(32,16) full pass best %.3f mean %.3f ms wall, z=%r counts=%s" % (name, best, mean, r[1], r[2]), flush=True)
worst, worst_mean = 0.0, 0.0
for lo, hi in [lpdist.balanced_shard_bounds(32, 16, q, 8) for q in range(8)]:
    bb, mm, _ = timed(p, lo, hi, 8)
    worst, worst_mean = max(worst, bb), max(worst_mean, mm)
print("   slowest of the 8 cost-balanced shards best %.3f mean %.3f ms  (full / slowest = %.2f)" % (worst, worst_mean, best / worst), flush=True)
p.free()
A, b, c, _ = capi.gen_lp(0, 14, 28)
p = ctx.enum_problem(A, b, c, True)
best, mean, r = timed(p, 0, p.total, 12)
print("   C(28,14) full pass best %.3f mean %.3f ms wall, z=%r counts=%s" % (best, mean, r[1], r[2]), flush=True)
