"""C(32,16) seed 0 full pass (and the 8-way shards' slowest) on the library named by LP_LIB_PATH: ms per pass,
counts and optimum for comparison between builds (scripts/README.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, 16, 32)
p = ctx.enum_problem(A, b, c, True)
for _ in range(3):
    r = p.range(0, p.total)
best = 1e9
for _ in range(10):
    t0 = time.perf_counter()
    r = p.range(0, p.total)
    best = min(best, time.perf_counter() - t0)
print("%s: full pass %.3f ms wall, kernel_ms %.3f, z=%r counts=%s" % (os.environ.get("LP_LIB_PATH", "default"), best * 1e3, r[3].kernel_ms if hasattr(r[3], "kernel_ms") else -1, r[1], r[2]), flush=True)
from simplexmethod_amd import dist as lpdist
worst = 0.0
for lo, hi in [lpdist.balanced_shard_bounds(32, 16, r, 8) for r in range(8)]:
    p.range(lo, hi)
    bb = 1e9
    for _ in range(8):
        t0 = time.perf_counter()
        p.range(lo, hi)
        bb = min(bb, time.perf_counter() - t0)
    worst = max(worst, bb)
print("   slowest of the 8 cost-balanced shards %.3f ms" % (worst * 1e3), flush=True)
