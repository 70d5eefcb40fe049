"""One shard of the 8-way cost-balanced cut of C(32,16) (LP_SHARD, default 3), five times: run under
`rocprofv3 --kernel-trace` and read with scripts/shard_timeline.py for the shard's kernel timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi, dist as lpdist
ctx = capi.Context(0)
m, n = (int(v) for v in os.environ.get("LP_SHAPE", "16,32").split(","))   # LP_SHAPE=14,28 LP_SHARD=-1: all of C(28,14)
A, b, c, _ = capi.gen_lp(0, m, n)
p = ctx.enum_problem(A, b, c, True)
r = int(os.environ.get("LP_SHARD", "3"))
lo, hi = lpdist.balanced_shard_bounds(n, m, r, 8) if r >= 0 else (0, p.total)
for _ in range(5):
    rc, z, counts, st = p.range(lo, hi)
print(rc, z, counts, st.kernel_ms)
