"""One shard of the 8-way cost-balanced cut of C(32,16) (LP_SHARD, default 3), five times: run under
`rocprofv3 --kernel-trace` and read with scripts/shard_timeline.py for the shard's kernel timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi, dist as lpdist
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, 16, 32)
p = ctx.enum_problem(A, b, c, True)
r = int(os.environ.get("LP_SHARD", "3"))
lo, hi = lpdist.balanced_shard_bounds(32, 16, r, 8)
for _ in range(5):
    rc, z, counts, st = p.range(lo, hi)
print(rc, z, counts, st.kernel_ms)
