"""One shard of an 8-way cut of C(32,16), 5 passes: workload for a rocprofv3 --kernel-trace run
(per-kernel times and the gaps between them inside a shard's pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi, dist as lpdist
m, n = 16, 32
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, m, n)
p = ctx.enum_problem(A, b, c, True)
shard = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lo, hi = lpdist.balanced_shard_bounds(n, m, shard, 8)
for _ in range(5):
    rc, z, counts, st = p.range(lo, hi)
    k = p.first_within(lo, hi, z)
print("shard", shard, lo, hi, "kernel ms", st.kernel_ms, "launches", st.launches)
