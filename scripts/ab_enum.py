"""A/B timing of the C(32,16) enumeration for the library named by LP_LIB_PATH."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
m, n = 16, 32
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, m, n)
p = ctx.enum_problem(A, b, c, True)
p.range(0, p.total)
best = 1e9
for _ in range(5):
    rc, z, counts, st = p.range(0, p.total)
    best = min(best, st.kernel_ms)
print(os.environ.get("LP_LIB_PATH", "default"), "kernel_ms %.3f" % best, "z", z, "counts", list(counts))
