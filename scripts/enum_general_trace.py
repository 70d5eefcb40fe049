import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from simplexmethod_amd import capi
ctx = capi.Context(0)
for m, n in [(16, 32), (18, 30), (12, 32)]:
    A, b, c, _ = capi.gen_lp(5, m, n)
    p = ctx.enum_problem(A, b, c, True)
    for _ in range(2):
        rc, z, counts, st = p.range(0, p.total, capi.ENUM_PREFIX)
    print(m, n, st.kernel_ms, st.launches)
    p.free()
