import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from simplexmethod_amd import capi
ctx = capi.Context(0)
m, n = 512, 1024
A, b, c, basis = capi.gen_lp(0, m, n)
p = ctx.simplex_problem(A, b, c, basis, True, n - m)
best = 1e9
for _ in range(30):
    p.reset()
    rc, st = p.run(algo=capi.SIMPLEX_RESIDENT)
    best = min(best, st.solve_ms)
print(os.environ.get("LP_LIB_PATH", "default"), "best solve_ms %.4f -> %.3f us/pivot" % (best, 1e3 * best / st.pivots))
