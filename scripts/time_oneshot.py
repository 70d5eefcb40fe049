"""Where the one-shot lp_simplex_solve spends its time at 512 x 1024: upload / run / download / free."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi

ctx = capi.Context(0)
m, n = 512, 1024
A, b, c, basis = capi.gen_lp(0, m, n)
ctx.simplex_solve(A, b, c, basis, True, n - m)
best = [1e9] * 5
for _ in range(5):
    t0 = time.perf_counter()
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    t1 = time.perf_counter()
    rc, st = p.run()
    t2 = time.perf_counter()
    d = p.download()
    t3 = time.perf_counter()
    p.free()
    t4 = time.perf_counter()
    r = ctx.simplex_solve(A, b, c, basis, True, n - m)
    t5 = time.perf_counter()
    cur = [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
    best = [min(a, b_) for a, b_ in zip(best, cur)]
print("upload %.3f ms | run %.3f ms (events %.3f) | download %.3f ms | free %.3f ms | one-shot lp_simplex_solve %.3f ms" %
      (best[0] * 1e3, best[1] * 1e3, st.solve_ms, best[2] * 1e3, best[3] * 1e3, best[4] * 1e3))
