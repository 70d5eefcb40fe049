import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from simplexmethod_amd import capi
ctx = capi.Context(0)
for m, n in [(18, 30), (20, 32), (24, 34), (12, 32), (16, 32)]:
    A, b, c, _ = capi.gen_lp(5, m, n)
    p = ctx.enum_problem(A, b, c, True)
    for lg in (8, 10, 12, 14, 16, 18):
        cnt = min(1 << lg, p.total)
        lo = p.total // 3
        res = {}
        for name, algo in (("prefix", capi.ENUM_PREFIX), ("direct", capi.ENUM_DIRECT)):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); r = p.range(lo, lo + cnt, algo); best = min(best, time.perf_counter() - t0)
            res[name] = best
        print(f"C({n},{m}) 2^{lg}: prefix {1e3*res['prefix']:.3f} ms  direct {1e3*res['direct']:.3f} ms", flush=True)
    p.free()
