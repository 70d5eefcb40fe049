import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import pyoracle as o
from simplexmethod_amd import capi
from tests import lpcases
ctx = capi.Context(0)
ok = True
for seed, m, n in [(0, 2, 5), (1, 8, 16), (3, 33, 71), (5, 128, 256), (6, 100, 1500), (0, 512, 1024), (44, 513, 700), (7, 1000, 1100)]:
    A, b, c, basis = lpcases.random_lp(seed, m, n)
    r = o.simplex_tableau(A, b, c, basis, True, n - m, trace_cap=1 << 14, want_tableau=True)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    rc, st = p.run(algo=capi.SIMPLEX_OVERLAP)
    g = p.download(trace_cap=max(st.pivots, 1), want_tableau=True)
    same = (rc == r["status"] and st.pivots == r["iters"] and np.array_equal(g["basis"], r["basis"])
            and np.array_equal(g["tableau"], r["tableau"]) and np.array_equal(g["x"], r["x"]))
    k = r["iters"]
    tr = list(zip(g["trace_enter"][:k].tolist(), g["trace_leave"][:k].tolist())) == r["trace"][:k]
    print("seed %d %dx%d rc=%d pivots=%d (oracle %d) algo_used=%d launches=%d %.3f us/pivot bit-exact=%s trace=%s" % (seed, m, n, rc, st.pivots, r["iters"], st.algo_used, st.launches, 1e3*st.solve_ms/max(st.pivots,1), same, tr), flush=True)
    ok &= same and tr
    # iteration limit (odd and even) leaves the same tableau as the launch path
    for lim in (1, 2, 7):
        if lim >= r["iters"]: continue
        p.reset(); rc1, st1 = p.run(algo=capi.SIMPLEX_OVERLAP, max_iter=lim); g1 = p.download(want_tableau=True)
        p.reset(); rc2, st2 = p.run(algo=capi.SIMPLEX_LAUNCH, max_iter=lim); g2 = p.download(want_tableau=True)
        e = rc1 == rc2 and st1.pivots == st2.pivots == lim and np.array_equal(g1["tableau"], g2["tableau"]) and np.array_equal(g1["basis"], g2["basis"])
        ok &= e
        if not e: print("   iteration limit", lim, "MISMATCH", rc1, rc2, st1.pivots, st2.pivots)
    p.free()
print("ALL OK" if ok else "FAILED")
for m, n in [(512, 1024), (1024, 2048), (2048, 4096), (4096, 8192)]:
    A, b, c, basis = capi.gen_lp(0, m, n)
    p = ctx.simplex_problem(A, b, c, basis, True, n - m)
    res = {}
    for name, algo in (("overlap", capi.SIMPLEX_OVERLAP), ("launch", capi.SIMPLEX_LAUNCH), ("lookahead", capi.SIMPLEX_LOOKAHEAD)):
        best = 1e9
        try:
            for _ in range(3):
                p.reset()
                rc, st = p.run(algo=algo, max_iter=300)
                best = min(best, st.solve_ms)
        except capi.LPError as e:
            print("  ", name, e); continue
        g = p.download(want_tableau=True)
        res[name] = g["tableau"]
        us = 1e3 * best / max(st.pivots, 1)
        print("%5d x %5d %-9s rc=%d pivots=%4d  %8.3f us/pivot  %6.2f TB/s" % (m, n, name, rc, st.pivots, us, 16.0 * m * (n + 1) / us / 1e6), flush=True)
    print("   same tableau as launch:", np.array_equal(res["overlap"], res["launch"]))
    p.free()
