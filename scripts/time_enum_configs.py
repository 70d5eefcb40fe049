"""BASELINE configs[2] (C(28,14)) and configs[3] (C(32,16)): device-resident pass time and the
one-shot host-buffer entry point (upload + tables + level buffers + solve + vertex)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
for m, n in [(14, 28), (16, 32)]:
    A, b, c, _ = capi.gen_lp(0, m, n)
    p = ctx.enum_problem(A, b, c, True)
    p.range(0, p.total)
    best = min(p.range(0, p.total)[3].kernel_ms for _ in range(5))
    t0 = time.perf_counter(); rc, z, counts, st = p.range(0, p.total); k = p.first_within(0, p.total, z); t1 = time.perf_counter()
    p.free()
    w = []
    for _ in range(3):
        t2 = time.perf_counter(); r = ctx.enum_solve(A, b, c, True, n - m); w.append(time.perf_counter() - t2)
    print(f"C({n},{m}) = {p.total}: kernels {best:.3f} ms ({p.total / best / 1e6:.1f} G subsets/s), pass1+pass2 wall {1e3 * (t1 - t0):.3f} ms, "
          f"one-shot lp_enum_solve {1e3 * min(w):.2f} ms (first {1e3 * w[0]:.2f}), rank {r['rank']} obj {r['obj']:.6f}")
