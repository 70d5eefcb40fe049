"""Diagnostic: where does a look-ahead selector pivot spend its cycles? (GPU box only)"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simplexmethod_amd import capi

m, n = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = capi.Context(0)
A, b, c, basis = capi.gen_lp(0, m, n)
p = ctx.simplex_problem(A, b, c, basis, True, n - m)
p.run(algo=capi.SIMPLEX_LOOKAHEAD)
p.reset()
cap = 400
ctx.lib.lp_debug_simplex_stamps(p.h, cap, None)
rc, st = p.run(algo=capi.SIMPLEX_LOOKAHEAD)
out = np.zeros(cap * 8, dtype=np.uint64)
ctx.lib.lp_debug_simplex_stamps(p.h, cap, out.ctypes.data_as(C.POINTER(C.c_uint64)))
s = out.reshape(cap, 8)[:min(cap, st.pivots)].astype(np.int64)
dt = np.diff(s, axis=1)
names = ["pricing chain", "column e (HBM+etas)", "ratio chain", "eta col + xB (under the row read)", "pivot row (etas)+d", "bookkeeping", "barrier"]
print("pivots", st.pivots, "solve_ms", st.solve_ms, "us/pivot", 1e3 * st.solve_ms / st.pivots)
print("median ticks per phase (s_memtime, 100 MHz?):")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {np.median(dt[:, k]):10.0f}  mean {dt[:, k].mean():10.0f}")
print("  total per pivot              ", np.median(s[:, 7] - s[:, 0]))
