"""One bench step of the enumeration (EnumProblem.solve_sharded on the whole range: pass 1, exchange, tie rule,
vertex), five times, for `rocprofv3 --kernel-trace`; scripts/shard_timeline.py reads the last one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
ctx = capi.Context(0)
A, b, c, _ = capi.gen_lp(0, 16, 32)
p = ctx.enum_problem(A, b, c, True)
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    r = p.solve_sharded(None, 16, want_vertex=True)
    ts.append(time.perf_counter() - t0)
print("step wall ms:", [round(1e3 * t, 3) for t in ts], r["rank"], r["counts"])
