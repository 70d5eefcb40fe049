"""Per-wave timeline of the leaf kernels (diagnostic build -DLP_LEAF_WAVELOG of enum_leaf.hip, named by LP_LIB_PATH):
every wave's start, first item and end on the 100 MHz clock, its items and steals, and where it ran.
  python scripts/ab_resident_variants.py build "wavelog@enum_leaf=-DLP_LEAF_WAVELOG"
  LP_LIB_PATH=ab_libs/libvar_wavelog.so python scripts/leaf_wavelog.py            # C(28,14); LP_SHAPE=16,32 LP_SHARD=3
Prints, per kernel, the distribution of start / first item / end (us after the first wave's start), busy time,
and per CU-slot idle time between the kernels."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi, dist as lpdist
ctx = capi.Context(0)
m, n = (int(v) for v in os.environ.get("LP_SHAPE", "14,28").split(","))
A, b, c, _ = capi.gen_lp(0, m, n)
p = ctx.enum_problem(A, b, c, True)
r = int(os.environ.get("LP_SHARD", "-1"))
lo, hi = lpdist.balanced_shard_bounds(n, m, r, 8) if r >= 0 else (0, p.total)
L = capi.load()
L.lp_debug_leaf_wavelog.restype = C.c_int
L.lp_debug_leaf_wavelog.argtypes = [C.c_void_p, C.c_int]
for _ in range(3):
    p.range(lo, hi)
L.lp_debug_leaf_wavelog(None, 0)   # empty the log
rc, z, counts, st = p.range(lo, hi)
buf = np.zeros((1 << 16, 6), dtype=np.uint64)
k = L.lp_debug_leaf_wavelog(buf.ctypes.data, 1 << 16)
w = buf[:k]
kern = (w[:, 0] & 0xFF).astype(int)
t0 = w[:, 2].astype(np.int64); t1 = w[:, 3].astype(np.int64); t2 = w[:, 4].astype(np.int64)
base = t0.min()
us = lambda t: (t - base) / 100.0
items = (w[:, 5] & 0xFFFFFFFF).astype(int); steals = (w[:, 5] >> 32).astype(int)
print("pass: %.3f ms kernel time, %d waves logged, counts %s" % (st.kernel_ms, k, list(counts)))
q = lambda a: "min %7.1f  p10 %7.1f  p50 %7.1f  p90 %7.1f  max %7.1f  mean %7.1f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max(), a.mean())
for kk, name in ((3, "k_enum_leaves<3>"), (1, "k_enum_leaves<1>"), (7, "k_enum_thin")):
    s = kern == kk
    if not s.any():
        continue
    print("%s: %d waves, items %d (per wave min %d max %d), steals %d" % (name, s.sum(), items[s].sum(), items[s].min(), items[s].max(), steals[s].sum()))
    print("   start      ", q(us(t0[s])))
    if kk != 7:
        print("   first item ", q(us(t1[s][items[s] > 0])))
    print("   end        ", q(us(t2[s])))
    print("   wave-time (end - start) ", q((t2[s] - t0[s]) / 100.0))
# slot occupancy: sum of wave-times of the two leaf kernels against 12 slots x 256 CUs x the span
leaf = (kern == 3) | (kern == 1)
span = us(t2[leaf]).max() - us(t0[leaf]).min()
print("leaf kernels: span %.1f us; sum of wave-times / (3072 slots x span) = %.3f" % (span, ((t2[leaf] - t0[leaf]) / 100.0).sum() / (3072 * span)))
# per workgroup of <3>: spread of its waves' ends
s3 = kern == 3
blk = (w[:, 0] >> 8).astype(int)
if s3.any():
    ends = {}
    for bkey, e in zip(blk[s3], us(t2[s3])):
        ends.setdefault(bkey, []).append(e)
    spread = np.array([max(v) - min(v) for v in ends.values()])
    lastm = np.array([max(v) - np.mean(v) for v in ends.values()])
    print("<3> per workgroup: spread of its waves' ends", q(spread), "; last - mean", q(lastm))
p.free()
