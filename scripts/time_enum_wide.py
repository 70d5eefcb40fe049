"""Wide enumeration shapes (m > 16 or n - m > 16): shared-prefix path (general leaf kernel) against the
direct kernel — wall time of pass 1 + tie rule, counts, optimum.

    python scripts/time_enum_wide.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from simplexmethod_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(0)
    shapes = [(18, 30), (12, 32), (20, 32), (10, 36), (16, 34), (24, 32)]
    if len(sys.argv) >= 3:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    for m, n in shapes:
        A, b, c, _ = capi.gen_lp(5, m, n)
        p = ctx.enum_problem(A, b, c, True)
        res = {}
        for name, algo in (("prefix", capi.ENUM_PREFIX), ("direct", capi.ENUM_DIRECT)):
            if name == "direct" and p.total > 3_000_000_000:
                continue
            best = 1e9
            for rep in range(2):
                t0 = time.perf_counter()
                rc, z, counts, st = p.range(0, p.total, algo)
                k = p.first_within(0, p.total, z) if rc == 0 else None
                best = min(best, time.perf_counter() - t0)
            res[name] = (rc, z, counts, k)
            print(f"C({n},{m}) = {p.total:>13d}  {name}: {1e3 * best:10.3f} ms  {p.total / best / 1e9:7.3f} G subsets/s  "
                  f"launches {st.launches}  rc={rc} z={z!r} rank={k} counts={counts}", flush=True)
        if len(res) == 2:
            print("    same answers:", res["prefix"] == res["direct"], flush=True)
        p.free()


if __name__ == "__main__":
    main()
