"""Interval timing of the chip-resident simplex kernel without disturbing it: for every (a, b) pair of marks a
diagnostic build of the library (-DRS_MARK_A=a -DRS_MARK_B=b: two clock reads per pivot) is linked into
ab_libs/, and each is run on 512 x 1024.

  python scripts/resident_marks.py build         (here: hipcc cross-compiles, a few at a time)
  python scripts/resident_marks.py run           (on the GPU box: prints cycles per pivot of every interval)

Marks of the communication wave: 0 loop top (poll starts), 1 all records fresh, 2 decision block written,
3 after the decision barrier, 4 after the pivot-row barrier, 5 after the ratio barrier, 6 record stored,
7 after the publication barrier.  Marks of row wave 0: 10 after the decision barrier, 11 decision read,
12 before / 13 after the pivot-row barrier, 14 priced, 15 eta entry known, 16 before / 17 after the ratio
barrier, 18 before / 19 after the publication barrier, 20 loop end."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PAIRS = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 0), (0, 0),
         (10, 11), (11, 12), (12, 13), (13, 14), (14, 15), (15, 16), (16, 17), (17, 18), (18, 19), (19, 20), (20, 10),
         (5, 30), (30, 31), (31, 32), (32, 6), (1, 34), (34, 2)]
if os.environ.get("LP_MARK_PAIRS"):   # e.g. LP_MARK_PAIRS=5-30,30-31
    PAIRS = [tuple(int(v) for v in t.split("-")) for t in os.environ["LP_MARK_PAIRS"].split(",")]
AB = os.path.join(ROOT, "ab_libs")


def build():
    from simplexmethod_amd import build as b
    b.build_hip()
    os.makedirs(AB, exist_ok=True)
    obj_dir = os.path.join(b.OUT, "obj")
    others = [os.path.join(obj_dir, f) for f in sorted(os.listdir(obj_dir)) if f.endswith(".o") and f != "simplex_resident.o"]
    flags = [f for f in b.HIPCC_FLAGS if f != "-shared"]

    def one(pair):
        a, c = pair
        obj = os.path.join(AB, "resident_%d_%d.o" % (a, c))
        lib = os.path.join(AB, "libmarks_%d_%d.so" % (a, c))
        subprocess.run([b.hipcc_path()] + flags + ["-DRS_MARK_A=%d" % a, "-DRS_MARK_B=%d" % c, "-c", "-o", obj,
                        os.path.join(b.CSRC, "simplex_resident.hip")], check=True)
        subprocess.run([b.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others + ["-ldl", "-pthread"], check=True)
        os.remove(obj)
        return lib
    with ThreadPoolExecutor(6) as pool:
        for lib in pool.map(one, PAIRS):
            print("built", lib, flush=True)


def run():
    code = ("import os,sys; sys.path.insert(0, %r); from simplexmethod_amd import capi; ctx = capi.Context(0); "
            "A,b,c,basis = capi.gen_lp(0,512,1024); p = ctx.simplex_problem(A,b,c,basis,True,512); "
            "[ (p.reset(), p.run(algo=capi.SIMPLEX_RESIDENT)) for _ in range(3)]; os.environ['LP_RESIDENT_MARKS']='1'; "
            "p.reset(); rc, st = p.run(algo=capi.SIMPLEX_RESIDENT); print('solve_ms %%.4f pivots %%d' %% (st.solve_ms, st.pivots))" % ROOT)
    for a, c in PAIRS:
        env = dict(os.environ, LP_LIB_PATH=os.path.join(AB, "libmarks_%d_%d.so" % (a, c)), LP_RESIDENT_STRICT="1")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        print("%2d -> %2d: %s | %s" % (a, c, (r.stderr.strip().splitlines() or ["?"])[-1], r.stdout.strip()), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
