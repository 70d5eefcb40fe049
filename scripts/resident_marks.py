"""Interval timing of the chip-resident simplex kernel without disturbing it: for every (a, b) pair of marks a
diagnostic build of the library (-DRS_MARK_A=a -DRS_MARK_B=b: two clock reads per pivot) is linked into
ab_libs/, and each is run on 512 x 1024.

  python scripts/resident_marks.py build         (here: hipcc cross-compiles, a few at a time)
  python scripts/resident_marks.py run           (on the GPU box: prints cycles per pivot of every interval)

Marks of the communication wave: 0 loop top (polling starts), 1 all pricing records fresh, 2 winner decided,
3 the winner's slice records fresh, 4 decision block written, 5 after the decision barrier.
Marks of row wave 0: 10 after the decision barrier, 11 decision read and pivot-row read issued, 16 quotients,
12 reduced costs, 13 priced and pricing record stored, 14 entering column arrived / eta entry, 15 candidate column
and slice record stored, 17 rank-1 update and mirror write done."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PAIRS = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0),
         (10, 11), (11, 16), (16, 12), (12, 13), (13, 14), (14, 15), (15, 17), (17, 10)]
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


NAMES = {(0, 1): "comm: poll until all pricing records are fresh", (1, 2): "comm: pricing decision (winner)",
         (2, 3): "comm: poll the winner's slice records (the hop on the critical path)",
         (3, 4): "comm: combine the slices, write the decision block", (4, 5): "comm: decision barrier", (5, 0): "comm: basis, trace, loop",
         (10, 11): "rows: read the decision block, column request, pivot-row read issued", (11, 16): "rows: quotients by the pivot element",
         (16, 12): "rows: pivot-row entries arrived, reduced costs", (12, 13): "rows: pricing, pricing record out",
         (13, 14): "rows: entering column arrived, eta entry, xB", (14, 15): "rows: candidate column, ratio, slice record and column out",
         (15, 17): "rows: rank-1 update, mirror write", (17, 10): "rows: wait for the next decision"}


def run():
    import json, re
    import bench
    out = {"kernel_source_hash": bench.kernel_source_hash(), "workload": "m=512 n=1024 seed 0, 345 pivots, default form",
           "what": "cycles per pivot between two marks of the communication wave (marks 0-9) or of row wave 0 (10-29), mean over "
                   "the workgroups; each interval from its own diagnostic build (two clock reads per pivot)", "intervals": {}}
    code = ("import os,sys; sys.path.insert(0, %r); from simplexmethod_amd import capi; ctx = capi.Context(0); "
            "A,b,c,basis = capi.gen_lp(0,512,1024); p = ctx.simplex_problem(A,b,c,basis,True,512); "
            "[ (p.reset(), p.run(algo=capi.SIMPLEX_RESIDENT)) for _ in range(3)]; "
            "p.reset(); rc, st = p.run(algo=capi.SIMPLEX_RESIDENT); print('solve_ms %%.4f pivots %%d' %% (st.solve_ms, st.pivots))" % ROOT)
    for a, c in PAIRS:
        env = dict(os.environ, LP_LIB_PATH=os.path.join(AB, "libmarks_%d_%d.so" % (a, c)), LP_RESIDENT_STRICT="1")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        last = (r.stderr.strip().splitlines() or ["?"])[-1]
        print("%2d -> %2d: %s | %s" % (a, c, last, r.stdout.strip()), flush=True)
        mm = re.search(r"mean (\d+)", last)
        if mm:
            out["intervals"]["%d->%d %s" % (a, c, NAMES.get((a, c), ""))] = int(mm.group(1))
    comm = [v for k2, v in out["intervals"].items() if int(k2.split("->")[0]) < 10]
    out["critical_path_cycles_per_pivot"] = sum(comm) if comm else None
    path = os.path.join(ROOT, "gpurun_out", "resident_marks.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
