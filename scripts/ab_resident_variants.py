"""Diagnostic builds of simplex_resident.hip with -D switches (scripts/ab_resident_variants.py build "NAME=-DFLAG ..."),
and best-of-8 kernel time per pivot at 512 x 1024 for each library in ab_libs/ (run; results may be wrong for
experiments that break the algorithm: the status and pivot count are printed)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "ab_libs")


def build(specs):
    from simplexmethod_amd import build as b
    b.build_hip()
    os.makedirs(AB, exist_ok=True)
    obj_dir = os.path.join(b.OUT, "obj")
    flags = [f for f in b.HIPCC_FLAGS if f != "-shared"]
    for spec in specs:
        name, _, defs = spec.partition("=")
        name, _, stem = name.partition("@")      # NAME@simplex_overlap=-D... builds a variant of another source file
        stem = stem or "simplex_resident"
        others = [os.path.join(obj_dir, f) for f in sorted(os.listdir(obj_dir)) if f.endswith(".o") and f != stem + ".o"]
        obj = os.path.join(AB, "var_%s.o" % name)
        lib = os.path.join(AB, "libvar_%s.so" % name)
        subprocess.run([b.hipcc_path()] + flags + defs.split() + ["-c", "-o", obj, os.path.join(b.CSRC, stem + ".hip")], check=True)
        subprocess.run([b.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others + ["-ldl", "-pthread"], check=True)
        os.remove(obj)
        print("built", lib, flush=True)


def run():
    code = ("import os,sys; sys.path.insert(0, %r); from simplexmethod_amd import capi; ctx = capi.Context(0); "
            "A,b,c,basis = capi.gen_lp(0,512,1024); p = ctx.simplex_problem(A,b,c,basis,True,512); best=1e9\n"
            "for _ in range(8):\n    p.reset(); rc, st = p.run(algo=capi.SIMPLEX_RESIDENT, max_iter=345); best=min(best, st.update_ms)\n"
            "print('rc %%d pivots %%d best kernel %%.4f ms = %%.3f us/pivot' %% (rc, st.pivots, best, 1e3*best/max(st.pivots,1)))" % ROOT)
    libs = [None] + sorted(f for f in os.listdir(AB) if f.startswith("libvar_"))
    for lib in libs:
        env = dict(os.environ, LP_RESIDENT_STRICT="1")
        if lib:
            env["LP_LIB_PATH"] = os.path.join(AB, lib)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        print("%-28s %s %s" % (lib or "(product)", r.stdout.strip(), r.stderr.strip()[-200:]), flush=True)


if __name__ == "__main__":
    build(sys.argv[2:]) if sys.argv[1:2] == ["build"] else run()
