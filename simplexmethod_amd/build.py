"""Builds the in-tree native libraries (hipcc for gfx950, g++ for the host classes).

The built files live under simplexmethod_amd/_build/ (git-ignored, but they travel
to the GPU box with the repo snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
HOST = os.path.join(_HERE, "host")
OUT = os.path.join(_HERE, "_build")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

HIP_LIB = os.path.join(OUT, "libsimplexmethod_hip.so")
HOST_LIB = os.path.join(OUT, "libsimplexmethod_host.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP hot path cannot be built")


def build_hip(force=False, verbose=False):
    """Every csrc/*.hip -> _build/obj/<name>.o (only the stale ones, a few at a time), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUT, exist_ok=True)
    obj_dir = os.path.join(OUT, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    srcs = _sources(CSRC, (".hip",))
    hdrs = _sources(CSRC, (".hpp",)) + _sources(INCLUDE, (".h",))
    extra = os.environ.get("LP_HIPCC_EXTRA", "").split()
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    # the flags are part of an object's identity (LP_HIPCC_EXTRA builds must not reuse plain objects)
    stamp = os.path.join(obj_dir, "flags.txt")
    flag_text = " ".join(flags + extra)
    if not os.path.exists(stamp) or open(stamp).read() != flag_text:
        force = True
    objs, todo = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + hdrs):
            todo.append((src, obj))
    for stale in set(_sources(obj_dir, (".o",))) - set(objs):
        os.remove(stale)
    if not todo and _newer(HIP_LIB, objs):
        return HIP_LIB
    hipcc = hipcc_path()

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + flags + extra + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    workers = max(1, min(len(todo), int(os.environ.get("LP_BUILD_JOBS", "0")) or min(6, os.cpu_count() or 1)))
    if todo:
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(compile_one, todo))
    with open(stamp, "w") as f:
        f.write(flag_text)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs + ["-ldl", "-pthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return HIP_LIB


def build_host(force=False, verbose=False):
    """C++ mirror of the reference's problem classes + Solver/EnumerationSolver wrappers."""
    os.makedirs(OUT, exist_ok=True)
    if not os.path.isdir(HOST):
        return None
    srcs = _sources(HOST, (".cpp",))
    if not srcs:
        return None
    deps = srcs + _sources(HOST, (".h",)) + _sources(INCLUDE, (".h",))
    if not force and _newer(HOST_LIB, deps):
        return HOST_LIB
    build_hip(force=False, verbose=verbose)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", INCLUDE,
           "-I", HOST, "-o", HOST_LIB] + srcs + ["-L", OUT, "-lsimplexmethod_hip",
                                                  "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return HOST_LIB


TESTS_CPP = os.path.join(os.path.dirname(_HERE), "tests", "cpp")
TESTS_OUT = os.path.join(TESTS_CPP, "_build")


def build_cpp_tests(force=False, verbose=False):
    """tests/cpp/*.cpp -> tests/cpp/_build/<name> (linked against the host + HIP libraries)."""
    host = build_host(force, verbose)
    if host is None or not os.path.isdir(TESTS_CPP):
        return []
    os.makedirs(TESTS_OUT, exist_ok=True)
    outs = []
    for src in _sources(TESTS_CPP, (".cpp",)):
        exe = os.path.join(TESTS_OUT, os.path.splitext(os.path.basename(src))[0])
        deps = [src, host, HIP_LIB] + _sources(TESTS_CPP, (".h",)) + _sources(HOST, (".h",))
        if force or not _newer(exe, deps):
            cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-DLP_HOST_TEST_HOOKS", "-I", INCLUDE, "-I", HOST, "-I", TESTS_CPP,
                   "-o", exe, src, "-L", OUT, "-lsimplexmethod_host", "-lsimplexmethod_hip",
                   "-pthread", "-Wl,-rpath,$ORIGIN/../../../simplexmethod_amd/_build"]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        outs.append(exe)
    return outs


def build_all(force=False, verbose=False):
    hip, host = build_hip(force, verbose), build_host(force, verbose)
    build_cpp_tests(force, verbose)
    return hip, host


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv, verbose=True))
