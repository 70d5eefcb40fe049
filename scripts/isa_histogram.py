"""Instruction histogram of the hot per-subset loop of a leaf kernel: compiles enum_leaf.hip to
gfx950 assembly (hipcc -S, here: no GPU needed), takes the innermost-but-one loop of k_enum_leaves<K,false> — the loop
over a work item's subsets with the per-subset routine inlined — and counts mnemonics by class.
    python scripts/isa_histogram.py [1|2] > profiles/r04_leaf_isa_histogram_<K>.txt"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = sys.argv[1] if len(sys.argv) > 1 else "1"
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "leaf.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "simplexmethod_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                    os.path.join(ROOT, "simplexmethod_amd", "csrc", "enum_leaf.hip")], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
name = "_ZN12_GLOBAL__N_113k_enum_leavesILi%sELb0EEEv7EnumDev9PrefixDevPKdyy:" % K
start = [i for i, l in enumerate(lines) if l.startswith(name)][0]
end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith(".amdhsa_kernel")][0]
body = lines[start:end]
hdr = [i for i, l in enumerate(body) if "Loop Header: Depth=2" in l and "Inner" not in l]
nxt = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and i > hdr[-1]]
seg = body[hdr[-1]:(nxt[0] if nxt else len(body))]
cnt = collections.Counter()
for l in seg:
    l = l.split(";")[0].strip()
    if not l or l.endswith(":") or l.startswith("."):
        continue
    cnt[l.split()[0]] += 1


def cls(op):
    if op.startswith(("v_fma", "v_mul_f64", "v_add_f64", "v_rcp", "v_fmac_f64")): return "fp64 arithmetic"
    if op.startswith(("v_max_f64", "v_min_f64")): return "fp64 min/max (pivot search, magnitude tracking)"
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_cndmask"): return "v_cndmask (row rotation, pivot-row choice)"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("s_"): return "SALU / control / waits"
    if op.startswith("v_"): return "other VALU (addresses, table look-ups, moves)"
    return "other"


tot = sum(cnt.values())
print("k_enum_leaves<%s,false>, static instructions of the per-subset loop (slow exits included): %d" % (K, tot))
c2 = collections.Counter()
for op, n in cnt.items():
    c2[cls(op)] += n
for k, v in c2.most_common():
    print("  %-52s %4d  %5.1f %%" % (k, v, 100.0 * v / tot))
print("by mnemonic:", ", ".join("%s %d" % kv for kv in cnt.most_common(24)))
