"""Static check of k_simplex_overlap<true>'s hand-counted ring (simplex_overlap.hip): between a slot's three asm loads
and the counted s_waitcnt that covers them (the next one for the same slot) no instruction may touch the loads' destination
registers — the compiler does not know that they are in flight, and a copy or a spill in between would take stale
data.  Compiles to gfx950 assembly here (no GPU needed); exit code 1 if a register is touched.
    python scripts/check_ring_asm.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "o.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "simplexmethod_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                    os.path.join(ROOT, "simplexmethod_amd", "csrc", "simplex_overlap.hip")], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN12_GLOBAL__N_117k_simplex_overlapILb1EEE")][0]
end = [i for i, l in enumerate(lines) if i > start and ".end_amdhsa_kernel" in l][0]
body = [l.split(";")[0].strip() for l in lines[start:end] if not l.strip().startswith(";;")]
body = [l for l in body if l and not l.startswith(".") or l.startswith(".LBB")]


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def touched(line):
    s = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        s |= regs(tok)
    return s


# the loop: from its header label to the backward branch
waits = [i for i, l in enumerate(body) if l == "s_waitcnt vmcnt(28)"]
assert len(waits) == 8, waits
head = max(i for i, l in enumerate(body[:waits[0]]) if l.startswith(".LBB"))
tail = min(i for i, l in enumerate(body) if i > waits[-1] and l.startswith("s_cbranch") and body[head].rstrip(":") in l)
loop = body[head + 1:tail]
seq = body[:head + 1] + loop + loop          # prologue + two turns of the ring
bad = 0
i = 0
nloads = 0
nwaits = 0   # counted waits seen so far; wait number w (0-based) serves slot w % 8
while i < len(seq):
    l = seq[i]
    if l == "s_waitcnt vmcnt(28)":
        nwaits += 1
    if l.startswith("global_load_dwordx4") and i + 2 < len(seq) and seq[i + 1].startswith("global_load_dwordx4") and seq[i + 2].startswith("global_load_dwordx2"):
        slot = nloads if nwaits == 0 else (nwaits - 1) % 8     # prologue: slots in order; loop: the slot just finished
        dest = set()
        for k in range(3):
            dest |= regs(seq[i + k].split()[1].rstrip(","))
        w, j = nwaits, i + 3
        while j < len(seq):
            if seq[j] == "s_waitcnt vmcnt(28)":
                if w % 8 == slot:
                    break
                w += 1
            elif touched(seq[j]) & dest:
                print("slot", slot, "in flight", sorted(dest), "touched by:", seq[j])
                bad += 1
            j += 1
        nloads += 1
        i += 3
    else:
        i += 1
# the prologue's padding loads: one destination register, touched by nothing else up to the end of the ring
pads = [l for l in seq if l.startswith("global_load_dword v")]
pdest = {l.split()[1].rstrip(",") for l in pads}
first = min(k for k, l in enumerate(seq) if l.startswith("global_load_dword v"))
others = [l for l in seq[first:] if not l.startswith("global_load_dword v") and any(touched(l) & regs(d) for d in pdest)]
if len(pads) != 8 or len(pdest) != 1 or others:
    print("padding loads:", len(pads), "destinations:", sorted(pdest), "also touched by:", others[:4])
    bad += 1
print("slot requests checked:", nloads, "violations:", bad)
sys.exit(1 if bad else 0)
