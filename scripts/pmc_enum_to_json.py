"""VALU issue utilisation of the enumeration kernels from one rocprofv3 --pmc pass of scripts/pmc_enum.py:

    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d <dir> -- python3 scripts/pmc_enum.py
    python3 scripts/pmc_enum_to_json.py <dir> > enum_valu.json      (third argument of pmc_to_json.py)

busy = SQ_INSTS_VALU / (SQ_WAVE_CYCLES / waves per SIMD): SQ_WAVE_CYCLES counts every resident wave's
cycles (in units of four), the kernels below run at 3 waves per SIMD (their register budget), and a SIMD
issues at most one VALU instruction per four cycles."""
import collections, csv, glob, json, sys

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    if "k_enum_" not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
out = {}
for k, c in sorted(acc.items()):
    if "SQ_INSTS_VALU" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"][0] > 0:
        valu = c["SQ_INSTS_VALU"][0] / c["SQ_INSTS_VALU"][1]
        cyc = c["SQ_WAVE_CYCLES"][0] / c["SQ_WAVE_CYCLES"][1]
        out[k] = {"SQ_INSTS_VALU": valu, "SQ_WAVE_CYCLES": cyc, "launches": c["SQ_INSTS_VALU"][1],
                  "busy_at_3_waves_per_simd": round(valu / (cyc / 3.0), 4)}
# optional second directory: the fp64 instruction counters of the same workload
#   rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 ...
# executed fp64 flops per subset = (ADD + MUL + 2*FMA + TRANS) wave-instructions x 64 lanes, summed over the
# kernels of one pass, / C(32,16) subsets (an upper bound: partially masked waves count as full)
if len(sys.argv) > 2:
    f2 = sorted(glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True))[-1]
    tot = collections.defaultdict(float)
    launches = collections.defaultdict(int)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f2)):
        if "k_enum_" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        per_kernel[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU_FMA_F64":
            launches[k] += 1
    passes = 2.0   # scripts/pmc_enum.py runs two C(32,16) passes
    subsets = 601080390.0
    winst = {c: tot.get("SQ_INSTS_VALU_" + c + "_F64", 0.0) / passes for c in ("ADD", "MUL", "FMA", "TRANS")}
    flops = (winst["ADD"] + winst["MUL"] + 2.0 * winst["FMA"] + winst["TRANS"]) * 64.0
    out["__fp64__"] = {
        "wave_instructions_per_pass": winst, "subsets_per_pass": subsets,
        "flops_per_subset": round(flops / subsets, 2),
        "formula": "(ADD + MUL + 2*FMA + TRANS) fp64 wave-instructions x 64 lanes / subsets, all k_enum_* kernels of a pass",
        "per_kernel_wave_instructions_per_pass": {k: {c: v / passes for c, v in d.items()} for k, d in sorted(per_kernel.items())}}
print(json.dumps(out, indent=1))
