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
print(json.dumps(out, indent=1))
