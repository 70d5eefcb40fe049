"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV (kernels whose name contains argv[2])."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_enum_leaves"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a["v"] += float(r["Counter_Value"]); a["n"] += 1
for k in sorted(acc):
    for c, v in sorted(acc[k].items()):
        print("%-22s %-24s %.4g per launch (%d launches)" % (k, c, v["v"] / v["n"], v["n"]))
