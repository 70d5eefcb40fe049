"""Durations of the breadth-first level kernels of the LAST C(32,16) pass in a rocprofv3 --kernel-trace CSV of
scripts/pmc_enum.py."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_enum_root" in r[2]]
tot = 0
for s, e, name in rows[starts[-1]:]:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if "expand" in short or "root" in short or "make_items" in short:
        print("%9.1f us  %s" % ((e - s) / 1e3, short))
        tot += e - s
print("sum %.1f us" % (tot / 1e3))
