"""Print the kernels of a rocprofv3 kernel-trace CSV in launch order with durations (us)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
