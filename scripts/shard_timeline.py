"""Kernel timeline of the LAST pass in a rocprofv3 --kernel-trace CSV of scripts/trace_shard.py: start (us after
the pass's first kernel), duration, gap to the previous kernel's end, per kernel; and the sums."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_enum_root" in r[2]]
last = rows[starts[-1]:]
t0 = last[0][0]
end_prev = t0
busy = 0
for s, e, name in last:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    print("%9.1f us  +%7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - end_prev) / 1e3, short[:70]))
    end_prev = max(end_prev, e)
print("pass: first kernel start -> last kernel end %.1f us" % ((max(e for _, e, _ in last) - t0) / 1e3))
