"""HBM bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction as in pmc_to_json.py), of every kernel
in two rocprofv3 --pmc passes (FETCH_SIZE dir, WRITE_SIZE dir) of one workload, launch by launch for kernels
matching argv[3] (default k_enum_expand)."""
import collections, csv, glob, sys
pat = sys.argv[3] if len(sys.argv) > 3 else "k_enum_expand"


def read(directory, counter):
    f = sorted(glob.glob(directory + "/**/*counter_collection.csv", recursive=True))[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and pat in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:], float(r["Counter_Value"])))
    rows.sort()
    return rows


fe, wr = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
for (i, name, f), (_, _, w) in zip(fe, wr):
    print("%5d %-40s fetch %10.1f KB (x2 = %8.1f MB)  write %10.1f KB   hbm bytes %8.1f MB" % (i, name, f, 2 * f / 1024, w, (2 * f + w) / 1024))
