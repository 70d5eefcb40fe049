"""profiles/r04_pmc_traffic.json from the two PMC passes of scripts/pmc_traffic.py.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KB, and on
gfx950 FETCH_SIZE reads exactly half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM
section) - the correction that guide prescribes; the raw counters are kept next to the result.
The file records the hash of the kernel sources it was taken on; bench.py reports the traffic only
when that hash matches the sources it runs."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash)

KERNELS = {"k_simplex_resident": "k_simplex_resident", "k_simplex_update": "k_simplex_update",
           "k_look_update": "k_look_update", "k_simplex_overlap": "k_simplex_overlap"}


def per_kernel(directory, counter):
    f = sorted(glob.glob(directory + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for key, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                a = acc[key]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"kernel_source_hash": bench.kernel_source_hash(),
       "workload": "scripts/pmc_traffic.py: m=512 n=1024 seed 0; k_simplex_overlap: m=2048 n=4096 seed 0, first 40 pivots "
                   "(one launch per pivot: the out-of-place rank-1 update of the 67 MB tableau + the next selection)",
       "formula": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
       "kernels": {}}
for k in KERNELS:
    if k in fetch and k in write:
        out["kernels"][k] = {
            "FETCH_SIZE_KB": round(fetch[k][0], 2), "WRITE_SIZE_KB": round(write[k][0], 2),
            "launches_sampled": [fetch[k][1], write[k][1]],
            "hbm_bytes_per_launch": round((2.0 * fetch[k][0] + write[k][0]) * 1024.0, 1)}
if len(sys.argv) > 3:   # optional: VALU issue utilisation (+ fp64 instruction counts) of the enumeration kernels
    ev = json.load(open(sys.argv[3]))
    fp = ev.pop("__fp64__", None)
    out["enum_valu_issue_busy"] = ev
    if fp is not None:
        out["enum_fp64"] = fp
if len(sys.argv) > 4:   # optional: a rocprofv3 --kernel-trace --stats directory; per-launch averages of the tableau kernels
    fs = sorted(glob.glob(sys.argv[4] + "/**/*kernel_stats.csv", recursive=True))
    if fs:
        for r in csv.DictReader(open(fs[-1])):
            for key, pat in KERNELS.items():
                if pat in r["Name"] and key in out["kernels"]:
                    out["kernels"][key]["rocprof_avg_launch_us"] = round(float(r["AverageNs"]) / 1e3, 3)
                    out["kernels"][key]["rocprof_calls"] = int(r["Calls"])
path = os.path.join(ROOT, "profiles", bench.PMC_PROFILE)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
