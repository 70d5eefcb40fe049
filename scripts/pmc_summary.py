import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "k_enum_leaves" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]["v"] += float(r["Counter_Value"]); acc[r["Counter_Name"]]["n"] += 1
for k, v in sorted(acc.items()):
    print("%-28s %.4g (per launch, %d launches)" % (k, v["v"] / v["n"], v["n"]))
