import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(list)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "fused_kernel" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "n=%d mean=%.4g" % (len(v), sum(v) / len(v)))
