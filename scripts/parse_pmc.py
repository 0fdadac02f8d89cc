"""Average a PMC counter per kernel from a rocprofv3 counter_collection.csv"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if any(s in k for s in sys.argv[2:]) or len(sys.argv) <= 2:
        print("%-62s %-12s n=%3d avg=%14.1f" % (k, c, len(v), sum(v) / len(v)))
