"""one line per phase exit: median kernel duration (kernel trace) and the SQ counters (per dispatch, median) of qbp_cell_kernel"""
import csv, glob, os, sys
import numpy as np
stop, tr, pm = sys.argv[1], sys.argv[2], sys.argv[3]
dur = []
for f in glob.glob(os.path.join(tr, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "qbp_cell_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = {}
for f in glob.glob(os.path.join(pm, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "qbp_cell_kernel" in r["Kernel_Name"]:
            ctr.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            ctr[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out = "stop %3s  n=%d  dur_us median %.2f min %.2f |" % (stop, len(dur), np.median(dur) if dur else -1, min(dur) if dur else -1)
for k in sorted(ctr):
    out += " %s %.0f" % (k, np.median(list(ctr[k].values())))
print(out)
