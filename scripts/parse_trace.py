"""Summarise a rocprofv3 kernel_trace.csv: per kernel-name consecutive-run groups with avg duration (us)."""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
group = []
def flush():
    if group:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in group]
        n = group[0]["Kernel_Name"][:70]
        print("%-70s n=%4d avg=%9.2f us min=%9.2f max=%9.2f" % (n, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
for r in rows:
    if pat and pat not in r["Kernel_Name"]:
        continue
    # a new group on a kernel-name change, or after an idle gap (a host sync between two bursts of one kernel)
    if group and (r["Kernel_Name"] != group[0]["Kernel_Name"] or int(r["Start_Timestamp"]) - int(group[-1]["End_Timestamp"]) > 100000):
        flush(); group = []
    group.append(r)
flush()
