"""split a kernel trace of scripts/qbp_phase_timing.py into its variants (55 launches each)"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if sys.argv[2] in r['Kernel_Name']]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 55
for i in range(0, len(rows), per):
    g = rows[i:i + per]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g[5:]]
    print("variant %2d  %-40s avg %8.2f us  (n=%d)" % (i // per, g[0]['Kernel_Name'][10:50], sum(d) / len(d) / 1e3, len(d)))
