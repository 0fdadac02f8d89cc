import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "bq_" in n or "qbp_cell" in n:
        d[(n[9:52], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items()):
    v=sorted(v); print(k, len(v), "median %.2f us"%v[len(v)//2], "min %.2f"%v[0])
