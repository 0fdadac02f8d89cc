import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True),key=lambda x:-len(open(x).read()))[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_Y"]) for r in rows if "bev_iou_kernel" in r["Kernel_Name"]]
i=0; k=0; names=["full 70000","full 65536","full 32768"]+[x for s in (1,2,3,4,5,6,7,8,0) for x in ("stop %d"%s,"stop %d half"%s)]
while i+210<=len(d):
    ch=[x[0] for x in d[i+10:i+210]]; print(names[k] if k<len(names) else k, d[i][1], round(sum(ch)/len(ch),2)); i+=210; k+=1
