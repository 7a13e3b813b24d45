import csv,sys,glob
d=sys.argv[1]
rows=[]
for f in glob.glob(d+'/**/*kernel_trace.csv',recursive=True): rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),'K '+r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))]
for f in glob.glob(d+'/**/*memory_copy_trace.csv',recursive=True): rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),'COPY '+r["Direction"]) for r in csv.DictReader(open(f))]
rows.sort()
t0=rows[0][0]
for s,e,n in rows:
    if (e-s)>100000: print('%10.1f (%8.1f us) %s'%((s-t0)/1e3,(e-s)/1e3,n))
