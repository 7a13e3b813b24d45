import csv,sys,glob,collections
d=sys.argv[1]
K=[];C=[]
for f in glob.glob(d+'/**/*kernel_trace.csv',recursive=True): K+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:40]) for r in csv.DictReader(open(f))]
for f in glob.glob(d+'/**/*memory_copy_trace.csv',recursive=True): C+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Direction"].replace("MEMORY_COPY_","")) for r in csv.DictReader(open(f))]
K.sort();C.sort()
t0=K[0][0]
ev=[(s,e,'K '+n) for s,e,n in K if 'copyBuffer' in n and e-s>100000]+[(s,e,'SDMA '+n) for s,e,n in C if e-s>100000]
ev.sort()
# bucket by 50 ms windows
b=collections.OrderedDict()
for s,e,n in ev:
    k=(s-t0)//100_000_000
    b.setdefault(k,collections.Counter())[n+' ~%dus'%(round((e-s)/1e5)*100)]+=1
for k,v in b.items(): print('t=%.1fs'%(k/10), dict(v))
