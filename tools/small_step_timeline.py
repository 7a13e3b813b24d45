import csv, sys, glob
d=sys.argv[1]
rows=[]
for f in glob.glob(d+'/**/*kernel_trace.csv',recursive=True): rows+=list(csv.DictReader(open(f)))
ker=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:50]) for r in rows)
ker=[k for k in ker if 'elementwise' not in k[2]]
# last occurrence of plane kernel = start of last step
# the first kernel of a step: the level-0 pre-pass (since round 5, for small calls, gabor_pre01_kernel: both pre-passes in one launch)
idx=[i for i,k in enumerate(ker) if 'gabor_plane' in k[2] or 'gabor_pre01' in k[2]]
i0=idx[-2]; i1=idx[-1]
t0=ker[i0][0]
prev=None
for s,e,n in ker[i0:i1]:
    print('%8.1f %8.1f (%6.1f) gap %5.1f %s'%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,0 if prev is None else (s-prev)/1e3,n))
    prev=max(prev or 0,e)
print('step span %.1f us, next step starts at %.1f'%((ker[i1-1][1]-t0)/1e3,(ker[i1][0]-t0)/1e3))
