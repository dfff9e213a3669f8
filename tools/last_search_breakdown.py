import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:80]))
rows.sort()
# last search = last 13 launches: find last 'select_topk' and go back until gap > 1 ms
end=len(rows)-1
i=end
while i>0 and rows[i][0]-rows[i-1][1] < 2_000_000: i-=1
seg=rows[i:]
t0=seg[0][0]
from collections import defaultdict
tot=defaultdict(float); cnt=defaultdict(int)
gaps=0.0; prev=None
for s,e,n in seg:
    tot[n]+= (e-s)/1e3; cnt[n]+=1
    if prev is not None and s>prev: gaps+=(s-prev)/1e3
    prev=max(prev or e, e)
wall=(seg[-1][1]-t0)/1e3
print(f"last search: {len(seg)} kernels, wall {wall/1e3:.3f} ms, idle gaps {gaps:.1f} us")
for n,v in sorted(tot.items(), key=lambda x:-x[1]): print(f"  {v/1e3:9.3f} ms  {100*v/wall:5.1f} %  x{cnt[n]:3d}  {n}")
