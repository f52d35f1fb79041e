import csv, glob, statistics, sys
rows=[]
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def kind(n):
    for k in ("k_hidden", "k_free_energy", "k_grad"):
        if k in n: return k + n[n.index(k)+len(k):][:34]
    return None
seq=[(s,e,kind(n)) for s,e,n in rows if kind(n)]
# group by consecutive triplets
from collections import defaultdict
d=defaultdict(list)
for i in range(len(seq)-2):
    a,b,c=seq[i],seq[i+1],seq[i+2]
    if a[2].startswith("k_hidden") and b[2].startswith("k_free") and c[2].startswith("k_grad"):
        key=(a[2],b[2],c[2])
        d[key].append((a[1]-a[0], b[1]-b[0], c[1]-c[0], c[1]-a[0]))
for k,v in d.items():
    med=[round(statistics.median(x[i] for x in v)/1e3,2) for i in range(4)]
    print(len(v), k, "hidden/fe/grad/total us:", med)
