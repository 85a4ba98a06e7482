import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tail=rows[-400:-60]
prev=None
d=collections.defaultdict(list); g=collections.defaultdict(list)
for r in tail:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=r['Kernel_Name'].replace('void ','').replace('svk::(anonymous namespace)::','')[:40]
    d[n].append((e-s)/1e3)
    if prev: g[n].append((s-prev)/1e3)
    prev=e
for n in d: print(n.ljust(42),'n=%3d dur %6.1f us  gap-before %6.2f us'%(len(d[n]),sum(d[n])/len(d[n]),sum(g[n])/max(1,len(g[n]))))
