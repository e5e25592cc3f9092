import sys,re,collections
rows=[]
for line in sys.stdin:
    m=re.match(r'trace (\w+)\s+(\d+) start\s+([\d.]+) us\s+dur\s+([\d.]+) us\s+swaps (\d+)',line)
    if m: rows.append((m.group(1),int(m.group(2)),float(m.group(3)),float(m.group(4))))
by=collections.defaultdict(list)
for k,i,st,du in rows: by[k].append(du)
for k,v in by.items():
    v2=sorted(v); n=len(v2)
    print(k, 'n',n,'median',v2[n//2],'p90',v2[int(n*0.9)],'p99',v2[int(n*0.99)],'max',v2[-1],'sum',sum(v2))
# gaps: time between consecutive trace entries' end and next start
rows.sort(key=lambda r:r[2])
gaps=[]
for a,b in zip(rows,rows[1:]):
    g=b[2]-(a[2]+a[3])
    gaps.append((g,a,b))
gaps.sort(reverse=True,key=lambda x:x[0])
print('largest untraced gaps:')
for g,a,b in gaps[:8]: print(round(g,1),a[:2],'->',b[:2])
print('total span',rows[-1][2]+rows[-1][3]-rows[0][2])
