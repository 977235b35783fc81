import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith('k_morton')]
i0=idx[-5]; i1=idx[-4]
t0=int(rows[i0]['Start_Timestamp'])
print("step length %.1f us"%((int(rows[i1]['Start_Timestamp'])-t0)/1e3))
for r in rows[i0:i1]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    nm=r['Kernel_Name'].split('(')[0][-40:]
    if e-s>60:
        print("%8.1f %8.1f %7.1f %s q%s"%(s,e,e-s,nm,r.get('Queue_Id')))
