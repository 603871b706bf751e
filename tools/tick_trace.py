import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'),key=lambda p:-__import__('os').path.getsize(p))[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r['Start_Timestamp']))
# ticks: split at k_heightfield launches
k1=[i for i,r in enumerate(rows) if 'k_heightfield' in r['Kernel_Name']]
print(len(rows),'kernels',len(k1),'region calls')
for a,b in list(zip(k1,k1[1:]))[-6:-3]:
    t0=int(rows[a]['Start_Timestamp']); seg=rows[a:b]
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e6
    print(f"--- period {(int(rows[b]['Start_Timestamp'])-t0)/1e6:.3f} ms, kernels {len(seg)}, sum of kernel time {busy:.3f} ms, last kernel ends at {(max(int(r['End_Timestamp']) for r in seg)-t0)/1e6:.3f}")
    for r in seg:
        s=(int(r['Start_Timestamp'])-t0)/1e6; e=(int(r['End_Timestamp'])-t0)/1e6
        print(f"  {s:7.3f} {e:7.3f} {e-s:6.3f} {r['Kernel_Name'].replace('void ','').split('(')[0][:40]}")
