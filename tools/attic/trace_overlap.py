import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void ','').replace('amdspeech::','')[:28],r['Queue_Id']) for r in rows]
ks.sort()
# take the last bwd chain: find last 1003 bwd_step launches
bw=[k for k in ks if k[2].startswith('lstm_bwd_step')][-1003:]
t0=bw[0][0]
gm=[k for k in ks if k[2].startswith('gemm') and k[0]>=t0]
print("bwd chain span ms",(bw[-1][1]-t0)/1e6)
# bwd step durations & periods in quarters
import statistics
for q in range(4):
    seg=bw[q*250:(q+1)*250]
    d=[e-s for s,e,_,_ in seg]; per=[seg[i+1][0]-seg[i][0] for i in range(len(seg)-1)]
    print("quarter",q,"dur us",statistics.mean(d)/1e3,"period us",statistics.mean(per)/1e3)
for s,e,n,q in gm: print(n,q,"start ms",(s-t0)/1e6,"dur us",(e-s)/1e3)
fw=[k for k in ks if k[2].startswith('lstm_fwd_step')][-1003:]
d=[e-s for s,e,_,_ in fw]; per=[fw[i+1][0]-fw[i][0] for i in range(len(fw)-1)]
print("fwd chain: dur us",statistics.mean(d)/1e3,"period us",statistics.mean(per)/1e3, "queue", fw[0][3], "bwd queue", bw[0][3])
