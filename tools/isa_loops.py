"""Dev tool: per-loop instruction counts of a kernel in hipcc's assembly (the evidence behind DESIGN.md 4.3 "what it took").
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -x hip rnn-speech_amd/csrc/lstm.hip -o /tmp/lstm.s
    python tools/isa_loops.py /tmp/lstm.s _ZN9amdspeech14lstm_bwd_flow2ILi4ELi0ELb1EEEvNS_11FlowBwdArgsE
prints, for every loop that contains MFMAs: length, MFMAs, SGPR-spill lane moves (v_readlane / v_writelane), scratch accesses,
barriers, s_waitcnt vmcnt."""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
def analyze(sym):
    start=[i for i,l in enumerate(lines) if l.startswith(sym+':')][0]
    end=start
    while not lines[end].startswith('.Lfunc_end'): end+=1
    lab={}
    for i in range(start,end):
        m=re.match(r'^(\.LBB\d+_\d+):',lines[i])
        if m: lab[m.group(1)]=i
    loops=[]
    for i in range(start,end):
        m=re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)',lines[i])
        if m:
            t=m.group(1) or m.group(2)
            if t in lab and lab[t] < i: loops.append((lab[t],i))
    print(sym[:60],'total lines',end-start)
    for a,b in sorted(loops):
        body=lines[a:b+1]
        mf=sum('v_mfma' in l for l in body)
        if mf==0: continue
        ln=sum(('v_writelane' in l or 'v_readlane' in l) for l in body)
        sc=sum('scratch_' in l for l in body)
        bar=sum('s_barrier' in l for l in body)
        wc=sum('s_waitcnt vmcnt' in l for l in body)
        print('  loop %6d..%6d len %5d mfma %4d lane-spill %3d scratch %2d barriers %d vmcnt-waits %d'%(a-start,b-start,b-a,mf,ln,sc,bar,wc))
for s in sys.argv[2:]: analyze(s)
