"""Wave-level cost model of the C5 rank selection (round 4): value-pivot rounds (mean key, shifted mean, secant steps) +
a popped finish, against the key-pivot rounds it replaced.  Cost in vector instructions per wave of 64 (row, rank) lanes,
lock step: a round costs its overhead + 62 per stride of the widest interval in the wave.  DESIGN.md 3.2."""
import numpy as np, math, sys
import os
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'selection_rounds_sim.py')).read().split("c=cell()")[0])
POP=62; POPSETUP=90; ROUND=150; STRIDE=62   # instr: pop, pop setup+final, value round overhead, per stride (4 VALU+DS per col ~ 75 -> now 60 + nop)
def task(cols,R,KV,E0):
    """returns per-round nb list (length KV, 0 = idle), pops"""
    lo=[0]*W; hi=[min(S,R)]*W
    mid=min(max(1,int(round(R/W))),hi[0])
    v=np.float32(np.mean([cols[j][mid-1] for j in range(W)]))
    hist=[]; nbs=[]; e=None
    for k in range(KV):
        w=max(hi[j]-lo[j] for j in range(W))
        if w==0 or (e is not None and abs(e)<=E0 and k>=2): nbs.append(0); continue
        nbs.append(nbits(w))
        pos=[count_above(cols[j],v) for j in range(W)]
        G=sum(pos); e=G-R; hist.append((v,G))
        if e==0: lo=pos[:];hi=pos[:]; continue
        if e>0:
            hi=[min(hi[j],pos[j]) for j in range(W)]; lo=[max(lo[j],pos[j]-e) for j in range(W)]
        else:
            lo=[max(lo[j],pos[j]) for j in range(W)]; hi=[min(hi[j],pos[j]-e) for j in range(W)]
        if len(hist)==1:
            sh=e/W; ks=[]
            for j in range(W):
                p=int(round(pos[j]-sh+0.5))
                if 1<=p<=S and 0<pos[j]<S: ks.append(float(cols[j][p-1])-float(v))
            v=np.float32(v+np.mean(ks)*W/len(ks)) if ks else v
        else:
            (v0,G0),(v1,G1)=hist[-2],hist[-1]
            if G1!=G0: v=np.float32(v1+(R-G1)*(float(v1)-float(v0))/(G1-G0))
    return nbs,abs(e)
c=cell()
rng2=np.random.default_rng(1)
for KV in (3,4,5):
  for E0 in (4,8,16):
    tot=[];fail=0
    for R in (150,750,1500,3000):
        tasks=[task([c[(d+k)%365] for k in range(-7,8)],R,KV,E0) for d in range(20,365,3)]
        # waves of 64 lanes: same rank (lanes = rows)
        for wv in range(0,len(tasks)-63,16):
            grp=tasks[wv:wv+64]
            cost=0
            for k in range(KV):
                m=max(g[0][k] for g in grp)
                if m: cost+=ROUND+STRIDE*m
            pm=max(g[1] for g in grp)
            fail+=sum(1 for g in grp if g[1]>E0)
            cost+=POPSETUP+POP*min(pm,E0)
            tot.append(cost)
    print(f"KV={KV} E0={E0}: wave cost mean {np.mean(tot):6.0f} max {max(tot)}  lanes over E0 {fail}")
def cur(cols,R):
    lo=[0]*W; hi=[min(S,R)]*W; nbs=[]
    mid=min(max(1,int(round(R/W))),hi[0])
    v=np.mean([cols[j][mid-1] for j in range(W)]); k=0
    while True:
        ww=[hi[j]-lo[j] for j in range(W)]; w=max(ww)
        if w<=0: break
        nbs.append(nbits(w))
        if k==0:
            pos=[count_above(cols[j],v) for j in range(W)]; G=sum(pos)
            if G==R: lo=pos[:];hi=pos[:]
            elif G>R: hi=[min(hi[j],pos[j]) for j in range(W)]; lo=[max(lo[j],pos[j]-(G-R)) for j in range(W)]
            else: lo=[max(lo[j],pos[j]) for j in range(W)]; hi=[min(hi[j],pos[j]+(R-G)) for j in range(W)]
            k=1; continue
        wj=int(np.argmax(ww)); mid=lo[wj]+((w+1)>>1); pkey=cols[wj][mid-1]
        pos=[(mid-1) if j==wj else min(max(count_above(cols[j],pkey),lo[j]),hi[j]) for j in range(W)]
        G=sum(pos)
        if G==R: lo=pos[:];hi=pos[:]
        elif G>R:
            hi=pos[:]; lo=[max(lo[j],pos[j]-(G-R)) for j in range(W)]
        else:
            lo=[pos[j]+(1 if j==wj else 0) for j in range(W)]; d=R-G-1
            hi=[min(hi[j],lo[j]+d) for j in range(W)]
    return nbs
tot=[]
for R in (150,750,1500,3000):
    tasks=[cur([c[(d+k)%365] for k in range(-7,8)],R) for d in range(20,365,3)]
    for wv in range(0,len(tasks)-63,16):
        grp=tasks[wv:wv+64]; L=max(len(g) for g in grp); cost=0
        for k in range(L):
            m=max((g[k] if k<len(g) else 0) for g in grp); cost+=190+STRIDE*m
        tot.append(cost)
print("current: wave cost mean %.0f max %d; rounds max per wave mean %.1f"%(np.mean(tot),max(tot),0))
