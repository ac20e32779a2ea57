import numpy as np, math
rng=np.random.default_rng(0)
S=1000; W=15; Y=100; M=10
T=36500; t=np.arange(T)
def cell():
    cols=[]
    for m in range(M):
        x=20+2*np.sin(2*np.pi*(270+t)/365.0)+0.7*rng.random(T)
        cols.append(x.reshape(Y,365).T)
    c=np.concatenate(cols,axis=1)  # [365][1000]
    return -np.sort(-c,axis=1)     # descending
def count_above(col,v):  # number of keys > v in a descending column
    return int(np.searchsorted(-col,-v,side='left'))
def nbits(w):
    return int(w).bit_length()
def select(cols,R,mode):
    """cols: list of W descending arrays. returns (rounds, stride_trips) for finding split with sum c_j = R"""
    lo=[0]*W; hi=[min(S,R)]*W
    rounds=0; trips=0
    first=True
    while True:
        ww=[hi[j]-lo[j] for j in range(W)]
        wj=int(np.argmax(ww)); w=ww[wj]
        if w<=0: break
        if mode>=1 and first:
            wj=W//2; mid=min(max(1,int(round(R/W))),hi[wj])   # centre column, expected position
        elif mode>=3:
            # proportional: remaining need split evenly
            need=R-sum(lo)
            wj=int(np.argmax(ww)); mid=lo[wj]+min(max(1,int(math.ceil(need/W))),ww[wj])
        else:
            mid=lo[wj]+((w+1)>>1)
        pkey=cols[wj][mid-1]   # 1-based position mid -> element index mid-1 ; count above it in own column = mid-1
        pos=[]
        for j in range(W):
            if j==wj: pos.append(mid-1)
            else:
                # ties: ignore (continuous data)
                c=count_above(cols[j],pkey); c=min(max(c,lo[j]),hi[j]); pos.append(c)
        # cost: strides cover widest interval among lanes -> approximate by this lane's widest interval
        trips+=nbits(max(ww)) if not (mode>=1 and first) else nbits(min(S,R))
        rounds+=1
        G=sum(pos)
        if G==R:
            lo=pos[:]; hi=pos[:]
        elif G>R:
            hi=pos[:]
            if mode>=2:
                for j in range(W): lo[j]=max(lo[j],pos[j]-(G-R))
        else:
            lo=[pos[j]+(1 if j==wj else 0) for j in range(W)]
            if mode>=2:
                d=R-G-1
                for j in range(W): hi[j]=min(hi[j],max(lo[j],pos[j]+(1 if j==wj else 0)+d))
        first=False
    assert sum(lo)==R
    return rounds,trips
c=cell()
for R in (150,750,1500,3000):
    for mode,name in ((0,"current"),(1,"smart first pivot"),(2,"+ |R-G| clamps"),(3,"+ proportional pivots")):
        rr=[];tt=[]
        for d in range(20,365,9):
            cols=[c[(d+k)%365] for k in range(-7,8)]
            r,tr=select(cols,R,mode); rr.append(r); tt.append(tr)
        print(f"R={R:5d} {name:24s} rounds mean {np.mean(rr):5.1f} max {max(rr):3d}   stride-trips mean {np.mean(tt):6.1f} max {max(tt)}")
