import numpy as np
rng=np.random.default_rng(1)
Y=100;T=36500;NL=64
t=np.arange(T)
DEFS=[(3,0,0),(3,1,1),(4,0,0),(4,1,1),(5,0,0),(5,1,1)]
PERC=np.arange(0.9,1.0,0.01)
def series(trend,beta=270):
    return (20+2*np.sin(2*np.pi*(beta+t)/365.0)+0.7*rng.random(T)+trend*t).astype(np.float32)
def thresholds(x):
    cols=x.reshape(Y,365).T  # [doy][year]
    out=np.empty((365,len(PERC)))
    for d in range(365):
        idx=[(d+k)%365 for k in range(-7,8)]
        out[d]=np.quantile(cols[idx].ravel().astype(np.float64),PERC)
    return out
def lane_trips(hot):
    """per-word trips of one lane following metrics_kernel_cells16's loop (year-aligned words)"""
    mmin=min(max(d[0],1) for d in DEFS); skip=min(mmin,64)
    trips=np.zeros(Y*6,dtype=np.int32)
    in_hw=[0]*len(DEFS); subs=[0]*len(DEFS)
    open_=0; s_open=0; e_prev=-(1<<30)
    w=0
    for y in range(Y):
        for j in range(6):
            t0=y*365+64*j; L=64 if j<5 else 365-320
            bits=hot[t0:t0+L]
            nxt=hot[t0+L:t0+L+64]
            # longs: bit i set iff days i..i+skip-1 hot
            ext=np.concatenate([bits,nxt,np.zeros(64,dtype=bool)])
            longs=np.ones(L,dtype=bool)
            for k in range(skip): longs&=ext[k:k+L]
            pos=0; n=0
            any_hw=any(in_hw)
            while True:
                n+=1
                if not open_:
                    src=bits if any_hw else longs
                    nz=np.flatnonzero(src[pos:])
                    if nz.size==0: break
                    pos+=nz[0]; s_open=t0+pos; open_=1
                    gap=s_open-e_prev
                    for k,(d,b,s) in enumerate(DEFS):
                        if gap>b: in_hw[k]=0
                nz=np.flatnonzero(~bits[pos:])
                if nz.size==0: break
                pos+=nz[0]; e=t0+pos; open_=0
                ln=e-s_open
                for k,(d,b,s) in enumerate(DEFS):
                    ge=ln>=d; sub=in_hw[k] and subs[k]<s
                    subs[k]=subs[k]+1 if sub else (0 if in_hw[k] else subs[k])
                    in_hw[k]=1 if (sub or ge) else 0
                e_prev=e; any_hw=any(in_hw)
            # the kernel skips the loop entirely if no lane has work; count trips as loop iterations incl. the final breaking one
            trips[w]=n; w+=1
    return trips
res=[]
for p_i in (0,5,9):
    all_tr=[]
    for lane in range(NL):
        base=series(0.0); meas=series(1.0/36500.0)
        thr=thresholds(base)[:,p_i].astype(np.float32)  # approx f32 round
        hot=meas>thr[t%365]
        all_tr.append(lane_trips(hot))
    A=np.array(all_tr)  # lanes x words
    # kernel: per word, loop runs max over lanes iterations (the last iteration is the 'break' detection for the slowest lane)
    perword=A.max(axis=0).sum(); mean=A.sum(axis=1).mean()
    peryear=A.reshape(NL,Y,6).sum(axis=2).max(axis=0).sum()
    per2=A.reshape(NL,Y*3,2).sum(axis=2).max(axis=0).sum()
    per3=A.reshape(NL,Y*2,3).sum(axis=2).max(axis=0).sum()
    whole=A.sum(axis=1).max()
    # real runs (iterations that closed a run) ~ trips minus 1 per word-with-work
    print(f"perc {PERC[p_i]:.2f}: hot frac {hot.mean():.3f} mean lane trips {mean:.0f}  lockstep per word {perword}  per 2 words {per2} per 3 words {per3} per year {peryear}  whole {whole}")
