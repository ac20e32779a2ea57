import numpy as np
rng = np.random.default_rng(0)
T=36500; t=np.arange(T)
def gen(kind):
    if kind=="bench": return 20+2*np.sin(2*np.pi*(270+t)/365.0)+0.7*rng.random(T)
    if kind=="flat": return 20+0.7*rng.random(T)
    if kind=="midlat": return 20+8*np.sin(2*np.pi*t/365.0)+rng.normal(0,3.0,T)
for kind in ("bench","flat","midlat"):
    allc=[]; 
    for cell in range(4):
        x=gen(kind); cols=x.reshape(100,365).T
        srt=-np.sort(-cols,axis=1)
        for d in range(365):
            idx=[(d+k)%365 for k in range(-7,8)]
            w=srt[idx]  # 15x100
            flat=np.sort(w.ravel())[::-1]
            v=flat[150]  # 151st largest
            c=(w>=v).sum(axis=1)
            allc.append(c)
    allc=np.array(allc)  # rows x 15
    mx=allc.max(axis=1)
    print(kind,"c_j mean",allc.mean(),"max c_j per row: mean",mx.mean(),"p50",np.percentile(mx,50),"p90",np.percentile(mx,90),"p99",np.percentile(mx,99),"max",mx.max())
    for K in (16,20,24,26,28,32,36,40,44,48,52,56,60):
        # steps that would touch the tail: sum over columns of max(c_j-K,0) per row
        tail=np.maximum(allc-K,0).sum(axis=1)
        print("  K",K,"rows with any tail",(mx>K).mean().round(4),"tail steps/row",tail.mean().round(3),"of 151")
