#!/usr/bin/env python3
"""CPU emulation of the split attention kernel's tile loop (fp32 statistics, fp16 probabilities) with an exact and a lagging
reference point: reproduces the growth of the context error with the lag that tools/attn_split_dbg.py measured on the GPU,
and shows which rounding is responsible."""
import numpy as np
rng=np.random.default_rng(3)
L,d=1370,64
NQ=256
q=(rng.standard_normal((NQ,d))*0.6*1.4427).astype(np.float32)   # log2 units like the test
k=rng.standard_normal((L,d)).astype(np.float32)
v=rng.standard_normal((L,d)).astype(np.float32).astype(np.float16).astype(np.float32)
S=(q.astype(np.float64)@k.astype(np.float64).T)
P=np.exp2(S-S.max(1,keepdims=True)); ref=(P@v.astype(np.float64))/P.sum(1,keepdims=True)
def run(lag):
    m2=np.zeros(NQ,np.float32); l=np.zeros(NQ,np.float32); o=np.zeros((NQ,d),np.float32)
    nt=(L+63)//64
    for t in range(nt):
        k0=t*64; k1=min(L,k0+64)
        s=(S[:,k0:k1].astype(np.float32)-m2[:,None]).astype(np.float32)
        mt=s.max(1)
        first=(t==0)
        if first or (mt>lag).any():
            delta=mt if first else np.maximum(mt,0).astype(np.float32)
            alpha=np.ones(NQ,np.float32) if first else np.exp2(-delta).astype(np.float32)
            m2=(m2+delta).astype(np.float32); l=(l*alpha).astype(np.float32); o=(o*alpha[:,None]).astype(np.float32)
            s=(s-delta[:,None]).astype(np.float32)
        p=np.exp2(s).astype(np.float32)
        l=(l+p.sum(1,dtype=np.float32)).astype(np.float32)
        p16=p.astype(np.float16).astype(np.float32)
        o=(o+(p16@v[k0:k1]).astype(np.float32)).astype(np.float32)
    out=o/l[:,None]
    return np.sqrt(np.mean((out-ref)**2)), np.abs(out-ref).max()
for lag in (0,1,4,8,12):
    print(lag, run(lag))
print("variants")
def run2(lag, round_p=True, odt=np.float32, sdt=np.float32):
    m2=np.zeros(NQ,sdt); l=np.zeros(NQ,odt); o=np.zeros((NQ,d),odt)
    nt=(L+63)//64
    for t in range(nt):
        k0=t*64; k1=min(L,k0+64)
        s=(S[:,k0:k1].astype(sdt)-m2[:,None]).astype(sdt)
        mt=s.max(1)
        first=(t==0)
        if first or (mt>lag).any():
            delta=mt if first else np.maximum(mt,0).astype(sdt)
            alpha=np.ones(NQ,odt) if first else np.exp2(-delta).astype(odt)
            m2=(m2+delta).astype(sdt); l=(l*alpha).astype(odt); o=(o*alpha[:,None]).astype(odt)
            s=(s-delta[:,None]).astype(sdt)
        p=np.exp2(s.astype(np.float64)).astype(np.float32)
        l=(l+p.astype(odt).sum(1)).astype(odt)
        pp=p.astype(np.float16).astype(np.float32) if round_p else p
        o=(o+(pp.astype(odt)@v[k0:k1].astype(odt))).astype(odt)
    out=o/l[:,None]
    return np.sqrt(np.mean((out-ref)**2))
for name,kw in (("no P rounding",dict(round_p=False)),("fp64 o,l",dict(odt=np.float64)),("fp64 s,m2",dict(sdt=np.float64)),("all 64 but P16",dict(odt=np.float64,sdt=np.float64))):
    print(name,[f"{run2(lag,**kw):.3e}" for lag in (0,4,8)])
