import numpy as np
from scipy.special import erfc, erf
np.set_printoptions(precision=10)
T=4.0
t=np.cos(np.linspace(0,np.pi,4001))*T/2+T/2
t=np.sort(t)
f=np.log2(erfc(t))
w0=erfc(t)*np.log(2)
for deg in (6,7,8,9):
    # fit p(t)=t*q(t) (p(0)=0 exactly)
    V=np.vander(t,deg,increasing=True)*t[:,None]   # t^1..t^deg
    w=w0.copy()
    lw=np.ones_like(t)
    for it in range(60):
        W=(w*lw)[:,None]
        c,*_=np.linalg.lstsq(V*W,f*w*lw,rcond=None)
        err=np.abs((V@c-f)*w)
        lw=lw*(err/err.max()+1e-3)**0.5; lw/=lw.mean()
    # float32 Horner evaluation
    c32=c.astype(np.float32)
    tt=np.linspace(0,6,200001).astype(np.float32)
    tc=np.minimum(tt,np.float32(T))
    p=np.zeros_like(tc)
    for k in range(deg-1,-1,-1): p=(p*tc+c32[k]).astype(np.float32)
    p=(p*tc).astype(np.float32)
    e=np.exp2(p.astype(np.float64))
    approx=1-e
    print(deg, 'max abs err erf', np.abs(approx-erf(tt.astype(np.float64))).max(), 'coef', c32)
