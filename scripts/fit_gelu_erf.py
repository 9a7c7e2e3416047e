"""How the coefficients of erf_gelu (tokenhmr_amd/csrc/common.h) were obtained: weighted Lawson/least-squares minimax fit of\nlog2(erfc(t)) = t*Q(t) on [0, 4] (weight = d erf / d r), rounded to fp32, then the complete fp32 GELU formula is emulated (fma\nvia fp64) and compared with fp64 GELU on the unit test's grid and on 2M normal samples.  CPU only: python scripts/fit_gelu_erf.py"""
import numpy as np
from scipy.special import erfc, erf
T=4.0
t=np.linspace(0,T,80001)[1:]
r=np.log2(erfc(t)); w=erfc(t)*np.log(2)
deg=8
A=np.vstack([t**k for k in range(1,deg+1)]).T
lw=np.ones_like(t)
for it in range(200):
    W=(w*lw)[:,None]
    c,*_=np.linalg.lstsq(A*W, r*w*lw, rcond=None)
    err=np.abs((A@c-r)*w)
    lw=lw*(err/err.max()+1e-3)**0.5; lw/=lw.max()
print('exact max abs err', err.max())
c32=c.astype(np.float32)
print('coeffs t^1..t^8:', [float.hex(float(x)) for x in c32])
print([repr(float(x)) for x in c32])
def f32(x): return np.asarray(x,dtype=np.float32)
def fma(a,b,cc): return (a.astype(np.float64)*b.astype(np.float64)+cc.astype(np.float64)).astype(np.float32)
def gelu_new(x):
    x=f32(x)
    a=f32(x*np.float32(0.70710678118654752440))
    tt=np.minimum(np.abs(a),np.float32(4.0))
    q=np.full_like(tt,c32[-1])
    for k in range(deg-2,-1,-1): q=fma(q,tt,np.full_like(tt,c32[k]))
    rr=f32(q*tt)
    ex=f32(np.exp2(rr.astype(np.float64)))      # hw exp2: 1 ulp; emulate exact-rounded here
    e=f32(np.float32(1.0)-ex)
    e=np.copysign(e,a)
    hx=f32(np.float32(0.5)*x)
    return f32(hx*f32(np.float32(1.0)+e))
import torch
M=8192
x=torch.cat([torch.linspace(-9,9,M-512,dtype=torch.float64),torch.linspace(-1.4,-1.2,256,dtype=torch.float64),torch.linspace(1.2,1.4,256,dtype=torch.float64)]).float()
ref=torch.nn.functional.gelu(x.double()).numpy()
out=gelu_new(x.numpy())
err=np.abs(out.astype(np.float64)-ref)
tol=1.2e-7*np.maximum(np.abs(x.numpy().astype(np.float64)),1.0)+2e-7*np.abs(ref)
print('test-grid max err/tol', (err/tol).max(), 'max abs err', err.max())
t32=np.abs(torch.nn.functional.gelu(x).double().numpy()-ref).max()
print('torch fp32 gelu max abs err', t32, ' ours', err.max(), ' second assert ok:', err.max()<=t32+2.5e-7)
# dense random check
xs=np.random.default_rng(0).normal(0,2.5,2_000_000).astype(np.float32)
refd=torch.nn.functional.gelu(torch.from_numpy(xs).double()).numpy()
outd=gelu_new(xs)
e2=np.abs(outd.astype(np.float64)-refd)
tol2=1.2e-7*np.maximum(np.abs(xs.astype(np.float64)),1.0)+2e-7*np.abs(refd)
print('dense max err/tol', (e2/tol2).max(), 'max abs', e2.max(), 'torch32 max abs', np.abs(torch.nn.functional.gelu(torch.from_numpy(xs)).double().numpy()-refd).max())
