import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from unsloth_amd import _lib
from unsloth_amd.kernels import utils as U
from unsloth_amd.kernels.utils import _group, _launch_gemm
DEV="cuda"; bf=torch.bfloat16
def run(fn, iters):
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e-3
def padded(t, pad):
    r,c=t.shape
    p=torch.empty(r,c+pad,device=DEV,dtype=t.dtype)[:, :c]; p.copy_(t); return p
L=_lib.lib(); U.GEMM256_MODE="on"
for form,(M,N,K) in (("NT down",(8192,4096,14336)),("NN gate-dX",(8192,4096,14336)),("NT gate",(8192,14336,4096)),("NN down-dX",(8192,14336,4096))):
    X=torch.randn(M,K,device=DEV,dtype=bf)
    nn=form.startswith("NN")
    W=(torch.randn(K,N,device=DEV)*0.02).to(bf) if nn else (torch.randn(N,K,device=DEV)*0.02).to(bf)
    out=torch.empty(M,N,device=DEV,dtype=bf)
    res={}
    for pa in (0,64,32,128):
        for pb in (0,64):
            Xp=padded(X,pa) if pa else X
            Wp=padded(W,pb) if pb else W
            def f():
                _launch_gemm(Xp,[_group(Wp,out,N,Wp.stride(0))],nf4=False,accumulate=False,nn=nn)
            run(f,3)
            t=min(run(f,10) for _ in range(4))
            res[f"A+{pa},B+{pb}"]=round(2.0*M*N*K/t/1e12,1)
    print(json.dumps({"form":form,"M":M,"N":N,"K":K,**res}),flush=True)
