import os, sys, torch
sys.path.insert(0, os.getcwd())
from unsloth_amd import _lib
from unsloth_amd.kernels import utils as U
from unsloth_amd.kernels.utils import _group, _launch_gemm
L=_lib.lib(); U.GEMM256_MODE="on"
DEV="cuda"; dt=torch.bfloat16
for (M,N,K,rank) in ((4096,8192,320,False),(4096,8192,320,True),(8192,4096,256,False)):
    g=torch.Generator().manual_seed(1)
    X=torch.randn(M,K,generator=g).to(dt).to(DEV)
    W=(torch.randn(N,K,generator=g)*0.05).to(dt).to(DEV)
    xk=torch.zeros(M,64,dtype=dt,device=DEV); xk[:,:16]=torch.randn(M,16,generator=g).to(dt).to(DEV)
    bk=torch.zeros(N,64,dtype=dt,device=DEV); bk[:,:16]=(torch.randn(N,16,generator=g)*0.05).to(dt).to(DEV)
    def run(knob):
        out=torch.full((M,N),0.25,dtype=dt,device=DEV)
        kw=dict(xa=xk,ld_xa=64,R=16,scale=1.0,xk=xk,bk=bk) if rank else {}
        L.uamd_set_tuning(6,0); L.uamd_set_tuning(7,0); L.uamd_set_tuning(11,knob)
        _launch_gemm(X,[_group(W,out,N,W.stride(0),**kw)],nf4=False,accumulate=False,nn=False)
        torch.cuda.synchronize()
        return out
    ref=run(0)
    for knob in (2,9):
        o=run(knob)
        bad=(o!=ref).view(M//256,256,N//256,256).any(dim=3).any(dim=1)
        print(f"M{M} N{N} K{K} rank{rank} knob{knob}: bad tiles {int(bad.sum())}/{bad.numel()}")
        if bad.any():
            idx=bad.nonzero()[:12].tolist(); print("  first bad (tm,tn):",idx)
            # linear dispatch index of tile? print count per tm row
            print("  per-row bad:", bad.sum(dim=1).tolist())

# what is wrong with the second tile of a walk? (M 4096, N 8192, K 320, no rank): tile (0, 4)
M,N,K=4096,8192,320
g=torch.Generator().manual_seed(1)
X=torch.randn(M,K,generator=g).to(dt).to(DEV); W=(torch.randn(N,K,generator=g)*0.05).to(dt).to(DEV)
def run(knob):
    out=torch.full((M,N),0.25,dtype=dt,device=DEV)
    L.uamd_set_tuning(6,0); L.uamd_set_tuning(7,0); L.uamd_set_tuning(11,knob)
    _launch_gemm(X,[_group(W,out,N,W.stride(0))],nf4=False,accumulate=False,nn=False); torch.cuda.synchronize(); return out
ref=run(0).float(); o=run(9).float()
for (tm,tn) in ((0,4),(0,5),(1,4)):
    r0,c0=tm*256,tn*256
    D=(o-ref)[r0:r0+256,c0:c0+256]
    Xf=X.float(); Wf=W.float()
    full=Xf[r0:r0+256]@Wf[c0:c0+256].t()
    print(f"tile ({tm},{tn}): |D|={float(D.norm()):.3f} |full|={float(full.norm()):.3f}")
    for kt in range(5):
        Pk=Xf[r0:r0+256,kt*64:(kt+1)*64]@Wf[c0:c0+256,kt*64:(kt+1)*64].t()
        # correlation of D with -Pk
        print(f"   kt{kt}: <D,-Pk>/|Pk|^2 = {float((D*(-Pk)).sum()/(Pk*Pk).sum()):.3f}")
