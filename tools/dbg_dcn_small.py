import sys, torch
sys.path.insert(0, '.')
import megreader_amd as mr
from megreader_amd._lib import load
from megreader_amd.assets.ops.dcn import modulated_deform_conv
from oracle.dcn import modulated_deform_conv2d
DEV='cuda'
def rel(a,b):
    a,b=a.double().cpu(),b.double().cpu(); return float((a-b).abs().max()/(b.abs().max()+1e-12))
for (N,C,Co,H,W,stride,oscale) in [(2,512,512,6,6,2,0.5),(2,512,512,6,6,2,0.0),(2,128,128,6,6,2,0.5),(2,512,512,6,6,1,0.5),(2,512,512,12,12,2,0.5),(1,64,64,6,6,2,0.5)]:
    for dtype in (torch.float32,):
        mr.set_compute_dtype(dtype)
        g=torch.Generator().manual_seed(1)
        Ho=(H+2-3)//stride+1; Wo=Ho
        x=torch.randn(N,C,H,W,generator=g)
        off=torch.randn(N,18,H,W,generator=g)*oscale
        msk=torch.sigmoid(torch.randn(N,9,H,W,generator=g))
        w=torch.randn(Co,C,3,3,generator=g)*0.02
        gy=torch.randn(N,Co,Ho,Wo,generator=g)
        xr=x.double().requires_grad_(True); offr=off.double().requires_grad_(True); mskr=msk.double().requires_grad_(True); wr=w.double().requires_grad_(True)
        yr=modulated_deform_conv2d(xr,offr,mskr,wr,None,stride,1,1); yr.backward(gy.double())
        def run():
            xd=x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            offd,mskd=off.to(DEV).requires_grad_(True),msk.to(DEV).requires_grad_(True)
            wd=w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y=modulated_deform_conv(xd,offd,mskd,wd,None,stride,1,1,1,1)
            y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
            return y.detach(),xd.grad,offd.grad,mskd.grad,wd.grad
        got=run()
        old=load().mr_set_dcn_fused(0)
        ref=run()
        load().mr_set_dcn_fused(old)
        names=("y","dx","doff","dmask","dw")
        print((N,C,Co,H,W,stride,oscale), "fused vs oracle:", ["%s %.1e"%(n,rel(a,b)) for n,a,b in zip(names,got,(yr,xr.grad,offr.grad,mskr.grad,wr.grad))],
              "general vs oracle:", ["%s %.1e"%(n,rel(a,b)) for n,a,b in zip(names,ref,(yr,xr.grad,offr.grad,mskr.grad,wr.grad))], flush=True)
