import os, sys, torch
sys.path.insert(0, '.')
import megreader_amd as mr
from megreader_amd._lib import load
from megreader_amd.backbones import deformable_resnet50
from oracle.dcn import perturb_offset_convs
from oracle.res50ppm import _Res50Dilated
DEV = 'cuda'
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
golden = torch.load('tests/golden/deformable_resnet50_golden.pt', weights_only=False)
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().cpu().abs().max() + 1e-12))
for block in ("layer4.0", "layer3.0"):
    torch.manual_seed(golden['weight_seed'])
    ora = _Res50Dilated(dilate=False, dcn=True)
    perturb_offset_convs(ora)
    ora.train()
    captured = {}
    mod_o = dict(ora.named_modules())[block]
    hk = mod_o.register_forward_pre_hook(lambda m, inp: captured.__setitem__('x', inp[0].detach().clone()))
    ora(golden['x'])
    hk.remove()
    x = captured['x']
    g = torch.randn(mod_o(x).shape, generator=torch.Generator().manual_seed(3))
    res = {}
    for prec in ("f64", "f32"):
        m = mod_o.double() if prec == "f64" else mod_o.float()
        xo = x.double().clone().requires_grad_(True) if prec == "f64" else x.clone().requires_grad_(True)
        m.zero_grad()
        yo = m(xo)
        yo.backward(g.double() if prec == "f64" else g)
        res[prec] = (yo.detach(), xo.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
    for fused in (1, 0):
        load().mr_set_dcn_fused(fused)
        mr.set_compute_dtype(torch.float32)
        torch.manual_seed(golden['weight_seed'])
        model = deformable_resnet50(pretrained=False)
        perturb_offset_convs(model)
        mod_m = dict(model.named_modules())[block].to(DEV).train()
        xm = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ym = mod_m(xm)
        ym.backward(g.to(DEV).contiguous(memory_format=torch.channels_last))
        y64, gx64, gp64 = res["f64"]
        y32, gx32, gp32 = res["f32"]
        print(block, "fused" if fused else "general", "y vs f64 %.2e (cpu f32 %.2e)  dx vs f64 %.2e (cpu f32 %.2e)" %
              (rel(ym, y64), rel(y32, y64), rel(xm.grad, gx64), rel(gx32, gx64)))
        worst = sorted(((rel(p.grad, gp64[k]), rel(gp32[k], gp64[k]), k) for k, p in mod_m.named_parameters()
                        if float(gp64[k].abs().max()) > 1e-7), reverse=True)[:4]
        print("   worst params (hip vs f64, cpu32 vs f64):", [("%.1e" % a, "%.1e" % b, k) for a, b, k in worst], flush=True)
    load().mr_set_dcn_fused(1)
