import torch, torch.nn.functional as TF
import megreader_amd as mr
from megreader_amd.nn import functional as F
for dtype in (torch.float32, torch.bfloat16):
    mr.set_compute_dtype(dtype)
    for (N, C, H, W, K) in [(4, 3, 32, 128, 64), (4, 4, 32, 128, 64), (4, 3, 8, 8, 64), (1, 3, 4, 4, 16), (4, 8, 32, 128, 64)]:
        torch.manual_seed(0)
        x = torch.randn(N, C, H, W)
        w = torch.randn(K, C, 3, 3) * 0.3
        b = torch.randn(K) * 0.1
        g = torch.randn(N, K, H, W)
        if dtype == torch.bfloat16:
            x, w, g = x.bfloat16().float(), w.bfloat16().float(), g.bfloat16().float()
        wr = w.double().requires_grad_(True); br = b.double().requires_grad_(True)
        yr = TF.conv2d(x.double(), wr, br, padding=1)
        yr.backward(g.double())
        wd = w.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True)
        y = F.conv2d(x.cuda(), wd, bd, (1, 1), (1, 1), (1, 1), False)
        y.backward(g.cuda().to(dtype))
        ey = float((y.double().cpu() - yr).abs().max() / yr.abs().max())
        ew = float((wd.grad.double().cpu() - wr.grad).abs().max() / wr.grad.abs().max())
        eb = float((bd.grad.double().cpu() - br.grad).abs().max() / br.grad.abs().max())
        print(dtype, (N, C, H, W, K), "y %.2e w %.2e b %.2e" % (ey, ew, eb))
        if ew > 1e-3 and dtype == torch.float32:
            d = (wd.grad.double().cpu() - wr.grad).abs()
            print("  worst idx", [int(i) for i in torch.nonzero(d == d.max())[0]], "per-c max", d.amax(dim=(0, 2, 3)).tolist(), "per-rs", d.amax(dim=(0, 1)).tolist())
