import torch, torch.nn.functional as TF
import megreader_amd as mr
from megreader_amd.nn import functional as F
mr.set_compute_dtype(torch.float32)
torch.manual_seed(3)
x = torch.randn(4, 3, 32, 128)
w0 = torch.randn(64, 3, 3, 3) * 0.2
b0 = torch.randn(64) * 0.1
g = torch.randn(4, 64, 16, 64)
def e(a, r): return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))
for cl in (False, True):
    w = w0.double().requires_grad_(True); b = b0.double().requires_grad_(True)
    c = TF.conv2d(x.double(), w, b, padding=1); c.retain_grad()
    yr = TF.max_pool2d(TF.relu(c), 2, 2); yr.backward(g.double())
    wd = w0.cuda()
    if cl: wd = wd.contiguous(memory_format=torch.channels_last)
    wd.requires_grad_(True); bd = b0.cuda().requires_grad_(True)
    yc = F.conv2d(x.cuda(), wd, bd, (1, 1), (1, 1), (1, 1), True, True); yc.retain_grad()
    yp = F.max_pool2d(yc, (2, 2), (2, 2), (0, 0), True)
    yp.backward(g.cuda())
    print("channels_last weight", cl, "y %.2e gconv %.2e w %.2e b %.2e" % (e(yp, yr), e(yc.grad, c.grad), e(wd.grad, w.grad), e(bd.grad, b.grad)), wd.grad.stride())
    # wgrad alone with the (verified) conv-level gradient
    wd2 = wd.detach().clone().requires_grad_(True)
    yc2 = F.conv2d(x.cuda(), wd2, None, (1, 1), (1, 1), (1, 1), False)
    yc2.backward(c.grad.float().cuda())
    print("   dense path with same g: w %.2e" % e(wd2.grad, w.grad), wd2.grad.stride(), wd2.stride())
