import torch, torch.nn.functional as TF
import megreader_amd as mr
from megreader_amd.backbones import crnn_backbone
from megreader_amd.nn import functional as F
mr.set_compute_dtype(torch.float32)
def e(a, r): return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))
for order in ("unfused_only", "fused_then_unfused", "functional_with_module_weights"):
    torch.manual_seed(3)
    net = crnn_backbone().cuda().train()
    stem = net.cnn[0]; conv = stem[0][0]
    x = torch.randn(4, 3, 32, 128, device="cuda")
    g = torch.randn(4, 64, 16, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    w = conv.weight.detach().double().cpu().requires_grad_(True); b = conv.bias.detach().double().cpu().requires_grad_(True)
    c = TF.conv2d(x.double().cpu(), w, b, padding=1); c.retain_grad()
    yr = TF.max_pool2d(TF.relu(c), 2, 2); yr.backward(g.double().cpu())
    if order == "fused_then_unfused":
        stem(x).backward(g); conv.weight.grad = None; conv.bias.grad = None
    if order == "functional_with_module_weights":
        yc = F.conv2d(x, conv.weight, conv.bias, (1, 1), (1, 1), (1, 1), True, True); yc.retain_grad()
        yp = F.max_pool2d(yc, (2, 2), (2, 2), (0, 0), True)
        yp.backward(g)
        print(order, "y %.2e gconv %.2e w %.2e" % (e(yp, yr), e(yc.grad, c.grad), e(conv.weight.grad, w.grad)))
        continue
    feats = {}
    def hook(m, i, o):
        o.retain_grad(); feats['c'] = o
    h = stem[0].register_forward_hook(hook)
    yu = torch.nn.Sequential.forward(stem, x)
    h.remove()
    yu.backward(g)
    print(order, "y %.2e gconv %.2e w %.2e b %.2e" % (e(yu, yr), e(feats['c'].grad, c.grad), e(conv.weight.grad, w.grad), e(conv.bias.grad, b.grad)),
          "relu_gd", conv.relu_grad_downstream, "fuse", conv.fuse_relu, "pool relu_input", stem[1].relu_input, stem[1].kernel_size, stem[1].stride, stem[1].padding)
