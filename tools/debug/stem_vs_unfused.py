import torch, torch.nn.functional as TF
import megreader_amd as mr
from megreader_amd.backbones import crnn_backbone
mr.set_compute_dtype(torch.float32)
torch.manual_seed(3)
net = crnn_backbone().cuda().train()
stem = net.cnn[0]
conv = stem[0][0]
x = torch.randn(4, 3, 32, 128, device="cuda")
y_f = stem(x)
g = torch.randn_like(y_f)
y_f.backward(g)
gw_f, gb_f = conv.weight.grad.clone(), conv.bias.grad.clone()
conv.weight.grad = None; conv.bias.grad = None
y_u = torch.nn.Sequential.forward(stem, x)
y_u.backward(g)
gw_u = conv.weight.grad.clone()
# torch reference
w = conv.weight.detach().double().cpu().requires_grad_(True); b = conv.bias.detach().double().cpu().requires_grad_(True)
yr = TF.max_pool2d(TF.relu(TF.conv2d(x.double().cpu(), w, b, padding=1)), 2, 2)
yr.backward(g.double().cpu())
def e(a, r): return float((a.double().cpu() - r).abs().max() / r.abs().max())
print("fused   vs torch: y %.2e w %.2e" % (e(y_f, yr), e(gw_f, w.grad)))
print("unfused vs torch: y %.2e w %.2e" % (e(y_u, yr), e(gw_u, w.grad)))
d = (gw_u.double().cpu() - w.grad).abs()
print("unfused err per c", d.amax(dim=(0, 2, 3)).tolist()); print("per rs", d.amax(dim=(0, 1)).tolist())
print("w.grad max", float(w.grad.abs().max()), "strides", conv.weight.stride(), gw_u.stride(), g.stride(), y_u.stride())
