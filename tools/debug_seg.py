import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr
from megreader_amd.decoders import SegDetector
from oracle.seg_detector import SegDetectorOracle
mr.set_compute_dtype(torch.float32)
chans = [16, 32, 64, 128]
torch.manual_seed(3)
ora = SegDetectorOracle(in_channels=chans, inner_channels=64, k=50, adaptive=True).train()
model = SegDetector(in_channels=chans, inner_channels=64, k=50, adaptive=True)
model.load_state_dict(ora.state_dict(), strict=True)
model.cuda().train()
g = torch.Generator().manual_seed(0)
feats = [torch.randn(2, c, 64 // s, 64 // s, generator=g) for c, s in zip(chans, (1, 2, 4, 8))]
outs = {}
def hook(store, name):
    def f(m, i, o):
        store[name] = o.detach().float().cpu() if torch.is_tensor(o) else None
    return f
so, sm = {}, {}
for name, m in ora.named_modules():
    if name: m.register_forward_hook(hook(so, name))
for name, m in model.named_modules():
    if name: m.register_forward_hook(hook(sm, name))
po = ora(feats); pm = model([f.cuda() for f in feats])
for name in so:
    if name in sm and so[name] is not None and sm[name] is not None and so[name].shape == sm[name].shape:
        d = float((so[name] - sm[name]).abs().max()); print("%-14s max|d| %.3e  (max|ref| %.3e) %s" % (name, d, float(so[name].abs().max()), tuple(so[name].shape)))
    else:
        print(name, "shape mismatch / missing", None if name not in sm or sm[name] is None else tuple(sm[name].shape), None if so[name] is None else tuple(so[name].shape))
