import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr
from megreader_amd.decoders import SegDetector, L1BalanceCELoss
from megreader_amd.synthetic import detection_batch
from oracle.seg_detector import SegDetectorOracle, l1_balance_ce_loss
mr.set_compute_dtype(torch.float32)
chans = [16, 32, 64, 128]
torch.manual_seed(3)
ora = SegDetectorOracle(in_channels=chans, inner_channels=64, k=50, adaptive=True).double().train()
model = SegDetector(in_channels=chans, inner_channels=64, k=50, adaptive=True)
model.load_state_dict({k: v.float() for k, v in ora.state_dict().items()}, strict=True)
model.cuda().train()
g = torch.Generator().manual_seed(0)
feats = [torch.randn(2, c, 64 // s, 64 // s, generator=g) for c, s in zip(chans, (1, 2, 4, 8))]
batch = detection_batch(2, 256, seed=1, boxes=3)
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu(); return float((a - b).abs().max() / (b.abs().max() + 1e-12))
mode = sys.argv[1] if len(sys.argv) > 1 else "loss"
fo = [f.double().requires_grad_(True) for f in feats]
po = ora(fo)
fd = [f.cuda().requires_grad_(True) for f in feats]
pd = model(fd)
if mode == "loss":
    lo = l1_balance_ce_loss(po, {k: v.double() for k, v in batch.items()}); lo.backward()
    ld, _ = L1BalanceCELoss()(pd, {k: v.cuda() for k, v in batch.items()}); ld.backward()
else:   # plain linear functional of the outputs: isolates the head's backward from the loss
    gg = torch.Generator().manual_seed(5)
    ws = {k: torch.randn(po[k].shape, generator=gg) for k in ("binary", "thresh")}
    sum((po[k] * ws[k].double()).sum() for k in ws).backward()
    sum((pd[k] * ws[k].cuda()).sum() for k in ws).backward()
for i, (a, b) in enumerate(zip(fd, fo)):
    print("feature c%d grad err %.2e" % (i + 2, rel(a.grad, b.grad)))
op = dict(ora.named_parameters())
for k, p in model.named_parameters():
    print("%-22s grad err %.2e" % (k, rel(p.grad, op[k].grad)))
