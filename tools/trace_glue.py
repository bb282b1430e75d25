"""Attribute the ATen 'glue' launches of one eager training step (fills, copies, adds, casts ...) to the Python line of
megreader_amd that issued them.  usage: python tools/trace_glue.py --workload db|fpn_attention|res50ppm|crnn"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(workload, dev):
    import megreader_amd as mr
    from megreader_amd.optim import FusedAdam, FusedSGD
    mr.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    if workload == "db":
        from megreader_amd.backbones import deformable_resnet50
        from megreader_amd.decoders import L1BalanceCELoss, SegDetector
        from megreader_amd.synthetic import detection_batch

        class M(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.backbone = deformable_resnet50(pretrained=False)
                self.decoder = SegDetector(in_channels=[256, 512, 1024, 2048], adaptive=True, k=50)
                self.criterion = L1BalanceCELoss()

            def forward(self, batch):
                return self.criterion(self.decoder(self.backbone(batch['image'])), batch)
        model = M().to(dev).train()
        opt = FusedSGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)
        batch = {k: v.to(dev) for k, v in detection_batch(2, 640, seed=0).items()}

        def step():
            opt.zero_grad()
            loss, _ = model(batch)
            loss.mean().backward()
            opt.step()
        return step
    from megreader_amd import synthetic
    if workload == "fpn_attention":
        from megreader_amd.backbones import Resnet50FPN
        from megreader_amd.decoders import AttentionDecoder
        bb, dec = Resnet50FPN(resnet_pretrained=False), AttentionDecoder(in_channels=256, gt_as_output=True)
        b = synthetic.recognition_batch(32, 64, 256, seed=0)
    elif workload == "res50ppm":
        from megreader_amd.backbones import resnet50dilated_ppm
        from megreader_amd.decoders import CTCDecoder2D
        bb, dec = resnet50dilated_ppm(), CTCDecoder2D(in_channels=256)
        b = synthetic.recognition_batch_2d(256, 32, 128, seed=0, max_len=3)
    else:
        from megreader_amd.backbones import crnn_backbone
        from megreader_amd.decoders import CRNNDecoder
        bb, dec = crnn_backbone(), CRNNDecoder(in_channels=512, inner_channels=256)
        b = synthetic.recognition_batch(256, 32, 128, seed=0)

    class M2(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone, self.decoder = bb, dec

        def forward(self, data, *a, **k):
            return self.decoder(self.backbone(data), *a, **k)
    model = M2().to(dev).train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    img, lab, ln = b['image'].to(dev), b['label'].to(dev), b['length'].to(dev).long()

    def step():
        opt.zero_grad()
        loss, _ = model(img, targets=lab, lengths=ln, train=True)
        loss.mean().backward()
        opt.step()
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="db")
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    step = build(args.workload, dev)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    cfg = torch._C._profiler._ExperimentalConfig(verbose=True)      # without it the Python stacks come back empty
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=cfg) as prof:
        step()
        torch.cuda.synchronize()
    counts = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
            continue    # top-level ATen calls only
        frames = [f for f in (ev.stack or []) if "megreader_amd" in f and "_lib.py" not in f]
        where = " < ".join(f.split("megreader_amd/")[-1] for f in frames[:2]) if frames else "(autograd engine)"
        counts[(ev.name, where)] += 1
    total = sum(counts.values())
    print("%s: %d top-level ATen calls in one eager step" % (args.workload, total))
    for (name, where), n in counts.most_common(args.top):
        print("%5d  %-28s %s" % (n, name, where[:150]))


if __name__ == "__main__":
    main()
