#!/usr/bin/env python
"""What the REFERENCE stack does on this GPU: the four benchmarked training steps with stock PyTorch-ROCm kernels (MIOpen
convolutions / BatchNorm / RNN, rocBLAS, torch's own CTC) plus the reference's OWN extensions for the two ops PyTorch does not
have -- its 2D-CTC kernels and its modulated deformable convolution, compiled for gfx950 from the reference's sources
(oracle/build_ref_ext.sh -> oracle/_ref/).  The model code is the oracle restatement of the reference models (oracle/*.py,
the same modules bench.py's cpu_baseline runs on the host cores), moved to the GPU.

This is a measurement tool, not part of the product path: it exists so that the numbers of bench.py have a same-hardware
baseline next to them ("a MegReader user who only installs PyTorch-ROCm and builds the repo's extensions").  fp32 is what the
reference trains in (train.py; apex amp is optional); the bf16 autocast column is the cheapest thing such a user could turn on.

    python tools/bench_reference_stack_gpu.py [--workloads crnn,res50ppm,fpn_attention,db] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
DEV = "cuda"


def ref_ext():
    d = os.path.join(REPO, "oracle", "_ref")
    if not os.path.isdir(d):
        raise SystemExit("oracle/_ref is missing: bash oracle/build_ref_ext.sh (where /root/reference exists)")
    if d not in sys.path:
        sys.path.insert(0, d)
    import ctc_2d_csrc
    import deform_conv_cuda
    return ctc_2d_csrc, deform_conv_cuda


class RefCTC2D(torch.autograd.Function):
    """The body of ops/ctc_2d/ctc_loss_2d.py:9-35 on the reference's extension module."""

    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank=0):
        ext, _ = ref_ext()
        ctx.blank = blank
        nll, log_alpha = ext.ctc2d_forward(log_probs, targets, input_lengths, target_lengths, blank, torch.finfo().tiny)
        ctx.save_for_backward(log_probs, targets, input_lengths, target_lengths, nll, log_alpha)
        return nll

    @staticmethod
    def backward(ctx, grad_output):
        ext, _ = ref_ext()
        log_probs, targets, input_lengths, target_lengths, nll, log_alpha = ctx.saved_tensors
        g = ext.ctc2d_backward(grad_output.contiguous(), log_probs, targets, input_lengths, target_lengths, nll, log_alpha,
                               ctx.blank)
        return g, None, None, None, None


def ref_ctc_loss_2d(pred, targets, il, tl, blank=0):
    pred = pred.float().contiguous()
    return RefCTC2D.apply(pred, targets.to(pred.device), il.to(pred.device), tl.to(pred.device), blank)


class RefMDCN(torch.autograd.Function):
    """ModulatedDeformConvFunction (assets/ops/dcn/functions/deform_conv.py:110-165) on the reference's extension module."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, stride, padding, dilation):
        _, ext = ref_ext()
        ctx.geom = (stride, padding, dilation)
        fake = input.new_empty(1)
        kh, kw = weight.shape[2:4]
        ho = (input.shape[2] + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
        wo = (input.shape[3] + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
        out = input.new_empty((input.size(0), weight.size(0), ho, wo))
        bufs = [input.new_empty(0), input.new_empty(0)]
        ext.modulated_deform_conv_cuda_forward(input, weight, fake, bufs[0], offset, mask, out, bufs[1], kh, kw, stride, stride,
                                               padding, padding, dilation, dilation, 1, 1, False)
        ctx.save_for_backward(input, offset, mask, weight, fake)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        _, ext = ref_ext()
        input, offset, mask, weight, fake = ctx.saved_tensors
        stride, padding, dilation = ctx.geom
        gi, go, gm = torch.zeros_like(input), torch.zeros_like(offset), torch.zeros_like(mask)
        gw, gb = torch.zeros_like(weight), torch.zeros_like(fake)
        bufs = [input.new_empty(0), input.new_empty(0)]
        ext.modulated_deform_conv_cuda_backward(input, weight, fake, bufs[0], offset, mask, bufs[1], gi, gw, gb, go, gm,
                                                grad_output.contiguous(), weight.shape[2], weight.shape[3], stride, stride,
                                                padding, padding, dilation, dilation, 1, 1, False)
        return gi, go, gm, gw, None, None, None


def build(workload):
    torch.manual_seed(0)
    if workload == "crnn":
        from oracle.crnn import CRNNOracle, synthetic_batch
        return CRNNOracle(), synthetic_batch(256, 32, 128, seed=0), 256, "adam"
    if workload == "res50ppm":
        import oracle.res50ppm as m
        m.oracle_ctc_loss_2d = ref_ctc_loss_2d            # the reference's own 2D-CTC kernels instead of the numpy restatement
        return m.Res50PPM2DCTCOracle(), m.synthetic_batch_2d(256, 32, 128, seed=0, max_len=3), 256, "adam"
    if workload == "fpn_attention":
        from oracle.crnn import synthetic_batch
        from oracle.fpn_attention import FPNAttentionOracle
        return FPNAttentionOracle(), synthetic_batch(32, 64, 256, seed=0), 32, "adam"
    import oracle.dcn as d
    from megreader_amd.synthetic import detection_batch
    from oracle.res50ppm import _Res50Dilated
    from oracle.seg_detector import SegDetectorOracle

    def mdcn_forward(self, x, offset, mask):              # the reference's own deformable-convolution kernels
        return RefMDCN.apply(x.float().contiguous(), offset.float(), mask.float(), self.weight.float(), self.stride,
                             self.padding, self.dilation)
    d.OracleModulatedDeformConv.forward = mdcn_forward

    class DB(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = _Res50Dilated(dilate=False, dcn=True)
            self.decoder = SegDetectorOracle(in_channels=[256, 512, 1024, 2048], adaptive=True, k=50)

        def forward(self, image):
            return self.decoder(self.backbone(image))
    return DB(), detection_batch(2, 640, seed=0), 2, "sgd"


def to_dev(b):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}


def run(workload, steps, warmup, autocast):
    model, batch, n, opt_kind = build(workload)
    model = model.to(DEV).train()
    batch = to_dev(batch)
    if opt_kind == "sgd":
        from oracle.seg_detector import l1_balance_ce_loss
        opt = torch.optim.SGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                pred = model(batch['image'])
            loss = l1_balance_ce_loss({k: v.float() for k, v in pred.items()} if isinstance(pred, dict) else pred, batch)
            loss.backward()
            opt.step()
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        lab, ln = batch['label'], batch['length'].long()

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                loss, _ = model(batch['image'], targets=lab, lengths=ln, train=True)
            loss.float().mean().backward()
            opt.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return ms, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="crnn,res50ppm,fpn_attention,db")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--find", action="store_true", help="torch.backends.cudnn.benchmark = True (MIOpen find mode: searches / "
                    "compiles per shape on first use -- minutes of warm-up for a ResNet-50)")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = bool(a.find)
    print("reference stack on %s: oracle restatement of the reference models + torch %s ROCm kernels + the reference's own "
          "2D-CTC / DCN extensions (oracle/_ref); %d timed eager steps after %d warm-up; MIOpen %s" %
          (torch.cuda.get_device_name(0), torch.__version__, a.steps, a.warmup,
           "find mode" if a.find else "immediate mode (PyTorch's default)"))
    for w in a.workloads.split(","):
        for autocast in (False, True):
            try:
                ms, n = run(w, a.steps, a.warmup, autocast)
                print("%-14s %-22s %9.2f ms/step  %10.1f images/s  (batch %d)" %
                      (w, "bf16 autocast" if autocast else "fp32 (as the reference)", ms, 1e3 * n / ms, n), flush=True)
            except Exception as e:  # noqa: BLE001 - a tool: report and go on with the next configuration
                print("%-14s %-22s FAILED: %s: %s" % (w, "bf16 autocast" if autocast else "fp32", type(e).__name__,
                                                     str(e).splitlines()[0][:200]), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
