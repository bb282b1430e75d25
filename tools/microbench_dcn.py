#!/usr/bin/env python
"""DCNv2 at the 13 deformable layers of `deformable_resnet50` for 640x640 inputs (reference backbones/resnet.py:295-309:
the 3x3 conv2 of every Bottleneck in layer2/3/4; seg_detector_db.yaml:53,73), through the C ABI (mr_dcn2_fwd / mr_dcn2_bwd,
NHWC bf16).  Per layer: time, algorithmic FLOP/s of the GEMM part and algorithmic bytes/s (input + offsets
+ mask + weights + output once each; backward: those plus every gradient once) against the MI355X rooflines, and the
ACTUAL HBM-side bytes the current implementation moves on top (the column matrix, written and re-read)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402
from megreader_amd._lib import call, ptr  # noqa: E402

HBM_TBS, MFMA_TFS = 8.0, 2500.0
# (name, count, C = Cout, H = W of the conv input at 640x640, stride): first block of a layer strides
LAYERS = [("layer2.0", 1, 128, 160, 2), ("layer2.1-3", 3, 128, 80, 1), ("layer3.0", 1, 256, 80, 2),
          ("layer3.1-5", 5, 256, 40, 1), ("layer4.0", 1, 512, 40, 2), ("layer4.1-2", 2, 512, 20, 1)]


def timeit(f, iters):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--offscale", type=float, default=0.5, help="std of the random offsets in pixels")
    ap.add_argument("--v1", type=int, default=1, help="general path: 1 = round-1 backward kernels, 0 = round-2 experiments")
    ap.add_argument("--layers", default="", help="comma-separated substrings of layer names to run (default: all)")
    ap.add_argument("--fused", type=int, default=1, help="1 (default): fused kernels (csrc/dcn_fused.hip), 0: general path")
    ap.add_argument("--reference", action="store_true",
                    help="also time the REFERENCE's own extension (oracle/_ref, built by oracle/build_ref_ext.sh from "
                         "assets/ops/dcn/src) on the same layers: NCHW float32, the call sequence of its Function file")
    a = ap.parse_args()
    dtype, dt, es = torch.bfloat16, 1, 2
    mr.set_compute_dtype(dtype)
    from megreader_amd._lib import load
    load().mr_set_dcn_v1_bwd(a.v1)
    load().mr_set_dcn_fused(a.fused)
    from megreader_amd._lib import dcn_workspace
    N = a.batch
    tot = {"fwd": 0.0, "bwd": 0.0, "ref_fwd": 0.0, "ref_bwd": 0.0}
    ref_ext = None
    if a.reference:
        from oracle.gen_golden_dcn import load_reference_extension     # tools/ may use the checker; the product path never does
        ref_ext, _ = load_reference_extension()
        if ref_ext is None:
            raise SystemExit("--reference: oracle/_ref holds no reference extension (bash oracle/build_ref_ext.sh)")
    print("DCNv2 3x3, batch %d, bf16 activations / f32 offsets+mask, offsets ~ N(0, %.1f px), backward kernels %s; "
          "rooflines %.0f TB/s, %.0f TFLOP/s" % (N, a.offscale, "fused (round 3)" if a.fused else ("round 1" if a.v1 else "round 2"), HBM_TBS, MFMA_TFS))
    for name, count, C, H, s in LAYERS:
        if a.layers and not any(k in name for k in a.layers.split(",")):
            continue
        W, Co, k, pad = H, C, 3, 1
        Ho = Wo = (H + 2 * pad - k) // s + 1
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
        w_n = (torch.randn(Co, k * k * C, device="cuda", generator=g) * 0.05).to(dtype)
        w_t = w_n.t().contiguous()
        off = torch.randn(N, 2 * k * k, Ho, Wo, device="cuda", generator=g) * a.offscale
        msk = torch.rand(N, k * k, Ho, Wo, device="cuda", generator=g)
        y = torch.empty(N, Ho, Wo, Co, device="cuda", dtype=dtype)
        lib = load()
        # forward column workspace (kept for the backward's weight gradient where the library says so, mr_dcn2_col_saved) and the
        # backward workspace (CSR + gcol), as megreader_amd/assets/ops/dcn/deform_conv.py allocates them
        colf = torch.empty(max(lib.mr_dcn2_ws_bytes(dt, N, H, W, C, Co, k, k, Ho, Wo, 0), 16), dtype=torch.uint8, device="cuda")
        col = torch.zeros(max(lib.mr_dcn2_ws_bytes(dt, N, H, W, C, Co, k, k, Ho, Wo, 1), 16), dtype=torch.uint8, device="cuda")
        col_saved = ptr(colf) if (a.fused and lib.mr_dcn2_col_saved(dt, H, W, C, Co, k, k)) else 0
        gy = torch.randn(N, Ho, Wo, Co, device="cuda", generator=g).to(dtype)
        dx32 = torch.zeros(N, H, W, C, device="cuda")
        doff, dmsk = torch.zeros_like(off), torch.zeros_like(msk)
        gw = torch.zeros(Co, k * k * C, device="cuda")

        def fwd():
            call("mr_dcn2_fwd", dt, ptr(x), ptr(w_n), 0, ptr(off), off[0].numel(), ptr(msk), msk[0].numel(), ptr(y),
                 ptr(colf), N, H, W, C, Co, k, k, s, pad, 1, Ho, Wo)

        def bwd():
            call("mr_dcn2_bwd3", dt, ptr(gy), ptr(x), ptr(w_t), ptr(off), off[0].numel(), ptr(msk), msk[0].numel(),
                 ptr(col), ptr(dx32), 0, 0, ptr(doff), ptr(dmsk), ptr(gw), 0, col_saved, N, H, W, C, Co, k, k, s, pad, 1, Ho, Wo)

        tf, tb = timeit(fwd, a.iters), timeit(bwd, a.iters)
        P = N * Ho * Wo
        flops = 2.0 * P * Co * k * k * C
        bx, by = es * N * H * W * C, es * P * Co
        boff = 4 * 27 * P
        bw = es * Co * k * k * C
        alg_f = bx + boff + bw + by
        alg_b = alg_f + (4 * N * H * W * C) + boff + 4 * Co * k * k * C      # + dx (f32), doffset/dmask, dw (f32)
        colb = 0 if a.fused else es * P * k * k * C
        print("%-11s x%d C=%3d %3dx%-3d s%d | fwd %7.1f us: %6.1f TFLOP/s (%.3f of MFMA), alg %6.2f MB -> %5.2f TB/s "
              "(%.3f of HBM); col matrix +%6.1f MB | bwd %7.1f us: %6.1f TFLOP/s (%.3f), alg %6.2f MB -> %5.2f TB/s (%.3f)"
              % (name, count, C, H, W, s, tf, flops / tf / 1e6, flops / tf / 1e6 / MFMA_TFS, alg_f / 1e6,
                 alg_f / tf / 1e6, alg_f / tf / 1e6 / HBM_TBS, 2 * colb / 1e6, tb, 2 * flops / tb / 1e6,
                 2 * flops / tb / 1e6 / MFMA_TFS, alg_b / 1e6, alg_b / tb / 1e6, alg_b / tb / 1e6 / HBM_TBS), flush=True)
        tot["fwd"] += count * tf
        tot["bwd"] += count * tb
        if ref_ext is not None:
            # the reference as it runs in its own model: NCHW float32 tensors, ModulatedDeformConvFunction's calls
            # (functions/deform_conv.py:110-165: caller-allocated output, zeroed gradients every backward)
            xr = x.float().permute(0, 3, 1, 2).contiguous()
            wr = w_n.float().view(Co, k, k, C).permute(0, 3, 1, 2).contiguous()
            gyr = gy.float().permute(0, 3, 1, 2).contiguous()
            fake, bufs = xr.new_empty(1), [xr.new_empty(0), xr.new_empty(0)]
            out = xr.new_empty((N, Co, Ho, Wo))

            def rfwd():
                ref_ext.modulated_deform_conv_cuda_forward(xr, wr, fake, bufs[0], off, msk, out, bufs[1], k, k, s, s, pad, pad, 1, 1,
                                                           1, 1, False)

            def rbwd():
                gi, go, gm = torch.zeros_like(xr), torch.zeros_like(off), torch.zeros_like(msk)
                gwr, gb = torch.zeros_like(wr), torch.zeros_like(fake)
                ref_ext.modulated_deform_conv_cuda_backward(xr, wr, fake, bufs[0], off, msk, bufs[1], gi, gwr, gb, go, gm, gyr, k, k,
                                                            s, s, pad, pad, 1, 1, 1, 1, False)

            rf, rb = timeit(rfwd, a.iters), timeit(rbwd, a.iters)
            tot["ref_fwd"] += count * rf
            tot["ref_bwd"] += count * rb
            print("%-11s    reference extension (f32 NCHW, im2col + GEMM per sample) | fwd %7.1f us (%.1fx)             "
                  "                                          | bwd %7.1f us (%.1fx)" % ("", rf, rf / tf, rb, rb / tb), flush=True)
    print("all 13 layers: fwd %.1f us, bwd %.1f us per step of batch %d" % (tot["fwd"], tot["bwd"], N))
    if ref_ext is not None:
        print("reference extension, same 13 layers on this GPU: fwd %.1f us, bwd %.1f us (%.1fx / %.1fx the HIP kernels)"
              % (tot["ref_fwd"], tot["ref_bwd"], tot["ref_fwd"] / tot["fwd"], tot["ref_bwd"] / tot["bwd"]))


if __name__ == "__main__":
    main()
