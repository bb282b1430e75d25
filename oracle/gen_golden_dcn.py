"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Golden vectors of the REFERENCE's deformable-convolution extension.

Runs the reference's own `deform_conv_cuda` / `deform_pool_cuda` modules -- built for gfx950 from the reference's sources where
they lie by oracle/build_ref_ext.sh into oracle/_ref/ -- on the MI355X, with the calling sequences of the reference's Function
files (assets/ops/dcn/functions/deform_conv.py:36-89,110-165, functions/deform_pool.py), and stores inputs + outputs as
float32 arrays in ONE npz:

    python oracle/gen_golden_dcn.py --out tests/golden/dcn_reference_ext.npz      # on a GPU box (the .so files travel there)

What the fixture pins (tests/test_oracle_dcn_pinned_cpu.py on CPU, tests/test_dcn_reference_gpu.py on the GPU):
  * oracle/dcn.py          (torch float64 restatement of deform_conv_cuda_kernel.cu:466-766) -- DCNv2 and, with a unit mask, v1;
    including the flat re-interpretation of a stride-1 offset map by a stride-2 layer (quirk Q10) and the non-contiguous
    offset slice of a 27-channel map (backbones/resnet.py:162-164);
  * oracle/deform_pool.py  (numpy restatement of deform_pool_cuda_kernel.cu:52-268);
  * the HIP kernels behind megreader_amd.assets.ops.dcn.{deform_conv_cuda, deform_pool_cuda} (same inputs, same calls).
Sample points are kept away from integer coordinates (fractional parts in [0.25, 0.75]) except in the `kink` case, which holds
integer and half-integer offsets on purpose: there the forward is still well defined (the kernels agree on floor()), only the
gradient with respect to the offset is not, and the fixture's consumers skip that gradient for this case.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

# N, C, Co, H, W, stride, pad, dil, offset-map size (None = output grid), offset scale, bias
DCN2_CASES = [
    ("plain", (2, 16, 24, 7, 9, 1, 1, 1, None, 2.0, True)),
    ("stride2_flat_offsets", (2, 32, 32, 12, 10, 2, 1, 1, (12, 10), 1.5, True)),      # quirk Q10
    ("dilated", (1, 16, 16, 9, 8, 1, 2, 2, None, 1.0, False)),
    ("large_offsets", (3, 8, 8, 5, 6, 1, 1, 1, None, 4.0, True)),
    ("c64_ragged", (2, 64, 32, 19, 21, 1, 1, 1, None, 1.0, False)),
    ("c128_stride2", (1, 128, 64, 20, 18, 2, 1, 1, (20, 18), 1.5, True)),
]
# v1 (DeformConv; no reference model uses it) runs with ONE sample per call: under current PyTorch the reference's v1 host code
# only works for batchSize == im2col_step == 1 -- with a step > 1 deform_conv_cuda.cpp:423-426 `.view`s the `zeros_like` of a
# transposed tensor (refused: zeros_like now preserves strides), and with batchSize / im2col_step > 1 the forward loop
# re-views `columns` on its second trip without having reshaped it back (deform_conv_cuda.cpp:225).
DCN1_CASES = [("v1_plain", (1, 16, 24, 9, 8, 1, 1, 1)), ("v1_stride2", (1, 8, 16, 11, 12, 2, 1, 1)),
              ("v1_dilated", (1, 16, 8, 10, 9, 1, 2, 2))]
POOL_CASES = [dict(), dict(no_trans=True), dict(group_size=1, pooled=7, part=7, spp=4, C=6, output_dim=6),
              dict(classes=2, output_dim=4, C=16, group_size=2), dict(output_dim=70, group_size=1, C=70, pooled=2, part=1),
              dict(R=1, B=1, pooled=1, part=1, spp=1, group_size=1, output_dim=3, C=3)]


def load_reference_extension():
    """The two modules of oracle/_ref (None, None when they have not been built / did not travel)."""
    if not os.path.isdir(REF_DIR) or not any(f.startswith("deform_conv_cuda") for f in os.listdir(REF_DIR)):
        return None, None
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    import deform_conv_cuda  # noqa: E402  (the reference's PYBIND11 module, deform_conv_cuda.cpp:681-695)
    import deform_pool_cuda  # noqa: E402
    return deform_conv_cuda, deform_pool_cuda


def out_size(H, W, k, stride, pad, dil):
    return (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1, (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def dcn2_inputs(name, case, kink=False):
    N, C, Co, H, W, stride, pad, dil, omap, oscale, with_bias = case
    g = torch.Generator().manual_seed(1000 + C + H + Co)
    Ho, Wo = out_size(H, W, 3, stride, pad, dil)
    oh, ow = omap if omap else (Ho, Wo)
    x = torch.randn(N, C, H, W, generator=g)
    om = torch.randn(N, 27, oh, ow, generator=g) * oscale          # the 27-channel map of backbones/resnet.py:162-164
    if kink:
        om[:, :18] = torch.round(om[:, :18] * 2) / 2               # integers and halves
    else:
        om[:, :18] = torch.floor(om[:, :18]) + 0.25 + 0.5 * torch.rand(N, 18, oh, ow, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.2
    b = torch.randn(Co, generator=g) if with_bias else None
    gy = torch.randn(N, Co, Ho, Wo, generator=g)
    return x, om, w, b, gy, stride, pad, dil


def run_dcn2(ext, x, om, w, b, gy, stride, pad, dil, dev="cuda"):
    """ModulatedDeformConvFunction.forward / .backward of the reference, call for call (functions/deform_conv.py:110-165)."""
    input, weight, grad_output = x.to(dev), w.to(dev), gy.to(dev)
    omd = om.to(dev)
    offset, mask = omd[:, :18], torch.sigmoid(omd[:, 18:27])        # non-contiguous slice / fresh tensor, as in resnet.py
    with_bias = b is not None
    bias = b.to(dev) if with_bias else input.new_empty(1)
    kh, kw = weight.shape[2:4]
    Ho, Wo = out_size(input.shape[2], input.shape[3], kh, stride, pad, dil)
    output = input.new_empty((input.size(0), weight.size(0), Ho, Wo))
    bufs = [input.new_empty(0), input.new_empty(0)]
    ext.modulated_deform_conv_cuda_forward(input, weight, bias, bufs[0], offset, mask, output, bufs[1], kh, kw, stride, stride,
                                           pad, pad, dil, dil, 1, 1, with_bias)
    gi, go, gm = torch.zeros_like(input), torch.zeros_like(offset), torch.zeros_like(mask)
    gw, gb = torch.zeros_like(weight), torch.zeros_like(bias)
    ext.modulated_deform_conv_cuda_backward(input, weight, bias, bufs[0], offset, mask, bufs[1], gi, gw, gb, go, gm, grad_output,
                                            kh, kw, stride, stride, pad, pad, dil, dil, 1, 1, with_bias)
    torch.cuda.synchronize()
    return dict(output=output, grad_input=gi, grad_offset=go, grad_mask=gm, grad_weight=gw,
                grad_bias=gb if with_bias else None)


def dcn1_inputs(name, case):
    N, C, Co, H, W, stride, pad, dil = case
    g = torch.Generator().manual_seed(2000 + C + H + Co)
    Ho, Wo = out_size(H, W, 3, stride, pad, dil)
    x = torch.randn(N, C, H, W, generator=g)
    off = torch.floor(torch.randn(N, 18, Ho, Wo, generator=g) * 1.5) + 0.25 + 0.5 * torch.rand(N, 18, Ho, Wo, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.2
    gy = torch.randn(N, Co, Ho, Wo, generator=g)
    return x, off, w, gy, stride, pad, dil


def run_dcn1(ext, x, off, w, gy, stride, pad, dil, dev="cuda"):
    """DeformConvFunction.forward / .backward of the reference (functions/deform_conv.py:36-89): width-first argument order,
    caller-allocated output and zeroed gradients, the weight gradient accumulated twice (scale 1 and 0.5) to pin `scale`."""
    input, offset, weight, grad_output = x.to(dev), off.to(dev), w.to(dev), gy.to(dev)
    N = input.shape[0]
    Ho, Wo = out_size(input.shape[2], input.shape[3], 3, stride, pad, dil)
    output = input.new_empty((N, weight.size(0), Ho, Wo))
    bufs = [input.new_empty(0), input.new_empty(0)]
    step = min(64, N)       # functions/deform_conv.py:43 (N == 1 here, see DCN1_CASES)
    ext.deform_conv_forward_cuda(input, weight, offset, output, bufs[0], bufs[1], weight.size(3), weight.size(2), stride, stride,
                                 pad, pad, dil, dil, 1, 1, step)
    gi, go = torch.zeros_like(input), torch.zeros_like(offset)
    ext.deform_conv_backward_input_cuda(input, offset, grad_output, gi, go, weight, bufs[0], weight.size(3), weight.size(2),
                                        stride, stride, pad, pad, dil, dil, 1, 1, step)
    gw = torch.zeros_like(weight)
    ext.deform_conv_backward_parameters_cuda(input, offset, grad_output, gw, bufs[0], bufs[1], weight.size(3), weight.size(2),
                                             stride, stride, pad, pad, dil, dil, 1, 1, 1, step)
    ext.deform_conv_backward_parameters_cuda(input, offset, grad_output, gw, bufs[0], bufs[1], weight.size(3), weight.size(2),
                                             stride, stride, pad, pad, dil, dil, 1, 1, 0.5, step)
    torch.cuda.synchronize()
    return dict(output=output, grad_input=gi, grad_offset=go, grad_weight_x1p5=gw)


def pool_args(kw):
    return (kw['no_trans'], kw['spatial_scale'], kw['output_dim'], kw['group_size'], kw['pooled_size'], kw['part_size'],
            kw['sample_per_part'], kw['trans_std'])


def run_pool(ext, data, rois, trans, kw, g, dev="cuda"):
    """DeformRoIPoolingFunction.forward / .backward of the reference (functions/deform_pool.py): caller-allocated `output`
    and `output_count`, zeroed gradients; `trans` is an empty tensor with no_trans."""
    d, r = torch.from_numpy(data).to(dev), torch.from_numpy(rois).to(dev)
    t = d.new_empty(0) if kw['no_trans'] else torch.from_numpy(trans).to(dev)
    n = rois.shape[0]
    out = d.new_empty((n, kw['output_dim'], kw['pooled_size'], kw['pooled_size']))
    cnt = d.new_empty((n, kw['output_dim'], kw['pooled_size'], kw['pooled_size']))
    ext.deform_psroi_pooling_cuda_forward(d, r, t, out, cnt, *pool_args(kw))
    dg, tg = torch.zeros_like(d), torch.zeros_like(t)
    ext.deform_psroi_pooling_cuda_backward(torch.from_numpy(g).to(dev), d, r, t, cnt, dg, tg, *pool_args(kw))
    torch.cuda.synchronize()
    return dict(out=out, count=cnt, data_grad=dg, trans_grad=None if kw['no_trans'] else tg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden", "dcn_reference_ext.npz"))
    args = ap.parse_args()
    conv_ext, pool_ext = load_reference_extension()
    if conv_ext is None:
        raise SystemExit("oracle/_ref holds no reference extension: run `bash oracle/build_ref_ext.sh` where /root/reference exists")
    if not torch.cuda.is_available():
        raise SystemExit("the reference extension is GPU-only (functions/deform_conv.py:40: `if not input.is_cuda: raise`)")
    sys.path.insert(0, os.path.join(HERE, ".."))
    from oracle.deform_pool import random_case
    blob = {}

    def put(prefix, d):
        for k, v in d.items():
            if v is not None:
                blob["%s/%s" % (prefix, k)] = (v.detach().float().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))

    for name, case in DCN2_CASES + [("kink", DCN2_CASES[0][1])]:
        x, om, w, b, gy, stride, pad, dil = dcn2_inputs(name, case, kink=(name == "kink"))
        put("dcn2/" + name, dict(x=x, offset_mask_map=om, weight=w, bias=b, grad_output=gy,
                                 geom=np.array([stride, pad, dil], np.int64)))
        put("dcn2/" + name, run_dcn2(conv_ext, x, om, w, b, gy, stride, pad, dil))
    for name, case in DCN1_CASES:
        x, off, w, gy, stride, pad, dil = dcn1_inputs(name, case)
        put("dcn1/" + name, dict(x=x, offset=off, weight=w, grad_output=gy, geom=np.array([stride, pad, dil], np.int64)))
        put("dcn1/" + name, run_dcn1(conv_ext, x, off, w, gy, stride, pad, dil))
    for i, c in enumerate(POOL_CASES):
        data, rois, trans, kw = random_case(10 + i, **c)
        n = rois.shape[0]
        g = np.random.default_rng(99).standard_normal((n, kw['output_dim'], kw['pooled_size'], kw['pooled_size'])).astype(np.float32)
        put("pool/%d" % i, dict(data=data, rois=rois, trans=trans, out_grad=g))
        put("pool/%d" % i, run_pool(pool_ext, data, rois, trans, kw, g))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **blob)
    print("wrote %s: %d arrays, %.1f KB" % (args.out, len(blob), os.path.getsize(args.out) / 1024.0))


if __name__ == "__main__":
    main()
