"""DCNv2 (assets.ops.dcn mirror) on HIP vs the float64 autograd oracle (oracle/dcn.py), including the reference's
flat re-interpretation of a larger offset map for stride-2 blocks."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.assets.ops.dcn import ModulatedDeformConv, modulated_deform_conv  # noqa: E402
from oracle.dcn import modulated_deform_conv2d  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _reset_dtype():
    yield
    mr.set_compute_dtype(torch.bfloat16)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


CASES = [  # N, C, Co, H, W, stride, pad, dil, offset-map size (None = output grid), offset scale
    (2, 16, 24, 7, 9, 1, 1, 1, None, 2.0),
    (2, 32, 32, 12, 10, 2, 1, 1, (12, 10), 1.5),   # stride-2 conv, stride-1 offset map (quirk Q10)
    (1, 16, 16, 9, 8, 1, 2, 2, None, 1.0),          # dilated
    (3, 8, 8, 5, 6, 1, 1, 1, None, 4.0),            # large offsets: many invalid samples / border touches
    # C a multiple of 64 (also what the opt-in round-2 backward kernels need: test_round2_backward_kernels)
    (2, 64, 32, 19, 21, 1, 1, 1, None, 1.0),        # tiles ragged in both directions, offsets inside the LDS patch
    (1, 128, 64, 20, 18, 2, 1, 1, (20, 18), 1.5),   # stride 2 (32-channel chunks), larger offset map
    (2, 64, 64, 11, 13, 1, 1, 1, None, 5.0),        # offsets beyond the patch margin: the direct-atomic path
    (1, 64, 64, 17, 16, 1, 2, 2, None, 1.0),        # dilated
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_forward_backward_vs_oracle(dtype, case):
    N, C, Co, H, W, stride, pad, dil, omap, oscale = case
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(C + H)
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    oh, ow = omap if omap else (Ho, Wo)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    # keep sample points away from exact integers (kink of the bilinear kernel; bf16 rounding could flip floor())
    off = (torch.randn(N, 18, oh, ow, generator=g) * oscale)
    off = torch.floor(off) + 0.25 + 0.5 * torch.rand(off.shape, generator=g)
    msk = torch.rand(N, 9, oh, ow, generator=g)
    w = (torch.randn(Co, C, 3, 3, generator=g) * 0.2)
    b = torch.randn(Co, generator=g)
    gy = torch.randn(N, Co, Ho, Wo, generator=g).to(dtype)

    xr = x.double().requires_grad_(True)
    offr = off.double().requires_grad_(True)
    mskr = msk.double().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = modulated_deform_conv2d(xr, offr, mskr, wr, br, stride, pad, dil)
    yr.backward(gy.double())

    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    offd = off.to(DEV).requires_grad_(True)
    mskd = msk.to(DEV).requires_grad_(True)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True)
    y = modulated_deform_conv(xd, offd, mskd, wd, bd, stride, pad, dil, 1, 1)
    assert y.shape == yr.shape
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert _rel(y, yr) < tol
    y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    gtol = 1e-4 if dtype == torch.float32 else 3e-2
    assert _rel(xd.grad, xr.grad) < gtol
    assert _rel(wd.grad, wr.grad) < gtol
    assert _rel(bd.grad, br.grad) < gtol
    assert _rel(offd.grad, offr.grad) < gtol
    assert _rel(mskd.grad, mskr.grad) < gtol
    if omap:  # entries outside the flat [18,Ho,Wo] window get exactly zero gradient, as in the reference
        flat = offd.grad.reshape(N, -1)
        assert float(flat[:, 18 * Ho * Wo:].abs().max()) == 0.0


# The 13 deformable layers of deformable_resnet50 at the DB detector's 640 x 640 input (backbones/resnet.py:113-181,
# 295-309; experiments/seg_detector/seg_detector_db.yaml:53): planes 128 / 256 / 512 at strides 8 / 16 / 32, the first block
# of every stage stride-2 with a stride-1 (input-sized) offset map read flat (quirk Q10).  (name, count, C, H, W, stride)
REAL_LAYERS = [
    ("layer2.0", 1, 128, 160, 160, 2), ("layer2.1-3", 3, 128, 80, 80, 1),
    ("layer3.0", 1, 256, 80, 80, 2), ("layer3.1-5", 5, 256, 40, 40, 1),
    ("layer4.0", 1, 512, 40, 40, 2), ("layer4.1-2", 2, 512, 20, 20, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layer", REAL_LAYERS, ids=[l[0] for l in REAL_LAYERS])
def test_real_layer_shapes_vs_oracle(dtype, layer):
    """DCNv2 forward + all five gradients at the real layer shapes of the published DB configuration (batch 2 per GPU),
    offsets ~ N(0, 1.5 px) kept off the integer grid, against the float64 autograd oracle.  These are the sizes at which
    the library picks its big-tile / split kernels and where the stride-2 blocks index a 4x larger offset map flat."""
    name, _count, C, H, W, stride = layer
    N, Co, pad, dil, oscale = 2, C, 1, 1, 1.5
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(C + H + stride)
    Ho = (H + 2 * pad - 3) // stride + 1
    Wo = (W + 2 * pad - 3) // stride + 1
    oh, ow = H, W                             # conv2_offset has stride 1 ALWAYS (backbones/resnet.py:136-142)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    off = torch.floor(torch.randn(N, 18, oh, ow, generator=g) * oscale) + 0.25 + 0.5 * torch.rand(N, 18, oh, ow, generator=g)
    msk = torch.sigmoid(torch.randn(N, 9, oh, ow, generator=g))
    w = torch.randn(Co, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    gy = torch.randn(N, Co, Ho, Wo, generator=g).to(dtype)
    xr = x.double().requires_grad_(True)
    offr = off.double().requires_grad_(True)
    mskr = msk.double().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    yr = modulated_deform_conv2d(xr, offr, mskr, wr, None, stride, pad, dil)
    yr.backward(gy.double())
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    offd = off.to(DEV).requires_grad_(True)
    mskd = msk.to(DEV).requires_grad_(True)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = modulated_deform_conv(xd, offd, mskd, wd, None, stride, pad, dil, 1, 1)
    assert y.shape == yr.shape
    errs = {"y": _rel(y, yr)}
    y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    errs.update(dx=_rel(xd.grad, xr.grad), dw=_rel(wd.grad, wr.grad), doff=_rel(offd.grad, offr.grad),
                dmask=_rel(mskd.grad, mskr.grad))
    print("DCN %s C=%d %dx%d s%d %s: " % (name, C, H, W, stride, str(dtype).split('.')[-1]) +
          ", ".join("%s %.2e" % kv for kv in errs.items()))
    tol, gtol = (2e-5, 1e-4) if dtype == torch.float32 else (2e-2, 3e-2)
    assert errs["y"] < tol
    for k in ("dx", "dw", "doff", "dmask"):
        assert errs[k] < gtol, (k, errs[k])
    if stride == 2:   # entries outside the flat [18,Ho,Wo] / [9,Ho,Wo] windows get exactly zero gradient
        assert float(offd.grad.reshape(N, -1)[:, 18 * Ho * Wo:].abs().max()) == 0.0
        assert float(mskd.grad.reshape(N, -1)[:, 9 * Ho * Wo:].abs().max()) == 0.0


def test_zero_offset_module_equals_conv():
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    m = ModulatedDeformConv(16, 24, 3, stride=1, padding=1, bias=True).to(DEV)
    x = torch.randn(2, 16, 6, 7, device=DEV)
    off = torch.zeros(2, 18, 6, 7, device=DEV)
    msk = torch.ones(2, 9, 6, 7, device=DEV)
    y = m(x, off, msk)
    ref = TF.conv2d(x.cpu().double(), m.weight.detach().cpu().double(), m.bias.detach().cpu().double(), 1, 1)
    assert _rel(y, ref) < 2e-5


def test_cpu_raises_like_reference():
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(torch.zeros(1, 8, 4, 4), torch.zeros(1, 18, 4, 4), torch.ones(1, 9, 4, 4),
                              torch.zeros(8, 8, 3, 3))


def _reference_function_calling_sequence(dcn_ext, input, offset, mask, weight, bias, stride, padding, dilation, gy):
    """The body of the reference's ModulatedDeformConvFunction.forward / .backward (assets/ops/dcn/functions/
    deform_conv.py:110-165) restated call for call against an extension module object: the caller allocates the output and
    the zeroed gradients and hands over scratch `ones` / `columns` tensors -- what running the reference's own Function
    file on top of `megreader_amd.assets.ops.dcn.deform_conv_cuda` does (the file itself is not on the GPU box)."""
    with_bias = bias is not None
    b = bias if with_bias else input.new_empty(1)
    kh, kw = weight.shape[2:4]
    n, co = input.size(0), weight.size(0)
    ho = (input.shape[2] + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    wo = (input.shape[3] + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    output = input.new_empty((n, co, ho, wo))
    bufs = [input.new_empty(0), input.new_empty(0)]
    dcn_ext.modulated_deform_conv_cuda_forward(input, weight, b, bufs[0], offset, mask, output, bufs[1], kh, kw, stride,
                                               stride, padding, padding, dilation, dilation, 1, 1, with_bias)
    grad_input, grad_offset, grad_mask = torch.zeros_like(input), torch.zeros_like(offset), torch.zeros_like(mask)
    grad_weight, grad_bias = torch.zeros_like(weight), torch.zeros_like(b)
    dcn_ext.modulated_deform_conv_cuda_backward(input, weight, b, bufs[0], offset, mask, bufs[1], grad_input, grad_weight,
                                                grad_bias, grad_offset, grad_mask, gy, kh, kw, stride, stride, padding,
                                                padding, dilation, dilation, 1, 1, with_bias)
    return output, grad_input, grad_offset, grad_mask, grad_weight, (grad_bias if with_bias else None)


@pytest.mark.parametrize("case", CASES[:3])
def test_extension_level_entry_points_vs_oracle(case):
    """`deform_conv_cuda.modulated_deform_conv_cuda_forward / _backward` (src/deform_conv_cuda.cpp:486-492,566-573): NCHW
    fp32 in, caller-allocated NCHW outputs / zeroed gradients written in place, the offset passed as the NON-contiguous
    channel slice of a 27-channel map exactly as backbones/resnet.py:162-164 produces it."""
    from megreader_amd.assets.ops.dcn import deform_conv_cuda
    N, C, Co, H, W, stride, pad, dil, omap, oscale = case
    mr.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(C + H + 1)
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    oh, ow = omap if omap else (Ho, Wo)
    x = torch.randn(N, C, H, W, generator=g)
    om = torch.randn(N, 27, oh, ow, generator=g) * oscale
    om[:, :18] = torch.floor(om[:, :18]) + 0.25 + 0.5 * torch.rand(N, 18, oh, ow, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    gy = torch.randn(N, Co, Ho, Wo, generator=g)
    omd = om.to(DEV)
    offset = omd[:, :18]                         # non-contiguous view across the batch
    mask = torch.sigmoid(omd[:, 18:27])
    assert N == 1 or not offset.is_contiguous()
    out = _reference_function_calling_sequence(deform_conv_cuda, x.to(DEV), offset, mask, w.to(DEV), b.to(DEV), stride,
                                               pad, dil, gy.to(DEV))
    xr = x.double().requires_grad_(True)
    offr = om[:, :18].double().contiguous().requires_grad_(True)
    mskr = torch.sigmoid(om[:, 18:27]).double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = modulated_deform_conv2d(xr, offr, mskr, wr, br, stride, pad, dil)
    yr.backward(gy.double())
    for got, want, name in zip(out, (yr, xr.grad, offr.grad, mskr.grad, wr.grad, br.grad),
                               ("output", "grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias")):
        assert got.shape == want.shape and got.dtype == torch.float32, name
        assert _rel(got, want) < (2e-5 if name == "output" else 1e-4), name
    with pytest.raises(RuntimeError):            # AT_CHECK(input.is_contiguous()), deform_conv_cuda.cpp:493
        deform_conv_cuda.modulated_deform_conv_cuda_forward(
            x.to(DEV).permute(0, 1, 3, 2), w.to(DEV), b.to(DEV), x.new_empty(0), offset, mask,
            torch.empty(N, Co, Ho, Wo, device=DEV), x.new_empty(0), 3, 3, stride, stride, pad, pad, dil, dil, 1, 1, True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dcn_v1_vs_oracle(dtype):
    """DeformConv (v1, reference functions/deform_conv.py:10-105): the v2 math with mask == 1."""
    from megreader_amd.assets.ops.dcn import DeformConv, deform_conv
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(5)
    N, C, Co, H, W = 2, 16, 24, 8, 9
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    off = torch.floor(torch.randn(N, 18, H, W, generator=g) * 1.5) + 0.25 + 0.5 * torch.rand(N, 18, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.2
    gy = torch.randn(N, Co, H, W, generator=g).to(dtype)
    xr = x.double().requires_grad_(True)
    offr = off.double().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    yr = modulated_deform_conv2d(xr, offr, torch.ones(N, 9, H, W, dtype=torch.float64), wr, None, 1, 1, 1)
    yr.backward(gy.double())
    xd = x.to(DEV).requires_grad_(True)
    offd = off.to(DEV).requires_grad_(True)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = deform_conv(xd, offd, wd, 1, 1, 1, 1, 1)
    tol, gtol = (2e-5, 1e-4) if dtype == torch.float32 else (2e-2, 3e-2)
    assert _rel(y, yr) < tol
    y.backward(gy.to(DEV))
    assert _rel(xd.grad, xr.grad) < gtol and _rel(offd.grad, offr.grad) < gtol and _rel(wd.grad, wr.grad) < gtol
    m = DeformConv(C, Co, 3, padding=1).to(DEV)
    assert m(xd.detach(), offd.detach()).shape == (N, Co, H, W)


def test_dcn_v1_extension_entry_points_vs_oracle():
    """`deform_conv_cuda.deform_conv_forward_cuda / _backward_input_cuda / _backward_parameters_cuda`
    (src/deform_conv_cuda.cpp:151-156,258-264,374-381) called exactly as the reference's DeformConvFunction does
    (assets/ops/dcn/functions/deform_conv.py:36-89: caller-allocated output, zeroed gradInput / gradOffset / gradWeight,
    width-first kW, kH, dW, dH, padW, padH, dilationW, dilationH argument order, scale = 1, im2col_step)."""
    from megreader_amd.assets.ops.dcn import deform_conv_cuda as ext
    mr.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(11)
    N, C, Co, H, W, stride, pad, dil = 2, 16, 24, 9, 8, 1, 1, 1
    x = torch.randn(N, C, H, W, generator=g)
    off = torch.floor(torch.randn(N, 18, H, W, generator=g) * 1.5) + 0.25 + 0.5 * torch.rand(N, 18, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.2
    gy = torch.randn(N, Co, H, W, generator=g)
    xr, offr, wr = x.double().requires_grad_(True), off.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = modulated_deform_conv2d(xr, offr, torch.ones(N, 9, H, W, dtype=torch.float64), wr, None, stride, pad, dil)
    yr.backward(gy.double())
    input, offset, weight, grad_output = x.to(DEV), off.to(DEV), w.to(DEV), gy.to(DEV)
    output = input.new_empty((N, Co, H, W))
    bufs = [input.new_empty(0), input.new_empty(0)]
    step = N
    assert ext.deform_conv_forward_cuda(input, weight, offset, output, bufs[0], bufs[1], weight.size(3), weight.size(2), stride,
                                        stride, pad, pad, dil, dil, 1, 1, step) == 1
    grad_input, grad_offset = torch.zeros_like(input), torch.zeros_like(offset)
    ext.deform_conv_backward_input_cuda(input, offset, grad_output, grad_input, grad_offset, weight, bufs[0], weight.size(3),
                                        weight.size(2), stride, stride, pad, pad, dil, dil, 1, 1, step)
    grad_weight = torch.zeros_like(weight)
    ext.deform_conv_backward_parameters_cuda(input, offset, grad_output, grad_weight, bufs[0], bufs[1], weight.size(3),
                                             weight.size(2), stride, stride, pad, pad, dil, dil, 1, 1, 1, step)
    ext.deform_conv_backward_parameters_cuda(input, offset, grad_output, grad_weight, bufs[0], bufs[1], weight.size(3),
                                             weight.size(2), stride, stride, pad, pad, dil, dil, 1, 1, 0.5, step)   # accumulates
    assert _rel(output, yr) < 2e-5
    assert _rel(grad_input, xr.grad) < 1e-4 and _rel(grad_offset, offr.grad) < 1e-4
    assert _rel(grad_weight, 1.5 * wr.grad) < 1e-4


@pytest.mark.parametrize("case", CASES[4:])
def test_round2_backward_kernels(case):
    """The opt-in round-2 backward kernels (8-lane vectorised coordinate gradient, LDS-tiled col2im; measured slower, kept
    behind mr_set_dcn_v1_bwd(0)) produce the same gradients as the default kernels."""
    from megreader_amd._lib import load
    N, C, Co, H, W, stride, pad, dil, omap, oscale = case
    mr.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(C + H)
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    oh, ow = omap if omap else (Ho, Wo)
    x = torch.randn(N, C, H, W, generator=g).bfloat16()
    off = torch.floor(torch.randn(N, 18, oh, ow, generator=g) * oscale) + 0.25 + 0.5 * torch.rand(N, 18, oh, ow, generator=g)
    msk = torch.rand(N, 9, oh, ow, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.2
    gy = torch.randn(N, Co, Ho, Wo, generator=g).bfloat16()

    def run():
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        offd, mskd = off.to(DEV).requires_grad_(True), msk.to(DEV).requires_grad_(True)
        wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        modulated_deform_conv(xd, offd, mskd, wd, None, stride, pad, dil, 1, 1).backward(
            gy.to(DEV).contiguous(memory_format=torch.channels_last))
        return xd.grad, offd.grad, mskd.grad

    fused = load().mr_set_dcn_fused(0)      # these channel counts would otherwise take the fused kernels
    try:
        ref = run()
        old = load().mr_set_dcn_v1_bwd(0)
        try:
            got = run()
        finally:
            load().mr_set_dcn_v1_bwd(old)
    finally:
        load().mr_set_dcn_fused(fused)
    for a, b in zip(got, ref):
        assert _rel(a, b) < 2e-3      # f32 atomics in a different order


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES[4:] + [(2, 256, 128, 14, 15, 1, 1, 1, None, 0.0), (1, 128, 192, 23, 9, 2, 1, 1, (23, 9), 8.0)])
def test_fused_path_equals_general_path(dtype, case):
    """The fused kernels (csrc/dcn_fused.hip: sample -> LDS -> MFMA forward, register-resident gcol for the offset / mask
    gradients, CSR gather-GEMM input gradient, sampled TN wgrad) against the general im2col / col2im kernels on the same
    inputs: Co != C, Co = 192 (64-column tiles), zero offsets (integer sample points: zero-weight corners produce no CSR
    entry), offsets far outside the image (invalid samples, empty CSR rows), stride 2 with the larger offset map."""
    from megreader_amd._lib import load
    N, C, Co, H, W, stride, pad, dil, omap, oscale = case
    mr.set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(C + H + 3)
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    oh, ow = omap if omap else (Ho, Wo)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    off = torch.randn(N, 18, oh, ow, generator=g) * oscale
    msk = torch.rand(N, 9, oh, ow, generator=g)
    msk[:, 4, ::3] = 0.0                              # exact zero masks: dropped from the CSR, still get a mask gradient
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    gy = torch.randn(N, Co, Ho, Wo, generator=g).to(dtype)

    def run():
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        offd, mskd = off.to(DEV).requires_grad_(True), msk.to(DEV).requires_grad_(True)
        wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bd = b.to(DEV).requires_grad_(True)
        y = modulated_deform_conv(xd, offd, mskd, wd, bd, stride, pad, dil, 1, 1)
        y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
        return y.detach(), xd.grad, offd.grad, mskd.grad, wd.grad, bd.grad

    lib = load()
    fused = C % 64 == 0 and Co % 64 == 0          # (case 4 has Co = 32: it stays on the general path, both runs equal)
    col_bytes = N * Ho * Wo * 9 * C * (2 if dtype == torch.bfloat16 else 4)       # what the general path's forward needs
    ws = lib.mr_dcn2_ws_bytes(1 if dtype == torch.bfloat16 else 0, N, H, W, C, Co, 3, 3, Ho, Wo, 0)
    # fused: nothing, or the nine tap groups' f32 slabs of a small layer (which may happen to equal col_bytes: C = 2 Co in bf16);
    # round 6: the bf16 materialised path (C in {64, 128, 256, 512}) writes the column matrix in the forward and keeps it
    col_fwd = fused and bool(lib.mr_dcn2_col_saved(1 if dtype == torch.bfloat16 else 0, H, W, C, Co, 3, 3))
    assert (ws == col_bytes if col_fwd else ws in (0, 9 * N * Ho * Wo * Co * 4)) if fused else ws == col_bytes
    got = run()
    old = lib.mr_set_dcn_fused(0)
    try:
        ref = run()
    finally:
        lib.mr_set_dcn_fused(old)
    names = ("y", "dx", "doffset", "dmask", "dw", "dbias")
    # same products, different summation order; in bf16 the general path also rounds gcol to bf16 before using it
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for a, r, nm in zip(got, ref, names):
        assert a.shape == r.shape, nm
        e = _rel(a, r)
        assert e < tol, (nm, e)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("stride", [1, 2])
def test_packed_operand_equals_unfused_expression(dtype, stride):
    """The deformable ResNet blocks call conv2(x, offset_mask[:, :18], offset_mask[:, -9:].sigmoid()) (reference
    backbones/resnet.py:125-142).  `forward_packed` is that expression as one autograd node (mr_dcn_unpack /
    mr_dcn_pack_grad + the zero-padded gradient hand-over to the offset conv's backward): outputs and every gradient --
    input, offset-conv weight and bias, DCN weight -- must agree with the unfused graph.  stride 2 = the reference's quirk
    Q10 (offset conv at stride 1 feeding a stride-2 DCN: a larger map indexed flat)."""
    from megreader_amd.nn import Conv2d
    mr.set_compute_dtype(dtype)
    try:
        torch.manual_seed(5)
        N, C, H, W = 2, 64, 12, 20
        conv_off = Conv2d(C, 27, kernel_size=3, padding=1).to(DEV)
        with torch.no_grad():
            conv_off.weight.mul_(3.0)          # offsets of a few pixels, mask logits away from 0
        dcn = ModulatedDeformConv(C, 64, 3, stride=stride, padding=1, bias=False).to(DEV)
        x0 = torch.randn(N, C, H, W, device=DEV)
        gy = None
        res = []
        for packed in (False, True):
            for p in list(conv_off.parameters()) + list(dcn.parameters()):
                p.grad = None
            x = x0.clone().requires_grad_(True)
            om = conv_off(x)
            if packed:
                y = dcn.forward_packed(x, om)
            else:
                y = dcn(x, om[:, :18, :, :], om[:, -9:, :, :].sigmoid())
            if gy is None:
                gy = torch.randn_like(y)
            y.backward(gy)
            res.append([y.detach().float(), x.grad.float(), conv_off.weight.grad.float().clone(),
                        conv_off.bias.grad.float().clone(), dcn.weight.grad.float().clone()])
        # f32: same kernels on the same f32 maps -> round-off of the sigmoid only; bf16: the unfused graph rounds the mask
        # (and its gradient) to bf16 between the launches, the fused one does not
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        for name, a, b in zip(("y", "dx", "d offset-conv weight", "d offset-conv bias", "d dcn weight"), res[0], res[1]):
            err = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-12)
            assert err < tol, (name, err)
    finally:
        mr.set_compute_dtype(torch.bfloat16)
