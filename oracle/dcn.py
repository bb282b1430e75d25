"""Oracle restatement (torch float64, differentiable) of the reference's CUDA-only modulated deformable conv v2.

Follows assets/ops/dcn/src/deform_conv_cuda_kernel.cu:
    :466-496  dmcn_im2col_bilinear (corners outside the image contribute 0)
    :569-632  modulated_deformable_im2col_gpu_kernel: sample at (h_in + i*dil + dh, w_in + j*dil + dw), valid iff
              h > -1 and w > -1 and h < H and w < W; offsets / mask are read as FLAT [.., Ho, Wo] arrays from the base
              pointer of offset[b] / mask[b] (:599-612) -- matters when the offset map is larger than the output
              grid (reference quirk Q10: stride-2 DCN conv with a stride-1 offset conv, backbones/resnet.py:136-142)
and the host GEMM of assets/ops/dcn/src/deform_conv_cuda.cpp:534-563.  The backward of the op is autograd through
this forward: SURVEY.md Appendix A.4 verified that the reference's explicit backward kernels (:634-766) equal it.

PARITY STATUS: pinned (round 4).  The reference has no CPU implementation and no tests of this op, but its OWN extension
(assets/ops/dcn/src/deform_conv_cuda.cpp + deform_conv_cuda_kernel.cu) compiles for gfx950 from the sources where they lie
(oracle/build_ref_ext.sh -> oracle/_ref/) and runs on the MI355X: oracle/gen_golden_dcn.py recorded its outputs for seven DCNv2
and three DCN v1 cases (flat stride-1 offset map under a stride-2 layer, the non-contiguous 27-channel slice, dilation, offsets
far outside the image, integer / half-integer offsets) in tests/golden/dcn_reference_ext.npz, and
tests/test_oracle_dcn_pinned_cpu.py holds this restatement to them (2e-5 forward, 1e-4 gradients: the reference is float32).
Older anchors (tests/test_oracle_dcn.py): zero offsets and unit mask == F.conv2d; integer offsets == shifted conv taps;
gradcheck of the restatement.
"""
import torch


def _flat_view(t, rows, Ho, Wo):
    """t [N, ch, Ho', Wo'] -> flat-reinterpreted [N, rows, Ho, Wo] exactly as the kernel indexes it."""
    N = t.shape[0]
    flat = t.reshape(N, -1)
    need = rows * Ho * Wo
    if flat.shape[1] < need:
        raise ValueError("offset/mask buffer smaller than the output grid needs")
    return flat[:, :need].reshape(N, rows, Ho, Wo)


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1):
    """x [N,C,H,W]; offset [N,2*kh*kw,Ho',Wo'] as (dh, dw) pairs per tap; mask [N,kh*kw,Ho',Wo']; weight [Co,C,kh,kw].
    groups = deformable_groups = 1 (the only configuration the reference models use)."""
    N, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    off = _flat_view(offset, 2 * kh * kw, Ho, Wo)
    msk = _flat_view(mask, kh * kw, Ho, Wo)
    ho = torch.arange(Ho, dtype=x.dtype).view(1, Ho, 1)
    wo = torch.arange(Wo, dtype=x.dtype).view(1, 1, Wo)
    xf = x.reshape(N, C, H * W)
    cols = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            ph = ho * stride - padding + i * dilation + off[:, 2 * k]
            pw = wo * stride - padding + j * dilation + off[:, 2 * k + 1]
            valid = (ph > -1) & (pw > -1) & (ph < H) & (pw < W)
            hl = torch.floor(ph)
            wl = torch.floor(pw)
            lh, lw = ph - hl, pw - wl
            hl, wl = hl.long(), wl.long()
            hh, wh = hl + 1, wl + 1

            def corner(hc, wc, ok):
                idx = (hc.clamp(0, H - 1) * W + wc.clamp(0, W - 1)).view(N, 1, Ho * Wo).expand(N, C, Ho * Wo)
                v = torch.gather(xf, 2, idx).view(N, C, Ho, Wo)
                return v * ok.view(N, 1, Ho, Wo).to(x.dtype)

            v1 = corner(hl, wl, (hl >= 0) & (wl >= 0))
            v2 = corner(hl, wh, (hl >= 0) & (wh <= W - 1))
            v3 = corner(hh, wl, (hh <= H - 1) & (wl >= 0))
            v4 = corner(hh, wh, (hh <= H - 1) & (wh <= W - 1))
            w1 = ((1 - lh) * (1 - lw)).unsqueeze(1)
            w2 = ((1 - lh) * lw).unsqueeze(1)
            w3 = (lh * (1 - lw)).unsqueeze(1)
            w4 = (lh * lw).unsqueeze(1)
            bil = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
            cols.append(bil * (valid.to(x.dtype) * msk[:, k]).unsqueeze(1))
    col = torch.stack(cols, dim=2)  # [N, C, kh*kw, Ho, Wo]
    y = torch.einsum('nckhw,ock->nohw', col, weight.reshape(Co, C, kh * kw))
    if bias is not None:
        y = y + bias.view(1, Co, 1, 1)
    return y


class OracleModulatedDeformConv(torch.nn.Module):
    """CPU stand-in with the constructor / parameters / init of assets/ops/dcn/modules/deform_conv.py:84-128, used
    (a) as the `assets.ops.dcn.ModulatedDeformConv` shim when the unmodified reference ResNet is executed on CPU for
    golden vectors and (b) inside the oracle's own deformable ResNet."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        import math
        assert groups == 1 and deformable_groups == 1
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, ks
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.weight = torch.nn.Parameter(torch.Tensor(out_channels, in_channels, *ks))
        if bias:
            self.bias = torch.nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        n = in_channels * ks[0] * ks[1]
        self.weight.data.uniform_(-1. / math.sqrt(n), 1. / math.sqrt(n))
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv2d(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                       self.dilation)


def perturb_offset_convs(model, seed=99, kink_safe=False):
    """The reference zero-initialises every conv2_offset (backbones/resnet.py:222-226), which would leave the
    deformable sampling trivial (offsets 0, mask 0.5).  Tests give them small seeded random values instead --
    applied identically to the reference model, the oracle and the HIP model (parameter order = named_parameters).

    kink_safe=True: the bilinear kernel has a kink at every integer sampling coordinate -- the forward value is
    continuous there, d/d(offset) and the set of pixels that receive gradient are not.  With feature-dependent offsets
    spread over the real line (the default: bias ~ N(0, 0.5), weights ~ N(0, 0.02)) some of the ~1e5 coordinates of a
    test batch always land within 1e-5 of an integer, and two arithmetically equivalent runs (another summation order of
    the BatchNorm statistics in front) then take floor() to different sides: ONE such flip moved a gradient by 0.13 of
    its maximum in round 3 (tools/diag_fast_paths.py, profiles/r04_diag_fast_paths_before.txt).  Tests that compare
    GRADIENTS of two runs use this mode: offset biases k + 0.5 +- 0.05 (k in {-1, 0, 1}) and weights ~ N(0, 1e-3), so
    every coordinate stays >= 0.2 away from an integer (the feature-dependent part has std ~0.05: BN + ReLU inputs, 9 * C
    taps); `min_kink_distance` checks that precondition on the actual model.  Mask channels as in the default mode."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'conv2_offset' in name:
                if name.endswith('weight'):
                    p.copy_(torch.randn(p.shape, generator=g) * (1e-3 if kink_safe else 0.02))
                elif kink_safe:
                    n_off = p.numel() * 2 // 3          # 18 offset channels, then 9 mask channels
                    b = torch.randn(p.shape, generator=g) * 0.5
                    k = torch.randint(-1, 2, (n_off,), generator=g).to(b.dtype)
                    b[:n_off] = k + 0.5 + (torch.rand((n_off,), generator=g) - 0.5) * 0.1
                    p.copy_(b)
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.5)


def min_kink_distance(model, x, n_offset_channels=18):
    """Smallest distance of any sampling coordinate to an integer row / column over the deformable blocks of `model` for
    the input `x` (integer base coordinates: only the fractional part of the offsets matters)."""
    dists, hooks = [], []
    for m in model.modules():
        conv = getattr(m, 'conv2_offset', None)
        if conv is not None:
            def hook(_m, _inp, out):
                off = out.detach().float()[:, :n_offset_channels]
                fr = off - off.floor()
                dists.append(float(torch.minimum(fr, 1 - fr).min()))
            hooks.append(conv.register_forward_hook(hook))
    # a TRAINING-mode forward (batch statistics, like the runs under test); running statistics restored afterwards.  Call it
    # on a copy when the model's first-forward behaviour matters (the HIP modules configure conv -> bn fusions on first use).
    was_training = model.training
    saved = {k: v.detach().clone() for k, v in model.named_buffers()}
    model.train()
    try:
        with torch.no_grad():
            model(x)
    finally:
        model.train(was_training)
        with torch.no_grad():
            for k, v in model.named_buffers():
                v.copy_(saved[k])
        for h in hooks:
            h.remove()
    return min(dists) if dists else float('inf')
