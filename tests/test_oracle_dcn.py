"""CPU anchors of the DCNv2 oracle (no reference CPU path exists for this CUDA-only op)."""
import torch
import torch.nn.functional as TF

from oracle.dcn import modulated_deform_conv2d


def test_zero_offset_unit_mask_is_conv2d():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 7, 9, generator=g, dtype=torch.float64)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(4, generator=g, dtype=torch.float64)
    for stride, pad, dil in ((1, 1, 1), (2, 1, 1), (1, 2, 2)):
        Ho = (7 + 2 * pad - (dil * 2 + 1)) // stride + 1
        Wo = (9 + 2 * pad - (dil * 2 + 1)) // stride + 1
        off = torch.zeros(2, 18, Ho, Wo, dtype=torch.float64)
        msk = torch.ones(2, 9, Ho, Wo, dtype=torch.float64)
        y = modulated_deform_conv2d(x, off, msk, w, b, stride, pad, dil)
        assert torch.allclose(y, TF.conv2d(x, w, b, stride, pad, dil), atol=1e-12)


def test_integer_offsets_shift_taps():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, 6, 6, generator=g, dtype=torch.float64)
    w = torch.randn(2, 3, 3, 3, generator=g, dtype=torch.float64)
    off = torch.zeros(1, 18, 6, 6, dtype=torch.float64)
    off[:, 0::2] = 1.0   # every tap samples one row lower == conv on an up-shifted image
    msk = torch.ones(1, 9, 6, 6, dtype=torch.float64)
    y = modulated_deform_conv2d(x, off, msk, w, None, 1, 1, 1)
    xs = torch.zeros_like(x)
    xs[:, :, :-1] = x[:, :, 1:]
    # (row 0 of the output also sees x[0] through the tap that ordinary zero padding would blank: compare rows >= 1)
    assert torch.allclose(y[:, :, 1:], TF.conv2d(xs, w, None, 1, 1)[:, :, 1:], atol=1e-12)


def test_flat_reinterpretation_of_larger_offset_map():
    """stride-2 conv with a stride-1 offset map (quirk Q10): only the first 18*Ho*Wo values are used, flat."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 2, 8, 8, generator=g, dtype=torch.float64)
    w = torch.randn(3, 2, 3, 3, generator=g, dtype=torch.float64)
    off_big = torch.randn(1, 18, 8, 8, generator=g, dtype=torch.float64)
    msk_big = torch.rand(1, 9, 8, 8, generator=g, dtype=torch.float64)
    y = modulated_deform_conv2d(x, off_big, msk_big, w, None, 2, 1, 1)
    off_small = off_big.reshape(1, -1)[:, :18 * 16].reshape(1, 18, 4, 4)
    msk_small = msk_big.reshape(1, -1)[:, :9 * 16].reshape(1, 9, 4, 4)
    assert torch.equal(y, modulated_deform_conv2d(x, off_small, msk_small, w, None, 2, 1, 1))


def test_gradcheck():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 2, 5, 5, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(2, 2, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    # keep sampling points away from integer coordinates (the bilinear kernel is not differentiable there)
    off = (torch.rand(1, 18, 5, 5, generator=g, dtype=torch.float64) * 0.6 + 0.2).requires_grad_(True)
    msk = torch.rand(1, 9, 5, 5, generator=g, dtype=torch.float64).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, o, m, ww: modulated_deform_conv2d(a, o, m, ww, None, 1, 1, 1),
                                    (x, off, msk, w), eps=1e-6, atol=1e-6)
