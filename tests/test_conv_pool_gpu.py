"""conv + bias + ReLU + max-pool as one forward launch (csrc/igemm_core.h: EpiPool, mr_conv2d_fwd_pool; round 6) against the two
launches it replaces (mr_conv2d_fwd + mr_maxpool_fwd) at the three pooled stages of the CRNN backbone (reference
backbones/crnn.py:14-33): pooled values and arg-max codes must be BIT-IDENTICAL (same bf16 rounding of the activation, same
first-maximum rule), hence so is everything the backward derives from them."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import megreader_amd as mr  # noqa: E402
from megreader_amd.nn import functional as F  # noqa: E402

DEV = "cuda"
# (name, N, Cin, H, W, Cout, pool kernel, stride, padding): conv1 (128-row tiles = one window row), conv3 (256-row tile = one image),
# conv5 (264 of 272 rows = two images; odd N: the last tile holds one image), and the same at the benchmarked batch
CASES = [("conv1", 8, 64, 16, 64, 128, (2, 2), (2, 2), (0, 0)), ("conv3", 4, 256, 8, 32, 256, (2, 2), (2, 1), (0, 1)),
         ("conv5", 5, 512, 4, 33, 512, (2, 2), (2, 1), (0, 1)), ("conv5_n256", 256, 512, 4, 33, 512, (2, 2), (2, 1), (0, 1)),
         ("conv5_n255", 255, 512, 4, 33, 512, (2, 2), (2, 1), (0, 1)), ("conv3_n256", 256, 256, 8, 32, 256, (2, 2), (2, 1), (0, 1)),
         ("conv1_n256", 256, 64, 16, 64, 128, (2, 2), (2, 2), (0, 0))]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fused_conv_relu_pool_equals_the_two_launches(case):
    name, N, Cin, H, W, Cout, pk, ps, pp = case
    mr.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(N + Cin + W)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    PH, PW = (H + 2 * pp[0] - pk[0]) // ps[0] + 1, (W + 2 * pp[1] - pk[1]) // ps[1] + 1
    gy = torch.randn(N, Cout, PH, PW, generator=g)

    def run(fused):
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).to(torch.bfloat16).requires_grad_(True)
        wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        if fused:
            ok = F.conv_relu_pool_eligible(xd, wd, (1, 1), (1, 1), (1, 1), pk, ps, pp)
            if not ok:
                # the fused launch rides on the 8-wave tile the plain forward would take: small problems take other tiles
                assert N < 256, name
                pytest.skip("%s at N = %d does not take an 8-wave tile" % (name, N))
            y = F.conv_relu_pool(xd, wd, bd, (1, 1), pk, ps, pp)
        else:
            z = F.conv2d(xd, wd, bd, (1, 1), (1, 1), (1, 1), relu=True, relu_grad_downstream=True)
            y = F.max_pool2d(z, pk, ps, pp, relu_input=True)
        y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last).to(y.dtype))
        torch.cuda.synchronize()
        return y.detach().clone(), xd.grad.clone(), wd.grad.clone(), bd.grad.clone()

    yf, dxf, dwf, dbf = run(True)
    yu, dxu, dwu, dbu = run(False)
    assert yf.shape == yu.shape == (N, Cout, PH, PW)
    assert torch.equal(yf, yu), float((yf.float() - yu.float()).abs().max())
    assert torch.equal(dxf, dxu)                      # same codes, same ReLU mask, the same dgrad kernel
    for a, r in ((dwf, dwu), (dbf, dbu)):             # split reductions with f32 atomics: equal up to their arrival order
        assert float((a - r).abs().max()) <= 2e-3 * float(r.abs().max()) + 1e-6
    # and against float64 on the bf16-rounded operands
    xr = x.bfloat16().double().requires_grad_(True)
    wr = w.bfloat16().double()
    zr = torch.relu(torch.nn.functional.conv2d(xr, wr, b.double(), 1, 1))
    yr = torch.nn.functional.max_pool2d(zr, pk, ps, pp)
    assert float((yf.double().cpu() - yr.detach()).abs().max()) <= 1.6e-2 * float(yr.abs().max())


def test_crnn_backbone_takes_the_fused_path_and_matches_the_unfused_one():
    from megreader_amd.backbones import crnn_backbone
    mr.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(3)
    net = crnn_backbone().to(DEV).train()
    x = torch.randn(6, 3, 32, 128, device=DEV)

    def run(env):
        old = os.environ.get("MEGREADER_CONV_POOL")
        os.environ["MEGREADER_CONV_POOL"] = env
        try:
            for p in net.parameters():
                p.grad = None
            y = net(x)
            y.float().square().mean().backward()
            torch.cuda.synchronize()
            return y.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}
        finally:
            if old is None:
                del os.environ["MEGREADER_CONV_POOL"]
            else:
                os.environ["MEGREADER_CONV_POOL"] = old

    yf, gf = run("1")
    yu, gu = run("0")
    assert torch.equal(yf, yu)
    for k in gf:
        # (the stem's weight gradient is a sum with heavy cancellation reduced with f32 atomics: 0.5 % of its largest element run to run)
        assert float((gf[k] - gu[k]).abs().max()) <= 2e-2 * float(gu[k].abs().max()) + 1e-7, k
