"""Where does the HIP float32 path lose accuracy relative to PyTorch's CPU float32 arithmetic?  (parity item 1: the
Res50-PPM forward error is a constant ~1.5x the CPU f32 oracle's at every stage.)  Single layers, float32, against float64:
convolutions with the reduction lengths of the ResNet bottlenecks, training-mode BatchNorm on conv-like and on large-mean inputs,
and conv -> BN chains with the statistics taken in the conv epilogue (bn_stats=True) or by the separate pass.
usage (GPU box): python tools/diag_f32_error.py"""
import os
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr
from megreader_amd.nn import functional as F

DEV = torch.device("cuda", 0)


def err(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


def conv_case(name, N, C, H, W, K, k, pad, g):
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, k, k, generator=g) / (C * k * k) ** 0.5
    y64 = TF.conv2d(x.double(), w.double(), None, 1, pad)
    y32 = TF.conv2d(x, w, None, 1, pad)
    yh = F.conv2d(x.to(DEV).contiguous(memory_format=torch.channels_last), w.to(DEV), None, (1, 1), (pad, pad))
    print("%-34s K=%5d   HIP f32 %.3e   CPU f32 %.3e   ratio %.2f" % (name, C * k * k, err(yh, y64), err(y32, y64),
                                                                       err(yh, y64) / err(y32, y64)))


def bn_case(name, x, g):
    C = x.shape[1]
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    y64 = TF.batch_norm(x.double(), None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)
    y32 = TF.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5)
    yh = F.batch_norm(x.to(DEV).contiguous(memory_format=torch.channels_last), gamma.to(DEV), beta.to(DEV), None, None, True,
                      0.1, 1e-5)
    print("%-34s           HIP f32 %.3e   CPU f32 %.3e   ratio %.2f" % (name, err(yh, y64), err(y32, y64),
                                                                       err(yh, y64) / err(y32, y64)))


def chain_case(name, N, C, H, W, K, k, pad, depth, g, epilogue):
    x = torch.randn(N, C, H, W, generator=g)
    ws = [torch.randn(K if i else K, C if i == 0 else K, k, k, generator=g) / ((C if i == 0 else K) * k * k) ** 0.5
          for i in range(depth)]
    gs = [torch.rand(K, generator=g) + 0.5 for _ in range(depth)]
    bs = [torch.randn(K, generator=g) * 0.1 for _ in range(depth)]

    def run(x, conv, bn, cast):
        for w, ga, be in zip(ws, gs, bs):
            x = bn(conv(x, cast(w)), cast(ga), cast(be))
        return x

    y64 = run(x.double(), lambda a, w: TF.conv2d(a, w, None, 1, pad),
              lambda a, ga, be: torch.relu(TF.batch_norm(a, None, None, ga, be, True, 0.1, 1e-5)), lambda t: t.double())
    y32 = run(x, lambda a, w: TF.conv2d(a, w, None, 1, pad),
              lambda a, ga, be: torch.relu(TF.batch_norm(a, None, None, ga, be, True, 0.1, 1e-5)), lambda t: t)
    yh = run(x.to(DEV).contiguous(memory_format=torch.channels_last),
             lambda a, w: F.conv2d(a, w, None, (1, 1), (pad, pad), bn_stats=epilogue),
             lambda a, ga, be: F.batch_norm(a, ga, be, None, None, True, 0.1, 1e-5, relu=True), lambda t: t.to(DEV))
    print("%-34s depth %d   HIP f32 %.3e   CPU f32 %.3e   ratio %.2f" % (name, depth, err(yh, y64), err(y32, y64),
                                                                        err(yh, y64) / err(y32, y64)))


def main():
    mr.set_compute_dtype(torch.float32)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    g = torch.Generator().manual_seed(0)
    print("max |d| / max |x64|, float32 single layers (the numbers of one run; ratio = HIP / CPU)")
    conv_case("conv 1x1  256 -> 64", 8, 256, 16, 16, 64, 1, 0, g)
    conv_case("conv 1x1 1024 -> 256", 8, 1024, 8, 16, 256, 1, 0, g)
    conv_case("conv 3x3  256 -> 256", 8, 256, 8, 16, 256, 3, 1, g)
    conv_case("conv 3x3  512 -> 512", 4, 512, 4, 16, 512, 3, 1, g)
    conv_case("conv 3x3 4096 -> 512 (PPM)", 2, 4096, 4, 16, 512, 3, 1, g)
    bn_case("bn, x ~ N(0, 1)", torch.randn(32, 256, 4, 16, generator=g), g)
    bn_case("bn, x ~ N(5, 1)", torch.randn(32, 256, 4, 16, generator=g) + 5.0, g)
    bn_case("bn, x ~ relu(N(0,1)) * 3", torch.relu(torch.randn(32, 256, 4, 16, generator=g)) * 3.0, g)
    for epi in (True, False):
        print("conv -> bn -> relu chains, statistics %s" % ("in the conv epilogue" if epi else "by the separate pass"))
        chain_case("3x3 256 -> 256 x depth", 16, 256, 4, 16, 256, 3, 1, 1, torch.Generator().manual_seed(1), epi)
        chain_case("3x3 256 -> 256 x depth", 16, 256, 4, 16, 256, 3, 1, 4, torch.Generator().manual_seed(1), epi)
        chain_case("3x3 256 -> 256 x depth", 16, 256, 4, 16, 256, 3, 1, 12, torch.Generator().manual_seed(1), epi)
        chain_case("1x1 512 -> 512 x depth", 16, 512, 4, 16, 512, 1, 0, 12, torch.Generator().manual_seed(2), epi)


if __name__ == "__main__":
    main()
