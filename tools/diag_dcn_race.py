"""Run-to-run stability of the fused DCNv2 backward (csrc/dcn_fused.hip): the same mr_dcn2_bwd inputs `reps` times; every
output (dx, doffset, dmask, dw) of every run against the first run and against the general kernels (mr_set_dcn_fused(0)).
Found with it (round 4): see DESIGN.md section 5 / profiles/r04_diag_dcn_race_*.txt.

    python tools/diag_dcn_race.py [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402
from megreader_amd._lib import load  # noqa: E402
from megreader_amd.assets.ops.dcn.deform_conv import modulated_deform_conv  # noqa: E402

DEV = "cuda"


def one(x, off, msk, w, g, stride):
    xs = x.clone().requires_grad_(True)
    o = off.clone().requires_grad_(True)
    m = msk.clone().requires_grad_(True)
    ws = w.clone().requires_grad_(True)
    y = modulated_deform_conv(xs, o, m, ws, None, stride, 1, 1, 1, 1)
    y.backward(g)
    return [t.detach().float().clone() for t in (y, xs.grad, o.grad, m.grad, ws.grad)]


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lib = load()
    names = ["y", "dx", "doffset", "dmask", "dw"]
    for dtype in (torch.float32, torch.bfloat16):
        mr.set_compute_dtype(dtype)
        for (N, C, H, W, stride) in ((2, 128, 12, 16, 1), (2, 128, 24, 32, 2), (2, 256, 40, 40, 1), (16, 128, 80, 80, 1)):
            gen = torch.Generator(device="cpu").manual_seed(5)
            Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
            x = torch.randn(N, C, H, W, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
            k = torch.randint(-1, 2, (N, 18, H, W), generator=gen).float()
            off = (k + 0.5 + (torch.rand(N, 18, H, W, generator=gen) - 0.5) * 0.1).to(DEV)   # stride-1 map (quirk Q10)
            msk = torch.sigmoid(torch.randn(N, 9, H, W, generator=gen)).to(DEV)
            w = (torch.randn(C, C, 3, 3, generator=gen) * 0.03).to(DEV).contiguous(memory_format=torch.channels_last)
            g = torch.randn(N, C, Ho, Wo, generator=gen).to(DEV).contiguous(memory_format=torch.channels_last)
            old = lib.mr_set_dcn_fused(0)
            ref = one(x, off, msk, w, g, stride)
            lib.mr_set_dcn_fused(1)
            first = one(x, off, msk, w, g, stride)
            worst = [0.0] * 5
            bad_runs = [0] * 5
            for r in range(reps):
                cur = one(x, off, msk, w, g, stride)
                for i in range(5):
                    e = rel(cur[i], first[i])
                    worst[i] = max(worst[i], e)
                    bad_runs[i] += int(e > (1e-3 if dtype == torch.float32 else 2e-2))
            lib.mr_set_dcn_fused(old)
            print("%s N=%d C=%d %dx%d stride %d:" % (str(dtype).replace("torch.", ""), N, C, H, W, stride))
            for i, nm in enumerate(names):
                print("   %-8s fused vs general %.2e | worst run-to-run %.2e | deviating runs %d / %d" %
                      (nm, rel(first[i], ref[i]), worst[i], bad_runs[i], reps))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
