"""Diagnosis of tests/test_deformable_resnet_gpu.py::test_fast_paths_equal_plain_autograd (red on the round-3 driver box,
green on the builder's): repeats the scenario with the forward difference printed, with each fast path switched off alone,
and with the caching allocator's free blocks poisoned (0xFF = NaN) so that a read of uninitialised memory shows.

    python tools/diag_fast_paths.py [reps] [--kink-safe]

Result (profiles/r04_diag_fast_paths_before.txt): a bilinear-kink flip, not a race -- the same 0.133 on `2.conv3.weight` in
~10 % of the runs whatever is switched off, in 10 of 10 runs on the general DCN kernels; forward difference always ~3e-6.
With --kink-safe (oracle/dcn.py:perturb_offset_convs(kink_safe=True), what the tests now use for gradient comparisons) no
run deviates (profiles/r04_diag_fast_paths_after.txt)."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402
from megreader_amd.backbones import deformable_resnet50  # noqa: E402
from megreader_amd.nn import modules as mrm  # noqa: E402
from megreader_amd.optim import FusedSGD  # noqa: E402
from megreader_amd._lib import load  # noqa: E402
from oracle.dcn import perturb_offset_convs  # noqa: E402

DEV = "cuda"


def poison():
    """Fill the allocator's free lists with NaN bit patterns (small pool and large pool)."""
    small = [torch.empty(128 * 1024, dtype=torch.int32, device=DEV).fill_(-1) for _ in range(2048)]   # 1 GB of 512 KB
    big = [torch.empty(64 * 1024 * 1024, dtype=torch.int32, device=DEV).fill_(-1) for _ in range(16)]  # 4 GB of 256 MB
    torch.cuda.synchronize()
    del small, big


def kink_distance(model, x):
    """Smallest distance of any sampling coordinate to an integer row / column over the four blocks."""
    d = []
    hooks = []
    for blk in model:
        def hook(m, inp, out, d=d):
            off = out.detach().float()[:, :18]
            fr = off - off.floor()
            d.append(float(torch.minimum(fr, 1 - fr).min()))
        hooks.append(blk.conv2_offset.register_forward_hook(hook))
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    return d


KINK_SAFE = '--kink-safe' in sys.argv


def scenario(use_opt=True, steps=2):
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(1)
    full = deformable_resnet50(pretrained=False)
    perturb_offset_convs(full, kink_safe=KINK_SAFE)
    plain = full.layer2.to(DEV).train()
    fast = copy.deepcopy(plain)
    x = torch.randn(2, 256, 24, 32, device=DEV)
    yp = plain(x)
    (yp.float() ** 2).mean().backward()
    ref = {k: p.grad.detach().clone() for k, p in plain.named_parameters() if p.grad is not None}
    opt = FusedSGD(fast.parameters(), lr=0.0, momentum=0.0) if use_opt else None
    for it in range(steps):
        if opt is not None:
            opt.zero_grad()
        else:
            for p in fast.parameters():
                p.grad = None
        yf = fast(x)
        (yf.float() ** 2).mean().backward()
    fwd = float((yf.float() - yp.float()).abs().max() / yp.float().abs().max())
    worst = (None, 0.0)
    for k, p in fast.named_parameters():
        scale = float(ref[k].abs().max())
        if scale < 1e-9:
            continue
        err = float((p.grad - ref[k]).abs().max()) / scale
        if not err <= worst[1]:
            worst = (k, err)
    return fwd, worst, float(x.double().sum())


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
    lib = load()
    print("device", torch.cuda.get_device_name(0))
    cases = [
        ("baseline", dict(), {}),
        ("poisoned", dict(), {"poison": True}),
        ("poisoned, epilogue off", dict(), {"poison": True, "epi": False}),
        ("poisoned, no optimizer", dict(use_opt=False), {"poison": True}),
        ("poisoned, dcn general", dict(), {"poison": True, "fused": 0}),
        ("poisoned, 1 step only (no epilogue yet)", dict(steps=1), {"poison": True}),
    ]
    for name, kw, env in cases:
        mrm.BN_EPILOGUE = env.get("epi", True)
        old = lib.mr_set_dcn_fused(env.get("fused", 1))
        rows = []
        for r in range(reps):
            if env.get("poison"):
                poison()
            rows.append(scenario(**kw))
        lib.mr_set_dcn_fused(old)
        mrm.BN_EPILOGUE = True
        fw = [a for a, _, _ in rows]
        ge = [b[1] for _, b, _ in rows]
        print("%-42s fwd diff max %.2e min %.2e | grad err max %.2e min %.2e (%s) | sum(x) %s" %
              (name, max(fw), min(fw), max(ge), min(ge), max(rows, key=lambda t: t[1][1])[1][0],
               sorted(set(round(c, 6) for _, _, c in rows))))
        print("   grad errs:", " ".join("%.1e" % g for g in ge))
    mr.set_compute_dtype(torch.float32)
    torch.manual_seed(1)
    full = deformable_resnet50(pretrained=False)
    perturb_offset_convs(full, kink_safe=KINK_SAFE)
    m = full.layer2.to(DEV).train()
    torch.manual_seed(1)
    x = torch.randn(2, 256, 24, 32, device=DEV)
    print("kink distance per block:", kink_distance(m, x))


if __name__ == "__main__":
    main()
