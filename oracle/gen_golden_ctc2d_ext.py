"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Golden vectors of the REFERENCE's 2D-CTC extension.

Runs the reference's own `ctc_2d_csrc` module (ops/ctc_2d/csrc/** compiled for gfx950 where it lies by
oracle/build_ref_ext.sh into oracle/_ref/) on the MI355X with the calling sequence of ops/ctc_2d/ctc_loss_2d.py:15-35
(`ctc2d_forward(log_probs, targets, input_lengths, target_lengths, blank, finfo.tiny)` -> (nll, log_alpha);
`ctc2d_backward(grad_out, log_probs, targets, input_lengths, target_lengths, nll, log_alpha, blank)` -> grad) and stores inputs and
outputs in one npz:

    python oracle/gen_golden_ctc2d_ext.py --out tests/golden/ctc2d_reference_ext.npz     # on a GPU box

Consumers: tests/test_oracle_ctc2d_ext_pinned_cpu.py (oracle/ctc2d.py against the CUDA kernels' outputs; the oracle was so far
pinned by the reference's PYTHON CTCLoss2D, decoders/ctc_loss2d.py) and tests/test_ctc2d_reference_gpu.py (the HIP kernels)."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

# T, H, N, C, S (padded target length), longest target
CASES = [(12, 4, 3, 10, 8, 5), (16, 4, 5, 38, 32, 6), (9, 1, 4, 7, 5, 4), (20, 8, 2, 38, 32, 6), (32, 8, 4, 38, 32, 12)]


def load_reference_extension():
    if not os.path.isdir(REF_DIR) or not any(f.startswith("ctc_2d_csrc") for f in os.listdir(REF_DIR)):
        return None
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    import ctc_2d_csrc  # noqa: E402  (the reference's PYBIND11 module, ops/ctc_2d/csrc/ctc2d.cpp)
    return ctc_2d_csrc


def inputs(case):
    """Seeded inputs: normalised log-probabilities [T, H, N, C] (log_softmax over classes + the height-attention term folded in
    as the reference's decoder produces them, oracle.ctc2d.synthetic_lp), padded targets with an adjacent repeat, one ragged
    input length."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from oracle.ctc2d import synthetic_lp
    T, H, N, C, S, lmax = case
    rng = np.random.RandomState(T * 7 + H + N)
    lengths = rng.randint(1, min(lmax, (T - 1) // 2) + 1, size=N).astype(np.int64)
    tg = np.zeros((N, S), dtype=np.int64)
    for i, L in enumerate(lengths):
        tg[i, :L] = rng.randint(1, C, size=L)
    if N > 2:
        lengths[2] = max(lengths[2], 2)
        tg[2, 1] = tg[2, 0] = max(int(tg[2, 0]), 1)
    il = np.full(N, T, dtype=np.int64)
    if N > 1:
        il[1] = T - 3
    lp, _, _ = synthetic_lp(T, H, N, C, seed=T + N)
    go = (rng.rand(N) + 0.5).astype(np.float32)
    return lp.astype(np.float32), tg, il, lengths, go


def run(ext, lp, tg, il, tl, go, dev="cuda"):
    x = torch.from_numpy(lp).to(dev)
    args = (torch.from_numpy(tg).to(dev), torch.from_numpy(il).to(dev), torch.from_numpy(tl).to(dev))
    nll, log_alpha = ext.ctc2d_forward(x, *args, 0, torch.finfo().tiny)
    grad = ext.ctc2d_backward(torch.from_numpy(go).to(dev), x, *args, nll, log_alpha, 0)
    torch.cuda.synchronize()
    return dict(nll=nll, log_alpha=log_alpha, grad=grad)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden", "ctc2d_reference_ext.npz"))
    args = ap.parse_args()
    ext = load_reference_extension()
    if ext is None:
        raise SystemExit("oracle/_ref holds no ctc_2d_csrc: run `bash oracle/build_ref_ext.sh` where /root/reference exists")
    if not torch.cuda.is_available():
        raise SystemExit("the reference extension is GPU-only (ops/ctc_2d/csrc/ctc2d.h:20: 'Not implemented on the CPU')")
    blob = {}
    for i, case in enumerate(CASES):
        lp, tg, il, tl, go = inputs(case)
        out = run(ext, lp, tg, il, tl, go)
        for k, v in dict(log_probs=lp, targets=tg, input_lengths=il, target_lengths=tl, grad_out=go).items():
            blob["%d/%s" % (i, k)] = v
        for k, v in out.items():
            blob["%d/%s" % (i, k)] = v.detach().float().cpu().numpy()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **blob)
    print("wrote %s: %d arrays, %.1f KB" % (args.out, len(blob), os.path.getsize(args.out) / 1024.0))


if __name__ == "__main__":
    main()
