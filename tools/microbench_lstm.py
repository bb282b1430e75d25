#!/usr/bin/env python
"""Timing of the BiLSTM layer kernels at CRNN shapes (T=33, N=256): recurrence fwd/bwd and the big GEMMs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megreader_amd as mr  # noqa: E402
from megreader_amd.nn import functional as F  # noqa: E402


def timeit(f, iters=10):
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
    from megreader_amd._lib import load
    dtype = torch.bfloat16
    mr.set_compute_dtype(dtype)
    T, N, H = 33, 256, 256
    variants = [(0, 0), (0, 16), (0, 32), (0, 64), (32, 32), (64, 32)] if "--sweep" in sys.argv else [(-1, -1)]
    persist = 0 if "--no-persist" in sys.argv else (2 if "--no-xcd" in sys.argv else 1)
    load().mr_set_lstm_persist(persist)
    for fv, bv in variants:
      load().mr_set_lstm_variant(fv, bv)
      print("variant fwd_bn=%d bwd_bn=%d persistent=%d" % (fv, bv, persist))
      for I in (512, 256):
        torch.manual_seed(0)
        ref = torch.nn.LSTM(I, H, bidirectional=True)
        params = [p.detach().cuda().requires_grad_(True) for p in ref.parameters()]
        x = torch.randn(T, N, I, device="cuda").to(dtype).requires_grad_(True)
        g = torch.randn(T, N, 2 * H, device="cuda").to(dtype)

        def fwd():
            with torch.no_grad():
                F.bilstm(x, *params)

        def fwdbwd():
            y = F.bilstm(x, *params)
            y.backward(g)

        # graph replay removes the host launch cost (the training step runs the same way)
        def graphed(fn):
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                fn()
            return gr.replay

        F.LSTM_LOCAL = []
        fwdbwd()
        torch.cuda.synchronize()
        print("  workgroups that found their batch group on one XCD (fwd, bwd): %s of %d" %
              ([int(w.view(torch.int32).item()) for w in F.LSTM_LOCAL], 2 * ((N + 15) // 16) * 4))
        F.LSTM_LOCAL = None
        if "--phases" in sys.argv:
            # the forward kernel's own phase clock (status words 8..11 of its workspace, armed by status word 2)
            orig = F._lstm_workspace
            seen = []

            def armed(dt, T_, N_, H_, dev):
                ws, size = orig(dt, T_, N_, H_, dev)
                if ws is not None:
                    ws[ws.numel() - 256:].view(torch.int32)[2] = 0x54494D45
                    seen.append(ws)
                return ws, size
            F._lstm_workspace = armed
            try:
                fwdbwd()
                torch.cuda.synchronize()
            finally:
                F._lstm_workspace = orig
            st = seen[0][seen[0].numel() - 256:].view(torch.int32).cpu()
            names = ["gather wait", "LDS + MFMA", "gate math + publish", "stores"]
            print("  forward phases (us per step): " + "  ".join("%s %.2f" % (n, int(st[8 + i]) * 0.01 / T) for i, n in enumerate(names)))
            if len(seen) > 1:
                st = seen[1][seen[1].numel() - 256:].view(torch.int32).cpu()
                names = ["gather wait + reduce", "gate algebra + LDS + barrier", "MFMA + publish", "barrier"]
                print("  backward phases (us per step): " + "  ".join("%s %.2f" % (n, int(st[12 + i]) * 0.01 / T) for i, n in enumerate(names)))
        tf = timeit(graphed(fwd))
        tfb = timeit(graphed(fwdbwd))
        print("  BiLSTM I=%d: fwd %.1f us, fwd+bwd %.1f us (bwd ~%.1f us)" % (I, tf, tfb, tfb - tf))


if __name__ == "__main__":
    main()
