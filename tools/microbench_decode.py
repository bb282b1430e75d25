"""Persistent decode forward (mr_decode_persist_fwd) beside the per-step launches it replaces, plus the kernel's own phase clock.
  python tools/microbench_decode.py [N T Ep S]
"""
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import test_decode_persist_gpu as t  # noqa: E402
from megreader_amd._lib import call, load, ptr  # noqa: E402

N, T, Ep, S = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (16, 64, 552, 32)
d = t._inputs(N, T, Ep, S, 38, 1)
H = t.H


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


COINS = len(sys.argv) > 5 and sys.argv[5] == "coins"
xc = t._coin_inputs(d, N, S, 38, 3)
b = t._buffers(N, T, Ep, S, d["h0"])
nbytes = load().mr_decode_persist_ws_bytes(N)
ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")


def persistent(timing=False):
    ws.zero_()
    if timing:
        ws[nbytes - 256:].view(torch.int32)[2] = 0x54494D45
    call("mr_decode_persist_fwd", ptr(d["cat_w"]), ptr(d["cat_b"]), ptr(d["ic_w"]), Ep, ptr(d["G"]), 3 * H, ptr(d["idx"]),
         ptr(xc["flags"]) if COINS else 0, ptr(xc["out_w"]), ptr(xc["out_b"]), 38, ptr(d["eproj"]), ptr(d["enc"]), ptr(d["v"]), ptr(b["H_all"]), ptr(b["HC_all"]), ptr(b["W_att"]), ptr(b["CTX_all"]),
         ptr(b["SAVE_all"]), ptr(ws), -nbytes, S, N, T, Ep)


# the per-step launches inside one hipGraph (as the training step replays them)
g = torch.cuda.CUDAGraph()
s_ = torch.cuda.Stream()
with torch.cuda.stream(s_):
    t._per_step(d, N, T, Ep, S)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s_):
        t._per_step(d, N, T, Ep, S)
print("per-step launches (graph replay): %.1f us  = %.2f us/step" % (timed(g.replay), timed(g.replay) / S))
us = timed(persistent)
print("persistent (incl. ws memset)    : %.1f us  = %.2f us/step" % (us, us / S))
persistent(True)
torch.cuda.synchronize()
st = ws[nbytes - 256:].view(torch.int32).cpu()
names = ["gemm1+sync", "energies+publish", "owner gather", "reduce+softmax", "ctx+publish", "ctx gather", "gemm2+sync",
         "gru+publish", "h gather"]
for g_ in (0, 1):
    v = [int(st[8 + 16 * g_ + i]) * 0.01 / S for i in range(9)]
    print("slice %d (%s): " % (g_, "owner" if g_ == 0 else "non-owner") +
          "  ".join("%s %.2f" % (n, x) for n, x in zip(names, v)) + "   sum %.2f us/step" % sum(v))
print("status", int(st[0]), "workgroups that found their group on one XCD:", int(st[1]))

# ---- backward
fw = t._per_step(d, N, T, Ep, S)
x = t._bwd_inputs(d, fw, N, T, Ep, S, seed=3, with_ga=False)
gb = torch.cuda.CUDAGraph()
with torch.cuda.stream(s_):
    t._bwd_per_step(d, fw, x, N, T, Ep, S)
    torch.cuda.synchronize()
    with torch.cuda.graph(gb, stream=s_):
        t._bwd_per_step(d, fw, x, N, T, Ep, S)
us = timed(gb.replay)
print("backward per-step launches (graph replay): %.1f us  = %.2f us/step" % (us, us / S))
bb = t._bwd_buffers(N, T, Ep, S)
nb = load().mr_decode_persist_bwd_ws_bytes(N)
wsb = torch.zeros((nb,), dtype=torch.uint8, device="cuda")


def persistent_bwd(timing=False):
    wsb.zero_()
    if timing:
        wsb[nb - 256:].view(torch.int32)[2] = 0x54494D45
    call("mr_decode_persist_bwd", ptr(x["cat_wt"]), ptr(x["ic_wt"]), 3 * H, ptr(d["eproj"]), ptr(d["enc"]), ptr(d["v"]),
         ptr(fw["H_all"]), ptr(fw["HC_all"]), ptr(fw["W_att"]), ptr(fw["SAVE_all"]), ptr(x["DHO"]), 0, S * T, ptr(bb["DGI"]),
         ptr(bb["DHC"]), ptr(bb["DCTX"]), ptr(bb["deproj"]), ptr(bb["dv"]), ptr(bb["denc"]), ptr(wsb), -nb, S, N, T, Ep)


us = timed(persistent_bwd)
print("backward persistent (incl. ws memset)     : %.1f us  = %.2f us/step" % (us, us / S))
persistent_bwd(True)
torch.cuda.synchronize()
st = wsb[nb - 256:].view(torch.int32).cpu()
names = ["dh gather+reduce", "gru bwd", "dctx mfma+publish", "dctx gather", "dctx reduce", "dw partial+publish", "dw gather",
         "softmax bwd", "tanh chain", "dh mfma+publish"]
for g_ in (0, 1):
    v = [int(st[8 + 16 * g_ + i]) * 0.01 / S for i in range(10)]
    print("slice %d: " % g_ + "  ".join("%s %.2f" % (n, x) for n, x in zip(names, v)) + "   sum %.2f us/step" % sum(v))
print("status", int(st[0]), "workgroups that found their group on one XCD:", int(st[1]))
