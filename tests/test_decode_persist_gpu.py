"""mr_decode_persist_fwd (csrc/decode_persist.hip): the teacher-forced attention-GRU decode loop of the reference
(decoders/attention_decoder.py:84-118 around AttentionRNNCell :146-231) as ONE persistent launch, against

  * the per-step launches it replaces (mr_gemm_nt + mr_attn_fwd2 + mr_gemm_gru_fwd through the C ABI), buffer by buffer: both round
    at the same points, so only f32 summation order (and one bf16 ulp where that order flips a rounding) separates them;
  * a float64 restatement of the same recurrence on the same bf16 inputs.

Shapes: the published one (N = 16, T = 64, H = 512, Ep = 552, S = 32), two batch groups, ragged groups / positions / channels.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from megreader_amd._lib import call, dtype_code, load, ptr  # noqa: E402

DEV = "cuda"
H = 512


def _inputs(N, T, Ep, S, C, seed):
    g = torch.Generator().manual_seed(seed)
    bf = torch.bfloat16
    d = {}
    d["cat_w"] = (torch.randn(4 * H, H, generator=g) * H ** -0.5).to(bf)
    d["cat_b"] = torch.cat([torch.zeros(H), torch.randn(3 * H, generator=g) * 0.1]).float()
    d["ic_w"] = (torch.randn(3 * H, Ep, generator=g) * Ep ** -0.5).to(bf)
    d["G"] = (torch.randn(C, 3 * H, generator=g) * 0.5).to(bf)
    d["idx"] = torch.randint(0, C, (S, N), generator=g, dtype=torch.int64)
    d["eproj"] = (torch.randn(N, T, H, generator=g) * 0.7).to(bf)
    d["enc"] = torch.randn(N, T, Ep, generator=g).to(bf)
    d["v"] = (torch.randn(H, generator=g) * H ** -0.5 * 4).float()
    d["h0"] = (torch.randn(N, H, generator=g) * 0.3).to(bf)
    return {k: v.to(DEV).contiguous() for k, v in d.items()}


def _buffers(N, T, Ep, S, h0):
    bf = torch.bfloat16
    b = {"H_all": torch.full((S + 1, N, H), float("nan"), dtype=bf, device=DEV),
         "HC_all": torch.full((S, N, 4 * H), float("nan"), dtype=bf, device=DEV),
         "W_att": torch.full((S, N, T), float("nan"), dtype=torch.float32, device=DEV),
         "CTX_all": torch.full((S, N, Ep), float("nan"), dtype=bf, device=DEV),
         "SAVE_all": torch.full((S, N, 3 * H), float("nan"), dtype=torch.float32, device=DEV)}
    b["H_all"][0].copy_(h0)
    return b


def _per_step(d, N, T, Ep, S):
    dt = dtype_code(torch.bfloat16)
    b = _buffers(N, T, Ep, S, d["h0"])
    HC = 4 * H
    for s in range(S):
        call("mr_gemm_nt", dt, ptr(b["H_all"][s]), H, ptr(d["cat_w"]), H, ptr(b["HC_all"][s]), HC, ptr(d["cat_b"]), 0, N, HC, H)
        call("mr_attn_fwd2", dt, ptr(b["HC_all"][s]), HC, ptr(d["eproj"]), ptr(d["v"]), ptr(d["enc"]), ptr(b["W_att"][s]),
             ptr(b["CTX_all"][s]), N, T, H, Ep)
        call("mr_gemm_gru_fwd", dt, ptr(b["CTX_all"][s]), Ep, ptr(d["ic_w"]), Ep, ptr(d["G"]), 3 * H, ptr(d["idx"][s]),
             ptr(b["HC_all"][s]) + H * 2, HC, ptr(b["H_all"][s]), ptr(b["H_all"][s + 1]), ptr(b["SAVE_all"][s]), N, H, Ep)
    return b


def _persistent(d, N, T, Ep, S, prezero):
    b = _buffers(N, T, Ep, S, d["h0"])
    nbytes = load().mr_decode_persist_ws_bytes(N)
    ws = torch.zeros((nbytes,), dtype=torch.uint8, device=DEV) if prezero else \
        torch.full((nbytes,), 0xAB, dtype=torch.uint8, device=DEV)
    call("mr_decode_persist_fwd", ptr(d["cat_w"]), ptr(d["cat_b"]), ptr(d["ic_w"]), Ep, ptr(d["G"]), 3 * H, ptr(d["idx"]),
         0, 0, 0, 0, ptr(d["eproj"]), ptr(d["enc"]), ptr(d["v"]), ptr(b["H_all"]), ptr(b["HC_all"]), ptr(b["W_att"]),
         ptr(b["CTX_all"]), ptr(b["SAVE_all"]), ptr(ws), -nbytes if prezero else nbytes, S, N, T, Ep)
    torch.cuda.synchronize()
    status = int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item())
    return b, status


def _f64(d, N, T, Ep, S):
    """The recurrence in float64 on the same (bf16-valued) inputs; no intermediate rounding."""
    f = {k: v.double() for k, v in d.items() if k != "idx"}
    h = f["h0"]
    out = {"H_all": [h], "W_att": [], "CTX_all": []}
    for s in range(S):
        hc = h @ f["cat_w"].t() + f["cat_b"]
        hproj, gh = hc[:, :H], hc[:, H:]
        energy = torch.tanh(hproj.unsqueeze(1) + f["eproj"]) @ f["v"]
        w = torch.softmax(energy, dim=1)
        ctx = torch.bmm(w.unsqueeze(1), f["enc"]).squeeze(1)
        gi = f["G"][d["idx"][s]] + ctx @ f["ic_w"].t()
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        out["H_all"].append(h)
        out["W_att"].append(w)
        out["CTX_all"].append(ctx)
    return {k: torch.stack(v) for k, v in out.items()}


CASES = [(16, 64, 552, 32, 38), (32, 64, 576, 6, 97), (5, 33, 64, 4, 11), (17, 20, 8, 3, 5), (1, 1, 16, 2, 3), (16, 64, 552, 1, 38)]


@pytest.mark.parametrize("N,T,Ep,S,C", CASES)
@pytest.mark.parametrize("prezero", [True, False])
def test_persistent_decode_matches_per_step_launches(N, T, Ep, S, C, prezero):
    assert load().mr_decode_persist_ok(dtype_code(torch.bfloat16), N, T, H, Ep) == 1
    d = _inputs(N, T, Ep, S, C, seed=N * 131 + T)
    ref = _per_step(d, N, T, Ep, S)
    got, status = _persistent(d, N, T, Ep, S, prezero)
    assert status == 0, "a hand-off of the persistent decode kernel timed out (code %d)" % status
    for k in ("H_all", "HC_all", "W_att", "CTX_all", "SAVE_all"):
        a, b = got[k].double(), ref[k].double()
        assert torch.isfinite(a).all(), k
        # one bf16 ulp of the largest magnitude, a few times over the steps (a flipped rounding feeds the next step)
        tol = 3e-2 if got[k].dtype == torch.bfloat16 else 2e-2
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), k
        # ... and almost everywhere far closer than that
        assert float(((a - b).abs() > 4e-3 * max(1.0, float(b.abs().max()))).double().mean()) < 0.02, k
    assert float((got["W_att"].sum(-1) - 1).abs().max()) < 1e-5


# N <= 32 runs groups of 4 rows, N <= 64 of 8 (decode_persist.hip: decode_rows); the per-step GRU launch stops at 32
@pytest.mark.parametrize("N,T,Ep,S,C", CASES[:4] + [(40, 64, 552, 3, 38), (64, 37, 576, 2, 7), (9, 64, 552, 5, 38)])
def test_persistent_decode_vs_f64(N, T, Ep, S, C):
    d = _inputs(N, T, Ep, S, C, seed=N * 17 + S)
    got, status = _persistent(d, N, T, Ep, S, True)
    assert status == 0
    ref = _f64(d, N, T, Ep, S)
    per = _per_step(d, N, T, Ep, S) if N <= 32 else None
    for k in ("H_all", "W_att", "CTX_all"):
        assert torch.isfinite(got[k].double()).all(), k
        e_new = float((got[k].double() - ref[k]).abs().max())
        scale = max(1.0, float(ref[k].abs().max()))
        assert e_new <= 3e-2 * scale, (k, e_new)
        if per is not None:
            e_old = float((per[k].double() - ref[k]).abs().max())
            assert e_new <= 2.0 * e_old + 4e-3 * scale, (k, e_new, e_old)   # no further from f64 than the launches it replaces


def test_persistent_decode_repeatable_and_shape_gate():
    N, T, Ep, S, C = 16, 64, 552, 8, 38
    d = _inputs(N, T, Ep, S, C, seed=5)
    a, sa = _persistent(d, N, T, Ep, S, True)
    b, sb = _persistent(d, N, T, Ep, S, False)
    assert sa == 0 and sb == 0
    for k in a:
        assert torch.equal(a[k], b[k]), k          # granule arrival order never enters the arithmetic
    lib = load()
    bf = dtype_code(torch.bfloat16)
    assert lib.mr_decode_persist_ok(dtype_code(torch.float32), N, T, H, Ep) == 0
    assert lib.mr_decode_persist_ok(bf, N, T, 256, Ep) == 0
    assert lib.mr_decode_persist_ok(bf, N, 65, H, Ep) == 0
    assert lib.mr_decode_persist_ok(bf, N, T, H, 580) == 0
    assert lib.mr_decode_persist_ok(bf, 65, T, H, Ep) == 0
    from megreader_amd._lib import set_tuning
    set_tuning(decode_persist=0)
    try:
        assert lib.mr_decode_persist_ok(bf, N, T, H, Ep) == 0
    finally:
        set_tuning(decode_persist=1)


# ------------------------------------------------------------------------------------------------------------------ backward
def _bwd_inputs(d, fw, N, T, Ep, S, seed, with_ga):
    g = torch.Generator().manual_seed(seed)
    x = {"cat_wt": d["cat_w"].t().contiguous(), "ic_wt": d["ic_w"].t().contiguous(),
         "DHO": (torch.randn(S, N, H, generator=g) * 0.05).to(torch.bfloat16).to(DEV),
         "ga": (torch.randn(N, S, T, generator=g) * 0.1).float().to(DEV) if with_ga else None}
    return x


def _bwd_buffers(N, T, Ep, S):
    bf = torch.bfloat16
    return {"DGI": torch.full((S, N, 3 * H), float("nan"), dtype=bf, device=DEV),
            "DHC": torch.full((S, N, 4 * H), float("nan"), dtype=bf, device=DEV),
            "DCTX": torch.full((S, N, Ep), float("nan"), dtype=bf, device=DEV),
            "denc": torch.full((N, T, Ep), float("nan"), dtype=bf, device=DEV),
            "deproj": torch.zeros((N, T, H), dtype=torch.float32, device=DEV),
            "dv": torch.zeros((H,), dtype=torch.float32, device=DEV)}


def _bwd_per_step(d, fw, x, N, T, Ep, S):
    """The launches of _DecodeLoopFn.backward (decoders/attention_decoder.py), fused flavour."""
    dt = dtype_code(torch.bfloat16)
    b = _bwd_buffers(N, T, Ep, S)
    HC, H3 = 4 * H, 3 * H
    dh_b = torch.empty((N, H), dtype=torch.bfloat16, device=DEV)
    ga = x["ga"]
    for s in range(S - 1, -1, -1):
        last = s == S - 1
        if not last:
            call("mr_gemm_gru_bwd", dt, ptr(b["DHC"][s + 1]), HC, ptr(x["cat_wt"]), HC, ptr(dh_b), ptr(x["DHO"][s]),
                 ptr(fw["SAVE_all"][s]), ptr(fw["HC_all"][s]) + H * 2, HC, ptr(fw["H_all"][s]), ptr(b["DGI"][s]),
                 ptr(b["DHC"][s]) + H * 2, HC, ptr(dh_b), N, H, HC)
        else:
            call("mr_gru_bwd2", dt, 0, 0, ptr(x["DHO"][s]), ptr(fw["SAVE_all"][s]), ptr(fw["HC_all"][s]) + H * 2, HC,
                 ptr(fw["H_all"][s]), ptr(b["DGI"][s]), ptr(b["DHC"][s]) + H * 2, HC, ptr(dh_b), N, H)
        call("mr_gemm_nt", dt, ptr(b["DGI"][s]), H3, ptr(x["ic_wt"]), H3, ptr(b["DCTX"][s]), Ep, 0, 0, N, Ep, H3)
        call("mr_attn_bwd2", dt, ptr(b["DCTX"][s]), (ptr(ga) + s * T * 4) if ga is not None else 0, S * T,
             ptr(fw["HC_all"][s]), HC, ptr(d["eproj"]), ptr(d["v"]), ptr(d["enc"]), ptr(fw["W_att"][s]), ptr(b["DHC"][s]), HC,
             ptr(b["deproj"]), ptr(b["dv"]), N, T, H, Ep)
    call("mr_attn_denc", dt, ptr(fw["W_att"]), ptr(b["DCTX"]), ptr(b["denc"]), S, N, T, Ep)
    return b


def _bwd_persistent(d, fw, x, N, T, Ep, S, prezero=True):
    b = _bwd_buffers(N, T, Ep, S)
    b["deproj"].fill_(float("nan"))                 # written, not accumulated
    nbytes = load().mr_decode_persist_bwd_ws_bytes(N)
    ws = torch.zeros((nbytes,), dtype=torch.uint8, device=DEV) if prezero else \
        torch.full((nbytes,), 0xCD, dtype=torch.uint8, device=DEV)
    ga = x["ga"]
    call("mr_decode_persist_bwd", ptr(x["cat_wt"]), ptr(x["ic_wt"]), 3 * H, ptr(d["eproj"]), ptr(d["enc"]), ptr(d["v"]),
         ptr(fw["H_all"]), ptr(fw["HC_all"]), ptr(fw["W_att"]), ptr(fw["SAVE_all"]), ptr(x["DHO"]),
         ptr(ga) if ga is not None else 0, S * T, ptr(b["DGI"]), ptr(b["DHC"]), ptr(b["DCTX"]), ptr(b["deproj"]), ptr(b["dv"]),
         ptr(b["denc"]), ptr(ws), -nbytes if prezero else nbytes, S, N, T, Ep)
    torch.cuda.synchronize()
    status = int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item())
    return b, status


def _f64_grads(d, x, N, T, Ep, S):
    """autograd through the float64 recurrence: gradients of sum_s <h'_s, DHO_s> + <w_s, ga_s> wrt eproj and v, plus the
    per-step gradient of the contexts."""
    f = {k: v.double() for k, v in d.items() if k != "idx"}
    eproj = f["eproj"].clone().requires_grad_(True)
    v = f["v"].clone().requires_grad_(True)
    h = f["h0"]
    total = 0.0
    ctxs = []
    for s in range(S):
        hc = h @ f["cat_w"].t() + f["cat_b"]
        hproj, gh = hc[:, :H], hc[:, H:]
        w = torch.softmax(torch.tanh(hproj.unsqueeze(1) + eproj) @ v, dim=1)
        ctx = torch.bmm(w.unsqueeze(1), f["enc"]).squeeze(1)
        ctx.retain_grad()
        ctxs.append(ctx)
        gi = f["G"][d["idx"][s]] + ctx @ f["ic_w"].t()
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        total = total + (h * x["DHO"][s].double()).sum()
        if x["ga"] is not None:
            total = total + (w * x["ga"][:, s].double()).sum()
    total.backward()
    return {"deproj": eproj.grad, "dv": v.grad, "DCTX": torch.stack([c.grad for c in ctxs])}


BWD_CASES = [(16, 64, 552, 32, 38, False), (16, 64, 552, 5, 38, True), (32, 64, 576, 6, 97, True), (5, 33, 64, 4, 11, False),
             (17, 20, 8, 3, 5, True), (1, 1, 16, 2, 3, False), (16, 64, 552, 1, 38, False)]


def _close(a, b, tol, frac_tol, name):
    a, b = a.double(), b.double()
    assert torch.isfinite(a).all(), name
    scale = max(1e-6, float(b.abs().max()))
    assert float((a - b).abs().max()) <= tol * scale, (name, float((a - b).abs().max()), scale)
    assert float(((a - b).abs() > frac_tol * scale).double().mean()) < 0.02, name


@pytest.mark.parametrize("N,T,Ep,S,C,with_ga", BWD_CASES)
def test_persistent_decode_backward_matches_per_step_launches(N, T, Ep, S, C, with_ga):
    assert load().mr_decode_persist_bwd_ok(dtype_code(torch.bfloat16), N, T, H, Ep) == 1
    d = _inputs(N, T, Ep, S, C, seed=N * 7 + T)
    fw = _per_step(d, N, T, Ep, S)
    x = _bwd_inputs(d, fw, N, T, Ep, S, seed=3, with_ga=with_ga)
    ref = _bwd_per_step(d, fw, x, N, T, Ep, S)
    got, status = _bwd_persistent(d, fw, x, N, T, Ep, S, prezero=(S % 2 == 0))
    assert status == 0, "a hand-off of the persistent decode backward timed out (code %d)" % status
    # both paths round dgi / dgh / dctx / dhproj to bf16 at the same points; the persistent kernel carries dh_b in f32 (the
    # launches round it to bf16 between steps) and sums its GEMMs in another order
    for k in ("DGI", "DHC", "DCTX", "denc"):
        _close(got[k], ref[k], 4e-2, 1e-2, k)
    _close(got["deproj"], ref["deproj"], 2e-2, 5e-3, "deproj")
    _close(got["dv"], ref["dv"], 2e-2, 1e-2, "dv")


@pytest.mark.parametrize("N,T,Ep,S,C,with_ga", BWD_CASES[1:5])
def test_persistent_decode_backward_vs_f64(N, T, Ep, S, C, with_ga):
    d = _inputs(N, T, Ep, S, C, seed=N * 3 + S)
    fw = _per_step(d, N, T, Ep, S)
    x = _bwd_inputs(d, fw, N, T, Ep, S, seed=9, with_ga=with_ga)
    got, status = _bwd_persistent(d, fw, x, N, T, Ep, S)
    assert status == 0
    per = _bwd_per_step(d, fw, x, N, T, Ep, S)
    ref = _f64_grads(d, x, N, T, Ep, S)
    for k in ("deproj", "dv", "DCTX"):
        scale = max(1e-6, float(ref[k].abs().max()))
        e_new = float((got[k].double() - ref[k]).abs().max()) / scale
        e_old = float((per[k].double() - ref[k]).abs().max()) / scale
        assert e_new <= 5e-2, (k, e_new)
        assert e_new <= 2.0 * e_old + 5e-3, (k, e_new, e_old)         # no further from f64 than the launches it replaces


def test_persistent_decode_backward_repeatable():
    N, T, Ep, S, C = 16, 64, 552, 6, 38
    d = _inputs(N, T, Ep, S, C, seed=11)
    fw = _per_step(d, N, T, Ep, S)
    x = _bwd_inputs(d, fw, N, T, Ep, S, seed=2, with_ga=True)
    a, sa = _bwd_persistent(d, fw, x, N, T, Ep, S, True)
    b, sb = _bwd_persistent(d, fw, x, N, T, Ep, S, False)
    assert sa == 0 and sb == 0
    for k in ("DGI", "DHC", "DCTX", "deproj", "denc"):
        assert torch.equal(a[k], b[k]), k
    assert float((a["dv"] - b["dv"]).abs().max()) <= 1e-5 * float(b["dv"].abs().max())     # atomics: order of the four waves
    assert load().mr_decode_persist_bwd_ok(dtype_code(torch.bfloat16), 33, T, H, Ep) == 0


# ------------------------------------------------------------------------------------------------------- arg-max feedback
def _coin_inputs(d, N, S, C, seed):
    g = torch.Generator().manual_seed(seed)
    x = {"out_w": (torch.randn(C, H, generator=g) * H ** -0.5 * 3).to(torch.bfloat16).to(DEV),
         "out_b": (torch.randn(C, generator=g) * 0.2).float().to(DEV),
         "targets": torch.randint(0, C, (S, N), generator=g, dtype=torch.int64).to(DEV),
         "flags": (torch.rand(S, generator=g) > 0.5).to(torch.int32).to(DEV)}
    x["idx0"] = torch.cat((torch.full((1, N), C - 1, dtype=torch.int64, device=DEV), x["targets"][:S - 1]), 0).contiguous()
    return x


def _per_step_coins(d, x, N, T, Ep, S, C):
    """_DecodeLoopFn.forward's fused launches with the per-step output layer + feedback word (mr_out_nll_fwd)."""
    dt = dtype_code(torch.bfloat16)
    b = _buffers(N, T, Ep, S, d["h0"])
    HC = 4 * H
    idx = torch.empty((S, N), dtype=torch.int64, device=DEV)
    idx[0].copy_(x["idx0"][0])
    lp = torch.empty((S, N, C), dtype=torch.float32, device=DEV)
    loss = torch.zeros((N,), dtype=torch.float32, device=DEV)
    am = torch.empty((S, N), dtype=torch.int64, device=DEV)
    mask = torch.ones((S, N), dtype=torch.float32, device=DEV)
    for s in range(S):
        call("mr_gemm_nt", dt, ptr(b["H_all"][s]), H, ptr(d["cat_w"]), H, ptr(b["HC_all"][s]), HC, ptr(d["cat_b"]), 0, N, HC, H)
        call("mr_attn_fwd2", dt, ptr(b["HC_all"][s]), HC, ptr(d["eproj"]), ptr(d["v"]), ptr(d["enc"]), ptr(b["W_att"][s]),
             ptr(b["CTX_all"][s]), N, T, H, Ep)
        call("mr_gemm_gru_fwd", dt, ptr(b["CTX_all"][s]), Ep, ptr(d["ic_w"]), Ep, ptr(d["G"]), 3 * H, ptr(idx[s]),
             ptr(b["HC_all"][s]) + H * 2, HC, ptr(b["H_all"][s]), ptr(b["H_all"][s + 1]), ptr(b["SAVE_all"][s]), N, H, Ep)
        last = s + 1 == S
        call("mr_out_nll_fwd", dt, ptr(b["H_all"][s + 1]), H, ptr(x["out_w"]), H, ptr(x["out_b"]), ptr(x["targets"][s]), 1,
             ptr(mask[s]), ptr(lp[s]), ptr(loss), ptr(am[s]), 0 if last else ptr(x["flags"]) + 4 * s,
             0 if last else ptr(idx[s + 1]), N, C, H, 1 if s else 0)
    b["idx"] = idx
    return b


def _persistent_coins(d, x, N, T, Ep, S, C):
    b = _buffers(N, T, Ep, S, d["h0"])
    nbytes = load().mr_decode_persist_ws_bytes(N)
    ws = torch.zeros((nbytes,), dtype=torch.uint8, device=DEV)
    idx = x["idx0"].clone()
    call("mr_decode_persist_fwd", ptr(d["cat_w"]), ptr(d["cat_b"]), ptr(d["ic_w"]), Ep, ptr(d["G"]), 3 * H, ptr(idx),
         ptr(x["flags"]), ptr(x["out_w"]), ptr(x["out_b"]), C, ptr(d["eproj"]), ptr(d["enc"]), ptr(d["v"]), ptr(b["H_all"]),
         ptr(b["HC_all"]), ptr(b["W_att"]), ptr(b["CTX_all"]), ptr(b["SAVE_all"]), ptr(ws), -nbytes, S, N, T, Ep)
    torch.cuda.synchronize()
    b["idx"] = idx
    return b, int(ws[nbytes - 256:nbytes - 252].view(torch.int32).item())


@pytest.mark.parametrize("N,T,Ep,S,C", [(16, 64, 552, 32, 38), (32, 64, 552, 12, 38), (5, 33, 64, 9, 11), (17, 20, 8, 7, 5),
                                        (3, 9, 16, 6, 256), (40, 64, 552, 5, 97)])
def test_persistent_decode_argmax_feedback(N, T, Ep, S, C):
    d = _inputs(N, T, Ep, S, C, seed=N * 5 + S)
    x = _coin_inputs(d, N, S, C, seed=S)
    assert int(x["flags"].sum()) < S          # some steps feed the arg-max
    got, status = _persistent_coins(d, x, N, T, Ep, S, C)
    assert status == 0
    if N <= 32:
        ref = _per_step_coins(d, x, N, T, Ep, S, C)
        # the fed words decide everything downstream: they must agree (random logits are never close to a tie)
        assert torch.equal(got["idx"], ref["idx"])
        for k in ("H_all", "HC_all", "W_att", "CTX_all", "SAVE_all"):
            a, b = got[k].double(), ref[k].double()
            assert torch.isfinite(a).all(), k
            assert float((a - b).abs().max()) <= 3e-2 * max(1.0, float(b.abs().max())), k
    # the words the kernel wrote are the arg-max of ITS OWN hidden states (first index on ties), the others were left alone
    logits = got["H_all"][1:].float() @ x["out_w"].float().t() + x["out_b"]
    am = logits.argmax(-1)
    for s in range(S - 1):
        if int(x["flags"][s]) == 0:
            top2 = logits[s].topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 1e-3
            assert torch.equal(got["idx"][s + 1][clear], am[s][clear]), s
        else:
            assert torch.equal(got["idx"][s + 1], x["targets"][s]), s
    assert torch.equal(got["idx"][0], x["idx0"][0])


def test_persistent_decode_beside_a_busy_side_stream():
    """The two kernels need every workgroup of a batch group co-resident (N = 32: 256 workgroups, one per CU).  Beside a foreign
    stream of chip-filling kernels (what a gradient all-reduce is to the scheduler; tests/test_persistent_beside_gpu.py) workgroups
    are dispatched late: the groups wait (bounded), nothing times out, the results do not change."""
    N, T, Ep, S, C = 32, 64, 552, 32, 38
    d = _inputs(N, T, Ep, S, C, seed=21)
    x = _coin_inputs(d, N, S, C, seed=4)
    ref, st = _persistent_coins(d, x, N, T, Ep, S, C)
    assert st == 0
    bx = _bwd_inputs(d, ref, N, T, Ep, S, seed=5, with_ga=False)
    bref, st = _bwd_persistent(d, ref, bx, N, T, Ep, S)
    assert st == 0
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    big2 = torch.empty_like(big)
    for it in range(12):
        with torch.cuda.stream(side):
            for _ in range(3):
                torch.mm(a, a)
                big2.copy_(big)
        for _ in range(3):
            got, st = _persistent_coins(d, x, N, T, Ep, S, C)
            assert st == 0, (it, st)
            bgot, st = _bwd_persistent(d, got, bx, N, T, Ep, S)
            assert st == 0, (it, st)
        for k in ("H_all", "HC_all", "W_att", "CTX_all", "SAVE_all", "idx"):
            assert torch.equal(got[k], ref[k]), (it, k)
        for k in ("DGI", "DHC", "DCTX", "deproj", "denc"):
            assert torch.equal(bgot[k], bref[k]), (it, k)
    torch.cuda.synchronize()
