"""Kernel-level tests of the attention decoder's step kernels (csrc/attention.hip, called through the C ABI) against
float64 torch restatements of the reference ops they replace (decoders/attention_decoder.py):

  mr_attn_step_fwd/bwd   Attn.forward :146-177 (energy = v . tanh(W[h; enc] + b), softmax over positions) + the context
                         bmm :207-209 -- with the Linear(1057 -> 512) split into its hidden / encoder halves
  mr_gru_gates_fwd/bwd   nn.GRUCell :192, gate order r, z, n
  mr_nll_step_fwd/bwd    log_softmax + NLLLoss(reduction='none') * mask + argmax :95-106
  mr_embed_rows_fwd/bwd  nn.Embedding(V, V) lookup :187-193, 203 and its scatter-add gradient

Sizes are the published configuration's (N = 32, T = 32, Hd = 512, Ep = 552 = 545 padded, C = 38) plus ragged ones.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from megreader_amd._lib import call, dtype_code, ptr  # noqa: E402

DEV = "cuda"


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _tol(dtype, f32, bf16):
    return f32 if dtype == torch.float32 else bf16


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,Tn,Hd,Ep", [(32, 32, 512, 552), (5, 17, 64, 40), (3, 64, 128, 72)])
def test_attn_step_fwd_bwd_vs_f64(dtype, N, Tn, Hd, Ep):
    g = torch.Generator().manual_seed(N * 1000 + Tn)
    hproj = (torch.randn(N, Hd, generator=g) * 0.7).to(dtype)
    eproj = (torch.randn(N, Tn, Hd, generator=g) * 0.7).to(dtype)
    enc = torch.randn(N, Tn, Ep, generator=g).to(dtype)
    v = torch.randn(Hd, generator=g) * (Hd ** -0.5) * 4
    gctx = torch.randn(N, Ep, generator=g).to(dtype)
    gw = torch.randn(N, Tn, generator=g)

    h64 = hproj.double().requires_grad_(True)
    e64 = eproj.double().requires_grad_(True)
    c64 = enc.double().requires_grad_(True)
    v64 = v.double().requires_grad_(True)
    energy = torch.tanh(h64.unsqueeze(1) + e64) @ v64
    w64 = torch.softmax(energy, dim=1)
    ctx64 = torch.bmm(w64.unsqueeze(1), c64).squeeze(1)
    (ctx64 * gctx.double()).sum().add((w64 * gw.double()).sum()).backward()

    hp, ep, en, vd = hproj.to(DEV), eproj.to(DEV).contiguous(), enc.to(DEV).contiguous(), v.to(DEV)
    w = torch.empty((N, Tn), dtype=torch.float32, device=DEV)
    ctx = torch.empty((N, Ep), dtype=dtype, device=DEV)
    call("mr_attn_step_fwd", dtype_code(dtype), ptr(hp), ptr(ep), ptr(vd), ptr(en), ptr(w), ptr(ctx), N, Tn, Hd, Ep)
    assert _rel(w, w64) < _tol(dtype, 2e-6, 2e-6) * 50          # inputs are identical (already rounded): f32 softmax
    assert _rel(ctx, ctx64) < _tol(dtype, 1e-5, 1e-2)
    assert float((w.sum(dim=1) - 1).abs().max()) < 1e-5

    deproj = torch.full((N, Tn, Hd), 0.5, dtype=torch.float32, device=DEV)    # += semantics: pre-filled
    denc = torch.full((N, Tn, Ep), -0.25, dtype=torch.float32, device=DEV)
    dv = torch.full((Hd,), 2.0, dtype=torch.float32, device=DEV)
    dh = torch.empty((N, Hd), dtype=dtype, device=DEV)
    gc, gwd = gctx.to(DEV), gw.to(DEV)
    call("mr_attn_step_bwd", dtype_code(dtype), ptr(gc), ptr(gwd), ptr(hp), ptr(ep), ptr(vd), ptr(en), ptr(w), ptr(dh),
         ptr(deproj), ptr(dv), ptr(denc), N, Tn, Hd, Ep)
    torch.cuda.synchronize()
    assert _rel(dh, h64.grad) < _tol(dtype, 2e-5, 2e-2)
    assert _rel(deproj - 0.5, e64.grad) < 2e-5
    assert _rel(denc + 0.25, c64.grad) < 2e-5
    assert _rel(dv - 2.0, v64.grad) < 2e-5
    # without an upstream gradient on the attention map (dweights = NULL)
    deproj.zero_(); denc.zero_(); dv.zero_()
    for t in (h64, e64, c64, v64):
        t.grad = None
    energy = torch.tanh(h64.unsqueeze(1) + e64) @ v64
    ctx64 = torch.bmm(torch.softmax(energy, dim=1).unsqueeze(1), c64).squeeze(1)
    (ctx64 * gctx.double()).sum().backward()
    call("mr_attn_step_bwd", dtype_code(dtype), ptr(gc), 0, ptr(hp), ptr(ep), ptr(vd), ptr(en), ptr(w), ptr(dh),
         ptr(deproj), ptr(dv), ptr(denc), N, Tn, Hd, Ep)
    assert _rel(dh, h64.grad) < _tol(dtype, 2e-5, 2e-2)
    assert _rel(deproj, e64.grad) < 2e-5 and _rel(dv, v64.grad) < 2e-5 and _rel(denc, c64.grad) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,two_inputs", [(32, 512, True), (7, 96, False), (256, 512, True)])
def test_gru_gates_vs_f64_grucell(dtype, N, H, two_inputs):
    """h' of nn.GRUCell given the three projections; the f64 reference IS torch.nn.GRUCell (weights = identity blocks so
    that gi / gh are the cell's pre-activations)."""
    g = torch.Generator().manual_seed(N + H)
    gi_a = torch.randn(N, 3 * H, generator=g).to(dtype)
    gi_b = (torch.randn(N, 3 * H, generator=g) * 0.5).to(dtype) if two_inputs else None
    gh = torch.randn(N, 3 * H, generator=g).to(dtype)
    h = torch.randn(N, H, generator=g).to(dtype)
    gout = torch.randn(N, H, generator=g).to(dtype)

    a64 = gi_a.double().requires_grad_(True)
    b64 = gi_b.double().requires_grad_(True) if two_inputs else None
    gh64 = gh.double().requires_grad_(True)
    h64 = h.double().requires_grad_(True)
    gi = a64 + b64 if two_inputs else a64
    r = torch.sigmoid(gi[:, :H] + gh64[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh64[:, H:2 * H])
    nn_ = torch.tanh(gi[:, 2 * H:] + r * gh64[:, 2 * H:])
    hn64 = (1 - z) * nn_ + z * h64
    # cross-check the restatement against torch.nn.GRUCell itself (x = gi through an identity weight_ih)
    cell = torch.nn.GRUCell(3 * H, H).double()
    with torch.no_grad():
        cell.weight_ih.copy_(torch.eye(3 * H, dtype=torch.float64))
        cell.bias_ih.zero_()
        cell.bias_hh.zero_()
        whh = torch.randn(3 * H, H, generator=g).double() * 0.1
        cell.weight_hh.copy_(whh)
    hh = torch.randn(4, H, generator=g).double()
    xi = torch.randn(4, 3 * H, generator=g).double()
    ghc = hh @ whh.t()
    rr = torch.sigmoid(xi[:, :H] + ghc[:, :H]); zz = torch.sigmoid(xi[:, H:2 * H] + ghc[:, H:2 * H])
    nc = torch.tanh(xi[:, 2 * H:] + rr * ghc[:, 2 * H:])
    assert float((cell(xi, hh) - ((1 - zz) * nc + zz * hh)).abs().max()) < 1e-12
    (hn64 * gout.double()).sum().backward()

    ad, ghd, hd = gi_a.to(DEV), gh.to(DEV), h.to(DEV)
    bd = gi_b.to(DEV) if two_inputs else None
    hnew = torch.empty_like(hd)
    save = torch.empty((N, 3 * H), dtype=torch.float32, device=DEV)
    call("mr_gru_gates_fwd", dtype_code(dtype), ptr(ad), ptr(bd), ptr(ghd), ptr(hd), ptr(hnew), ptr(save), N, H)
    assert _rel(hnew, hn64) < _tol(dtype, 2e-6, 1e-2)
    dgi = torch.empty((N, 3 * H), dtype=dtype, device=DEV)
    dgh = torch.empty((N, 3 * H), dtype=dtype, device=DEV)
    dh = torch.empty_like(hd)
    gd = gout.to(DEV)
    call("mr_gru_gates_bwd", dtype_code(dtype), ptr(gd), ptr(save), ptr(ghd), ptr(hd), ptr(dgi), ptr(dgh), ptr(dh), N, H)
    tol = _tol(dtype, 1e-5, 2e-2)
    assert _rel(dgi, a64.grad) < tol
    if two_inputs:
        assert _rel(dgi, b64.grad) < tol
    assert _rel(dgh, gh64.grad) < tol
    assert _rel(dh, h64.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,ld", [(32, 38, 40), (9, 38, 38), (130, 100, 104)])
def test_nll_step_vs_f64(dtype, N, C, ld):
    g = torch.Generator().manual_seed(N + C)
    logits = torch.zeros(N, ld)
    logits[:, :C] = torch.randn(N, C, generator=g) * 3
    logits[:, C:] = 1e4                       # padding columns must be ignored
    logits = logits.to(dtype)
    logits[1, :C] = logits[1, 0]              # an all-ties row: first index wins the arg-max
    target = torch.randint(0, C, (N, 3), generator=g)          # read with a stride (targets[:, t] view)
    mask = (torch.rand(N, generator=g) > 0.3).float()
    gl = torch.randn(N, generator=g)

    x64 = logits[:, :C].double().requires_grad_(True)
    lp64 = torch.log_softmax(x64, dim=1)
    loss64 = torch.nn.functional.nll_loss(lp64, target[:, 1], reduction='none') * mask.double()
    (loss64 * gl.double()).sum().backward()

    ld_, td, md = logits.to(DEV), target.to(DEV), mask.to(DEV)
    tcol = td[:, 1]
    lp = torch.empty((N, C), dtype=torch.float32, device=DEV)
    loss = torch.empty((N,), dtype=torch.float32, device=DEV)
    am = torch.empty((N,), dtype=torch.int64, device=DEV)
    call("mr_nll_step_fwd", dtype_code(dtype), ptr(ld_), ld, ptr(tcol), tcol.stride(0), ptr(md), ptr(lp), ptr(loss),
         ptr(am), N, C, 0, 0)
    assert float((lp.cpu().double() - lp64.detach()).abs().max()) < 2e-5
    assert float((loss.cpu().double() - loss64.detach()).abs().max()) < 2e-5
    assert torch.equal(am.cpu(), logits[:, :C].float().argmax(dim=1)) and int(am[1]) == 0
    # accumulate = 1 adds to the loss buffer; softmax_out = 1 writes probabilities (eval path)
    call("mr_nll_step_fwd", dtype_code(dtype), ptr(ld_), ld, ptr(tcol), tcol.stride(0), ptr(md), ptr(lp), ptr(loss),
         ptr(am), N, C, 1, 1)
    assert float((loss.cpu().double() - 2 * loss64.detach()).abs().max()) < 4e-5
    assert float((lp.cpu().double() - lp64.detach().exp()).abs().max()) < 2e-6
    call("mr_nll_step_fwd", dtype_code(dtype), ptr(ld_), ld, ptr(tcol), tcol.stride(0), ptr(md), ptr(lp), ptr(loss),
         ptr(am), N, C, 0, 0)
    d = torch.full((N, ld), 7.0, dtype=dtype, device=DEV)
    gld = gl.to(DEV)
    call("mr_nll_step_bwd", dtype_code(dtype), ptr(gld), ptr(lp), ptr(tcol), tcol.stride(0), ptr(md), ptr(d), ld, N, C)
    assert _rel(d[:, :C], x64.grad) < _tol(dtype, 1e-5, 1e-2)
    assert bool((d[:, C:] == 7.0).all())      # padding columns are the caller's (the Function pre-zeroes them)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_rows_vs_f64(dtype):
    N, V, ldo = 64, 38, 40
    g = torch.Generator().manual_seed(5)
    table = torch.eye(V) + 0.1 * torch.randn(V, V, generator=g)       # trainable, initialised to the identity (:190-191)
    idx = torch.randint(0, V, (N,), generator=g)
    idx[:8] = 3                                                       # many samples on one row: the atomics path
    gout = torch.randn(N, ldo, generator=g).to(dtype)
    t64 = table.double().requires_grad_(True)
    emb64 = torch.nn.functional.embedding(idx, t64)
    (emb64 * gout[:, :V].double()).sum().backward()
    td, idd = table.to(DEV), idx.to(DEV)
    out = torch.full((N, ldo), 9.0, dtype=dtype, device=DEV)
    call("mr_embed_rows_fwd", dtype_code(dtype), ptr(idd), ptr(td), ptr(out), N, V, V, ldo)
    assert _rel(out[:, :V], emb64) < _tol(dtype, 1e-7, 4e-3)
    assert bool((out[:, V:] == 0).all())                              # padded input columns of word_linear are zero
    dt = torch.full((V, V), 1.0, dtype=torch.float32, device=DEV)     # accumulated into
    gd = gout.to(DEV)
    call("mr_embed_rows_bwd", dtype_code(dtype), ptr(idd), ptr(gd), ptr(dt), N, V, V, ldo)
    assert _rel(dt - 1.0, t64.grad) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# Round-3 decode-loop kernels (strided operands, table gather, (N, Hd/64)-parallel backward, deferred denc)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,Tn,Hd,Ep,S", [(32, 32, 512, 552, 3), (5, 17, 64, 40, 2), (3, 64, 128, 72, 4)])
def test_attn_fwd2_bwd2_denc_vs_f64(dtype, N, Tn, Hd, Ep, S):
    """S steps sharing eproj / enc: hproj is a column slice (leading dimension 4*Hd) of a wider buffer, deproj / dv accumulate
    over the steps inside mr_attn_bwd2, the encoder gradient comes from ONE mr_attn_denc call; reference = float64 autograd of
    Attn.forward + the context bmm (decoders/attention_decoder.py:146-177, 207-209)."""
    g = torch.Generator().manual_seed(N * 100 + Tn + S)
    ld = 4 * Hd
    hc = (torch.randn(S, N, ld, generator=g) * 0.7).to(dtype)          # hproj = hc[..., :Hd]
    eproj = (torch.randn(N, Tn, Hd, generator=g) * 0.7).to(dtype)
    enc = torch.randn(N, Tn, Ep, generator=g).to(dtype)
    v = torch.randn(Hd, generator=g) * (Hd ** -0.5) * 4
    gctx = torch.randn(S, N, Ep, generator=g).to(dtype)
    gw = torch.randn(N, S, Tn, generator=g)                               # upstream gradient of the returned attention map

    h64 = hc[..., :Hd].double().requires_grad_(True)
    e64 = eproj.double().requires_grad_(True)
    c64 = enc.double().requires_grad_(True)
    v64 = v.double().requires_grad_(True)
    tot = 0
    w_ref, ctx_ref = [], []
    for s in range(S):
        w = torch.softmax(torch.tanh(h64[s].unsqueeze(1) + e64) @ v64, dim=1)
        ctx = torch.bmm(w.unsqueeze(1), c64).squeeze(1)
        w_ref.append(w.detach())
        ctx_ref.append(ctx.detach())
        tot = tot + (ctx * gctx[s].double()).sum() + (w * gw[:, s].double()).sum()
    tot.backward()

    hcd, ep, en, vd = hc.to(DEV), eproj.to(DEV), enc.to(DEV), v.to(DEV)
    W = torch.empty((S, N, Tn), dtype=torch.float32, device=DEV)
    CTX = torch.empty((S, N, Ep), dtype=dtype, device=DEV)
    dt = dtype_code(dtype)
    for s in range(S):
        call("mr_attn_fwd2", dt, ptr(hcd[s]), ld, ptr(ep), ptr(vd), ptr(en), ptr(W[s]), ptr(CTX[s]), N, Tn, Hd, Ep)
        assert _rel(W[s], w_ref[s]) < 1e-4 and _rel(CTX[s], ctx_ref[s]) < _tol(dtype, 1e-5, 1e-2)
    deproj = torch.zeros((N, Tn, Hd), dtype=torch.float32, device=DEV)
    dv = torch.zeros((Hd,), dtype=torch.float32, device=DEV)
    dhc = torch.full((S, N, ld), 3.0, dtype=dtype, device=DEV)             # only the [:Hd] slice may be written
    gc, gwd = gctx.to(DEV), gw.to(DEV).contiguous()
    for s in range(S - 1, -1, -1):
        call("mr_attn_bwd2", dt, ptr(gc[s]), ptr(gwd) + s * Tn * 4, S * Tn, ptr(hcd[s]), ld, ptr(ep), ptr(vd), ptr(en),
             ptr(W[s]), ptr(dhc[s]), ld, ptr(deproj), ptr(dv), N, Tn, Hd, Ep)
    denc = torch.empty((N, Tn, Ep), dtype=dtype, device=DEV)
    call("mr_attn_denc", dt, ptr(W), ptr(gc), ptr(denc), S, N, Tn, Ep)
    torch.cuda.synchronize()
    assert _rel(dhc[..., :Hd], h64.grad) < _tol(dtype, 2e-5, 2e-2)
    assert bool((dhc[..., Hd:] == 3.0).all())
    assert _rel(deproj, e64.grad) < 2e-5 and _rel(dv, v64.grad) < 2e-5
    assert _rel(denc, c64.grad) < _tol(dtype, 2e-5, 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,V", [(32, 512, 38), (7, 96, 5)])
def test_gru2_table_gather_strided_vs_f64(dtype, N, H, V):
    """mr_gru_fwd2 / mr_gru_bwd2: gi_a rows gathered from a [V, 3H] table through idx, gh a column slice (leading dimension
    4H, offset H) of the stacked projection, dh' = dh_a + dh_b + dh_c."""
    g = torch.Generator().manual_seed(N + H)
    table = torch.randn(V, 3 * H, generator=g).to(dtype)
    idx = torch.randint(0, V, (N,), generator=g)
    gi_b = (torch.randn(N, 3 * H, generator=g) * 0.5).to(dtype)
    hc = torch.randn(N, 4 * H, generator=g).to(dtype)                    # gh = hc[:, H:]
    h = torch.randn(N, H, generator=g).to(dtype)
    parts = [torch.randn(N, H, generator=g).to(dtype) for _ in range(3)]

    t64 = table.double().requires_grad_(True)
    b64 = gi_b.double().requires_grad_(True)
    hc64 = hc.double().requires_grad_(True)
    h64 = h.double().requires_grad_(True)
    gi = t64[idx] + b64
    gh = hc64[:, H:]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    nn_ = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    hn64 = (1 - z) * nn_ + z * h64
    gout = sum(p.double() for p in parts)
    (hn64 * gout).sum().backward()

    dt = dtype_code(dtype)
    es = 2 if dtype == torch.bfloat16 else 4
    td, idd, bd, hcd, hd = table.to(DEV), idx.to(DEV), gi_b.to(DEV), hc.to(DEV), h.to(DEV)
    hnew = torch.empty_like(hd)
    save = torch.empty((N, 3 * H), dtype=torch.float32, device=DEV)
    call("mr_gru_fwd2", dt, ptr(td), 3 * H, ptr(idd), ptr(bd), ptr(hcd) + H * es, 4 * H, ptr(hd), ptr(hnew), ptr(save), N, H)
    assert _rel(hnew, hn64) < _tol(dtype, 2e-6, 1e-2)
    dgi = torch.empty((N, 3 * H), dtype=dtype, device=DEV)
    dhc = torch.full((N, 4 * H), 5.0, dtype=dtype, device=DEV)
    dhp = torch.empty((N, H), dtype=dtype, device=DEV)
    pd = [p.to(DEV) for p in parts]
    call("mr_gru_bwd2", dt, ptr(pd[0]), ptr(pd[1]), ptr(pd[2]), ptr(save), ptr(hcd) + H * es, 4 * H, ptr(hd), ptr(dgi),
         ptr(dhc) + H * es, 4 * H, ptr(dhp), N, H)
    tol = _tol(dtype, 1e-5, 2e-2)
    assert _rel(dgi, b64.grad) < tol
    assert _rel(dhc[:, H:], hc64.grad[:, H:]) < tol and bool((dhc[:, :H] == 5.0).all())
    assert _rel(dhp, h64.grad) < tol
    # nullable parts: only dh_c
    call("mr_gru_bwd2", dt, 0, 0, ptr(pd[2]), ptr(save), ptr(hcd) + H * es, 4 * H, ptr(hd), ptr(dgi), ptr(dhc) + H * es,
         4 * H, ptr(dhp), N, H)
    for t in (t64, b64, hc64, h64):
        t.grad = None
    gi = t64[idx] + b64
    gh = hc64[:, H:]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H]); z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    nn_ = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    (((1 - z) * nn_ + z * h64) * parts[2].double()).sum().backward()
    assert _rel(dgi, b64.grad) < tol and _rel(dhp, h64.grad) < tol
    # gradient of the table gather: rows scatter-added with atomics
    dtab = torch.full((V, 3 * H), 1.0, dtype=torch.float32, device=DEV)
    call("mr_rows_scatter_add", dt, ptr(idd), ptr(dgi), 3 * H, ptr(dtab), N, V, 3 * H)
    assert _rel(dtab - 1.0, t64.grad) < _tol(dtype, 1e-5, 2e-2)


# ---------------------------------------------------------------------------------------------------------------------
# Round-4 decode-step fusions (csrc/gemm_skinny.hip): [GEMM(context) + GRU gates], [GEMM + GRU backward], [out + NLL] against
# f64 torch restatements of the same reference lines (attention_decoder.py:92-115,195-231: nn.GRUCell over
# cat([word, context]) split into the table gather + the context GEMM, nn.Linear out, log_softmax, NLLLoss, topk(1)).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,H,K,V", [(32, 512, 552, 38), (7, 96, 40, 5), (17, 48, 72, 3)])
def test_gemm_gru_fwd_bwd_fused_vs_f64(dtype, N, H, K, V):
    g = torch.Generator().manual_seed(N + H + K)
    vec = 4 if dtype == torch.float32 else 8
    assert K % vec == 0
    table = torch.randn(V, 3 * H, generator=g).to(dtype)
    idx = torch.randint(0, V, (N,), generator=g)
    ctx = torch.randn(N, K, generator=g).to(dtype)
    w_ic = (torch.randn(3 * H, K, generator=g) * K ** -0.5).to(dtype)
    hc = torch.randn(N, 4 * H, generator=g).to(dtype)                    # gh = hc[:, H:]
    h = torch.randn(N, H, generator=g).to(dtype)

    t64, c64, w64 = table.double(), ctx.double().requires_grad_(True), w_ic.double().requires_grad_(True)
    hc64, h64 = hc.double().requires_grad_(True), h.double().requires_grad_(True)
    gi = t64[idx] + c64 @ w64.t()
    gi.retain_grad()
    gh = hc64[:, H:]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    nn_ = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    hn64 = (1 - z) * nn_ + z * h64

    dt = dtype_code(dtype)
    es = 2 if dtype == torch.bfloat16 else 4
    td, idd, cd, wd, hcd, hd = (t.to(DEV) for t in (table, idx, ctx, w_ic, hc, h))
    hnew = torch.empty_like(hd)
    save = torch.empty((N, 3 * H), dtype=torch.float32, device=DEV)
    call("mr_gemm_gru_fwd", dt, ptr(cd), K, ptr(wd), K, ptr(td), 3 * H, ptr(idd), ptr(hcd) + H * es, 4 * H, ptr(hd), ptr(hnew),
         ptr(save), N, H, K)
    assert _rel(hnew, hn64) < _tol(dtype, 1e-5, 1e-2)
    assert _rel(save[:, :H], r) < _tol(dtype, 1e-5, 1e-2) and _rel(save[:, 2 * H:], nn_) < _tol(dtype, 1e-5, 2e-2)
    # ... equals the unfused pair up to the bf16 rounding of the intermediate gi_c it no longer stores
    gic = torch.empty((N, 3 * H), dtype=dtype, device=DEV)
    call("mr_gemm_nt", dt, ptr(cd), K, ptr(wd), K, ptr(gic), 3 * H, 0, 0, N, 3 * H, K)
    hnew2, save2 = torch.empty_like(hd), torch.empty_like(save)
    call("mr_gru_fwd2", dt, ptr(td), 3 * H, ptr(idd), ptr(gic), ptr(hcd) + H * es, 4 * H, ptr(hd), ptr(hnew2), ptr(save2), N, H)
    assert _rel(hnew, hnew2) < _tol(dtype, 2e-6, 1e-2)

    # backward: dh_a = dhc_next @ w_t^T (w_t = [H, Kb] image of a [Kb, H] weight), then the GRU backward of THIS step
    Kb = 4 * H
    dhc_next = (torch.randn(N, Kb, generator=g) * 0.3).to(dtype)
    w_t = (torch.randn(H, Kb, generator=g) * Kb ** -0.5).to(dtype)
    dh_b, dh_c = torch.randn(N, H, generator=g).to(dtype), torch.randn(N, H, generator=g).to(dtype)
    gout = dhc_next.double() @ w_t.double().t() + dh_b.double() + dh_c.double()
    (hn64 * gout).sum().backward()
    dn, wt, bb, bc = (t.to(DEV) for t in (dhc_next, w_t, dh_b, dh_c))
    save64 = torch.cat([r, z, nn_], 1).float().to(DEV)                   # exact gates: isolates the backward
    dgi = torch.empty((N, 3 * H), dtype=dtype, device=DEV)
    dhc = torch.full((N, 4 * H), 5.0, dtype=dtype, device=DEV)
    dhp = bb.clone()                                                     # dh_prev aliases dh_b, as in the decode loop
    call("mr_gemm_gru_bwd", dt, ptr(dn), Kb, ptr(wt), Kb, ptr(dhp), ptr(bc), ptr(save64), ptr(hcd) + H * es, 4 * H, ptr(hd),
         ptr(dgi), ptr(dhc) + H * es, 4 * H, ptr(dhp), N, H, Kb)
    tol = _tol(dtype, 2e-5, 2e-2)
    assert _rel(dgi, gi.grad) < tol
    assert _rel(dhc[:, H:], hc64.grad[:, H:]) < tol and bool((dhc[:, :H] == 5.0).all())
    assert _rel(dhp, h64.grad) < tol
    # nullable dh_b / dh_c
    call("mr_gemm_gru_bwd", dt, ptr(dn), Kb, ptr(wt), Kb, 0, 0, ptr(save64), ptr(hcd) + H * es, 4 * H, ptr(hd), ptr(dgi),
         ptr(dhc) + H * es, 4 * H, ptr(dhp), N, H, Kb)
    for t in (c64, w64, hc64, h64):
        t.grad = None
    gi2 = t64[idx] + c64 @ w64.t()
    gh = hc64[:, H:]
    r2 = torch.sigmoid(gi2[:, :H] + gh[:, :H]); z2 = torch.sigmoid(gi2[:, H:2 * H] + gh[:, H:2 * H])
    n2 = torch.tanh(gi2[:, 2 * H:] + r2 * gh[:, 2 * H:])
    (((1 - z2) * n2 + z2 * h64) * (dhc_next.double() @ w_t.double().t())).sum().backward()
    assert _rel(dhp, h64.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,K", [(32, 38, 512), (5, 7, 64), (3, 200, 96)])
@pytest.mark.parametrize("feed", [0, 1, None])
def test_out_nll_fused_vs_f64(dtype, N, C, K, feed):
    g = torch.Generator().manual_seed(N + C + K)
    h = torch.randn(N, K, generator=g).to(dtype)
    W = (torch.randn(C, K, generator=g) * K ** -0.5 * 3).to(dtype)
    b = torch.randn(C, generator=g)
    target = torch.randint(0, C, (N,), generator=g)
    mask = (torch.rand(N, generator=g) > 0.3).float()
    logits = h.double() @ W.double().t() + b.double()
    lp64 = torch.log_softmax(logits, 1)
    loss64 = -lp64[torch.arange(N), target] * mask.double()
    am64 = logits.argmax(1)
    top2 = logits.topk(2, 1).values
    clear = (top2[:, 0] - top2[:, 1]) > _tol(dtype, 1e-4, 5e-2)          # rows whose arg-max survives the compute dtype

    dt = dtype_code(dtype)
    hd, Wd, bd, td, md = h.to(DEV), W.to(DEV), b.to(DEV), target.to(DEV), mask.to(DEV)
    lp = torch.empty((N, C), dtype=torch.float32, device=DEV)
    loss = torch.full((N,), 2.0, dtype=torch.float32, device=DEV)
    am = torch.empty((N,), dtype=torch.int64, device=DEV)
    flag = torch.tensor([feed or 0], dtype=torch.int32, device=DEV)
    fidx = torch.full((N,), -1, dtype=torch.int64, device=DEV)
    call("mr_out_nll_fwd", dt, ptr(hd), K, ptr(Wd), K, ptr(bd), ptr(td), 1, ptr(md), ptr(lp), ptr(loss), ptr(am),
         0 if feed is None else ptr(flag), 0 if feed is None else ptr(fidx), N, C, K, 1)
    assert _rel(lp, lp64) < _tol(dtype, 2e-6, 2e-6) * 10                 # f32 accumulation of already rounded operands
    assert _rel(loss - 2.0, loss64) < 1e-5                               # accumulate = 1
    assert bool((am.cpu()[clear] == am64[clear]).all())
    if feed is None:
        assert bool((fidx == -1).all())
    elif feed:
        assert bool((fidx.cpu() == target).all())
    else:
        assert bool((fidx == am).all())
