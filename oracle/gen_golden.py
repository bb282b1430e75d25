"""Generate tests/golden/*.pt by executing the UNMODIFIED reference on CPU (this container only).

    python oracle/gen_golden.py

For every fixture the script also runs the oracle restatement on the same weights/inputs and refuses to write
the file unless the two agree bit-for-bit (they execute the same torch CPU kernels) -- that is the pin of the
oracle to the reference.  Weights are too large to commit (33 MB), so a fixture stores the RNG seed that
reproduces them plus per-tensor checksums; inputs and outputs are stored in full.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import refimport  # noqa: E402
from oracle.crnn import CRNNOracle, synthetic_batch  # noqa: E402
from oracle.decode import greedy_decode  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")
WEIGHT_SEED = 1234


def checksums(state):
    return {k: (float(v.double().sum()), float(v.double().abs().sum())) for k, v in state.items()}


def crnn_fixture():
    torch.set_num_threads(4)
    sm = refimport.import_reference()
    from concern.charsets import EnglishCharset
    from structure.representers.ctc_representer import CTCRepresenter
    charset = EnglishCharset()
    args = {'backbone': 'crnn_backbone', 'decoder': 'CRNNDecoder',
            'decoder_args': {'in_channels': 512, 'inner_channels': 256, 'need_reduce': False, 'charset': charset}}

    torch.manual_seed(WEIGHT_SEED)
    ref = sm.SequenceRecognitionModel(args, torch.device('cpu'))  # structure/model.py:160-181 (DataParallel on CPU)
    torch.manual_seed(WEIGHT_SEED)
    ora = CRNNOracle(num_classes=len(charset))
    ref_state = {k.replace('model.module.', ''): v for k, v in ref.state_dict().items()}
    assert list(ref_state.keys()) == list(ora.state_dict().keys()), "state_dict keys differ"
    for k, v in ora.state_dict().items():
        assert torch.equal(v, ref_state[k]), "seeded init differs at %s" % k

    batch = synthetic_batch(3, height=32, width=64, seed=7)
    out = {'weight_seed': WEIGHT_SEED, 'batch': batch, 'state_checksums': checksums(ref_state),
           'state_keys': list(ref_state.keys()), 'state_shapes': {k: tuple(v.shape) for k, v in ref_state.items()}}

    # ---- training forward / backward through the reference wrapper (batch dict in, (loss, pred) out)
    ref.train()
    ora.train()
    loss_r, pred_r = ref.forward(dict(batch), training=True)
    loss_r.mean().backward()
    loss_o, pred_o = ora(batch['image'], targets=batch['label'], lengths=batch['length'].long(), train=True)
    loss_o.mean().backward()
    assert torch.equal(loss_r, loss_o) and torch.equal(pred_r, pred_o), "oracle forward != reference"
    grads_r = {k.replace('model.module.', ''): p.grad for k, p in ref.named_parameters()}
    for k, p in ora.named_parameters():
        assert torch.equal(p.grad, grads_r[k]), "oracle grad != reference at %s" % k
    out['train_loss'] = loss_r.detach().clone()
    out['train_log_probs'] = pred_r.detach().clone()
    out['grad_stats'] = {k: (float(g.double().norm()), g.flatten()[:8].clone()) for k, g in grads_r.items()}
    # BN running stats after one training forward
    out['bn_after'] = {k.replace('model.module.', ''): v.clone() for k, v in ref.state_dict().items()
                       if 'running' in k}

    # ---- three Adam steps (experiments/recognition/crnn.yaml:82-89: Adam, lr 1e-3) -- loss trajectory
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3)
    opt_o = torch.optim.Adam(ora.parameters(), lr=1e-3)
    traj_r, traj_o = [], []
    for _ in range(3):
        opt_r.zero_grad()
        l, _p = ref.forward(dict(batch), training=True)
        l = l.mean()
        l.backward()
        opt_r.step()
        traj_r.append(float(l))
        opt_o.zero_grad()
        l2, _p = ora(batch['image'], targets=batch['label'], lengths=batch['length'].long(), train=True)
        l2 = l2.mean()
        l2.backward()
        opt_o.step()
        traj_o.append(float(l2))
    assert traj_r == traj_o, "oracle Adam trajectory != reference: %s vs %s" % (traj_r, traj_o)
    out['adam_losses'] = traj_r

    # ---- eval forward + greedy decode (structure/representers/ctc_representer.py:20-34)
    ref.eval()
    ora.eval()
    with torch.no_grad():
        ev_r = ref.forward(dict(batch), training=False)
        ev_o = ora(batch['image'], train=False)
    assert torch.equal(ev_r, ev_o), "oracle eval != reference"
    rep = CTCRepresenter(charset=charset)
    strings = rep.represent(batch, ev_r)
    dec = greedy_decode(ev_r.numpy())
    assert [charset.label_to_string(d) for d in dec] == [s['pred_string'] for s in strings], "decode differs"
    out['eval_pred'] = ev_r.clone()
    out['eval_decode'] = torch.from_numpy(dec)
    out['eval_strings'] = [s['pred_string'] for s in strings]
    out['label_strings'] = [s['label_string'] for s in strings]
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, "crnn_golden.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; loss", float(loss_r), "adam", traj_r)


def ctc2d_fixture():
    """Pin oracle/ctc2d.py to the reference's own pure-python 2D-CTC (decoders/ctc_loss2d.py:86-154), which takes
    log-mask and log-classify separately and is valid while NLL <~ 80 (SURVEY.md §2b notes)."""
    import warnings
    import numpy as np
    refimport.import_reference()
    from decoders.ctc_loss2d import CTCLoss2D
    from oracle.ctc2d import ctc2d, synthetic_lp
    T, H, N, C, S = 14, 4, 5, 12, 8   # the python class indexes count_computable[t]: needs T <= 2S+1
    rng = np.random.RandomState(11)
    tl = rng.randint(1, 5, size=N).astype(np.int64)
    tg = np.zeros((N, S), dtype=np.int64)
    for i, L in enumerate(tl):
        tg[i, :L] = rng.randint(1, C, size=L)
    tg[1, 1] = tg[1, 0]
    tl[1] = max(tl[1], 2)
    il = np.full(N, T, dtype=np.int64)
    lp, log_mask, log_cls = synthetic_lp(T, H, N, C, seed=4, peak=4.0, targets=tg, target_lengths=tl)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = CTCLoss2D(reduction='none')
        nll_ref = ref(torch.from_numpy(log_mask), torch.from_numpy(log_cls), torch.from_numpy(tg),
                      torch.from_numpy(il), torch.from_numpy(tl))
        # ---- gradient pin (VERDICT r2 "missing" 1): the python class is plain differentiable torch, so autograd of it
        # w.r.t. the log-classify input is  -occupancy[t,h,n,c]  (posterior probability that the path is at pixel
        # (t,h) emitting class c) and w.r.t. log-mask  -sum_c occupancy.  Run in float64 (module buffers converted)
        # so the pin is not limited by f32 round-off.  ctc2d_cuda_kernel.cu:498-515 returns exp(lp) - occupancy on
        # the classes of the extended target whose collected log(alpha*beta) is finite, and 0 elsewhere.
        ref64 = CTCLoss2D(reduction='none').double()
        lm = torch.from_numpy(log_mask).double().requires_grad_()
        lc = torch.from_numpy(log_cls).double().requires_grad_()
        nll64 = ref64(lm, lc, torch.from_numpy(tg), torch.from_numpy(il), torch.from_numpy(tl))
        nll64.sum().backward()
    occ_ref = (-lc.grad).numpy()
    occ_mask_ref = (-lm.grad).numpy()
    o = ctc2d(lp, tg, il, tl)
    diff = float(np.abs(nll_ref.numpy() - o['nll']).max())
    assert float(o['nll'].max()) < 60 and diff < 2e-5, (o['nll'], nll_ref, diff)
    nz = o['grad'] != 0
    recon = np.where(nz, np.exp(lp.astype(np.float64)) - o['grad'], 0.0)
    gdiff = float(np.abs(recon - occ_ref).max())
    assert gdiff < 1e-6, "oracle gradient convention != exp(lp) - occupancy of the reference: %g" % gdiff
    assert float(np.abs(occ_ref[~nz]).max()) == 0.0, "reference occupancy is non-zero where the oracle returns 0"
    assert float(np.abs(recon.sum(axis=3) - occ_mask_ref).max()) < 1e-6
    out = {'lp': torch.from_numpy(lp), 'targets': torch.from_numpy(tg), 'input_lengths': torch.from_numpy(il),
           'target_lengths': torch.from_numpy(tl), 'nll_reference_python': nll_ref.float(),
           'occupancy_reference_python': torch.from_numpy(occ_ref),
           'mask_occupancy_reference_python': torch.from_numpy(occ_mask_ref),
           'nll_oracle': torch.from_numpy(o['nll']), 'grad_oracle': torch.from_numpy(o['grad'])}
    path = os.path.join(GOLDEN, "ctc2d_golden.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; max |oracle - reference python| nll", diff, "occupancy", gdiff)


def res50ppm_fixture():
    """ResNet50-dilated + PPM + CTCDecoder2D (experiments/recognition/res50-ppm-2d-ctc.yaml) executed by the
    UNMODIFIED reference modules on CPU; the only substitution is the CUDA-only `ops.ctc_loss_2d`, replaced by the
    float64 oracle op (oracle/res50ppm.py:OracleCTC2D).  Dropout2d(0.1) is set to p = 0 on both sides (its RNG
    stream cannot be reproduced by another implementation)."""
    import types
    from oracle.res50ppm import Res50PPM2DCTCOracle, synthetic_batch_2d, oracle_ctc_loss_2d
    torch.set_num_threads(4)
    ops = types.ModuleType("ops")
    ops.ctc_loss_2d = oracle_ctc_loss_2d
    sm = refimport.import_reference(ops_module=ops)
    from concern.charsets import EnglishCharset
    charset = EnglishCharset()
    args = {'backbone': 'resnet50dilated_ppm', 'decoder': 'CTCDecoder2D',
            'decoder_args': {'in_channels': 256, 'charset': charset}}
    torch.manual_seed(WEIGHT_SEED)
    ref = sm.SequenceRecognitionModel(args, torch.device('cpu'))
    torch.manual_seed(WEIGHT_SEED)
    ora = Res50PPM2DCTCOracle(num_classes=len(charset))
    ref_state = {k.replace('model.module.', ''): v for k, v in ref.state_dict().items()}
    assert list(ref_state.keys()) == list(ora.state_dict().keys()), "state_dict keys differ"
    for k, v in ora.state_dict().items():
        assert torch.equal(v, ref_state[k]), "seeded init differs at %s" % k
    for m in list(ref.modules()) + list(ora.modules()):
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    batch = synthetic_batch_2d(2, 32, 64, seed=3)
    out = {'weight_seed': WEIGHT_SEED, 'batch': batch, 'state_checksums': checksums(ref_state),
           'state_keys': list(ref_state.keys()), 'state_shapes': {k: tuple(v.shape) for k, v in ref_state.items()}}
    ref.train()
    ora.train()
    loss_r, pred_r = ref.forward(dict(batch), training=True)
    loss_r.mean().backward()
    loss_o, pred_o = ora(batch['image'], targets=batch['label'], lengths=batch['length'].long(), train=True)
    loss_o.mean().backward()
    assert torch.equal(loss_r, loss_o) and torch.equal(pred_r, pred_o), "oracle forward != reference"
    grads_r = {k.replace('model.module.', ''): p.grad for k, p in ref.named_parameters()}
    out['grad_stats'] = {}
    for k, p in ora.named_parameters():
        if p.grad is None:
            assert grads_r[k] is None, k
            out['grad_stats'][k] = None
            continue
        assert torch.equal(p.grad, grads_r[k]), "oracle grad != reference at %s" % k
        out['grad_stats'][k] = (float(p.grad.double().norm()), p.grad.flatten()[:8].clone())
    out['train_loss'] = loss_r.detach().clone()
    out['train_pred'] = pred_r.detach().clone()
    ref.eval()
    ora.eval()
    with torch.no_grad():
        cls_r, mask_r = ref.forward(dict(batch), training=False)
        cls_o, mask_o = ora(batch['image'], train=False)
    assert torch.equal(cls_r, cls_o) and torch.equal(mask_r, mask_o), "oracle eval != reference"
    out['eval_classify'] = cls_r.clone()
    out['eval_mask'] = mask_r.clone()
    path = os.path.join(GOLDEN, "res50ppm_golden.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; loss", loss_r.tolist())


def fpn_attention_fixture():
    """ResNet50-FPN + AttentionDecoder (experiments/recognition/fpn50-attention-decoder.yaml with
    backbone_args resnet_pretrained=False -- no network -- and gt_as_output=True for determinism), all of it the
    unmodified reference on CPU.  Input 64x256 (the decoder's conv encoder requires it)."""
    from oracle.fpn_attention import FPNAttentionOracle
    from oracle.crnn import synthetic_batch
    torch.set_num_threads(8)
    sm = refimport.import_reference()
    from concern.charsets import EnglishCharset
    charset = EnglishCharset()
    args = {'backbone': 'Resnet50FPN', 'backbone_args': {'resnet_pretrained': False}, 'decoder': 'AttentionDecoder',
            'decoder_args': {'in_channels': 256, 'charset': charset, 'gt_as_output': True}}
    torch.manual_seed(WEIGHT_SEED)
    ref = sm.SequenceRecognitionModel(args, torch.device('cpu'))
    torch.manual_seed(WEIGHT_SEED)
    ora = FPNAttentionOracle(len(charset))
    ref_state = {k.replace('model.module.', ''): v for k, v in ref.state_dict().items()}
    assert list(ref_state.keys()) == list(ora.state_dict().keys()), \
        [(a, b) for a, b in zip(ref_state.keys(), ora.state_dict().keys()) if a != b][:5]
    for k, v in ora.state_dict().items():
        assert torch.equal(v, ref_state[k]), "seeded init differs at %s" % k
    batch = synthetic_batch(2, 64, 256, seed=5)
    out = {'weight_seed': WEIGHT_SEED, 'batch': batch, 'state_checksums': checksums(ref_state),
           'state_keys': list(ref_state.keys()), 'state_shapes': {k: tuple(v.shape) for k, v in ref_state.items()}}
    ref.train()
    ora.train()
    loss_r, att_r = ref.forward(dict(batch), training=True)
    loss_r.mean().backward()
    loss_o, att_o = ora(batch['image'], targets=batch['label'], lengths=batch['length'].long(), train=True)
    loss_o.mean().backward()
    assert torch.allclose(loss_r, loss_o, rtol=1e-6, atol=1e-6) and torch.allclose(att_r, att_o, atol=1e-6), \
        (loss_r, loss_o)
    grads_r = {k.replace('model.module.', ''): p.grad for k, p in ref.named_parameters()}
    out['grad_stats'] = {}
    for k, p in ora.named_parameters():
        if p.grad is None:
            assert grads_r[k] is None, k
            out['grad_stats'][k] = None
            continue
        gr = grads_r[k]
        assert float((p.grad - gr).abs().max()) <= 1e-5 * max(1e-6, float(gr.abs().max())) + 1e-8, k
        out['grad_stats'][k] = (float(gr.double().norm()), gr.flatten()[:8].clone())
    out['train_loss'] = loss_r.detach().clone()
    out['train_attention'] = att_r.detach().clone()
    ref.eval()
    ora.eval()
    with torch.no_grad():
        pred_r = ref.forward(dict(batch), training=False)
        pred_o = ora(batch['image'], train=False)
    assert torch.equal(pred_r, pred_o)
    out['eval_pred'] = pred_r.clone()
    path = os.path.join(GOLDEN, "fpn_attention_golden.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; loss", loss_r.tolist())


def deformable_resnet_fixture():
    """deformable_resnet50 (backbones/resnet.py:295-309, the DB detector's backbone) executed by the unmodified
    reference graph code on CPU, with `assets.ops.dcn.ModulatedDeformConv` (CUDA-only) substituted by the float64-
    capable oracle module (oracle/dcn.py).  Exercises the stride-2 / stride-1-offset-map quirk (Q10) end to end."""
    import types
    from oracle.dcn import OracleModulatedDeformConv, perturb_offset_convs
    from oracle.res50ppm import _Res50Dilated
    torch.set_num_threads(8)
    dcn = types.ModuleType("assets.ops.dcn")
    dcn.ModulatedDeformConv = OracleModulatedDeformConv
    refimport.import_reference(dcn_module=dcn)
    import backbones
    torch.manual_seed(WEIGHT_SEED)
    ref = backbones.deformable_resnet50(pretrained=False)
    torch.manual_seed(WEIGHT_SEED)
    ora = _Res50Dilated(dilate=False, dcn=True)
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys())
    for k, v in ora.state_dict().items():
        assert torch.equal(v, ref.state_dict()[k]), "seeded init differs at %s" % k
    init_checksums = checksums(ref.state_dict())
    perturb_offset_convs(ref)
    perturb_offset_convs(ora)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(2, 3, 96, 96, generator=g)   # 3x3 maps at stride 32: keeps batch-statistics BN well conditioned
    ref.train()
    ora.train()
    fr, fo = ref(x), ora(x)
    for a, b in zip(fr, fo):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
    loss_r = sum(f.square().mean() for f in fr)
    loss_r.backward()
    sum(f.square().mean() for f in fo).backward()
    out = {'weight_seed': WEIGHT_SEED, 'x': x, 'state_checksums': init_checksums,
           'state_keys': list(ref.state_dict().keys()), 'features': [f.detach().clone() for f in fr[2:]], 'feature_norms': [float(f.norm()) for f in fr],
           'loss': float(loss_r), 'grad_stats': {}}
    go = dict(ora.named_parameters())
    for k, p in ref.named_parameters():
        if p.grad is None:
            out['grad_stats'][k] = None
            continue
        assert float((p.grad - go[k].grad).abs().max()) <= 1e-5 * max(1e-6, float(p.grad.abs().max())) + 1e-9, k
        out['grad_stats'][k] = (float(p.grad.double().norm()), p.grad.flatten()[:8].clone())
    path = os.path.join(GOLDEN, "deformable_resnet50_golden.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; loss", float(loss_r))


if __name__ == "__main__":
    if not refimport.available():
        raise SystemExit("reference not available: golden vectors can only be regenerated in the build container")
    os.chdir("/tmp")
    which = sys.argv[1:] or ["crnn", "ctc2d", "res50ppm", "fpn_attention", "deformable_resnet"]
    if "crnn" in which:
        crnn_fixture()
    if "ctc2d" in which:
        ctc2d_fixture()
    if "res50ppm" in which:
        res50ppm_fixture()
    if "fpn_attention" in which:
        fpn_attention_fixture()
    if "deformable_resnet" in which:
        deformable_resnet_fixture()
