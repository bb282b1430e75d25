"""CPU pins of the larger oracle models against the golden vectors produced by the unmodified reference
(oracle/gen_golden.py): ResNet50-dilated+PPM+2D-CTC and ResNet50-FPN+attention decoder."""
import os

import pytest
import torch

from oracle.fpn_attention import FPNAttentionOracle
from oracle.res50ppm import Res50PPM2DCTCOracle


def _check_state(model, golden):
    state = model.state_dict()
    assert list(state.keys()) == golden['state_keys']
    for k, v in state.items():
        s, a = golden['state_checksums'][k]
        assert abs(float(v.double().sum()) - s) <= 1e-9 * max(1.0, a), k


def test_res50ppm_oracle_matches_reference_golden(golden_dir):
    torch.set_num_threads(4)
    g = torch.load(os.path.join(golden_dir, "res50ppm_golden.pt"), weights_only=False)
    torch.manual_seed(g['weight_seed'])
    m = Res50PPM2DCTCOracle(dropout=0.0).train()
    _check_state(m, g)
    assert len(m.state_dict()) == 377 and sum(p.numel() for p in m.parameters()) == 52746727
    b = g['batch']
    loss, pred = m(b['image'], targets=b['label'], lengths=b['length'].long(), train=True)
    assert float((loss - g['train_loss']).abs().max()) < 1e-4 * float(g['train_loss'].abs().max())
    loss.mean().backward()
    for k, p in m.named_parameters():
        gs = g['grad_stats'][k]
        if gs is None:
            assert p.grad is None, k
        elif gs[0] > 1e-6:
            assert abs(float(p.grad.double().norm()) - gs[0]) <= 1e-3 * gs[0], k


def test_fpn_attention_oracle_matches_reference_golden(golden_dir):
    torch.set_num_threads(4)
    g = torch.load(os.path.join(golden_dir, "fpn_attention_golden.pt"), weights_only=False)
    torch.manual_seed(g['weight_seed'])
    m = FPNAttentionOracle().train()
    _check_state(m, g)
    assert len(m.state_dict()) == 339 + 63
    b = g['batch']
    loss, att = m(b['image'], targets=b['label'], lengths=b['length'].long(), train=True)
    assert float((loss - g['train_loss']).abs().max()) < 1e-4 * float(g['train_loss'].abs().max())
    assert float((att - g['train_attention']).abs().max()) < 1e-5


def test_product_mirrors_share_the_reference_state_dicts(golden_dir):
    """The HIP mirrors (constructed on CPU, no forward) have the reference's keys, shapes and seeded default init."""
    from megreader_amd.backbones import resnet50dilated_ppm, Resnet50FPN
    from megreader_amd.decoders import CTCDecoder2D, AttentionDecoder

    class M(torch.nn.Module):
        def __init__(self, b, d):
            super().__init__()
            self.backbone, self.decoder = b, d

    for name, make in (("res50ppm_golden.pt", lambda: M(resnet50dilated_ppm(), CTCDecoder2D(in_channels=256))),
                       ("fpn_attention_golden.pt",
                        lambda: M(Resnet50FPN(resnet_pretrained=False), AttentionDecoder(in_channels=256)))):
        g = torch.load(os.path.join(golden_dir, name), weights_only=False)
        torch.manual_seed(g['weight_seed'])
        m = make()
        _check_state(m, g)
        for k, v in m.state_dict().items():
            assert tuple(v.shape) == g['state_shapes'][k], k


def test_seg_detector_oracle_and_mirror_vs_reference():
    """DB head + loss: the oracle restatement is bit-identical to the unmodified reference SegDetector / L1BalanceCELoss
    on CPU (same seeded init); the HIP mirror has the same state_dict keys, shapes and initial values, and its restated
    loss module equals the reference's."""
    from oracle import refimport
    if not refimport.available():
        pytest.skip("reference tree not present")
    refimport.import_reference()
    from decoders.seg_detector import SegDetector as RefSeg
    from decoders.seg_detector_loss import L1BalanceCELoss as RefLoss
    from megreader_amd.decoders import L1BalanceCELoss, SegDetector
    from megreader_amd.synthetic import detection_batch
    from oracle.seg_detector import SegDetectorOracle, l1_balance_ce_loss
    chans = [16, 32, 64, 128]
    torch.manual_seed(3)
    ref = RefSeg(in_channels=chans, inner_channels=64, k=50, adaptive=True)
    torch.manual_seed(3)
    ora = SegDetectorOracle(in_channels=chans, inner_channels=64, k=50, adaptive=True)
    torch.manual_seed(3)
    ours = SegDetector(in_channels=chans, inner_channels=64, k=50, adaptive=True)
    rs = ref.state_dict()
    assert list(rs.keys()) == list(ora.state_dict().keys()) == list(ours.state_dict().keys())
    for k, v in rs.items():
        assert torch.equal(v, ora.state_dict()[k]), k
        assert v.shape == ours.state_dict()[k].shape and torch.equal(v, ours.state_dict()[k]), k
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(2, c, 64 // s, 64 // s, generator=g) for c, s in zip(chans, (1, 2, 4, 8))]
    ref.train()
    ora.train()
    pr, po = ref(feats), ora(feats)
    for k in pr:
        assert torch.equal(pr[k], po[k]), k
    batch = detection_batch(2, 256, seed=1, boxes=3)
    lr_, _ = RefLoss()(pr, batch)
    lm_, _ = L1BalanceCELoss()(pr, batch)
    assert torch.equal(lr_, lm_) and torch.equal(lr_, l1_balance_ce_loss(po, batch))


@pytest.mark.parametrize("reduce_func,loss_func", [("conv", "pytorch"), ("pooling", "custom")])
def test_crnn_decoder_variants_oracle_vs_reference(reduce_func, loss_func):
    """CRNNDecoder(need_reduce=True, reduce_func=...) and loss_func != 'pytorch' (decoders/crnn.py:36-50): the oracle
    equals the unmodified reference on CPU -- state_dict keys and seeded init, training loss and log-probabilities
    (the reference's python CTCLoss, decoders/ctc_loss.py, against its restatement as per-sample nll / length: 1e-9),
    eval softmax -- and the HIP mirror has the same state_dict."""
    from oracle import refimport
    if not refimport.available():
        pytest.skip("reference tree not present")
    refimport.import_reference()
    from decoders.crnn import CRNNDecoder as RefDecoder
    from megreader_amd.decoders import CRNNDecoder
    from megreader_amd.charsets import DefaultCharset
    from oracle.crnn import CRNNDecoderOracle
    # 'pooling' keeps the channel count while the LSTM expects inner_channels: the reference only works with equal counts
    cin = 48 if reduce_func == "conv" else 32
    kw = dict(inner_channels=32, in_channels=cin, need_reduce=True, reduce_func=reduce_func, loss_func=loss_func)
    torch.manual_seed(5)
    ref = RefDecoder(**kw)
    torch.manual_seed(5)
    ora = CRNNDecoderOracle(num_classes=len(DefaultCharset()), **kw)
    torch.manual_seed(5)
    ours = CRNNDecoder(**kw)
    assert list(ref.state_dict()) == list(ora.state_dict()) == list(ours.state_dict())
    for k, v in ref.state_dict().items():
        assert torch.equal(v, ora.state_dict()[k]) and torch.equal(v, ours.state_dict()[k]), k
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(3, cin, 8, 20, generator=g)
    labels = torch.randint(2, 38, (3, 6), generator=g, dtype=torch.int32)
    lengths = torch.tensor([6, 3, 4], dtype=torch.int32)
    ref.train(), ora.train()
    lr_, pr = ref(feat, targets=labels, lengths=lengths, train=True)
    lo, po = ora(feat, targets=labels, lengths=lengths, train=True)
    assert torch.equal(pr, po)
    assert lr_.shape == lo.shape and (lr_.detach().double() - lo.detach()).abs().max() < 1e-9 * max(1.0, float(lo.detach().abs().max()))
    ref.eval(), ora.eval()
    assert torch.equal(ref(feat), ora(feat))


def test_ctc_decoder_oracle_and_mirror_vs_reference():
    """decoders.CTCDecoder (decoders/ctc_decoder.py:13-66): the oracle restatement is bit-identical to the unmodified reference
    module on CPU (state_dict keys + seeded init, training loss / log-probabilities / every gradient, eval softmax) and the HIP
    mirror has the same state_dict."""
    from oracle import refimport
    if not refimport.available():
        pytest.skip("reference tree not present")
    refimport.import_reference()
    from decoders.ctc_decoder import CTCDecoder as RefDecoder
    from megreader_amd.decoders import CTCDecoder
    from megreader_amd.charsets import DefaultCharset
    from oracle.ctc_decoder import CTCDecoderOracle
    torch.manual_seed(7)
    ref = RefDecoder(in_channels=24, inner_channels=32)
    torch.manual_seed(7)
    ora = CTCDecoderOracle(24, num_classes=len(DefaultCharset()), inner_channels=32)
    torch.manual_seed(7)
    ours = CTCDecoder(in_channels=24, inner_channels=32)
    assert list(ref.state_dict()) == list(ora.state_dict()) == list(ours.state_dict())
    for k, v in ref.state_dict().items():
        assert torch.equal(v, ora.state_dict()[k]) and torch.equal(v, ours.state_dict()[k]), k
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(3, 24, 16, 64, generator=g)
    labels = torch.randint(1, 38, (3, 8), generator=g, dtype=torch.long)
    lengths = torch.tensor([8, 3, 5], dtype=torch.long)
    ref.train(), ora.train()
    lr_, pr = ref(feat, targets=labels, lengths=lengths, train=True)
    lo, po = ora(feat, targets=labels, lengths=lengths, train=True)
    assert torch.equal(pr, po) and torch.equal(lr_, lo) and lr_.dim() == 0 and pr.shape == (3, 38, 32)
    lr_.backward()
    lo.backward()
    for (k, p), (_, q) in zip(ref.named_parameters(), ora.named_parameters()):
        assert torch.equal(p.grad, q.grad), k
    ref.eval(), ora.eval()
    assert torch.equal(ref(feat), ora(feat))
