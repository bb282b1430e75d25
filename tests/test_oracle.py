"""CPU tests that pin the oracle: against the committed golden vectors (produced by the unmodified reference,
oracle/gen_golden.py), against torch's own CTC (what the reference calls), and -- when the reference tree is
present (build container only) -- against the reference modules executed live."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import refimport
from oracle.crnn import CRNNOracle, synthetic_batch, train_step
from oracle.ctc import ctc_1d
from oracle.decode import greedy_decode


@pytest.fixture(scope="module")
def golden(golden_dir):
    return torch.load(os.path.join(golden_dir, "crnn_golden.pt"), weights_only=False)


@pytest.fixture(scope="module")
def oracle_model(golden):
    torch.manual_seed(golden['weight_seed'])
    return CRNNOracle()


def test_seeded_weights_match_reference_checksums(golden, oracle_model):
    state = oracle_model.state_dict()
    assert list(state.keys()) == golden['state_keys']
    assert len(state) == 49
    for k, v in state.items():
        s, a = golden['state_checksums'][k]
        assert abs(float(v.double().sum()) - s) <= 1e-9 * max(1.0, a), k
    assert sum(p.numel() for p in oracle_model.parameters()) == 8332966


def test_training_forward_backward_matches_reference_golden(golden, oracle_model):
    torch.set_num_threads(4)
    m = oracle_model
    m.train()
    m.zero_grad()
    b = golden['batch']
    loss, logp = m(b['image'], targets=b['label'], lengths=b['length'].long(), train=True)
    assert loss.dtype == torch.float64
    assert abs(float(loss) - float(golden['train_loss'])) < 1e-9
    assert float((logp - golden['train_log_probs']).abs().max()) < 1e-6
    loss.mean().backward()
    for k, p in m.named_parameters():
        norm, head = golden['grad_stats'][k]
        assert abs(float(p.grad.double().norm()) - norm) <= 1e-5 * max(norm, 1e-9), k
        assert float((p.grad.flatten()[:8] - head).abs().max()) <= 1e-5 * max(float(head.abs().max()), 1e-9), k


def test_adam_trajectory_matches_reference_golden(golden):
    torch.set_num_threads(4)
    torch.manual_seed(golden['weight_seed'])
    m = CRNNOracle().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = [train_step(m, opt, golden['batch']) for _ in range(3)]
    for a, b in zip(losses, golden['adam_losses']):
        assert abs(a - b) < 1e-4 * max(1.0, abs(b)), (losses, golden['adam_losses'])


def test_greedy_decode_rules():
    # hand-made path "_ A A _ A ? B B _" -> "AAB"  (SURVEY.md §8c): blank=0, unknown=1, A=12, B=13
    path = [0, 12, 12, 0, 12, 1, 13, 13, 0]
    pred = np.zeros((1, 38, 1, len(path)), dtype=np.float32)
    for t, c in enumerate(path):
        pred[0, c, 0, t] = 1.0
    out = greedy_decode(pred)
    assert out[0, :3].tolist() == [12, 12, 13] and out[0, 3:].tolist() == [0] * 6
    # an unknown between two identical symbols does not reset `previous`
    path = [12, 1, 12]
    pred = np.zeros((1, 38, 1, 3), dtype=np.float32)
    for t, c in enumerate(path):
        pred[0, c, 0, t] = 1.0
    assert greedy_decode(pred)[0].tolist() == [12, 0, 0]


def test_greedy_decode_matches_reference_golden(golden):
    dec = greedy_decode(golden['eval_pred'].numpy())
    assert (torch.from_numpy(dec) == golden['eval_decode']).all()


def test_explicit_ctc_matches_torch_ctc():
    g = torch.Generator().manual_seed(2)
    T, N, C, S = 14, 5, 11, 6
    logits = torch.randn(T, N, C, generator=g) * 1.5
    lengths = torch.tensor([3, 1, 6, 0, 4])
    targets = torch.zeros(N, S, dtype=torch.int64)
    for i, L in enumerate(lengths.tolist()):
        targets[i, :L] = torch.randint(1, C, (L,), generator=g)
    targets[2, 1] = targets[2, 0]
    x = logits.clone().requires_grad_(True)
    lp = TF.log_softmax(x, dim=2).double()
    loss = TF.ctc_loss(lp, targets, torch.full((N,), T), lengths, zero_infinity=True)
    loss.backward()
    o = ctc_1d(logits.numpy(), targets.numpy(), lengths.numpy())
    assert abs(o['loss'] - float(loss)) < 1e-6   # f32 log-softmax round-off (numpy vs torch summation order)
    assert float((torch.from_numpy(o['grad_logits']).float() - x.grad).abs().max()) < 2e-7
    nll = TF.ctc_loss(lp, targets, torch.full((N,), T), lengths, reduction='none')
    assert float((torch.from_numpy(o['nll']) - nll.detach()).abs().max()) < 1e-5


def test_explicit_ctc_infeasible_sample_zero_infinity():
    T, N, C = 3, 2, 5
    logits = torch.zeros(T, N, C)
    targets = torch.tensor([[1, 1, 1], [2, 0, 0]])
    lengths = torch.tensor([3, 1])     # "111" needs T >= 5: infeasible
    o = ctc_1d(logits.numpy(), targets.numpy(), lengths.numpy())
    assert np.isinf(o['nll'][0]) and np.isfinite(o['nll'][1])
    assert np.abs(o['grad_logits'][:, 0]).max() == 0.0


def test_synthetic_batch_contract():
    b = synthetic_batch(8, 32, 128, seed=0)
    assert b['image'].shape == (8, 3, 32, 128) and b['image'].dtype == torch.float32
    assert b['label'].shape == (8, 32) and b['label'].dtype == torch.int32
    assert b['length'].dtype == torch.int32 and int(b['length'].min()) >= 3 and int(b['length'].max()) <= 10
    for i in range(8):
        L = int(b['length'][i])
        assert int(b['label'][i, :L].min()) >= 2 and int(b['label'][i, L:].abs().sum()) == 0


@pytest.mark.skipif(not refimport.available(), reason="reference tree only exists in the build container")
def test_oracle_equals_live_reference():
    sm = refimport.import_reference()
    from concern.charsets import EnglishCharset
    charset = EnglishCharset()
    args = {'backbone': 'crnn_backbone', 'decoder': 'CRNNDecoder',
            'decoder_args': {'in_channels': 512, 'inner_channels': 256, 'need_reduce': False, 'charset': charset}}
    torch.manual_seed(5)
    ref = sm.BasicModel(args).train()
    ora = CRNNOracle(len(charset)).train()
    ora.load_state_dict(ref.state_dict(), strict=True)
    b = synthetic_batch(2, 32, 48, seed=9)
    lr_, pr_ = ref(b['image'], targets=b['label'], lengths=b['length'].long(), train=True)
    lo_, po_ = ora(b['image'], targets=b['label'], lengths=b['length'].long(), train=True)
    assert torch.equal(lr_, lo_) and torch.equal(pr_, po_)
