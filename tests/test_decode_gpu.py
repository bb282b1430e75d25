"""GPU greedy decode + metrics (csrc/pipeline.hip, megreader_amd.ops.decode / structure.representers / measurers)
bit-exact against the golden vectors produced by the unmodified reference and against oracle/decode.py on adversarial
and random inputs (ties, unknown runs, T > 64 = several wave chunks, non-contiguous layouts, bf16 / f64 scores)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from megreader_amd.ops.decode import ctc2d_greedy_decode, ctc_greedy_decode, sequence_measure  # noqa: E402
from megreader_amd.structure.measurers import SequenceRecognitionMeasurer  # noqa: E402
from megreader_amd.structure.representers import CTCRepresenter, CTCRepresenter2D  # noqa: E402
from oracle.decode import greedy_decode, greedy_decode_2d, measure  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def golden(golden_dir):
    return torch.load(os.path.join(golden_dir, "decode_golden.pt"), weights_only=False)


def test_decode_1d_matches_reference_golden(golden):
    ids, ln = ctc_greedy_decode(golden['pred_1d'].to(DEV))
    assert torch.equal(ids.cpu(), golden['decode_1d'])
    assert ln.cpu().tolist() == [int((r != 0).sum()) for r in golden['decode_1d']]
    out = CTCRepresenter().represent({'label': golden['labels']}, golden['pred_1d'].to(DEV))
    assert [o['pred_string'] for o in out] == golden['pred_strings_1d']
    assert [o['label_string'] for o in out] == golden['label_strings']
    m = SequenceRecognitionMeasurer().measure({'label': golden['labels']}, out)
    assert m['accuracy'] == golden['accuracy_1d'] and m['edit_distance'] == golden['edit_distance_1d']


def test_decode_2d_matches_reference_golden(golden):
    cl, mk = golden['classify'].to(DEV), golden['mask'].to(DEV)
    ids, _ = ctc2d_greedy_decode(cl, mk)
    assert torch.equal(ids.cpu(), golden['decode_2d'])
    # NHWC-strided inputs (what the HIP head produces): same answer
    cl2 = cl.contiguous(memory_format=torch.channels_last)
    mk2 = mk.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    ids2, _ = ctc2d_greedy_decode(cl2, mk2)
    assert torch.equal(ids2.cpu(), golden['decode_2d'])
    out = CTCRepresenter2D().represent({'label': golden['labels']}, (cl, mk))
    assert [o['pred_string'] for o in out] == golden['pred_strings_2d']
    m = SequenceRecognitionMeasurer().measure({'label': golden['labels']}, out)
    assert m['accuracy'] == golden['accuracy_2d'] and m['edit_distance'] == golden['edit_distance_2d']


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float64])
@pytest.mark.parametrize("N,C,T", [(256, 38, 33), (7, 5, 1), (3, 38, 64), (5, 38, 65), (4, 100, 200)])
def test_decode_1d_random_vs_oracle(dtype, N, C, T):
    g = torch.Generator().manual_seed(N * 1000 + T)
    p = torch.rand(N, C, 1, T, generator=g)
    p[:, :3] *= 1.6                                       # plenty of blank / unknown / class-2 runs
    p = (p * 8).round() / 8                               # coarse grid: many exact ties, exactly representable in bf16
    p = p.to(dtype)
    want = greedy_decode(p.double().numpy())
    ids, ln = ctc_greedy_decode(p.to(DEV))
    assert np.array_equal(ids.cpu().numpy(), want)
    assert ln.cpu().tolist() == [int((r != 0).sum()) for r in want]
    # permuted memory layout ([T, N, C] storage viewed as [N, C, 1, T])
    q = p[:, :, 0, :].permute(2, 0, 1).contiguous().to(DEV).permute(1, 2, 0).unsqueeze(2)
    ids2, _ = ctc_greedy_decode(q)
    assert np.array_equal(ids2.cpu().numpy(), want)


@pytest.mark.parametrize("N,C,H,W", [(256, 38, 8, 32), (5, 38, 4, 16), (3, 7, 1, 70), (2, 38, 3, 130)])
def test_decode_2d_random_vs_oracle(N, C, H, W):
    g = torch.Generator().manual_seed(N + W)
    cl = ((torch.rand(N, C, H, W, generator=g) * 8).round() / 8)
    mk = ((torch.rand(N, 1, H, W, generator=g) * 4).round() / 4)
    want = greedy_decode_2d(cl.numpy(), mk.numpy())
    ids, ln = ctc2d_greedy_decode(cl.to(DEV), mk.to(DEV))
    assert np.array_equal(ids.cpu().numpy(), want)
    assert ln.cpu().tolist() == [int((r != 0).sum()) for r in want]


def test_measure_random_vs_oracle():
    g = torch.Generator().manual_seed(9)
    N = 300
    lab = torch.randint(0, 12, (N, 32), generator=g, dtype=torch.int32)
    pred = lab.clone()
    # edits: substitutions, deletions (-> blank), unknowns, shifted copies, empty labels, full-length sequences
    noise = torch.rand(N, 32, generator=g)
    pred[noise < 0.15] = torch.randint(0, 12, (int((noise < 0.15).sum()),), generator=g, dtype=torch.int32)
    pred[5] = 0
    lab[6] = 0
    lab[7] = torch.arange(32, dtype=torch.int32) % 10 + 2
    pred[7] = lab[7].roll(3)
    pred[8] = lab[8]
    pred2 = torch.cat([pred, torch.randint(0, 12, (N, 8), generator=g, dtype=torch.int32)], dim=1)   # S2 != S
    for p in (pred, pred2):
        acc, eds = measure(lab.numpy(), p.numpy(), charset=[None, None] + list("ABCDEFGHIJ"))
        m = sequence_measure(lab.to(DEV), p.to(DEV))
        assert m['accuracy'].cpu().tolist() == acc
        assert m['edit_distance'].cpu().tolist() == eds        # identical IEEE double operations
    # case folding table: ids 2..6 <-> 7..11 are the same letters in another case
    fold = torch.arange(12, dtype=torch.int32)
    fold[7:12] = torch.arange(2, 7, dtype=torch.int32)
    cs = [None, None] + list("ABCDE") + list("abcde")
    acc, eds = measure(lab.numpy(), pred.numpy(), charset=cs)
    m = sequence_measure(lab.to(DEV), pred.to(DEV), fold=fold)
    assert m['accuracy'].cpu().tolist() == acc and m['edit_distance'].cpu().tolist() == eds
