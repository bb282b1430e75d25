"""CPU: oracle/decode.py (greedy 1-D / 2-D CTC decode, label strings, accuracy / edit distance) against the golden
vectors produced by the UNMODIFIED reference representers and measurer (oracle/gen_golden_decode.py), and -- in the
build container, where /root/reference exists -- against the live reference classes on fresh random inputs."""
import os

import numpy as np
import pytest
import torch

from oracle.decode import greedy_decode, greedy_decode_2d, label_to_string, levenshtein, measure


@pytest.fixture(scope="module")
def golden(golden_dir):
    return torch.load(os.path.join(golden_dir, "decode_golden.pt"), weights_only=False)


def test_decode_oracle_matches_reference_golden(golden):
    d1 = greedy_decode(golden['pred_1d'].numpy())
    assert np.array_equal(d1, golden['decode_1d'].numpy())
    assert [label_to_string(r) for r in d1] == golden['pred_strings_1d']
    d2 = greedy_decode_2d(golden['classify'].numpy(), golden['mask'].numpy())
    assert np.array_equal(d2, golden['decode_2d'].numpy())
    assert [label_to_string(r) for r in d2] == golden['pred_strings_2d']
    assert [label_to_string(r) for r in golden['labels'].numpy()] == golden['label_strings']


def test_measure_oracle_matches_reference_golden(golden):
    acc, eds = measure(golden['labels'].numpy(), golden['decode_1d'].numpy())
    assert acc == golden['accuracy_1d'] and eds == golden['edit_distance_1d']
    acc, eds = measure(golden['labels'].numpy(), golden['decode_2d'].numpy())
    assert acc == golden['accuracy_2d'] and eds == golden['edit_distance_2d']
    assert any(acc_ for acc_ in golden['accuracy_1d']) and not all(golden['accuracy_1d'])


def test_levenshtein_known_answers():
    assert levenshtein("", "") == 0 and levenshtein("ABC", "") == 3 and levenshtein("", "AB") == 2
    assert levenshtein("KITTEN", "SITTING") == 3 and levenshtein("FLAW", "LAWN") == 2
    assert levenshtein("A" * 30, "B" * 31) == 31


def test_decode_oracle_vs_live_reference():
    from oracle import refimport
    if not refimport.available():
        pytest.skip("reference tree not present (GPU box)")
    refimport.import_reference()
    from concern.charsets import EnglishCharset
    from structure.representers.ctc_representer import CTCRepresenter
    from structure.representers.ctc_representer2d import CTCRepresenter2D
    charset = EnglishCharset()
    g = torch.Generator().manual_seed(3)
    p = torch.rand(9, 38, 1, 70, generator=g)               # T > 64: two chunks in the GPU kernel
    p[:, 1, 0, ::3] = 5.0                                   # many unknowns
    labels = torch.zeros(9, 32, dtype=torch.int32)
    out = CTCRepresenter(charset=charset).represent({'label': labels}, p.clone())
    assert [o['pred_string'] for o in out] == [label_to_string(r) for r in greedy_decode(p.numpy())]
    cl = torch.rand(9, 38, 4, 70, generator=g)
    mk = torch.rand(9, 1, 4, 70, generator=g)
    out = CTCRepresenter2D(charset=charset).represent({'label': labels}, (cl.clone(), mk.clone()))
    assert [o['pred_string'] for o in out] == [label_to_string(r) for r in greedy_decode_2d(cl.numpy(), mk.numpy())]
