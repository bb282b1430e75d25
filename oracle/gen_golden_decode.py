"""Generate tests/golden/decode_golden.pt by executing the UNMODIFIED reference representers / measurer on CPU
(this container only):

    python oracle/gen_golden_decode.py

  structure/representers/ctc_representer.py:20-34      greedy 1-D CTC decode
  structure/representers/ctc_representer2d.py:27-51    2-D decode (row pick, then class pick)
  structure/measurers/sequence_recognition_measurer.py:66-72,101-112   accuracy / normalised edit distance

Inputs are adversarial on purpose: exact ties between classes and between rows, `unknown` (1) runs, repeats split by
blanks and by unknowns, all-blank rows, full-length outputs.  The script refuses to write the file unless the oracle
restatement (oracle/decode.py) agrees with the reference on every case.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import refimport  # noqa: E402
from oracle.decode import greedy_decode, greedy_decode_2d, label_to_string, measure  # noqa: E402


def make_cases():
    g = torch.Generator().manual_seed(11)
    N, C, T, H = 12, 38, 33, 8
    # --- 1-D: softmax-like scores with planted structure
    p = torch.rand(N, C, 1, T, generator=g)
    path = torch.randint(0, C, (N, T), generator=g)
    path[0] = 0                                             # all blank
    path[1] = torch.arange(T) % 36 + 2                      # every step a new symbol: full-length output
    path[2, :] = 5                                          # one long repeat
    path[3] = torch.tensor(([7, 0, 7, 1, 7, 7, 0, 1, 1, 8] * 4)[:T])   # repeats split by blank / unknown
    path[4] = torch.tensor(([1] * 5 + [9] * 3 + [1, 9, 0, 9] * 7)[:T])  # unknown does not update `previous`
    for n in range(N):
        for t in range(T):
            p[n, path[n, t], 0, t] = 2.0
    # exact ties: first index must win
    p[5, :, 0, 0] = 0.5
    p[5, 3, 0, 1] = 2.0
    p[5, 9, 0, 1] = 2.0
    p[6, 0, 0, :] = 2.0                                     # tie between blank and the planted class
    # --- 2-D
    cl = torch.rand(N, C, H, 16, generator=g)
    cl = cl / cl.sum(dim=1, keepdim=True)
    mk = torch.rand(N, 1, H, 16, generator=g)
    mk = mk / mk.sum(dim=2, keepdim=True)
    cl[0, :, :, :] = 1.0 / C                                # total tie over classes
    mk[0] = 1.0 / H                                         # and rows
    cl[1, 0] = 0.9                                          # blank everywhere
    mk[2, 0, 3, :] = 1.0                                    # one dominant row
    cl[2, 11, 3, :] = 0.99
    cl[3, 1, :, ::2] = 0.95                                 # unknown on even columns
    return p, cl, mk


def main():
    refimport.import_reference()
    from concern.charsets import EnglishCharset
    from structure.representers.ctc_representer import CTCRepresenter
    from structure.representers.ctc_representer2d import CTCRepresenter2D
    from structure.measurers.sequence_recognition_measurer import SequenceRecognitionMeasurer

    charset = EnglishCharset()
    p, cl, mk = make_cases()
    g = torch.Generator().manual_seed(5)
    labels = torch.zeros(p.shape[0], 32, dtype=torch.int32)
    for i in range(p.shape[0]):
        L = int(torch.randint(0, 11, (1,), generator=g))
        labels[i, :L] = torch.randint(2, 38, (L,), generator=g, dtype=torch.int32)
    batch = {'label': labels}

    rep1 = CTCRepresenter(charset=charset)
    out1 = rep1.represent(batch, p.clone())
    dec1 = greedy_decode(p.numpy())
    # make some predictions equal their label so accuracy is not all-False: decode row i as label for even i
    for i in range(0, p.shape[0], 2):
        labels[i] = torch.from_numpy(dec1[i][:32])
    out1 = rep1.represent(batch, p.clone())
    assert [o['pred_string'] for o in out1] == [label_to_string(r) for r in dec1], "1-D decode oracle != reference"
    assert [o['label_string'] for o in out1] == [label_to_string(r) for r in labels.numpy()]

    rep2 = CTCRepresenter2D(charset=charset)
    out2 = rep2.represent(batch, (cl.clone(), mk.clone()))
    dec2 = greedy_decode_2d(cl.numpy(), mk.numpy())
    assert [o['pred_string'] for o in out2] == [label_to_string(r) for r in dec2], "2-D decode oracle != reference"

    meas = SequenceRecognitionMeasurer()
    m1 = meas.measure(batch, out1)
    acc, eds = measure(labels.numpy(), dec1)
    assert [bool(a) for a in m1['accuracy']] == acc, "accuracy oracle != reference"
    assert np.allclose(m1['edit_distance'], eds, rtol=0, atol=0), "edit distance oracle != reference"
    m2 = meas.measure(batch, out2)
    acc2, eds2 = measure(labels.numpy(), dec2)
    assert [bool(a) for a in m2['accuracy']] == acc2 and np.allclose(m2['edit_distance'], eds2, rtol=0, atol=0)

    out = {'pred_1d': p, 'classify': cl, 'mask': mk, 'labels': labels,
           'decode_1d': torch.from_numpy(dec1), 'decode_2d': torch.from_numpy(dec2),
           'pred_strings_1d': [o['pred_string'] for o in out1], 'pred_strings_2d': [o['pred_string'] for o in out2],
           'label_strings': [o['label_string'] for o in out1],
           'accuracy_1d': [bool(a) for a in m1['accuracy']], 'edit_distance_1d': [float(e) for e in m1['edit_distance']],
           'accuracy_2d': [bool(a) for a in m2['accuracy']], 'edit_distance_2d': [float(e) for e in m2['edit_distance']]}
    path = os.path.join(REPO, "tests", "golden", "decode_golden.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes; accuracy_1d", sum(out['accuracy_1d']), "/", len(acc))


if __name__ == "__main__":
    main()
