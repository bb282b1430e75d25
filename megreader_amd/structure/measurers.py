"""Mirror of structure/measurers/sequence_recognition_measurer.py:11-112 (`measure` / `validate_measure` /
`gather_measure` contract) scoring the id sequences on the GPU (mr_seq_measure): accuracy = upper-cased strings equal,
edit_distance = 1 - min(len, levenshtein) / len (0 for an empty label)."""
import numpy as np
import torch

from ..ops.decode import sequence_measure


class AverageMeter(object):
    """concern/average_meter.py semantics (val / avg / sum / count)."""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        return self


class SequenceRecognitionMeasurer(object):
    def __init__(self, charset=None, blank=0, unknown=1, cmd=None, **kwargs):
        from . import member_from_config
        charset = member_from_config(charset, cmd)
        self.blank, self.unknown = blank, unknown
        self.fold = None
        if charset is not None and hasattr(charset, "_charset"):
            # `.upper()` of the strings: ids whose characters upper-case to the same character compare equal
            canon, fold = {}, []
            for i, ch in enumerate(charset._charset):
                key = ch.upper() if isinstance(ch, str) else ("#", i)
                fold.append(canon.setdefault(key, i))
            if any(f != i for i, f in enumerate(fold)):
                self.fold = torch.tensor(fold, dtype=torch.int32)

    def measure(self, batch, output):
        pred = torch.stack([o['pred_ids'] for o in output])
        label = torch.stack([o['label_ids'] for o in output]).to(pred.device)
        m = sequence_measure(label, pred, self.blank, self.unknown, self.fold)
        return dict(accuracy=m['accuracy'].cpu().tolist(), edit_distance=m['edit_distance'].cpu().tolist())

    def validate_measure(self, batch, output):
        return self.measure(batch, output), []

    evaluate_measure = validate_measure

    def gather_measure(self, raw_metrics, logger=None):
        def gather(key):
            meter = AverageMeter()
            for m in raw_metrics:
                v = m[key]
                meter.update(np.array(v).sum() / len(v), len(v))
            return meter
        return dict(accuracy=gather('accuracy'), edit_distance=gather('edit_distance'))
