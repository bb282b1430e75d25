"""Mirrors of structure/representers/ctc_representer.py:8-45 and ctc_representer2d.py:7-62: same `represent(batch,
pred)` contract (a list of {'label_string', 'pred_string'} dicts, the 2-D one also carries 'mask' / 'classify'), with
the arg-max + collapse done by one HIP kernel launch instead of a Python loop per sample and step.  The id tensors stay
on the GPU under 'pred_ids' / 'label_ids' so that megreader_amd.structure.measurers can score them there."""
import torch

from ..charsets import EnglishCharset
from ..ops.decode import ctc2d_greedy_decode, ctc_greedy_decode


class _Base(object):
    def __init__(self, charset=None, cmd=None, **kwargs):
        from . import member_from_config
        charset = member_from_config(charset, cmd)     # built from the YAML by the reference's config system
        self.charset = charset if charset is not None else EnglishCharset()

    def label_to_string(self, label):
        return self.charset.label_to_string(label)

    def _result(self, labels, ids, extra=None):
        ids_host = ids.cpu().tolist()
        labels_host = labels.cpu().tolist()
        out = []
        for i in range(labels.shape[0]):
            d = {'label_string': self.label_to_string(labels_host[i]),
                 'pred_string': self.label_to_string(ids_host[i]),
                 'pred_ids': ids[i], 'label_ids': labels[i]}
            if extra is not None:
                d.update(extra(i))
            out.append(d)
        return out


class CTCRepresenter(_Base):
    def represent(self, batch, pred):
        """pred: (N, C, 1, W) class scores (decoders/crnn.py:100-104 eval output)."""
        ids, _ = ctc_greedy_decode(pred, self.charset.blank, self.charset.unknown)
        return self._result(batch['label'], ids)


class CTCRepresenter2D(_Base):
    max_size = 32

    def represent(self, batch, pred):
        classify, mask = pred
        ids, _ = ctc2d_greedy_decode(classify, mask, self.charset.blank, self.charset.unknown)
        cl, mk = classify.to('cpu'), mask.to('cpu')
        return self._result(batch['label'], ids, lambda i: {'mask': mk[i][0], 'classify': cl[i]})
