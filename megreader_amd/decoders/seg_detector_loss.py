"""Losses of the DB detector -- restatement of reference decoders/seg_detector_loss.py:157-185 (`L1BalanceCELoss`),
balance_cross_entropy_loss.py:29-56, l1_loss.py:5-11 (MaskL1Loss) and dice_loss.py:28-42 (DiceLoss): elementwise /
reduction torch ops on the 1-channel 640 x 640 maps (they are not MFMA or kernel material: one pass over 3 x 6.5 MB).
The reference counts positives / negatives on the HOST (`int(positive.float().sum())`, a device synchronisation per step
that also makes the step impossible to capture in a hipGraph); on GPU tensors the same quantities stay on the device:
the hard-negative top-k with its data-dependent k becomes "sort descending, sum the first k" with k a device scalar --
the same set of elements, the same loss and gradient (ties inside the sort carry equal values)."""
import torch
import torch.nn as nn


class BalanceCrossEntropyLoss(nn.Module):
    def __init__(self, negative_ratio=3.0, eps=1e-6, host_counts=None):
        super().__init__()
        self.negative_ratio = negative_ratio
        self.eps = eps
        self.host_counts = host_counts   # None: host counts for CPU tensors (the reference's code path), device counts on GPU

    def forward(self, pred, gt, mask, return_origin=False):
        positive = (gt * mask).byte()
        negative = ((1 - gt) * mask).byte()
        loss = nn.functional.binary_cross_entropy(pred, gt, reduction='none')[:, 0, :, :]
        positive_loss = loss * positive.float()
        negative_loss = loss * negative.float()
        host = (not pred.is_cuda) if self.host_counts is None else self.host_counts
        if host:   # balance_cross_entropy_loss.py:38-52 verbatim
            positive_count = int(positive.float().sum())
            negative_count = min(int(negative.float().sum()), int(positive_count * self.negative_ratio))
            negative_loss, _ = torch.topk(negative_loss.view(-1), negative_count)
            balance_loss = (positive_loss.sum() + negative_loss.sum()) / (positive_count + negative_count + self.eps)
        else:
            # integer counts (exact beyond 2^24 elements, i.e. batch >= 41 at 640 x 640), the reference's
            # `min(int(neg), int(pos * ratio))` with the product formed in double like Python's
            pc = positive.sum(dtype=torch.int64)
            nc = torch.minimum(negative.sum(dtype=torch.int64), torch.floor(pc.double() * self.negative_ratio).long())
            flat = negative_loss.view(-1)
            srt, _ = torch.sort(flat, descending=True)
            take = (torch.arange(flat.numel(), device=flat.device, dtype=torch.int64) < nc).to(srt.dtype)
            balance_loss = (positive_loss.sum() + (srt * take).sum()) / ((pc + nc).double() + self.eps).to(srt.dtype)
        if return_origin:
            return balance_loss, loss
        return balance_loss


class MaskL1Loss(nn.Module):
    def forward(self, pred, gt, mask):
        loss = (torch.abs(pred[:, 0] - gt) * mask).sum() / mask.sum()
        return loss, dict(l1_loss=loss)


class DiceLoss(nn.Module):
    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, pred, gt, mask, weights=None):
        assert pred.dim() == 4, pred.dim()
        pred = pred[:, 0, :, :]
        gt = gt[:, 0, :, :]
        assert pred.shape == gt.shape and pred.shape == mask.shape
        if weights is not None:
            mask = weights * mask
        intersection = (pred * gt * mask).sum()
        union = (pred * mask).sum() + (gt * mask).sum() + self.eps
        return 1 - 2.0 * intersection / union


class _DBLossFn(torch.autograd.Function):
    """The whole L1BalanceCELoss on the device in 7 launches forward / 1 backward (csrc/db_loss.hip) instead of ~100 torch
    launches: returns f32 [4] = (loss, bce, l1, dice); only element 0 carries a gradient."""

    @staticmethod
    def forward(ctx, binary, thresh, tbinary, gt, mask, tmap, tmask, ratio, eps, l1_scale, bce_scale):
        from .._lib import call, load, ptr
        from ..nn.functional import ZeroArena
        N = binary.shape[0]
        HW = binary.numel() // N
        dev = binary.device
        maps = [t.contiguous().float() if (t.dtype != torch.float32 or not t.is_contiguous()) else t
                for t in (binary, thresh, tbinary, gt, mask, tmap, tmask)]
        nbytes = load().mr_db_loss_ws_bytes()
        arena = ZeroArena.take(dev, (nbytes + 7) // 8)
        ws = arena if arena is not None else torch.zeros(((nbytes + 7) // 8,), dtype=torch.float64, device=dev)
        negloss = torch.empty((N * N * HW,), dtype=torch.float32, device=dev)
        out = torch.empty((16,), dtype=torch.float32, device=dev)
        call("mr_db_loss_fwd", *[ptr(t) for t in maps], ptr(negloss), ptr(ws), ptr(out), N, HW, float(ratio), float(eps),
             float(l1_scale), float(bce_scale))
        ctx.save_for_backward(out, *maps)
        ctx.scales = (float(l1_scale), float(bce_scale))
        ctx.shapes = (binary.shape, thresh.shape, tbinary.shape)
        return out[:4]

    @staticmethod
    def backward(ctx, g):
        from .._lib import call, ptr
        out, binary, thresh, tbinary, gt, mask, tmap, tmask = ctx.saved_tensors
        N = binary.shape[0]
        HW = binary.numel() // N
        gl = g[0:1].contiguous().float()          # metrics (elements 1..3) are for logging only
        gb, gth, gtb = torch.empty_like(binary), torch.empty_like(thresh), torch.empty_like(tbinary)
        call("mr_db_loss_bwd", ptr(binary), ptr(thresh), ptr(tbinary), ptr(gt), ptr(mask), ptr(tmap), ptr(tmask), ptr(out),
             ptr(gl), ptr(gb), ptr(gth), ptr(gtb), N, HW, ctx.scales[0], ctx.scales[1])
        s0, s1, s2 = ctx.shapes
        return (gb.view(s0), gth.view(s1), gtb.view(s2)) + (None,) * 8


def _fused_ok(pred, batch):
    try:
        b, t, tb = pred['binary'], pred['thresh'], pred['thresh_binary']
        gt, mask, tmap, tmask = batch['gt'], batch['mask'], batch['thresh_map'], batch['thresh_mask']
    except KeyError:
        return False
    if not all(isinstance(x, torch.Tensor) and x.is_cuda for x in (b, t, tb, gt, mask, tmap, tmask)):
        return False
    N = b.shape[0]
    return (b.dim() == 4 and b.shape[1] == 1 and t.shape == b.shape and tb.shape == b.shape and gt.shape == b.shape and
            mask.dim() == 3 and mask.shape[0] == N and mask.shape[1:] == b.shape[2:] and tmap.shape == mask.shape and
            tmask.shape == mask.shape)


FUSED_DB_LOSS = __import__("os").environ.get("MEGREADER_DB_LOSS_FUSED", "1") != "0"     # A/B: 0 = the torch restatement


class L1BalanceCELoss(nn.Module):
    """Balanced cross entropy on `binary`, masked L1 on `thresh`, Dice on `thresh_binary`."""

    def __init__(self, eps=1e-6, l1_scale=10, bce_scale=5):
        super().__init__()
        self.dice_loss = DiceLoss(eps=eps)
        self.l1_loss = MaskL1Loss()
        self.bce_loss = BalanceCrossEntropyLoss()
        self.l1_scale = l1_scale
        self.bce_scale = bce_scale

    def forward(self, pred, batch):
        # (the fused kernels use ONE eps for the balanced-BCE denominator and the Dice union: only when the two modules agree --
        # the reference builds DiceLoss(eps=eps) from the constructor argument and BalanceCrossEntropyLoss() with its own default,
        # decoders/seg_detector_loss.py:160-170; a non-default L1BalanceCELoss(eps=...) takes the unfused path -- ADVICE r5)
        if FUSED_DB_LOSS and getattr(self.dice_loss, "eps", None) == self.bce_loss.eps and _fused_ok(pred, batch):
            r = _DBLossFn.apply(pred['binary'], pred['thresh'], pred['thresh_binary'], batch['gt'], batch['mask'],
                                batch['thresh_map'], batch['thresh_mask'], self.bce_loss.negative_ratio, self.bce_loss.eps,
                                self.l1_scale, self.bce_scale)
            m = r.detach()
            return r[0], dict(bce_loss=m[1], thresh_loss=m[3], l1_loss=m[2])
        bce_loss = self.bce_loss(pred['binary'], batch['gt'], batch['mask'])
        metrics = dict(bce_loss=bce_loss)
        l1_loss, l1_metric = self.l1_loss(pred['thresh'], batch['thresh_map'], batch['thresh_mask'])
        dice_loss = self.dice_loss(pred['thresh_binary'], batch['gt'], batch['mask'])
        metrics['thresh_loss'] = dice_loss
        loss = dice_loss + self.l1_scale * l1_loss + bce_loss * self.bce_scale
        metrics.update(**l1_metric)
        return loss, metrics
