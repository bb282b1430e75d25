"""`ops.ctc_loss_2d` on MI355X -- mirror of reference ops/ctc_2d/ctc_loss_2d.py:7-37.

Same call signature, same saved tensors, same errors: CPU tensors raise NotImplementedError
(ctc_loss_2d.py:12-13), non-contiguous log_probs / blank out of range / wrong length sizes raise RuntimeError
(csrc/cuda/ctc2d_cuda.cu:35-42).  The returned gradient follows the reference's collect kernel
(SURVEY.md Appendix A.1), not d nll / d log_probs.
"""
import torch
from torch.autograd import Function

from .._lib import call, dtype_code, ptr


def _check(log_probs, targets, input_lengths, target_lengths, blank):
    if not log_probs.is_cuda:
        raise NotImplementedError
    if log_probs.dim() != 4:
        raise RuntimeError("log_probs must be [T, H, N, C]")
    if not log_probs.is_contiguous():
        raise RuntimeError("log_probs tensor has to be contiguous")
    T, H, N, C = log_probs.shape
    if not (0 <= blank < C):
        raise RuntimeError("blank must be in label range")
    if input_lengths.shape[0] != N:
        raise RuntimeError("input_lengths must be of size batch_size")
    if target_lengths.shape[0] != N:
        raise RuntimeError("target_lengths must be of size batch_size")
    if targets.dim() != 2 or targets.shape[0] != N:
        raise RuntimeError("targets must be padded [N, S]")
    if 2 * targets.shape[1] + 1 > 8192:
        raise RuntimeError("target too long")


class CTCLoss2DFunction(Function):

    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank=0):
        ctx.blank = blank
        _check(log_probs, targets, input_lengths, target_lengths, blank)
        T, H, N, C = log_probs.shape
        S = targets.shape[1]
        dev = log_probs.device
        targets = targets.to(device=dev, dtype=torch.int64).contiguous()
        input_lengths = input_lengths.to(device=dev, dtype=torch.int64).contiguous()
        target_lengths = target_lengths.to(device=dev, dtype=torch.int64).contiguous()
        nll = torch.empty((N,), dtype=torch.float32, device=dev)
        log_alpha = torch.empty((N, T, H, 2 * S + 1), dtype=torch.float32, device=dev)
        call("mr_ctc2d_fwd", dtype_code(log_probs.dtype), ptr(log_probs), ptr(targets), ptr(input_lengths),
             ptr(target_lengths), T, H, N, C, S, int(blank), ptr(nll), ptr(log_alpha))
        if log_probs.requires_grad:
            ctx.save_for_backward(log_probs, targets, input_lengths, target_lengths, nll, log_alpha)
        return nll if log_probs.dtype == torch.float32 else nll.to(log_probs.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        log_probs, targets, input_lengths, target_lengths, nll, log_alpha = ctx.saved_tensors
        grad_log_probs = None
        if ctx.needs_input_grad[0]:
            T, H, N, C = log_probs.shape
            S = targets.shape[1]
            go = grad_output.to(torch.float32).contiguous()
            log_beta = torch.empty_like(log_alpha)
            grad_log_probs = torch.empty_like(log_probs)
            call("mr_ctc2d_bwd", dtype_code(log_probs.dtype), ptr(go), ptr(log_probs), ptr(targets),
                 ptr(input_lengths), ptr(target_lengths), ptr(nll), ptr(log_alpha), ptr(log_beta),
                 ptr(grad_log_probs), T, H, N, C, S, int(ctx.blank))
        return grad_log_probs, None, None, None, None


ctc_loss_2d = CTCLoss2DFunction.apply
