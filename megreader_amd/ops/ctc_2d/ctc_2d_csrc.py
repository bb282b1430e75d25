"""`ctc_2d_csrc` with the reference extension's two entry points (ops/ctc_2d/csrc/ctc2d.h:7-43, bound by
csrc/vision.cpp; implemented in CUDA by csrc/cuda/ctc2d_cuda.cu:20-88 + ctc2d_cuda_kernel.cu) on top of the C ABI
(`mr_ctc2d_fwd` / `mr_ctc2d_bwd`, include/megreader_hip.h).  Same argument order, same return values, same ownership
(outputs allocated here like `at::zeros` in ctc2d_cuda_kernel.cu:222-228), same errors: AT_CHECK -> RuntimeError for
non-contiguous log_probs / blank out of range / wrong length sizes (ctc2d_cuda.cu:35-42), AT_ERROR("Not implemented on
the CPU") -> RuntimeError for CPU tensors (ctc2d.h:20,42).  TINY is accepted and unused, as in the reference (Q8)."""
import torch

from ..._lib import call, dtype_code, ptr


def _check(log_probs, targets, input_lengths, target_lengths, BLANK):
    if not log_probs.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    if log_probs.dim() != 4:
        raise RuntimeError("log_probs must be [T, H, N, C]")
    if not log_probs.is_contiguous():
        raise RuntimeError("log_probs tensor has to be contiguous")
    T, H, N, C = log_probs.shape
    if not (0 <= BLANK < C):
        raise RuntimeError("blank must be in label range")
    if input_lengths.shape[0] != N:
        raise RuntimeError("input_lengths must be of size batch_size")
    if target_lengths.shape[0] != N:
        raise RuntimeError("target_lengths must be of size batch_size")
    if targets.dim() != 2 or targets.shape[0] != N:
        raise RuntimeError("targets must be padded [batch_size, max_target_length]")
    if 2 * targets.shape[1] + 1 > 8192:
        raise RuntimeError("target too long")


def _i64(t, dev):
    return t.to(device=dev, dtype=torch.int64).contiguous()


def ctc2d_forward(log_probs, targets, input_lengths, target_lengths, BLANK, TINY):
    """-> (neg_log_likelihood [N], log_alpha [N, T, H, 2S+1]) in log_probs' dtype domain (f32 accumulate)."""
    _check(log_probs, targets, input_lengths, target_lengths, int(BLANK))
    T, H, N, C = log_probs.shape
    S = targets.shape[1]
    dev = log_probs.device
    nll = torch.empty((N,), dtype=torch.float32, device=dev)
    log_alpha = torch.empty((N, T, H, 2 * S + 1), dtype=torch.float32, device=dev)
    call("mr_ctc2d_fwd", dtype_code(log_probs.dtype), ptr(log_probs), ptr(_i64(targets, dev)),
         ptr(_i64(input_lengths, dev)), ptr(_i64(target_lengths, dev)), T, H, N, C, S, int(BLANK), ptr(nll),
         ptr(log_alpha))
    return (nll if log_probs.dtype == torch.float32 else nll.to(log_probs.dtype)), log_alpha


def ctc2d_backward(grad_out, log_probs, targets, input_lengths, target_lengths, neg_log_likelihood, log_alpha, BLANK):
    """-> grad [T, H, N, C] (the reference's collect-kernel convention)."""
    _check(log_probs, targets, input_lengths, target_lengths, int(BLANK))
    T, H, N, C = log_probs.shape
    S = targets.shape[1]
    dev = log_probs.device
    go = grad_out.to(device=dev, dtype=torch.float32).contiguous()
    nll = neg_log_likelihood.to(torch.float32).contiguous()
    la = log_alpha.to(torch.float32).contiguous()
    log_beta = torch.empty_like(la)
    grad = torch.empty_like(log_probs)
    call("mr_ctc2d_bwd", dtype_code(log_probs.dtype), ptr(go), ptr(log_probs), ptr(_i64(targets, dev)),
         ptr(_i64(input_lengths, dev)), ptr(_i64(target_lengths, dev)), ptr(nll), ptr(la), ptr(log_beta), ptr(grad), T, H,
         N, C, S, int(BLANK))
    return grad
