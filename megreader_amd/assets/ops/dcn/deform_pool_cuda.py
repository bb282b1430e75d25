"""Extension-level mirror of the reference's `deform_pool_cuda` pybind module (assets/ops/dcn/src/deform_pool_cuda.cpp:
30-52 `deform_psroi_pooling_cuda_forward`, :54-77 `deform_psroi_pooling_cuda_backward`, bound at
assets/ops/dcn/functions/deform_pool.py:4) over the C ABI (`mr_deform_psroi_fwd/bwd`, csrc/deform_pool.hip).

Same argument order, same contract: contiguous NCHW float32 CUDA tensors, outputs and gradient buffers allocated (and,
for the gradients, zero-filled) by the caller and written in place, `RuntimeError` on a shape mismatch."""
import torch

from ...._lib import call, ptr, require_cuda


def _check(t, name):
    require_cuda(t)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s tensor has to be contiguous" % name)


def deform_psroi_pooling_cuda_forward(input, bbox, trans, out, top_count, no_trans, spatial_scale, output_dim,
                                      group_size, pooled_size, part_size, sample_per_part, trans_std):
    _check(input, "input"), _check(bbox, "bbox"), _check(out, "out"), _check(top_count, "top_count")
    no_trans = int(bool(no_trans))
    if not no_trans:
        _check(trans, "trans")
    B, C, H, W = input.shape
    channels_trans = 2 if no_trans else trans.size(1)
    R = bbox.size(0)
    if R != out.size(0):
        raise RuntimeError("Output shape and bbox number wont match: (%d vs %d)." % (out.size(0), R))
    call("mr_deform_psroi_fwd", ptr(input), ptr(bbox), 0 if no_trans else ptr(trans), ptr(out), ptr(top_count), B, C, H,
         W, R, channels_trans, no_trans, float(spatial_scale), int(output_dim), int(group_size), int(pooled_size),
         int(part_size), int(sample_per_part), float(trans_std))


def deform_psroi_pooling_cuda_backward(out_grad, input, bbox, trans, top_count, input_grad, trans_grad, no_trans,
                                       spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part,
                                       trans_std):
    _check(out_grad, "out_grad"), _check(input, "input"), _check(bbox, "bbox"), _check(top_count, "top_count")
    _check(input_grad, "input_grad")
    no_trans = int(bool(no_trans))
    if not no_trans:
        _check(trans, "trans"), _check(trans_grad, "trans_grad")
    B, C, H, W = input.shape
    channels_trans = 2 if no_trans else trans.size(1)
    R = bbox.size(0)
    if R != out_grad.size(0):
        raise RuntimeError("Output shape and bbox number wont match: (%d vs %d)." % (out_grad.size(0), R))
    call("mr_deform_psroi_bwd", ptr(out_grad), ptr(input), ptr(bbox), 0 if no_trans else ptr(trans), ptr(top_count),
         ptr(input_grad), 0 if no_trans else ptr(trans_grad), B, C, H, W, R, channels_trans, no_trans,
         float(spatial_scale), int(output_dim), int(group_size), int(pooled_size), int(part_size),
         int(sample_per_part), float(trans_std))
