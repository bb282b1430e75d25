"""`deform_conv_cuda` -- the extension module the reference's own assets/ops/dcn/functions/deform_conv.py:5 binds
(src/deform_conv_cuda.cpp:681-695 PYBIND11 table), on top of the single-call C ABI `mr_dcn2_fwd` / `mr_dcn2_bwd`
(include/megreader_hip.h).  Same argument order and the reference's ownership rule: the CALLER allocates outputs and
gradients (functions/deform_conv.py:135-137,150-154: `new_empty` / `zeros_like`) and passes scratch `ones` / `columns`
tensors; results are written in place into those NCHW tensors.  Errors: AT_CHECK -> RuntimeError (non-contiguous input or
weight, deform_conv_cuda.cpp:493-494; shape mismatches :507-513), CPU tensors -> RuntimeError.

Layout: the reference extension works on NCHW fp32; the HIP kernels are NHWC in the package's compute dtype
(megreader_amd.set_compute_dtype; fp32 for parity runs), so every call converts in and out -- this entry point exists for
drop-in compatibility of the reference's Function file, the fast path is megreader_amd.assets.ops.dcn.ModulatedDeformConv.
groups = deformable_groups = 1 (all the reference's models)."""
import torch

from .... import get_compute_dtype
from ...._lib import dcn_workspace, call, dtype_code, ptr, vec_of


def _check(input, weight, offset, mask, group, deformable_group):
    if not input.is_cuda:
        raise RuntimeError("modulated_deform_conv: not implemented on the CPU")
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    if group != 1 or deformable_group != 1:
        raise NotImplementedError("groups / deformable_groups > 1 are not used by any reference model")
    if weight.shape[1] != input.shape[1]:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (input.shape[1], weight.shape[1]))


def _geometry(input, weight, stride_h, pad_h, dilation_h):
    N, C, H, W = input.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * pad_h - (dilation_h * (kh - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_h - (dilation_h * (kw - 1) + 1)) // stride_h + 1
    return N, C, H, W, Co, kh, kw, Ho, Wo


def _nhwc(x, dtype):
    N, C, H, W = x.shape
    v = vec_of(dtype)
    if C % v:
        raise RuntimeError("modulated_deform_conv: channels (%d) must be a multiple of %d" % (C, v))
    out = torch.empty((N, H, W, C), dtype=dtype, device=x.device)
    call("mr_nchw_to_nhwc", dtype_code(dtype), ptr(x.float().contiguous()), ptr(out), N, C, H, W, C)
    return out


def _weight_images(weight, dtype):
    Co, C, kh, kw = weight.shape
    K = kh * kw * C
    wk = weight.detach().float().permute(0, 2, 3, 1).contiguous()          # KRSC f32
    w_n = torch.empty((Co, K), dtype=dtype, device=weight.device)
    w_t = torch.empty((K, Co), dtype=dtype, device=weight.device)
    call("mr_prep_matrix", dtype_code(dtype), ptr(wk), K, ptr(w_n), K, ptr(w_t), Co, Co, K, 0)
    return w_n, w_t


def modulated_deform_conv_cuda_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w,
                                       stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group,
                                       with_bias):
    _check(input, weight, offset, mask, group, deformable_group)
    if (kernel_h, kernel_w) != tuple(weight.shape[2:4]):
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)." %
                           (kernel_h, kernel_w, weight.shape[2], weight.shape[3]))
    if stride_h != stride_w or pad_h != pad_w or dilation_h != dilation_w:
        raise NotImplementedError("anisotropic stride / padding / dilation are not used by any reference model")
    dtype = get_compute_dtype()
    dt = dtype_code(dtype)
    N, C, H, W, Co, kh, kw, Ho, Wo = _geometry(input, weight, stride_h, pad_h, dilation_h)
    if tuple(output.shape) != (N, Co, Ho, Wo) or not output.is_contiguous():
        raise RuntimeError("output must be a contiguous [N, Cout, Ho, Wo] tensor allocated by the caller")
    xi = _nhwc(input, dtype)
    w_n, _ = _weight_images(weight, dtype)
    off = offset.detach().float().contiguous()      # per-sample FLAT [2*kh*kw][Ho][Wo] indexing from each sample's base
    msk = mask.detach().float().contiguous()
    col = dcn_workspace(dtype, N, H, W, C, Co, kh, kw, Ho, Wo, False, input.device)
    y = torch.empty((N, Ho, Wo, Co), dtype=dtype, device=input.device)
    b = bias.detach().float().contiguous() if with_bias else None
    call("mr_dcn2_fwd", dt, ptr(xi), ptr(w_n), ptr(b), ptr(off), off[0].numel(), ptr(msk), msk[0].numel(), ptr(y), ptr(col),
         N, H, W, C, Co, kh, kw, stride_h, pad_h, dilation_h, Ho, Wo)
    out32 = output if output.dtype == torch.float32 else torch.empty_like(output, dtype=torch.float32)
    call("mr_nhwc_to_nchw", dt, ptr(y), ptr(out32), N, Co, Ho, Wo, Co)
    if out32 is not output:
        output.copy_(out32)


def modulated_deform_conv_cuda_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight,
                                        grad_bias, grad_offset, grad_mask, grad_output, kernel_h, kernel_w, stride_h,
                                        stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group,
                                        with_bias):
    _check(input, weight, offset, mask, group, deformable_group)
    if stride_h != stride_w or pad_h != pad_w or dilation_h != dilation_w:
        raise NotImplementedError("anisotropic stride / padding / dilation are not used by any reference model")
    dtype = get_compute_dtype()
    dt = dtype_code(dtype)
    N, C, H, W, Co, kh, kw, Ho, Wo = _geometry(input, weight, stride_h, pad_h, dilation_h)
    xi = _nhwc(input, dtype)
    gy = _nhwc(grad_output.contiguous(), dtype)
    _, w_t = _weight_images(weight, dtype)
    off = offset.detach().float().contiguous()
    msk = mask.detach().float().contiguous()
    dev = input.device
    col = dcn_workspace(dtype, N, H, W, C, Co, kh, kw, Ho, Wo, True, dev)
    dx32 = torch.zeros((N, H, W, C), dtype=torch.float32, device=dev)
    doff = torch.zeros_like(off)
    dmsk = torch.zeros_like(msk)
    gw = torch.zeros((Co, kh, kw, C), dtype=torch.float32, device=dev)
    gb = torch.zeros((Co,), dtype=torch.float32, device=dev) if with_bias else None
    call("mr_dcn2_bwd", dt, ptr(gy), ptr(xi), ptr(w_t), ptr(off), off[0].numel(), ptr(msk), msk[0].numel(), ptr(col),
         ptr(dx32), ptr(doff), ptr(dmsk), ptr(gw), ptr(gb), N, H, W, C, Co, kh, kw, stride_h, pad_h, dilation_h, Ho, Wo)
    # in place into the caller's (zero-initialised) gradient tensors, reference layout
    gi = torch.empty((N, C, H, W), dtype=torch.float32, device=dev)
    call("mr_nhwc_to_nchw", 0, ptr(dx32), ptr(gi), N, C, H, W, C)
    grad_input.add_(gi.to(grad_input.dtype))
    grad_weight.add_(gw.permute(0, 3, 1, 2).to(grad_weight.dtype))
    grad_offset.add_(doff.view_as(grad_offset).to(grad_offset.dtype))
    grad_mask.add_(dmsk.view_as(grad_mask).to(grad_mask.dtype))
    if with_bias:
        grad_bias.add_(gb.to(grad_bias.dtype))


# ---------------------------------------------------------------------------------------------------------------------
# DCN v1 entry points (deform_conv_cuda.cpp:151-156,258-264,374-381; called from functions/deform_conv.py:46-51,70-86):
# the v2 kernels with a mask of ones (same flat offset indexing and validity rule, deform_conv_cuda_kernel.cu:254-263 vs
# 617).  NOTE the reference's argument order: kW, kH, dW, dH, padW, padH, dilationW, dilationH (width first).
# ---------------------------------------------------------------------------------------------------------------------
def _v1_ones(offset, kh, kw):
    N, _, Ho, Wo = offset.shape
    return torch.ones((N, kh * kw, Ho, Wo), dtype=torch.float32, device=offset.device)


def deform_conv_forward_cuda(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                             dilationH, group, deformable_group, im2col_step):
    """output <- deformable convolution of `input` (no bias); `output` is allocated by the caller and written in place."""
    modulated_deform_conv_cuda_forward(input, weight, None, ones, offset, _v1_ones(offset, kH, kW), output, columns, kH, kW,
                                       dH, dW, padH, padW, dilationH, dilationW, group, deformable_group, False)
    return 1


def deform_conv_backward_input_cuda(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW,
                                    padH, dilationW, dilationH, group, deformable_group, im2col_step):
    """gradInput / gradOffset (caller-allocated, zero-initialised: functions/deform_conv.py:62-63) <- the data and offset
    gradients."""
    gw = torch.zeros_like(weight, dtype=torch.float32)
    gm = torch.zeros((offset.shape[0], kH * kW, offset.shape[2], offset.shape[3]), dtype=torch.float32, device=offset.device)
    gi = torch.zeros_like(gradInput)
    go = torch.zeros_like(gradOffset)
    modulated_deform_conv_cuda_backward(input, weight, None, None, offset, _v1_ones(offset, kH, kW), columns, gi, gw, None, go,
                                        gm, gradOutput, kH, kW, dH, dW, padH, padW, dilationH, dilationW, group,
                                        deformable_group, False)
    gradInput.copy_(gi)
    gradOffset.copy_(go)
    return 1


def deform_conv_backward_parameters_cuda(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH,
                                         dilationW, dilationH, group, deformable_group, scale, im2col_step):
    """gradWeight += scale * weight gradient (deform_conv_cuda.cpp:466-470 accumulates with `addmm_`).  The weight values
    do not enter the weight gradient, so a zero weight of the right shape feeds the shared backward call."""
    w0 = torch.zeros_like(gradWeight, dtype=torch.float32).contiguous()
    gw = torch.zeros_like(w0)
    gi = torch.zeros_like(input, dtype=torch.float32)
    go = torch.zeros_like(offset, dtype=torch.float32)
    gm = torch.zeros((offset.shape[0], kH * kW, offset.shape[2], offset.shape[3]), dtype=torch.float32, device=offset.device)
    modulated_deform_conv_cuda_backward(input, w0, None, None, offset, _v1_ones(offset, kH, kW), columns, gi, gw, None, go, gm,
                                        gradOutput, kH, kW, dH, dW, padH, padW, dilationH, dilationW, group,
                                        deformable_group, False)
    gradWeight.add_((gw * float(scale)).to(gradWeight.dtype))
    return 1
