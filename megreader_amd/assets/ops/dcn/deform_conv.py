"""Modulated deformable convolution (DCNv2) on HIP kernels.

Mirror of reference assets/ops/dcn/functions/deform_conv.py:108-177 and modules/deform_conv.py:84-160: same
constructor / call signatures, parameter names and initialisation, same errors (CPU tensors -> NotImplementedError,
deform_conv.py:130-131), no shape check on offset / mask (the v2 host code has none, which is what makes the
reference's stride-2 blocks work on a larger offset map -- SURVEY.md §3.3; this implementation indexes the maps the
same flat way).  groups = deformable_groups = 1 only (all the reference's models).

Channels that are multiples of 64 (every DCN layer of deformable_resnet50) take the fused kernels of csrc/dcn_fused.hip:
forward = sample -> LDS -> MFMA without a column matrix; backward = offset / mask gradients from gcol tiles kept in the
MFMA accumulators, input gradient as a CSR-inverted gather-GEMM (no f32 atomics), dW / dbias by a TN GEMM sampled on the fly.
Other shapes: deformable im2col over the whole batch + one MFMA GEMM forward; gcol = dy * W, offset/mask gradient kernel,
input gradient scatter (f32 atomics), dW/dbias on a recomputed column matrix backward.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from .... import get_compute_dtype
from ...._lib import call, dcn_backward_workspace, dcn_workspace, dtype_code, load, ptr, vec_of
from ....nn import prep
from ....nn.functional import (ZeroArena, _grad_internal, grad_sink, mark_zero_padded, notify_grad_ready, to_internal)


def _dcn_forward(ctx, input, off, msk, weight, bias, stride, padding, dilation, groups, deformable_groups):
    """Shared forward of the two Functions below.  off / msk: contiguous f32 NCHW maps (the flat per-sample indexing of the
    reference).  Returns the NHWC output of the compute dtype; saves what `_dcn_backward` needs on ctx."""
    if groups != 1 or deformable_groups != 1:
        raise NotImplementedError("groups / deformable_groups > 1 are not used by any reference model")
    dtype = get_compute_dtype()
    dt = dtype_code(dtype)
    v = vec_of(dtype)
    xi = to_internal(input, dtype)
    N, H, W, C = xi.shape
    Co, Ci, kh, kw = weight.shape
    if Ci != C or C % v or Co % v:
        raise RuntimeError("modulated_deform_conv: channels (%d -> %d) must be multiples of %d" % (C, Co, v))
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    off_bs, msk_bs = off[0].numel(), msk[0].numel()
    K = kh * kw * C
    wk = weight.detach().permute(0, 2, 3, 1)
    if wk.is_contiguous():
        # physical KRSC parameter (the modules keep their weight channels_last): the [Co][K] / [K][Co] operand images
        # live in the prepared-weight cache (nn/prep.py) -- rebuilt only when the parameter changes, and refreshed
        # together with every other layer's images by the fused optimizers' ONE batched launch per step
        def build(old):
            if old is None:
                w_n_ = torch.empty((Co, K), dtype=dtype, device=xi.device)
                w_t_ = torch.empty((K, Co), dtype=dtype, device=xi.device)
            else:
                w_n_, w_t_ = old
            return (w_n_, w_t_), [prep.matrix_job(ptr(wk), K, ptr(w_n_), K, ptr(w_t_), Co, Co, K, 0)]
        w_n, w_t = prep.prepared((weight,), ("dcn", K, Co), build, dtype)
    else:
        wk = wk.contiguous()   # physical KRSC f32
        w_n = torch.empty((Co, K), dtype=dtype, device=xi.device)
        w_t = torch.empty((K, Co), dtype=dtype, device=xi.device)
        call("mr_prep_matrix", dt, ptr(wk), K, ptr(w_n), K, ptr(w_t), Co, Co, K, 0)
    col = dcn_workspace(dtype, N, H, W, C, Co, kh, kw, Ho, Wo, False, xi.device)   # the reference's `columns` scratch
    y = torch.empty((N, Ho, Wo, Co), dtype=dtype, device=xi.device)
    call("mr_dcn2_fwd", dt, ptr(xi), ptr(w_n), ptr(bias), ptr(off), off_bs, ptr(msk), msk_bs, ptr(y), ptr(col), N, H,
         W, C, Co, kh, kw, stride, padding, dilation, Ho, Wo)
    # bf16 materialised path (round 6): the forward's column matrix stays alive for the backward's weight gradient
    keep_col = col is not None and bool(load().mr_dcn2_col_saved(dt, H, W, C, Co, kh, kw))
    ctx.save_for_backward(xi, off, msk, w_t, col if keep_col else None)
    ctx.geom = (N, H, W, C, Co, kh, kw, stride, padding, dilation, Ho, Wo, off_bs, msk_bs)
    ctx.params = (weight, bias)
    ctx.dtype = dtype
    return y


def _dcn_backward(ctx, grad_output, want_dx, want_dw, want_db, scratch_ok=False):
    """Shared backward: (grad_input NCHW view | None, grad_offset f32 NCHW, grad_mask f32 NCHW, grad_weight | None,
    grad_bias | None).  Parameter gradients go straight into the fused optimizers' gradient sinks when those exist
    (nn/functional.py grad_sink: no temporary, no `grad += tmp` launch) -- the returned gradient is then None."""
    xi, off, msk, w_t, col_saved = ctx.saved_tensors
    N, H, W, C, Co, kh, kw, stride, padding, dilation, Ho, Wo, off_bs, msk_bs = ctx.geom
    dtype = ctx.dtype
    dt = dtype_code(dtype)
    dev = xi.device
    K = kh * kw * C
    g = _grad_internal(grad_output, dtype)
    col, ws_flags = dcn_backward_workspace(dtype, N, H, W, C, Co, kh, kw, Ho, Wo, dev)   # CSR of the scatter / column matrix
    weight, bias = ctx.params
    w_sink = grad_sink(weight, (Co, kh, kw, C)) if want_dw else None
    b_sink = grad_sink(bias, (Co,)) if want_db else None
    # the input gradient straight in the compute dtype where the fused kernel runs un-split (every layer of the detector at
    # batch 2): no zero fill in front of it, no conversion pass behind it
    direct = bool(want_dx and load().mr_dcn2_dx_direct(dt, N, H, W, C, Co, kh, kw))
    # every accumulated output of mr_dcn2_bwd must arrive zeroed: ONE zero-fill for all of them (separate torch.zeros
    # were 52 fill launches per DB step) -- or none: scratch_ok (the packed node consumes the offset / mask gradients before
    # it returns) takes them from the pre-zeroed arena that zero_grad() clears with the gradient buffers
    sizes = [off.numel(), msk.numel(), N * H * W * C if (want_dx and not direct) else 0,
             Co * K if (want_dw and w_sink is None) else 0, Co if (want_db and b_sink is None) else 0]
    offs = [0]
    for n_ in sizes:
        offs.append(offs[-1] + (n_ + 63) // 64 * 64)
    zbuf = None
    if scratch_ok and not any(sizes[2:]):
        arena = ZeroArena.take(dev, (offs[-1] + 1) // 2)
        if arena is not None:
            zbuf = arena.view(torch.float32)[:offs[-1]]
    if zbuf is None:
        zbuf = torch.zeros((offs[-1],), dtype=torch.float32, device=dev)
    grad_offset = zbuf[offs[0]:offs[0] + sizes[0]].view(off.shape)
    grad_mask = zbuf[offs[1]:offs[1] + sizes[1]].view(msk.shape)
    dx32 = zbuf[offs[2]:offs[2] + sizes[2]].view(N, H, W, C) if (want_dx and not direct) else None
    dxi = torch.empty((N, H, W, C), dtype=dtype, device=dev) if direct else None
    gw = gb = None
    if want_dw:
        gw = w_sink if w_sink is not None else zbuf[offs[3]:offs[3] + sizes[3]].view(Co, kh, kw, C)
    if want_db:
        gb = b_sink if b_sink is not None else zbuf[offs[4]:offs[4] + sizes[4]]
    call("mr_dcn2_bwd3", dt, ptr(g), ptr(xi), ptr(w_t), ptr(off), off_bs, ptr(msk), msk_bs, ptr(col), ptr(dx32), ptr(dxi),
         ws_flags, ptr(grad_offset), ptr(grad_mask), ptr(gw), ptr(gb), ptr(col_saved), N, H, W, C, Co, kh, kw, stride, padding,
         dilation, Ho, Wo)
    grad_input = None
    if want_dx:
        if direct:
            pass
        elif dtype == torch.float32:
            dxi = dx32
        else:
            dxi = torch.empty((N, H, W, C), dtype=dtype, device=dev)
            call("mr_cast", 0, ptr(dx32), dt, ptr(dxi), dx32.numel())
        grad_input = dxi.permute(0, 3, 1, 2)
    grad_weight = grad_bias = None
    if want_dw:
        if w_sink is not None:
            notify_grad_ready(weight)
        else:
            grad_weight = gw.permute(0, 3, 1, 2)
    if want_db:
        if b_sink is not None:
            notify_grad_ready(bias)
        else:
            grad_bias = gb
    return grad_input, grad_offset, grad_mask, grad_weight, grad_bias


class ModulatedDeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1):
        if not input.is_cuda:
            raise NotImplementedError
        # offsets / mask as contiguous f32 NCHW (they are small); flat per-sample indexing like the reference
        off = offset.detach().to(torch.float32).contiguous()
        msk = mask.detach().to(torch.float32).contiguous()
        y = _dcn_forward(ctx, input, off, msk, weight, bias, stride, padding, dilation, groups, deformable_groups)
        ctx.with_bias = bias is not None
        ctx.off_meta = (offset.dtype, mask.dtype)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        grad_input, grad_offset, grad_mask, grad_weight, grad_bias = _dcn_backward(
            ctx, grad_output, ctx.needs_input_grad[0], ctx.needs_input_grad[3],
            ctx.with_bias and ctx.needs_input_grad[4])
        odtype, mdtype = ctx.off_meta
        return (grad_input, grad_offset.to(odtype), grad_mask.to(mdtype), grad_weight, grad_bias, None, None, None,
                None, None)


class ModulatedDeformConvPackedFunction(Function):
    """modulated_deform_conv(x, offset_mask[:, :2k], offset_mask[:, -k:].sigmoid(), ...) -- the way the reference's
    deformable ResNet blocks call the op (backbones/resnet.py:125-142) -- as ONE autograd node taking the offset conv's raw
    output.  Same arithmetic as the unfused expression (sigmoid evaluated in f32); what it removes is glue: the slices,
    casts, `contiguous`, sigmoid, sigmoid_backward, two slice_backward (zero-fill + copy each), their sum and the re-padding
    of the offset conv's incoming gradient were ~20 of the ~30 launches of a DCN layer of the DB step.  The gradient it
    returns for offset_mask is a view of a zero-padded NHWC buffer that nn.Conv2d's backward consumes as is."""

    @staticmethod
    def forward(ctx, input, offset_mask, weight, bias, stride, padding, dilation):
        if not input.is_cuda:
            raise NotImplementedError
        dtype = get_compute_dtype()
        Co, Ci, kh, kw = weight.shape
        n_msk = kh * kw
        n_off = 2 * n_msk
        Nr, Cr, Hm, Wm = offset_mask.shape
        if Cr != n_off + n_msk:
            raise RuntimeError("packed offset_mask must have %d channels, got %d" % (n_off + n_msk, Cr))
        rp = offset_mask.detach().permute(0, 2, 3, 1)
        ld = rp.stride(2)
        if not (rp.dtype == dtype and rp.stride(3) == 1 and ld >= Cr and rp.stride(1) == Wm * ld and
                rp.stride(0) == Hm * Wm * ld):
            rp = rp.contiguous()                # foreign layout: dense NHWC copy
            if rp.dtype != dtype:
                rp = rp.to(dtype)
            ld = Cr
        off = torch.empty((Nr, n_off, Hm, Wm), dtype=torch.float32, device=rp.device)
        msk = torch.empty((Nr, n_msk, Hm, Wm), dtype=torch.float32, device=rp.device)
        call("mr_dcn_unpack", dtype_code(dtype), ptr(rp), ld, ptr(off), ptr(msk), Nr, Hm * Wm, n_off, n_msk)
        y = _dcn_forward(ctx, input, off, msk, weight, bias, stride, padding, dilation, 1, 1)
        ctx.with_bias = bias is not None
        ctx.raw_meta = (offset_mask.dtype, n_off, n_msk)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        grad_input, grad_offset, grad_mask, grad_weight, grad_bias = _dcn_backward(
            ctx, grad_output, ctx.needs_input_grad[0], ctx.needs_input_grad[2],
            ctx.with_bias and ctx.needs_input_grad[3], scratch_ok=True)
        g_raw = None
        if ctx.needs_input_grad[1]:
            msk = ctx.saved_tensors[2]
            rdtype, n_off, n_msk = ctx.raw_meta
            dtype = ctx.dtype
            Nr, _, Hm, Wm = msk.shape
            Cr = n_off + n_msk
            ld = (Cr + vec_of(dtype) - 1) // vec_of(dtype) * vec_of(dtype)   # the offset conv's padded channel count
            graw = torch.empty((Nr, Hm, Wm, ld), dtype=dtype, device=msk.device)
            call("mr_dcn_pack_grad", dtype_code(dtype), ptr(grad_offset), ptr(grad_mask), ptr(msk), ptr(graw), ld, Nr,
                 Hm * Wm, n_off, n_msk)
            mark_zero_padded(graw)
            g_raw = graw[..., :Cr].permute(0, 3, 1, 2)
            if rdtype != dtype:
                g_raw = g_raw.to(rdtype)
        return grad_input, g_raw, grad_weight, grad_bias, None, None, None


def modulated_deform_conv_packed(input, offset_mask, weight, bias=None, stride=1, padding=0, dilation=1):
    return ModulatedDeformConvPackedFunction.apply(input, offset_mask, weight, bias, stride, padding, dilation)


modulated_deform_conv = ModulatedDeformConvFunction.apply


class ModulatedDeformConv(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super(ModulatedDeformConv, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.with_bias = bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)  # physical KRSC

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)

    def forward_packed(self, x, offset_mask):
        """== self(x, offset_mask[:, :2k], offset_mask[:, -k:].sigmoid()) with k = kh * kw taps (k == offset_mask channels / 3),
        as one fused autograd node (ModulatedDeformConvPackedFunction)."""
        if self.groups != 1 or self.deformable_groups != 1:
            raise NotImplementedError("groups / deformable_groups > 1 are not used by any reference model")
        return modulated_deform_conv_packed(x, offset_mask, self.weight, self.bias, self.stride, self.padding,
                                            self.dilation)


class ModulatedDeformConvPack(ModulatedDeformConv):

    def __init__(self, *args, **kwargs):
        super(ModulatedDeformConvPack, self).__init__(*args, **kwargs)
        from ....nn import Conv2d
        self.conv_offset_mask = Conv2d(self.in_channels,
                                       self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                       kernel_size=self.kernel_size, stride=_pair(self.stride),
                                       padding=_pair(self.padding), bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, x):
        # reference modules/deform_conv.py:150-157: o1, o2, mask = chunk(out, 3); offset = cat(o1, o2); mask = sigmoid(mask)
        # -- channels [0, 2k) and [2k, 3k) of `out`: exactly the packed operand
        return self.forward_packed(x, self.conv_offset_mask(x))


class DeformConvFunction(Function):
    """DCN v1 (reference functions/deform_conv.py:10-105; kernels deform_conv_cuda_kernel.cu:189-464): the v2 kernels with
    a mask of ones -- same flat offset indexing, same validity rule (deform_conv_cuda_kernel.cu:254-263 / 617).
    groups = deformable_groups = 1; `im2col_step` only tiles the reference's per-image loop and is ignored."""

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        stride, padding, dilation = _pair(stride)[0], _pair(padding)[0], _pair(dilation)[0]
        kh, kw = weight.shape[2:4]
        Ho = (input.shape[2] + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
        Wo = (input.shape[3] + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
        if not (Ho > 0 and Wo > 0):
            raise ValueError("convolution input is too small (output would be {}x{})".format(Ho, Wo))
        ones = torch.ones((input.shape[0], kh * kw, Ho, Wo), dtype=torch.float32, device=input.device)
        return ModulatedDeformConvFunction.apply(input, offset, ones, weight, None, stride, padding, dilation, groups,
                                                 deformable_groups)

    @staticmethod
    def backward(ctx, *g):  # pragma: no cover - forward returns the v2 Function's graph
        raise RuntimeError("unreachable")


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
    return DeformConvFunction.forward(None, input, offset, weight, stride, padding, dilation, groups, deformable_groups,
                                      im2col_step)


class DeformConv(nn.Module):
    """reference modules/deform_conv.py:11-57"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super(DeformConv, self).__init__()
        assert not bias
        assert in_channels % groups == 0 and out_channels % groups == 0
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // self.groups, *self.kernel_size))
        self.reset_parameters()
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)  # physical KRSC

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class DeformConvPack(DeformConv):
    """reference modules/deform_conv.py:60-81"""

    def __init__(self, *args, **kwargs):
        super(DeformConvPack, self).__init__(*args, **kwargs)
        from ....nn import Conv2d
        self.conv_offset = Conv2d(self.in_channels,
                                  self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                  kernel_size=self.kernel_size, stride=_pair(self.stride), padding=_pair(self.padding),
                                  bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        offset = self.conv_offset(x)
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)
