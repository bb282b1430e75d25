"""Deformable position-sensitive RoI pooling on MI355X -- mirror of the reference's
assets/ops/dcn/functions/deform_pool.py:9-66 (`DeformRoIPoolingFunction`, `deform_roi_pooling`) and
assets/ops/dcn/modules/deform_pool.py:6-172 (`DeformRoIPooling`, `DeformRoIPoolingPack`,
`ModulatedDeformRoIPoolingPack`): same constructor arguments, `forward` contracts and `state_dict` keys.

The pooling runs in csrc/deform_pool.hip through the extension-level module `deform_pool_cuda` (the reference's own
binding layer).  The offset / mask branches of the *Pack modules are fully-connected stacks on [R, P*P*C] vectors: the
Linear layers are the package's MFMA GEMMs (megreader_amd.nn.Linear); the ReLU / Sigmoid between them are torch
elementwise ops on [R, 1024] tensors (glue, as in the reference)."""
import torch
from torch import nn
from torch.autograd import Function

from . import deform_pool_cuda
from ....nn import Linear


class DeformRoIPoolingFunction(Function):

    @staticmethod
    def forward(ctx, data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                sample_per_part=4, trans_std=.0):
        ctx.spatial_scale = spatial_scale
        ctx.out_size = out_size
        ctx.out_channels = out_channels
        ctx.no_trans = no_trans
        ctx.group_size = group_size
        ctx.part_size = out_size if part_size is None else part_size
        ctx.sample_per_part = sample_per_part
        ctx.trans_std = trans_std
        assert 0.0 <= ctx.trans_std <= 1.0
        if not data.is_cuda:
            raise NotImplementedError
        data = data.detach().float().contiguous()      # the extension's contract: contiguous NCHW float32
        rois = rois.detach().float().contiguous()
        offset = offset.detach().float().contiguous()
        n = rois.shape[0]
        output = data.new_empty(n, out_channels, out_size, out_size)
        output_count = data.new_empty(n, out_channels, out_size, out_size)
        deform_pool_cuda.deform_psroi_pooling_cuda_forward(
            data, rois, offset, output, output_count, ctx.no_trans, ctx.spatial_scale, ctx.out_channels,
            ctx.group_size, ctx.out_size, ctx.part_size, ctx.sample_per_part, ctx.trans_std)
        ctx.save_for_backward(data, rois, offset)
        ctx.output_count = output_count
        return output

    @staticmethod
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        data, rois, offset = ctx.saved_tensors
        grad_input = torch.zeros_like(data)
        grad_offset = torch.zeros_like(offset)
        deform_pool_cuda.deform_psroi_pooling_cuda_backward(
            grad_output.float().contiguous(), data, rois, offset, ctx.output_count, grad_input, grad_offset,
            ctx.no_trans, ctx.spatial_scale, ctx.out_channels, ctx.group_size, ctx.out_size, ctx.part_size,
            ctx.sample_per_part, ctx.trans_std)
        return (grad_input, None, grad_offset, None, None, None, None, None, None, None, None)


deform_roi_pooling = DeformRoIPoolingFunction.apply


class DeformRoIPooling(nn.Module):

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0):
        super(DeformRoIPooling, self).__init__()
        self.spatial_scale = spatial_scale
        self.out_size = out_size
        self.out_channels = out_channels
        self.no_trans = no_trans
        self.group_size = group_size
        self.part_size = out_size if part_size is None else part_size
        self.sample_per_part = sample_per_part
        self.trans_std = trans_std

    def _pool(self, data, rois, offset, no_trans):
        return deform_roi_pooling(data, rois, offset, self.spatial_scale, self.out_size, self.out_channels, no_trans,
                                  self.group_size, self.part_size, self.sample_per_part, self.trans_std)

    def forward(self, data, rois, offset):
        if self.no_trans:
            offset = data.new_empty(0)
        return self._pool(data, rois, offset, self.no_trans)


def _fc_stack(n_fcs, in_features, hidden, out_features, final_sigmoid=False):
    seq, ic = [], in_features
    for i in range(n_fcs):
        last = i == n_fcs - 1
        oc = out_features if last else hidden
        seq.append(Linear(ic, oc))
        ic = oc
        if not last:
            seq.append(nn.ReLU())   # not in place: the Linear output is a view produced by a custom Function
        elif final_sigmoid:
            seq.append(nn.Sigmoid())
    return nn.Sequential(*seq)


class DeformRoIPoolingPack(DeformRoIPooling):

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, num_offset_fcs=3, deform_fc_channels=1024):
        super(DeformRoIPoolingPack, self).__init__(spatial_scale, out_size, out_channels, no_trans, group_size,
                                                   part_size, sample_per_part, trans_std)
        self.num_offset_fcs = num_offset_fcs
        self.deform_fc_channels = deform_fc_channels
        if not no_trans:
            self.offset_fc = _fc_stack(num_offset_fcs, out_size * out_size * out_channels, deform_fc_channels,
                                       out_size * out_size * 2)
            self.offset_fc[-1].weight.data.zero_()
            self.offset_fc[-1].bias.data.zero_()

    def _offset(self, data, rois):
        n = rois.shape[0]
        x = self._pool(data, rois, data.new_empty(0), True)
        return x.view(n, -1)

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self._pool(data, rois, data.new_empty(0), True)
        x = self._offset(data, rois)
        offset = self.offset_fc(x).view(rois.shape[0], 2, self.out_size, self.out_size)
        return self._pool(data, rois, offset.float(), False)


class ModulatedDeformRoIPoolingPack(DeformRoIPoolingPack):

    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, num_offset_fcs=3, num_mask_fcs=2, deform_fc_channels=1024):
        super(ModulatedDeformRoIPoolingPack, self).__init__(
            spatial_scale, out_size, out_channels, no_trans, group_size, part_size, sample_per_part, trans_std,
            num_offset_fcs, deform_fc_channels)
        self.num_mask_fcs = num_mask_fcs
        if not no_trans:
            self.mask_fc = _fc_stack(num_mask_fcs, out_size * out_size * out_channels, deform_fc_channels,
                                     out_size * out_size, final_sigmoid=True)
            self.mask_fc[-2].weight.data.zero_()
            self.mask_fc[-2].bias.data.zero_()

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self._pool(data, rois, data.new_empty(0), True)
        n = rois.shape[0]
        x = self._offset(data, rois)
        offset = self.offset_fc(x).view(n, 2, self.out_size, self.out_size)
        mask = self.mask_fc(x).view(n, 1, self.out_size, self.out_size)
        return self._pool(data, rois, offset.float(), False) * mask.float()
