"""DB detection head on the HIP layers -- mirror of reference decoders/seg_detector.py:7-147 (`SegDetector`, the decoder
of experiments/seg_detector/seg_detector_db.yaml: adaptive=True, k=50).  Same constructor, same module / parameter names
and shapes (in5..in2, out5..out2, binarize.{0,1,3,4,6}, thresh.{0,1,3,4,6}) and the same initialisation
(kaiming_normal_ on every conv / deconv weight, BatchNorm weight 1 / bias 1e-4), so checkpoints interchange.

The 1x1 / 3x3 convolutions and BatchNorms run on the MFMA / HIP kernels; nn.Upsample(nearest) (+ the top-down add) is
mr_nearest_up_fwd / _bwd; ConvTranspose2d(k=2, s=2) is a GEMM (one output pixel quad = a [Cin] x [Cin, 4*Cout] product)
followed by a depth-to-space pass (megreader_amd.nn.functional.conv_transpose2x2).  The final sigmoid (always float32, `_SigmoidF32`) and the differentiable
binarisation 1 / (1 + exp(-k (x - y))) are elementwise torch ops on 1-channel maps.
`smooth=True` / `serial=True` are not used by any reference YAML and raise NotImplementedError."""
from collections import OrderedDict

import torch
import torch.nn as nn

from ..nn import BatchNorm2d, Conv2d, FusedReLU
from ..nn import functional as F


FUSED_TAIL = __import__("os").environ.get("MEGREADER_DB_TAIL_FUSED", "1") != "0"     # A/B: 0 = the torch expression


class ConvTranspose2x2(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(cin, cout, 2, 2) (weight [cin, cout, 2, 2], bias [cout]) computed as a GEMM:
    out[n, co, 2h+i, 2w+j] = sum_ci x[n, ci, h, w] * W[ci, co, i, j] + b[co]."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, 2, 2)

    def forward(self, x):
        if self.weight.is_cuda:
            return F.conv_transpose2x2(x, self.weight, self.bias)     # GEMM + depth-to-space, one autograd node (nn/functional.py)
        raise NotImplementedError("megreader_amd ops run only on an AMD GPU (HIP); there is no CPU fallback")


class _SigmoidF32(nn.Module):
    """The heads' final nn.Sigmoid of the reference (seg_detector.py:77-79), evaluated in float32 whatever the compute
    dtype: in bf16 sigmoid(x) rounds to exactly 1.0 from x ~ 6.2 (fp32: ~17), which puts confident pixels on BCE's
    log(1 - p) clamp (loss 100 per pixel, zero gradient through y * (1 - y)) and quantises `binary` / `thresh` to 2^-8
    in front of the k = 50 step function.  The maps have one channel: the cost is nil."""

    def forward(self, x):
        return torch.sigmoid(x.float())


class _Up(nn.Upsample):
    def forward(self, x):
        return F.upsample_nearest(x, int(self.scale_factor))


class SegDetector(nn.Module):
    def __init__(self, in_channels=[64, 128, 256, 512], inner_channels=256, k=10, bias=False, adaptive=False,
                 smooth=False, serial=False, *args, **kwargs):
        super(SegDetector, self).__init__()
        if smooth or serial:
            raise NotImplementedError("SegDetector(smooth / serial) is not used by any reference experiment")
        self.k = k
        self.serial = serial
        self.up5 = _Up(scale_factor=2, mode='nearest')
        self.up4 = _Up(scale_factor=2, mode='nearest')
        self.up3 = _Up(scale_factor=2, mode='nearest')
        self.in5 = Conv2d(in_channels[-1], inner_channels, 1, bias=bias)
        self.in4 = Conv2d(in_channels[-2], inner_channels, 1, bias=bias)
        self.in3 = Conv2d(in_channels[-3], inner_channels, 1, bias=bias)
        self.in2 = Conv2d(in_channels[-4], inner_channels, 1, bias=bias)
        q = inner_channels // 4
        self.out5 = nn.Sequential(Conv2d(inner_channels, q, 3, padding=1, bias=bias), _Up(scale_factor=8, mode='nearest'))
        self.out4 = nn.Sequential(Conv2d(inner_channels, q, 3, padding=1, bias=bias), _Up(scale_factor=4, mode='nearest'))
        self.out3 = nn.Sequential(Conv2d(inner_channels, q, 3, padding=1, bias=bias), _Up(scale_factor=2, mode='nearest'))
        self.out2 = Conv2d(inner_channels, q, 3, padding=1, bias=bias)
        self.binarize = self._head(inner_channels, bias)
        self.binarize.apply(self.weights_init)
        self.adaptive = adaptive
        if adaptive:
            self.thresh = self._head(inner_channels, bias)
            self.thresh.apply(self.weights_init)
        for m in (self.in5, self.in4, self.in3, self.in2, self.out5, self.out4, self.out3, self.out2):
            m.apply(self.weights_init)

    @staticmethod
    def _head(inner_channels, bias):
        q = inner_channels // 4
        return nn.Sequential(Conv2d(inner_channels, q, 3, padding=1, bias=bias), BatchNorm2d(q, fuse_relu=True),
                             FusedReLU(), ConvTranspose2x2(q, q), BatchNorm2d(q, fuse_relu=True), FusedReLU(),
                             ConvTranspose2x2(q, 1), _SigmoidF32())

    def weights_init(self, m):
        classname = m.__class__.__name__
        if classname.find('Conv') != -1:
            # same random stream as the reference: draw into a row-major tensor (the HIP Conv2d keeps its weight in
            # channels_last memory, where normal_() takes another CPU kernel and consumes the generator differently)
            w = torch.empty(m.weight.shape, dtype=m.weight.dtype)
            nn.init.kaiming_normal_(w)
            m.weight.data.copy_(w)
        elif classname.find('BatchNorm') != -1:
            m.weight.data.fill_(1.)
            m.bias.data.fill_(1e-4)

    def forward(self, features, gt=None, masks=None, training=False):
        c2, c3, c4, c5 = features
        in5 = self.in5(c5)
        in4 = self.in4(c4)
        in3 = self.in3(c3)
        in2 = self.in2(c2)
        out4 = F.upsample_nearest(in5, 2, in4)    # self.up5(in5) + in4, 1/16
        out3 = F.upsample_nearest(out4, 2, in3)   # 1/8
        out2 = F.upsample_nearest(out3, 2, in2)   # 1/4
        p5 = self.out5(in5)
        p4 = self.out4(out4)
        p3 = self.out3(out3)
        p2 = self.out2(out2)
        fuse = F.cat_channels([p5, p4, p3, p2])
        if (self.adaptive and FUSED_TAIL and fuse.is_cuda and isinstance(self.binarize[-1], _SigmoidF32) and
                isinstance(self.thresh[-1], _SigmoidF32)):
            # the two heads up to their last deconvolution, then sigmoid / sigmoid / step function as ONE launch each way
            xb, xt = self.binarize[:-1](fuse), self.thresh[:-1](fuse)
            binary, thresh, thresh_binary = F.db_head_tail(xb, xt, self.k)
            return OrderedDict(binary=binary, thresh=thresh, thresh_binary=thresh_binary)
        binary = self.binarize(fuse).float()
        result = OrderedDict(binary=binary)
        if self.adaptive:
            thresh = self.thresh(fuse).float()
            thresh_binary = self.step_function(binary, thresh)
            result.update(thresh=thresh, thresh_binary=thresh_binary)
        return result

    def step_function(self, x, y):
        return torch.reciprocal(1 + torch.exp(-self.k * (x - y)))
