"""nn.Module mirrors whose parameters keep the reference's names, shapes and default initialisation
(they subclass the torch modules the reference instantiates) while `forward` runs the HIP kernels."""
import math

import torch
import torch.nn as nn

from . import functional as F


# A/B switch (tools/): MEGREADER_BN_EPILOGUE=0 keeps BatchNorm's own statistics pass
BN_EPILOGUE = __import__("os").environ.get("MEGREADER_BN_EPILOGUE", "1") != "0"


# A/B switch (tools/): MEGREADER_FORK_RESIDUAL=0 leaves the shortcut's gradient add to autograd (an ATen add kernel per block)
FORK_RESIDUAL = __import__("os").environ.get("MEGREADER_FORK_RESIDUAL", "1") != "0"


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class Conv2d(nn.Conv2d):
    """nn.Conv2d drop-in (groups=1, zero padding) -- implicit-GEMM MFMA kernel, optional fused ReLU."""

    def __init__(self, *args, fuse_relu=False, relu_grad_downstream=False, **kwargs):
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise NotImplementedError("megreader_amd.nn.Conv2d supports groups=1 and explicit zero padding only")
        self.fuse_relu = fuse_relu
        # True only when the sole consumer is MaxPool2d(relu_input=True), which then applies the ReLU mask
        self.relu_grad_downstream = relu_grad_downstream and fuse_relu
        # physical KRSC layout: the wgrad kernel's output is then adopted as .grad without a re-layout
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    # set by the BatchNorm2d that consumed this layer's output in training mode (see BatchNorm2d.forward): from the next
    # forward on, the batch statistics come out of this convolution's GEMM epilogue
    feeds_batch_norm = False
    # set by the module that builds the graph (ResNet blocks, the CRNN backbone) when this convolution is the ONLY consumer of a
    # BatchNorm's output: the BatchNorm's backward reductions then ride in this convolution's dgrad epilogue (F.conv2d)
    sole_consumer_of_bn = False
    # the same guarantee for forward_fork(): x is consumed by this convolution and -- through the forked alias it hands back --
    # by nothing that autograd does not route through this node (identity shortcut, deformable conv)
    sole_consumer_when_forked = False

    def forward(self, x):
        y = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.fuse_relu,
                     self.relu_grad_downstream, bn_stats=self.feeds_batch_norm and self.training and BN_EPILOGUE,
                     sole_consumer_of_bn=self.sole_consumer_of_bn)
        y._mr_producer = self
        return y

    def forward_fork(self, x):
        """(self(x), x') with x' = x as a second output of this convolution's autograd node: a residual block passes x' to its
        identity shortcut, and the shortcut's gradient is then added in the epilogue of this convolution's dgrad
        (F.conv2d(fork=True)) instead of by an elementwise kernel over the block input's gradient."""
        if not FORK_RESIDUAL:     # x then has a second consumer outside this node: no BatchNorm-backward sums here
            y = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.fuse_relu,
                         self.relu_grad_downstream, bn_stats=self.feeds_batch_norm and self.training and BN_EPILOGUE)
            y._mr_producer = self
            return y, x
        y, xr = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.fuse_relu,
                         self.relu_grad_downstream, bn_stats=self.feeds_batch_norm and self.training and BN_EPILOGUE,
                         fork=True, sole_consumer_of_bn=self.sole_consumer_when_forked)
        y._mr_producer = self
        return y, xr


class FusedReLU(nn.Module):
    """Placeholder that keeps the reference's Sequential indices; the ReLU itself runs in the epilogue of
    the preceding Conv2d / BatchNorm2d (constructed with fuse_relu=True)."""

    def forward(self, x):
        return x


class BatchNorm2d(nn.BatchNorm2d):
    def __init__(self, *args, fuse_relu=False, **kwargs):
        super().__init__(*args, **kwargs)
        if not (self.affine and self.track_running_stats):
            raise NotImplementedError("megreader_amd.nn.BatchNorm2d requires affine=True, track_running_stats=True")
        if self.momentum is None:
            raise NotImplementedError("megreader_amd.nn.BatchNorm2d: momentum=None (cumulative moving average) is not "
                                      "used by any reference model; pass a float")
        self.fuse_relu = fuse_relu

    def forward(self, x, residual=None):
        momentum = self.momentum
        if self.training:
            # conv -> bn fusion is self-configuring: the first training forward marks the producing Conv2d, every later
            # one receives the batch statistics from that convolution's epilogue (F.conv2d(bn_stats=True)) and skips the
            # statistics pass over its input
            producer = getattr(x, "_mr_producer", None)
            if producer is not None and not producer.feeds_batch_norm and not producer.fuse_relu:
                producer.feeds_batch_norm = True
        # num_batches_tracked is advanced by the statistics kernel itself (no separate tiny launch per BN layer)
        return F.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, momentum,
                            self.eps, self.fuse_relu, residual,
                            self.num_batches_tracked if self.training else None)


class MaxPool2d(nn.MaxPool2d):
    def __init__(self, *args, relu_input=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.relu_input = relu_input  # input is a ReLU output whose backward mask this op applies

    def forward(self, x):
        if self.ceil_mode or _pair(self.dilation) != (1, 1) or self.return_indices:
            raise NotImplementedError("megreader_amd.nn.MaxPool2d: ceil_mode / dilation / return_indices unsupported")
        stride = self.kernel_size if self.stride is None else self.stride
        return F.max_pool2d(x, _pair(self.kernel_size), _pair(stride), _pair(self.padding), self.relu_input)


class Linear(nn.Linear):
    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class LSTM(nn.Module):
    """Single-layer bidirectional LSTM with nn.LSTM's parameter names (weight_ih_l0, ..., bias_hh_l0_reverse),
    gate order (i, f, g, o) and default initialisation U(-1/sqrt(H), 1/sqrt(H))."""

    def __init__(self, input_size, hidden_size, num_layers=1, bias=True, batch_first=False, dropout=0.0,
                 bidirectional=False):
        super().__init__()
        if num_layers != 1 or not bias or batch_first or dropout or not bidirectional:
            raise NotImplementedError("megreader_amd.nn.LSTM implements nn.LSTM(nIn, nHidden, bidirectional=True)")
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.bidirectional = 1, True
        for suffix in ("", "_reverse"):
            self.register_parameter("weight_ih_l0" + suffix, nn.Parameter(torch.empty(4 * hidden_size, input_size)))
            self.register_parameter("weight_hh_l0" + suffix, nn.Parameter(torch.empty(4 * hidden_size, hidden_size)))
            self.register_parameter("bias_ih_l0" + suffix, nn.Parameter(torch.empty(4 * hidden_size)))
            self.register_parameter("bias_hh_l0" + suffix, nn.Parameter(torch.empty(4 * hidden_size)))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.hidden_size) if self.hidden_size > 0 else 0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def flatten_parameters(self):  # API parity with nn.LSTM (reference decoders/crnn.py:91-92); nothing to do
        pass

    def forward(self, x, hx=None):
        if hx is not None:
            raise NotImplementedError("initial states are not used on the reference path")
        out = F.bilstm(x, self.weight_ih_l0, self.weight_hh_l0, self.bias_ih_l0, self.bias_hh_l0,
                       self.weight_ih_l0_reverse, self.weight_hh_l0_reverse, self.bias_ih_l0_reverse,
                       self.bias_hh_l0_reverse)
        return out, None


class AdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    def forward(self, x):
        size = self.output_size if isinstance(self.output_size, (tuple, list)) else (self.output_size,) * 2
        return F.adaptive_avg_pool2d(x, size)


class Dropout2d(nn.Dropout2d):
    def forward(self, x):
        return F.dropout2d(x, self.p, self.training)
