"""Inference-time graph rewrites of the detector around the hot path.

``fuse_conv_bn``: fold every BatchNorm2d that follows a convolution into that convolution's
weight and bias -- what the reference's benchmark and test tools do with ``--fuse-conv-bn``
(tools/analysis_tools/benchmark.py:9,31,88-89; tools/test.py:11,33: mmcv.cnn.fuse_conv_bn).
``fuse_epilogues``: run what remains after each convolution (bias, ReLU, the bottleneck's residual
add) as ONE in-place pass (ops/epilogue.py) instead of two or three elementwise launches.
Both keep fp32 and change results only by rounding (tests/test_model.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops.epilogue import bias_act_
from .backbone import Bottleneck, ConvModule, ResNet50


def _fold(conv, bn):
    """mmcv.cnn.utils.fuse_conv_bn._fuse_conv_bn: w' = w * gamma / sqrt(var + eps),
    b' = (b - mean) * gamma / sqrt(var + eps) + beta."""
    w = conv.weight
    b = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
    factor = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                      conv.dilation, conv.groups, bias=True).to(w.device)
    fused.weight = nn.Parameter(w * factor.reshape(-1, 1, 1, 1), requires_grad=False)
    fused.bias = nn.Parameter((b - bn.running_mean) * factor + bn.bias, requires_grad=False)
    return fused


@torch.no_grad()
def fuse_conv_bn(model):
    for m in model.modules():
        if isinstance(m, Bottleneck):
            for i in (1, 2, 3):
                conv, bn = getattr(m, f'conv{i}'), getattr(m, f'bn{i}')
                if isinstance(bn, nn.BatchNorm2d):
                    setattr(m, f'conv{i}', _fold(conv, bn))
                    setattr(m, f'bn{i}', nn.Identity())
            if m.downsample is not None and isinstance(m.downsample[1], nn.BatchNorm2d):
                m.downsample = nn.Sequential(_fold(m.downsample[0], m.downsample[1]), nn.Identity())
        elif isinstance(m, ResNet50) and isinstance(m.bn1, nn.BatchNorm2d):
            m.conv1 = _fold(m.conv1, m.bn1)
            m.bn1 = nn.Identity()
    return model


def _conv_nobias(conv, x):
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def _bottleneck_forward(self, x):
    if self.downsample is None:
        idt = x
    else:
        d = self.downsample[0]
        idt = bias_act_(_conv_nobias(d, x), d.bias, None, relu=False)
    out = bias_act_(_conv_nobias(self.conv1, x), self.conv1.bias)
    out = bias_act_(_conv_nobias(self.conv2, out), self.conv2.bias)
    return bias_act_(_conv_nobias(self.conv3, out), self.conv3.bias, idt)


def _stem_forward(self, x):
    x = self.maxpool(bias_act_(_conv_nobias(self.conv1, x), self.conv1.bias))
    outs = []
    for i in range(4):
        x = getattr(self, f'layer{i + 1}')(x)
        outs.append(x)
    return tuple(outs)


def _convmodule_forward(self, x):
    return bias_act_(_conv_nobias(self.conv, x), self.conv.bias, None, relu=self.act)


@torch.no_grad()
def fuse_epilogues(model):
    """Requires fuse_conv_bn (every backbone convolution then has a bias)."""
    for m in model.modules():
        if isinstance(m, Bottleneck) and isinstance(m.bn1, nn.Identity):
            m.forward = _bottleneck_forward.__get__(m)
        elif isinstance(m, ResNet50) and isinstance(m.bn1, nn.Identity):
            m.forward = _stem_forward.__get__(m)
        elif isinstance(m, ConvModule) and m.conv.bias is not None:
            m.forward = _convmodule_forward.__get__(m)
    return model


def fuse_for_inference(model):
    return fuse_epilogues(fuse_conv_bn(model.eval()))
