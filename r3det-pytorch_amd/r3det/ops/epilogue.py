"""Convolution epilogue for inference (include/r3det_hip.h: r3det_bias_act):
``y = act(y + bias[c] (+ residual))`` in place, one pass, NCHW or channels_last.

Not one of the reference's extension ops: it serves the model around the hot path
(models/fuse.py), after BatchNorm has been folded into the convolutions the way the reference's
own benchmark does (tools/analysis_tools/benchmark.py:88-89, mmcv.cnn.fuse_conv_bn).
"""
import torch

from .. import _C


def bias_act_(y, bias, residual=None, relu=True):
    """In place on ``y`` (N, C, H, W).  Shapes / layouts the kernel does not take (and CPU
    tensors) go through the equivalent torch ops."""
    if y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and bias is not None:
        N, C, H, W = y.shape
        if y.is_contiguous():
            outer, inner, fmt = N, H * W, torch.contiguous_format
        elif y.is_contiguous(memory_format=torch.channels_last):
            outer, inner, fmt = N * H * W, 1, torch.channels_last
        else:
            outer = None
        if outer is not None and (inner > 1 or C % 4 == 0) and (inner == 1 or N * C <= 65535):
            res = None
            if residual is not None:
                res = residual if residual.is_contiguous(memory_format=fmt) else residual.contiguous(memory_format=fmt)
            with torch.cuda.device(y.device):
                rc = _C.lib().r3det_bias_act(_C.ptr(y), _C.ptr(bias), _C.ptr(res), outer, C, inner, int(relu),
                                             _C.stream())
            if rc == 0:
                return y
            if rc != -1:  # R3DET_EINVAL = shape not taken; anything else is a real failure
                _C.check(rc, "r3det_bias_act")
    y.add_(bias.view(1, -1, 1, 1))
    if residual is not None:
        y.add_(residual)
    return y.relu_() if relu else y


def mix_to_nchw(a, b=None, bias_a=None, bias_b=None):
    """channels_last (N, C, H, W) tensors -> a new NCHW-contiguous ``(a + bias_a) + (b + bias_b)``
    (r3det_frm_mix_nchw); b and the biases are optional.  None when the inputs are not dense
    channels_last fp32 CUDA tensors (the caller then uses the torch ops)."""
    ok = lambda t: (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4  # noqa: E731
                    and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous())
    if not ok(a) or (b is not None and (not ok(b) or b.shape != a.shape)):
        return None
    N, C, H, W = a.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _C.check(_C.lib().r3det_frm_mix_nchw(_C.ptr(a), _C.ptr(b), _C.ptr(bias_a), _C.ptr(bias_b), N, C, H, W,
                                             _C.ptr(out), _C.stream()), "r3det_frm_mix_nchw")
    return out
