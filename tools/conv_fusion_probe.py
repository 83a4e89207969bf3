"""Does MIOpen's fused convolution + bias + ReLU (torch.miopen_convolution_relu: a fusion plan) beat the plain convolution
followed by the one-pass epilogue (r3det_bias_act) on the model's own shapes?  fp32, channels_last, batch 4 x 1024^2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from r3det.ops.epilogue import bias_act_  # noqa: E402

torch.backends.cudnn.benchmark = bool(int(os.environ.get("BENCHMARK", "1")))
dev = torch.device("cuda")
N = 4
shapes = [  # (Cin, Cout, k, stride, H_in, what)
    (3, 64, 7, 2, 1024, "stem"),
    (64, 64, 1, 1, 256, "l1 1x1a"), (64, 64, 3, 1, 256, "l1 3x3"), (64, 256, 1, 1, 256, "l1 1x1b"), (256, 64, 1, 1, 256, "l1 1x1a'"),
    (256, 128, 1, 1, 256, "l2 1x1a"), (128, 128, 3, 2, 256, "l2 3x3 s2"), (128, 128, 3, 1, 128, "l2 3x3"), (128, 512, 1, 1, 128, "l2 1x1b"), (512, 128, 1, 1, 128, "l2 1x1a'"),
    (256, 256, 3, 1, 64, "l3 3x3"), (256, 1024, 1, 1, 64, "l3 1x1b"), (1024, 256, 1, 1, 64, "l3 1x1a'"),
    (512, 512, 3, 1, 32, "l4 3x3"), (512, 2048, 1, 1, 32, "l4 1x1b"), (2048, 512, 1, 1, 32, "l4 1x1a'"),
    (256, 256, 3, 1, 128, "head P3"), (256, 256, 3, 1, 64, "head P4"), (256, 256, 3, 1, 32, "head P5"), (256, 256, 3, 1, 16, "head P6"),
]


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n


tot = [0.0, 0.0, 0.0]
for cin, cout, k, s, h, what in shapes:
    x = torch.randn(N, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev)
    pad = k // 2
    plain = lambda: F.conv2d(x, w, None, s, pad)
    ours = lambda: bias_act_(F.conv2d(x, w, None, s, pad), b)
    try:
        fused = lambda: torch.miopen_convolution_relu(x, w, b, [s, s], [pad, pad], [1, 1], 1)
        y1, y2 = ours(), fused()
        err = float((y1 - y2).abs().max() / y1.abs().max().clamp_min(1e-6))
        t = (timed(plain), timed(ours), timed(fused))
        cl = y2.is_contiguous(memory_format=torch.channels_last)
        print(f"{what:10s} {cin:4d}->{cout:4d} k{k} s{s} {h:4d}^2: conv {t[0]:7.1f}  conv+epilogue {t[1]:7.1f}  fused {t[2]:7.1f} us   rel.err {err:.1e}  out channels_last {cl}", flush=True)
        for i in range(3):
            tot[i] += t[i]
    except Exception as e:  # noqa: BLE001
        print(f"{what:10s}: fused form failed: {str(e)[:120]}", flush=True)
print(f"sum: conv {tot[0]:.0f}  conv+epilogue {tot[1]:.0f}  fused {tot[2]:.0f} us")
