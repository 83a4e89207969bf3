"""The convolution epilogue (r3det_bias_act, channels_last) alone, on the activation sizes of the ResNet-50 at batch
4 x 1024^2, buffers rotating beyond the Infinity Cache: us per call and bytes / s against the 8 (12 with a residual)
bytes per element it has to move."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from r3det.ops.epilogue import bias_act_  # noqa: E402

dev = torch.device("cuda")
N = 4
cases = [(256, 256, True), (512, 128, True), (1024, 64, True), (2048, 32, True), (64, 256, False), (64, 512, False),
         (128, 128, False), (256, 128, False), (256, 64, False), (512, 32, False)]
for C, H, res in cases:
    elems = N * C * H * H
    nbuf = max(2, int(1.2e9 // (elems * 4 * (2 if res else 1))))
    ys = [torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)]
    rs = [torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(nbuf)] if res else None
    b = torch.randn(C, device=dev)
    for i in range(nbuf):
        bias_act_(ys[i], b, rs[i] if res else None)
    torch.cuda.synchronize()
    reps = max(20, nbuf)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(reps):
        bias_act_(ys[i % nbuf], b, rs[i % nbuf] if res else None)
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) * 1000 / reps
    byts = elems * (12 if res else 8)
    print(f"C={C:5d} {H:4d}^2 residual={res!s:5}: {us:7.1f} us per call  {byts / 1e6:7.1f} MB  {byts / us / 1e6:6.2f} TB/s", flush=True)
