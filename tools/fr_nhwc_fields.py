import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")): sys.path.insert(0, p)
import torch
from r3det import synthetic as syn
from r3det.ops.feature_refine import fr_module_nhwc, fr_forward_nhwc
dev = torch.device("cuda")
N, C, H = 4, 256, 128
cl = torch.channels_last
# FR_PAD=bytes: a pad allocation between the tensors, so that a / b / res / out do not start 2^26 bytes apart
# (the same DRAM channel / bank bits for the four rows of a position)
PAD = int(os.environ.get("FR_PAD", "0"))
_pads = []
def _mk():
    if PAD:
        _pads.append(torch.empty(PAD, dtype=torch.uint8, device=dev))
    return torch.randn(N, C, H, H, device=dev).contiguous(memory_format=cl)
sets = [tuple(_mk() for _ in range(4)) for _ in range(3)]
print("pad", PAD, "address deltas of set 0 (bytes):", [sets[0][i + 1].data_ptr() - sets[0][i].data_ptr() for i in range(3)], flush=True)
ba, bb = torch.randn(C, device=dev), torch.randn(C, device=dev)
ONLY = os.environ.get("FR_FIELD", "")  # run one field only (PMC passes average over a kernel's launches)
def run(name, boxes, scale=0.125):
    if ONLY and ONLY not in name:
        return
    for i in range(6):
        a, b, r, o = sets[i % 3]; fr_module_nhwc(a, b, ba, bb, r, boxes, scale, 1, o)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(30):
        a, b, r, o = sets[i % 3]; fr_module_nhwc(a, b, ba, bb, r, boxes, scale, 1, o)
    e.record(); torch.cuda.synchronize()
    print(f"{name:34s} {s.elapsed_time(e) * 1000 / 30:7.1f} us", flush=True)
boxes = syn.fr_level_boxes(N, H, H, 8, 3, device=dev)
run("jittered field (bench)", boxes)
ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(H, device=dev), indexing="ij")
reg = torch.zeros(N, H, H, 5, device=dev)
reg[..., 0] = (xs + 0.5) * 8; reg[..., 1] = (ys + 0.5) * 8; reg[..., 2] = 30; reg[..., 3] = 10
run("regular field (transposed taps)", reg.view(-1, 5).contiguous())
sw = reg.clone(); sw[..., 0] = (ys + 0.5) * 8; sw[..., 1] = (xs + 0.5) * 8
run("swapped centres (taps = own cell)", sw.view(-1, 5).contiguous())
far = reg.clone(); far[..., 0] = -1000; far[..., 1] = -1000
run("all samples out of range", far.view(-1, 5).contiguous())
z = reg.clone(); z[..., 0] = 4; z[..., 1] = 4
run("all sample the same cell", z.view(-1, 5).contiguous())
