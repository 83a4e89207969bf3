"""Host time of the training-side FR node (FeatureRefineModuleLevelsFunction, five levels, N = 2, C = 256): per call host /
wall time, the forward alone, cProfile of 50 calls, and an autograd node of the same signature that launches nothing --
torch's own cost for the call pattern (bench.py reports the same floor as fr_null_node_ms_wall).  CL=1: channels_last."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "r3det-pytorch_amd")):
    sys.path.insert(0, p)
import torch
from r3det import synthetic as syn
from r3det.ops.feature_refine import feature_refine_module_levels
dev = torch.device("cuda")
feats, boxes = syn.fr_pyramid(2, 256, 9, device=dev)
if os.environ.get("CL") == "1":
    feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
xs = [f.clone().requires_grad_(True) for f in feats]
gs = [torch.randn_like(f) for f in feats]
as_ = [torch.randn_like(f).requires_grad_(True) for f in feats]
bs_ = [torch.randn_like(f).requires_grad_(True) for f in feats]
scales = [1.0 / s for s in syn.STRIDES]
def fr():
    for t in xs + as_ + bs_:
        t.grad = None
    torch.autograd.backward(feature_refine_module_levels(as_, bs_, xs, boxes, scales, 1), gs)
def fwd_only():
    with torch.no_grad():
        feature_refine_module_levels(as_, bs_, xs, boxes, scales, 1)
for _ in range(5): fr()
torch.cuda.synchronize()
def wall(fn, n=50):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    return th / n * 1e3, (time.perf_counter() - t) / n * 1e3
print("fr(): host %.3f ms, wall %.3f ms per call" % wall(fr))
print("forward only (no_grad): host %.3f ms, wall %.3f ms" % wall(fwd_only))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): fr()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)

class Null(torch.autograd.Function):
    """The same signature with no library call: what torch's autograd machinery costs for a node of this shape."""
    @staticmethod
    def forward(ctx, scales, points, n, *tensors):
        ctx.n = n
        return tuple(torch.empty_like(t) for t in tensors[2 * n:3 * n])
    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        ds = tuple(torch.empty_like(g) for g in grads)
        return (None, None, None) + ds + ds + tuple(grads) + (None,) * n
def null():
    for t in xs + as_ + bs_:
        t.grad = None
    torch.autograd.backward(Null.apply(tuple(scales), 1, 5, *as_, *bs_, *xs, *boxes), gs)
for _ in range(5): null()
print("null node of the same signature: host %.3f ms, wall %.3f ms per call" % wall(null))
def clear():
    for t in xs + as_ + bs_:
        t.grad = None
print("clearing 15 grads: host %.3f ms" % wall(clear)[0])
