"""Diagnostic: every aten mm / addmm / bmm of one two-stage inference batch (config 5's own sizes) with its shape, device time and
achieved TFLOP/s -- which layer the long library GEMM of profiles/r03_two_stage_kernel_stats.csv is."""
import os, sys, collections
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd.two_stage import TwoStageDetector
from bench import kitti_frustum, B, N0, IMG_H, IMG_W, IMG_C, KITTI_P2
torch.manual_seed(0)
det = TwoStageDetector().cuda().eval()
xyz = torch.from_numpy(kitti_frustum(np.random.default_rng(0), B, N0)).cuda()
inten = torch.zeros(B, N0, 1, device="cuda")
img = torch.randn(B, IMG_H, IMG_W, IMG_C, device="cuda")
cal = torch.from_numpy(KITTI_P2).cuda().repeat(B, 1, 1).contiguous()
recs = []
class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if name.split(".")[0] in ("mm", "addmm", "bmm", "baddbmm"):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); out = func(*args, **(kwargs or {})); e1.record()
            ts = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            recs.append((name.split(".")[0], tuple(ts), e0, e1))
            return out
        return func(*args, **(kwargs or {}))
with torch.no_grad():
    for _ in range(2):
        det(xyz, inten, img, cal)
    torch.cuda.synchronize()
    with Rec():
        det(xyz, inten, img, cal)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, ts, e0, e1 in recs:
    a = agg.setdefault((name, ts), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
def flops(ts):
    a, b = ts[-2], ts[-1]
    return 2.0 * (a[0] if len(a) == 3 else 1) * a[-2] * a[-1] * b[-1]
tot = sum(v[1] for v in agg.values())
print("library GEMM calls %d, total %.0f us" % (len(recs), tot))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-6s %-48s x%d %8.0f us %7.1f TFLOP/s" % (k[0], str(k[1]), v[0], v[1], flops(k[1]) * v[0] / (v[1] * 1e-6) / 1e12))
