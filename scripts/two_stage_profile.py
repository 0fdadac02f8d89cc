"""rocprofv3 --kernel-trace target: pipelined two-stage inference at config 5's own sizes, 6 batches of 8 frames"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heterofusionrcnn_amd.two_stage import TwoStageDetector
from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
from bench import kitti_frustum, B, N0, IMG_H, IMG_W, IMG_C, KITTI_P2
torch.manual_seed(0)
det = TwoStageDetector().cuda().eval()
xyz = torch.from_numpy(kitti_frustum(np.random.default_rng(0), B, N0)).cuda()
inten = torch.zeros(B, N0, 1, device="cuda")
img = torch.randn(B, IMG_H, IMG_W, IMG_C, device="cuda")
cal = torch.from_numpy(KITTI_P2).cuda().repeat(B, 1, 1).contiguous()
pf = GeometryPrefetcher(det.geometry, depth=2)
pf.submit(xyz); pf.submit(xyz)
for i in range(8):
    if i == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    geo = pf.get(); pf.submit(xyz)
    det(xyz, inten, img, cal, geometry=geo)
torch.cuda.synchronize()
print("ms per batch", 1e3 * (time.perf_counter() - t0) / 6)
