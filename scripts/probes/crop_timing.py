"""Diagnostic: pc_crop_and_sample at the bench shape (512 RoIs x 512 points x 288 channels from 8 clouds of 16384 points), device
time per call and output bandwidth.  A/B against another build: HFOPS_LIBRARY=<path> python scripts/probes/crop_timing.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import modules
from bench import kitti_uniform, time_op, B, N0
rng = np.random.default_rng(0)
xyz = torch.from_numpy(kitti_uniform(rng, B, N0)).cuda()
nroi = 64 * B
b3 = torch.zeros(nroi, 7, device="cuda")
b3[:, 0] = torch.from_numpy(rng.uniform(-35, 35, nroi).astype(np.float32)).cuda()
b3[:, 1] = 3.0
b3[:, 2] = torch.from_numpy(rng.uniform(5, 65, nroi).astype(np.float32)).cuda()
b3[:, 3:6] = torch.tensor([5.9, 3.6, 8.0], device="cuda")
b3[:, 6] = torch.from_numpy(rng.uniform(-np.pi, np.pi, nroi).astype(np.float32)).cuda()
boxes8 = modules.box_3d_to_box_8co(b3).contiguous()
box_ind = (torch.arange(nroi, device="cuda") // 64).to(torch.int32)
inten = torch.rand(B, N0, 1, device="cuda")
msk = torch.rand(B, N0, device="cuda") < 0.1
for c in (288, 64, 6):
    fts = torch.randn(B, N0, c, device="cuda")
    us = time_op(lambda: hf.pc_crop_and_sample(xyz, fts, inten, msk, boxes8, box_ind, 512), iters=20, warm=3)
    print("C = %3d: %.1f us, %.2f TB/s of output" % (c, us, nroi * 512 * (3 + c + 1) * 4 / us / 1e6), flush=True)
