import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import _lib
from bench import kitti_uniform
L = _lib.lib()
rng = np.random.default_rng(0)
for B in (8, 80):
    xyz = torch.from_numpy(kitti_uniform(rng, B, 16384)).cuda()
    m, K = 4096, 32
    new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(m, xyz))
    idx = torch.empty((B, m, K), dtype=torch.int32, device="cuda"); cnt = torch.empty((B, m), dtype=torch.int32, device="cuda")
    grouped = torch.empty((B, m, K, 3), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    args = (B, 16384, m, 0.5, K, xyz.data_ptr(), new_xyz.data_ptr(), 1, idx.data_ptr(), cnt.data_ptr(), grouped.data_ptr(), st)
    ref = None
    for q in ("128", "256", "64"):
        os.environ["HF_QBP_QPW"] = q
        idx.fill_(-1); rc = L.hf_query_ball_group_xyz(*args); torch.cuda.synchronize()
        out = (idx.clone(), cnt.clone(), grouped.clone())
        if ref is None: ref = out
        same = all(torch.equal(a, b) for a, b in zip(ref, out))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): L.hf_query_ball_group_xyz(*args)
        e1.record(); torch.cuda.synchronize()
        print("B", B, "qpw", q, "rc", rc, "same", same, "us", round(10 * e0.elapsed_time(e1), 2))
