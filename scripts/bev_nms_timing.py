"""compute_bev_iou at 70 000 x 64 (BASELINE config 3) and oriented_nms at 9000 boxes (rpn pre_nms_size): bursts for
`rocprofv3 --kernel-trace` (scripts/parse_trace.py splits bev_iou_kernel / nms_mask_kernel / nms_sweep_kernel)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import heterofusionrcnn_amd as hf
from bench import rand_bev, time_op
rng = np.random.default_rng(3)
a = torch.from_numpy(rand_bev(rng, 70000)).cuda()
g = torch.from_numpy(rand_bev(rng, 64)).cuda()
from heterofusionrcnn_amd import _lib
L = _lib.lib()
def burst_bev(na, launches=200):
    """direct C-ABI launches, outputs preallocated: kernel + launch gap (the Python wrapper costs more than the kernel)"""
    aa = a[:na].contiguous()
    ov = torch.empty((na, 64), dtype=torch.float32, device="cuda"); io = torch.empty_like(ov)
    st = torch.cuda.current_stream().cuda_stream
    args = (na, aa.data_ptr(), 64, g.data_ptr(), ov.data_ptr(), io.data_ptr(), st)
    for _ in range(10):
        assert L.hf_compute_bev_iou(*args) == 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        L.hf_compute_bev_iou(*args)
    e1.record(); torch.cuda.synchronize()
    return round(1e3 * e0.elapsed_time(e1) / launches, 2)
res = {"bev_iou_70000x64_us": burst_bev(70000), "bev_iou_65536x64_us": burst_bev(65536), "bev_iou_32768x64_us": burst_bev(32768)}
res["bev_iou_Gboxpairs_per_s"] = round(70000 * 64 / res["bev_iou_70000x64_us"] / 1e3, 1)
for seed, thresh in ((4, 0.8), (4, 0.01)):
    r2 = np.random.default_rng(seed)
    base = rand_bev(r2, 300)
    boxes = np.repeat(base, 30, 0)
    boxes[:, [0, 2]] += r2.normal(0, 0.3, (9000, 1)).astype(np.float32)
    boxes[:, [1, 3]] += r2.normal(0, 0.3, (9000, 1)).astype(np.float32)
    boxes[:, 4] += r2.normal(0, 0.1, 9000).astype(np.float32)
    nb = torch.from_numpy(boxes.astype(np.float32)).cuda()
    res["oriented_nms_9000_clustered_t%.2f_us" % thresh] = round(time_op(lambda: hf.oriented_nms(nb, thresh), iters=10, warm=2), 1)
nb = torch.from_numpy(rand_bev(rng, 9000)).cuda()
res["oriented_nms_9000_uniform_t0.80_us"] = round(time_op(lambda: hf.oriented_nms(nb, 0.8), iters=10, warm=2), 1)
import time
def wall(fn, iters=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return round(1e6 * (time.perf_counter() - t0) / iters, 1)
res["oriented_nms_9000_uniform_t0.80_wallclock_us"] = wall(lambda: hf.oriented_nms(nb, 0.8))
os.environ["HF_NMS_STOP"] = "1"
res["nms_mask_alone_uniform_us"] = wall(lambda: hf.oriented_nms(nb, 0.8))
nbc = torch.from_numpy(boxes.astype(np.float32)).cuda()
res["nms_mask_alone_clustered_us"] = wall(lambda: hf.oriented_nms(nbc, 0.01))
os.environ.pop("HF_NMS_STOP")
print(json.dumps(res, indent=1))
if os.environ.get("PHASES"):
    # cumulative kernel time up to a phase boundary (HF_BEV_STOP: 1 launch only, 2 + box precompute, 3 + circle filter and
    # zero stores, 4 + separating-axis filter, 0 everything)
    # 5: + edge crossings / corners of the clip, 6: + centroid, 7: + angles
    # 8: everything but the zero stores
    for stop in (1, 2, 3, 4, 5, 6, 7, 8, 0):
        os.environ["HF_BEV_STOP"] = str(stop)
        print("bev_iou stop", stop, burst_bev(70000), "us", " half grid:", burst_bev(32768), "us")
    os.environ.pop("HF_BEV_STOP")
