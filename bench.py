#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the hot path.

  python bench.py --gpus N --steps K --warmup W [--scaling auto|weak|strong]

Default (--scaling auto) = BASELINE config 4's shape: a GLOBAL batch of 8 frames split over the N ranks (strong scaling; 8 frames on
one GPU, one frame per GPU at N = 8), with the reference's own fixed-per-rank-batch mode (weak: 8 frames per GPU) measured
right after it and reported under extra.weak_scaling when N > 1.

A "step" is one RPN train step (BASELINE.json: "KITTI frames/sec RPN train step") of hf/configs/rpn_multiclass.config
(BASELINE.json configs[3]) over one batch of synthetic KITTI-shaped frames resident in HBM: the PointCNN backbone (5 X-Conv +
6 X-DeConv layers, fc 256/256), the projection of the points into the image and the gather of the image branch's feature map
under them, path drop + 'concat' fusion, the segmentation head, the three-class bin-based box head, the target encoding and the
focal / softmax / smooth-L1 losses of hf/core/models/rpn_model.py, backward (down to the gradient of the image feature map),
Adam -- heterofusionrcnn_amd/rpn.py, fp32.  The image branch's VGG pyramid (a stock convolutional network, no custom op) is
NOT run: its output, a (B,360,1200,32) feature map, is a resident synthetic input and the step ends with its gradient.
The step is replayed from a captured hipGraph (heterofusionrcnn_amd/graph_step.py; --no-graph: eager launches).
N > 1: one process per GPU (this script starts them itself when no launcher did), gradients averaged over RCCL
(hf/experiments/mpi_run_training.sh + hvd.DistributedOptimizer, hf/core/trainer.py:71).  --workload rpn selects the
PointNet++ configuration (rpn_cars_pointnet_paper.config), --workload stack the round-1 workload (BASELINE.json configs[1]).

Rank 0 prints ONE JSON line: frames/s (whole job), plus
  roofline      the fused query_ball_point+group_point kernel at the headline shape
                (B=8, N=16384, M=4096, K=32), timed live with HIP events on its launch stream;
  cpu_baseline  the CPU oracle's op chain for the same backbone (no GPU), 1 core and all cores, bounded sample;
  extra         per-op device times and the bev_iou Gboxpairs/s figure BASELINE.json also names.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SURVEY_TWO_OP_BYTES = 24641536  # query_ball_point 6 291 456 + group_point 18 350 080 at B=8, N=16384, M=4096, K=32, C=3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

B, N0, KNN = 8, 16384, 32
SA = ((4096, 0.5, 32, (32, 32, 64)), (1024, 1.0, 32, (64, 96, 128)), (256, 2.0, 32, (128, 196, 256)))
FP = ((256, 256), (256, 256), (128, 128))


def kitti_uniform(rng, b, n):
    lo = np.array([-40.0, -5.0, 0.0], np.float32)
    hi = np.array([40.0, 3.0, 70.0], np.float32)
    return (lo + (hi - lo) * rng.random((b, n, 3), dtype=np.float32)).astype(np.float32)


KITTI_P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]],
                    np.float32)   # the left colour camera's projection matrix of a KITTI object frame
IMG_H, IMG_W, IMG_C = 360, 1200, 32   # rpn_multiclass.config:14-15 and vgg_conv1[1] (:122), the width of pyramid_fusion1


def kitti_frustum(rng, b, n, p2=KITTI_P2, h=IMG_H, w=IMG_W):
    """uniform points of the KITTI extents that project INSIDE the image, as the reference's loader keeps them
    (kitti_dataset.py:341-371 filters the cloud to the camera's field of view before sampling pc_sample_pts points)"""
    out = np.empty((b, n, 3), np.float32)
    for i in range(b):
        got = 0
        while got < n:
            c = kitti_uniform(rng, 1, 4 * n)[0]
            d = p2[2, 0] * c[:, 0] + p2[2, 1] * c[:, 1] + p2[2, 2] * c[:, 2] + p2[2, 3]
            u = (p2[0, 0] * c[:, 0] + p2[0, 1] * c[:, 1] + p2[0, 2] * c[:, 2] + p2[0, 3]) / d
            v = (p2[1, 0] * c[:, 0] + p2[1, 1] * c[:, 1] + p2[1, 2] * c[:, 2] + p2[1, 3]) / d
            c = c[(d > 0.1) & (u >= 0) & (u < w) & (v >= 0) & (v < h)]
            take = min(n - got, len(c))
            out[i, got:got + take] = c[:take]
            got += take
    return out


def rand_bev(rng, n):
    cx, cz = rng.uniform(-40, 40, n), rng.uniform(0, 70, n)
    l, w = np.clip(rng.normal(3.9, 0.4, n), 0.5, None), np.clip(rng.normal(1.6, 0.1, n), 0.5, None)
    ry = rng.uniform(-np.pi, np.pi, n)
    return np.stack([cx - l / 2, cz - w / 2, cx + l / 2, cz + w / 2, ry], 1).astype(np.float32)


def ball_group_bytes(b, n, m, k):
    """algorithmic bytes of ONE fused launch: read xyz1, xyz2 once; write idx, pts_cnt, grouped_xyz once
    (qbp 4*3*B*(N+M) + 4*B*M*(K+1), plus the grouped (B,M,K,3) output; the fused kernel
    never re-reads idx or xyz, so those two terms of the two-op sum 24 641 536 B are not counted)."""
    return 4 * 3 * b * (n + m) + 4 * b * m * (k + 1) + 4 * b * m * k * 3


class EventTimer:
    """HIP events on the current torch stream around selected launches inside the timed region."""

    def __init__(self):
        self.pairs = []
        self.enabled = False

    def wrap(self, fn, match):
        def inner(*args, **kwargs):
            if self.enabled and match(*args, **kwargs):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*args, **kwargs)
                e1.record()
                self.pairs.append((e0, e1))
                return out
            return fn(*args, **kwargs)
        return inner

    def mean_us(self):
        if not self.pairs:
            return None
        return 1e3 * sum(a.elapsed_time(b) for a, b in self.pairs) / len(self.pairs)


def time_op(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters  # us


def time_op_stats(fn, iters=30, warm=5):
    """median / p10 / p90 of per-call device time (one event pair per call; includes the launch gap)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    us = np.array([1e3 * a.elapsed_time(b) for a, b in evs])
    return {"median": round(float(np.median(us)), 2), "p10": round(float(np.percentile(us, 10)), 2),
            "p90": round(float(np.percentile(us, 90)), 2)}


def batched_kernel_burst(hf, clouds=80, launches=100):
    """the same op on the batched shape the train step launches it with (the clouds of a geometry group in one call), alone on
    the device, through hf_query_ball_group_xyz_ws with a workspace as the Python layer calls it: at this size the library
    builds the cell structure once per cloud (ballquery_sorted.hip: two kernels per call); the single-launch kernel is timed
    next to it"""
    from heterofusionrcnn_amd import _lib
    L = _lib.lib()
    m = SA[0][0]
    xyz = torch.from_numpy(kitti_uniform(np.random.default_rng(2000), clouds, N0)).cuda()
    new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(m, xyz))
    idx = torch.empty((clouds, m, KNN), dtype=torch.int32, device="cuda")
    cnt = torch.empty((clouds, m), dtype=torch.int32, device="cuda")
    grouped = torch.empty((clouds, m, KNN, 3), dtype=torch.float32, device="cuda")
    nbytes = L.hf_ball_query_workspace(clouds, N0)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    out = {"clouds_per_launch": clouds}
    nbytes_alg = SURVEY_TWO_OP_BYTES * clouds / B
    for name, variant in (("auto", 0), ("single_launch_cell_kernel", 1)):
        args = (variant, clouds, N0, m, SA[0][1], KNN, xyz.data_ptr(), new_xyz.data_ptr(), 1, idx.data_ptr(), cnt.data_ptr(),
                grouped.data_ptr(), ws.data_ptr(), nbytes, st)
        for _ in range(5):
            assert L.hf_query_ball_group_xyz_ws(*args) == 0
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            L.hf_query_ball_group_xyz_ws(*args)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / launches
        r = {"avg_call_us": round(us, 2), "us_per_cloud": round(us / clouds, 3),
             "achieved_GBs": round(nbytes_alg / (us * 1e-6) / 1e9, 1), "frac": round(nbytes_alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if name == "auto":
            out.update(r)
            out["kernels_per_call"] = "bq_build_kernel + bq_query_kernel (cell-sorted path)"
        else:
            out[name] = r
    return out


def headline_kernel_burst(hf, xyz, launches=200):
    """The roofline kernel alone on the device: `launches` back-to-back launches of hf_query_ball_group_xyz at
    the headline shape straight through the C ABI (outputs preallocated, nothing else in flight), bracketed by
    two HIP events on the launch stream.  Average = kernel duration + the ~1-2 us inter-kernel gap; the
    rocprofv3 kernel trace of the same command (profiles/) gives the bare kernel duration."""
    from heterofusionrcnn_amd import _lib
    L = _lib.lib()
    m = SA[0][0]
    new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(m, xyz))
    idx = torch.empty((B, m, KNN), dtype=torch.int32, device="cuda")
    cnt = torch.empty((B, m), dtype=torch.int32, device="cuda")
    grouped = torch.empty((B, m, KNN, 3), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    args = (B, N0, m, SA[0][1], KNN, xyz.data_ptr(), new_xyz.data_ptr(), 1, idx.data_ptr(), cnt.data_ptr(),
            grouped.data_ptr(), st)
    for _ in range(10):
        assert L.hf_query_ball_group_xyz(*args) == 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        L.hf_query_ball_group_xyz(*args)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / launches, launches


def kernel_source_stamp():
    """sha256 of the roofline kernel's source: profiles/roofline_traffic.json carries the stamp of the build its PMC
    passes were collected on (scripts/collect_evidence.sh)"""
    import hashlib
    with open(os.path.join(ROOT, "heterofusionrcnn_amd", "csrc", "ballquery.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def measured_traffic():
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE cannot share a pass on gfx950, so they are collected by scripts/roofline_kernel.py under
    `rocprofv3 --pmc ...` and stored in profiles/roofline_traffic.json with their provenance).  null when that file
    was collected on a different version of the kernel source: a stale figure is not reported."""
    path = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_source_stamp") != kernel_source_stamp():
            return None
        return d["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def bev_iou_alu_fraction():
    """ALU-bound fraction of compute_bev_iou (vector-instruction issue cycles / SIMD cycles of the launch) from the committed
    rocprofv3 PMC pass (profiles/bev_iou_alu.json, scripts/make_alu_json.py); null when collected on another version of bev_iou.hip"""
    import hashlib
    try:
        with open(os.path.join(ROOT, "profiles", "bev_iou_alu.json")) as f:
            d = json.load(f)
        with open(os.path.join(ROOT, "heterofusionrcnn_amd", "csrc", "bev_iou.hip"), "rb") as f:
            if d.get("kernel_source_stamp") != hashlib.sha256(f.read()).hexdigest()[:16]:
                return None
        return d["alu_bound_frac"]
    except (OSError, KeyError, ValueError):
        return None


def per_op_table(hf, xyz):
    """device time of each op at the headline shapes (us per launch)"""
    t = {}
    fps = hf.farthest_point_sample(4096, xyz)
    new_xyz = hf.gather_point(xyz, fps)
    t["fps_16384_to_4096_us"] = time_op(lambda: hf.farthest_point_sample(4096, xyz), iters=5, warm=1)
    t["fps_us_per_round"] = t["fps_16384_to_4096_us"] / 4095.0
    t["gather_point_us"] = time_op(lambda: hf.gather_point(xyz, fps))
    t["query_ball_point_us"] = time_op(lambda: hf.query_ball_point(0.5, KNN, xyz, new_xyz))
    idx, _ = hf.query_ball_point(0.5, KNN, xyz, new_xyz)
    t["group_point_c3_us"] = time_op(lambda: hf.group_point(xyz, idx))
    t["ball_group_fused_us"] = time_op(lambda: hf.query_ball_group(0.5, KNN, xyz, new_xyz, True))
    # per-call distribution through the C ABI with preallocated outputs (one event pair per launch)
    from heterofusionrcnn_amd import _lib
    L = _lib.lib()
    o_idx = torch.empty((B, 4096, KNN), dtype=torch.int32, device="cuda")
    o_cnt = torch.empty((B, 4096), dtype=torch.int32, device="cuda")
    o_grp = torch.empty((B, 4096, KNN, 3), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    fused_args = (B, N0, 4096, 0.5, KNN, xyz.data_ptr(), new_xyz.data_ptr(), 1, o_idx.data_ptr(), o_cnt.data_ptr(), o_grp.data_ptr(), st)
    t["ball_group_fused_per_call_us"] = time_op_stats(lambda: L.hf_query_ball_group_xyz(*fused_args), iters=100, warm=10)
    f64 = torch.randn(B, N0, 64, device="cuda")
    t["group_point_c64_us"] = time_op(lambda: hf.group_point(f64, idx))
    t["group_point_c64_GBs"] = (4 * B * N0 * 64 + 4 * B * 4096 * KNN + 4 * B * 4096 * KNN * 64) / t["group_point_c64_us"] / 1e3
    t["three_nn_us"] = time_op(lambda: hf.three_nn(xyz, new_xyz))
    dist, i3 = hf.three_nn(xyz, new_xyz)
    w = torch.rand(B, N0, 3, device="cuda")
    p256 = torch.randn(B, 4096, 256, device="cuda")
    t["three_interpolate_c256_us"] = time_op(lambda: hf.three_interpolate(p256, i3, w))
    t["three_interpolate_c256_GBs"] = (4 * B * 4096 * 256 + 2 * 4 * 3 * B * N0 + 4 * B * N0 * 256) / t["three_interpolate_c256_us"] / 1e3
    rng = np.random.default_rng(3)
    a = torch.from_numpy(rand_bev(rng, 70000)).cuda()
    g = torch.from_numpy(rand_bev(rng, 64)).cuda()
    # straight through the C ABI with the outputs preallocated: the Python wrapper (two 18 MB allocations per call) costs
    # about as much as the kernel
    from heterofusionrcnn_amd import _lib
    ov = torch.empty((70000, 64), dtype=torch.float32, device="cuda")
    io = torch.empty_like(ov)
    st = torch.cuda.current_stream().cuda_stream
    bev_args = (70000, a.data_ptr(), 64, g.data_ptr(), ov.data_ptr(), io.data_ptr(), st)
    t["bev_iou_70000x64_us"] = time_op(lambda: _lib.lib().hf_compute_bev_iou(*bev_args), iters=100, warm=10)
    t["bev_iou_Gboxpairs_per_s"] = 70000 * 64 / t["bev_iou_70000x64_us"] / 1e3
    t["bev_iou_frac_of_8B_per_pair_hbm_ceiling"] = 8 * 70000 * 64 / (t["bev_iou_70000x64_us"] * 1e-6) / (HBM_PEAK_GBS * 1e9)
    t["bev_iou_frac_alu_bound"] = bev_iou_alu_fraction()   # "HBM- and ALU-bound fractions"
    nb = torch.from_numpy(rand_bev(rng, 9000)).cuda()
    t["oriented_nms_9000_us"] = time_op(lambda: hf.oriented_nms(nb, 0.8), iters=10, warm=2)
    # 300 clusters of 30 near-duplicates (what an RPN hands to NMS looks like this, not like uniform boxes)
    base = rand_bev(rng, 300)
    cl = np.repeat(base, 30, 0)
    cl[:, [0, 2]] += rng.normal(0, 0.3, (9000, 1)).astype(np.float32)
    cl[:, [1, 3]] += rng.normal(0, 0.3, (9000, 1)).astype(np.float32)
    cl[:, 4] += rng.normal(0, 0.1, 9000).astype(np.float32)
    nbc = torch.from_numpy(cl.astype(np.float32)).cuda()
    t["oriented_nms_9000_clustered_t0.8_us"] = time_op(lambda: hf.oriented_nms(nbc, 0.8), iters=10, warm=2)
    # kNN (PointCNN / knn-mode SA): K=8 neighbours of 4096 queries in 16384 points, no (B,M,N) matrix
    t["knn_k8_16384x4096_us"] = time_op(lambda: hf.knn_point(8, xyz, new_xyz), iters=10, warm=2)
    t["knn_k8_16384x4096_all_pairs_us"] = time_op(lambda: hf.knn_point(8, xyz, new_xyz, all_pairs=True), iters=5, warm=1)
    # the PointCNN RPN's neighbour search (rpn_multiclass.config: K=8 among all 16384 points of the cloud, pointfly.py:185-212)
    t["knn_k8_16384x16384_us"] = time_op(lambda: hf.knn_point(8, xyz, xyz), iters=5, warm=1)
    # RoI crop at the RCNN shape: 64 boxes per frame, R=512, C=288 features (rcnn_multiclass.config:42,297)
    from heterofusionrcnn_amd import modules
    nroi = 64 * B
    b3 = torch.zeros(nroi, 7, device="cuda")
    b3[:, 0] = torch.from_numpy(rng.uniform(-35, 35, nroi).astype(np.float32)).cuda()
    b3[:, 1] = 3.0
    b3[:, 2] = torch.from_numpy(rng.uniform(5, 65, nroi).astype(np.float32)).cuda()
    b3[:, 3:6] = torch.tensor([5.9, 3.6, 8.0], device="cuda")       # 3.9 x 1.6 car + 1 m context each side
    b3[:, 6] = torch.from_numpy(rng.uniform(-np.pi, np.pi, nroi).astype(np.float32)).cuda()
    boxes8 = modules.box_3d_to_box_8co(b3).contiguous()
    box_ind = (torch.arange(nroi, device="cuda") // 64).to(torch.int32)
    fts = torch.randn(B, N0, 288, device="cuda")
    inten = torch.rand(B, N0, 1, device="cuda")
    msk = torch.rand(B, N0, device="cuda") < 0.1
    t["pc_crop_512roi_R512_C288_us"] = time_op(lambda: hf.pc_crop_and_sample(xyz, fts, inten, msk, boxes8, box_ind, 512),
                                               iters=10, warm=2)
    t["pc_crop_out_GBs"] = nroi * 512 * (3 + 288 + 1) * 4 / t["pc_crop_512roi_R512_C288_us"] / 1e3
    # glue either side of the ops: the LiDAR -> image fusion gather, the bin-based box decode of the RPN head
    from heterofusionrcnn_amd import box_codec
    from heterofusionrcnn_amd.fusion import project_gather
    calib = torch.tensor([[700.0, 0, 300.0, 20.0], [0, 700.0, 90.0, -0.1], [0, 0, 1, 0.003]], device="cuda").repeat(B, 1, 1)
    cam = torch.stack([xyz[..., 0] * 0.25, xyz[..., 1] * 0.25, xyz[..., 2] + 2.0], dim=-1).contiguous()
    img_fts = torch.randn(B, 180, 600, 64, device="cuda")
    t["project_gather_c64_us"] = time_op(lambda: project_gather(cam, calib, img_fts))
    t["project_gather_c64_GBs"] = B * N0 * (12 + 2 * 64 * 4) / t["project_gather_c64_us"] / 1e3
    kcls = 3
    dec_in = [torch.randint(0, 12, (B, N0, kcls), device="cuda", dtype=torch.int32), torch.rand(B, N0, kcls, device="cuda") - .5,
              torch.randint(0, 12, (B, N0, kcls), device="cuda", dtype=torch.int32), torch.rand(B, N0, kcls, device="cuda") - .5,
              torch.randint(0, 12, (B, N0, kcls), device="cuda", dtype=torch.int32), torch.rand(B, N0, kcls, device="cuda") - .5,
              torch.randn(B, N0, kcls, device="cuda"), torch.randn(B, N0, kcls, 3, device="cuda") * .1,
              torch.rand(B, N0, kcls, 3, device="cuda") + 1]
    t["bin_box_decode_16384x3_us"] = time_op(lambda: box_codec.decode(xyz, 0, *dec_in, [3.0] * kcls, [0.5] * kcls,
                                                                     0.25 * np.pi, 0.5 * np.pi / 12))
    # two-stage inference of BASELINE config 5 at its own sizes (RPN of rpn_multiclass.config with image fusion -> 9000 boxes ->
    # NMS 0.8 -> 100 proposals -> 512-point crops of 288 channels -> the RCNN of rcnn_multiclass.config -> NMS 0.01), random
    # weights, 8 frames per batch; frames filtered to the camera's field of view
    from heterofusionrcnn_amd.two_stage import TwoStageDetector
    torch.manual_seed(0)
    det = TwoStageDetector().cuda().eval()
    fx = torch.from_numpy(kitti_frustum(np.random.default_rng(7), B, N0)).cuda()
    inten0 = torch.from_numpy(np.random.default_rng(8).uniform(-0.5, 0.5, (B, N0, 1)).astype(np.float32)).cuda()
    img = torch.randn(B, IMG_H, IMG_W, IMG_C, device="cuda")
    cal = torch.from_numpy(KITTI_P2).cuda().repeat(B, 1, 1).contiguous()
    us = time_op(lambda: det(fx, inten0, img, cal), iters=3, warm=1)
    t["two_stage_infer_ms_per_batch8"] = us / 1e3
    t["two_stage_infer_frames_per_s"] = B / (us * 1e-6)
    t["two_stage_rpn_stage_ms"] = time_op(lambda: det.rpn_stage(fx, inten0, img, cal), iters=3, warm=1) / 1e3
    # the same flow with the RPN geometry of the next batches computed ahead on side streams (two_stage.run_sharded)
    from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
    pf = GeometryPrefetcher(det.geometry, depth=2)
    pf.submit(fx); pf.submit(fx)

    def piped():
        geo = pf.get()
        pf.submit(fx)
        det(fx, inten0, img, cal, geometry=geo)
    us = time_op(piped, iters=6, warm=2)
    t["two_stage_infer_pipelined_ms_per_batch8"] = us / 1e3
    t["two_stage_infer_pipelined_frames_per_s"] = B / (us * 1e-6)
    t["two_stage_config"] = "rpn_multiclass.config + rcnn_multiclass.config at their own sizes (pre-NMS 9000, 100 RoIs/frame, R=512, C=288), random weights"
    return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()}


def _cpu_chain(frames, seed, levels, fp_channels):
    """the CPU oracle's custom-op chain of the backbone for `frames` frames: per SA level FPS + gather, per scale ball
    query + group (xyz and features) + group gradient; per FP level three_nn + interpolate + its gradient"""
    import oracle
    rng = np.random.default_rng(seed)
    xyz0 = kitti_uniform(rng, frames, N0)
    t0 = time.perf_counter()
    xyzs = [xyz0]
    cin = 1
    for (npoint, scales) in levels:
        x = xyzs[-1]
        nx = oracle.gather_point(x, oracle.farthest_point_sample(npoint, x))
        feats = np.zeros((frames, x.shape[1], cin), np.float32)
        for (radius, ns, cout) in scales:
            idx, _ = oracle.query_ball_point(radius, ns, x, nx)
            oracle.group_point(x, idx)
            g = oracle.group_point(feats, idx)
            oracle.group_point_grad(feats.shape, idx, g)
        cin = sum(c for (_, _, c) in scales)
        xyzs.append(nx)
    for lvl in range(len(levels) - 1, -1, -1):
        dist, i3 = oracle.three_nn(xyzs[lvl], xyzs[lvl + 1])
        pts = np.zeros((frames, xyzs[lvl + 1].shape[1], fp_channels[lvl]), np.float32)
        w = np.full(dist.shape, 1.0 / 3.0, np.float32)
        out = oracle.three_interpolate(pts, i3, w)
        oracle.three_interpolate_grad(pts.shape, i3, w, out)
    return time.perf_counter() - t0, xyz0, xyzs


def _pointcnn_plan(cfg):
    """(K, data level, query level, gathered feature width) of every X-Conv / X-DeConv layer and the points per level, from the
    PointCnnConfig (the same bookkeeping as pointcnn.PointCnnBackbone.__init__)"""
    npts, chans = [N0], [cfg.in_channel]
    enc = []
    for li, (k, d, p, c) in enumerate(cfg.xconv):
        same = p == -1 or (li > 0 and p == cfg.xconv[li - 1][2])
        npts.append(npts[-1] if same else p)
        enc.append((k * d, li, li + 1, chans[-1]))
        chans.append(c + (c // 4 if (cfg.with_global and li == len(cfg.xconv) - 1) else 0))
    dec, c_last = [], chans[-1]
    for li, (k, d, pi, qi) in enumerate(cfg.xdconv):
        dec.append((k * d, pi + 1, qi + 1, chans[pi + 1] if li == 0 else c_last))
        c_last = cfg.xconv[qi][3]
    return npts, enc, dec


def _cpu_chain_pointcnn(frames, seed, img_c):
    """the CPU oracle's custom-op chain of the rpn_multiclass step for `frames` frames: FPS + gather per encoder level, per
    X-Conv / X-DeConv layer kNN + group (coordinates and features) + the group gradient, then the image projection + gather +
    its scatter gradient"""
    import oracle
    from heterofusionrcnn_amd.pointcnn import PointCnnConfig
    rng = np.random.default_rng(seed)
    xyz0 = kitti_frustum(rng, frames, N0)
    img = np.zeros((frames, IMG_H, IMG_W, img_c), np.float32) if img_c else None
    npts, enc, dec = _pointcnn_plan(PointCnnConfig())
    t0 = time.perf_counter()
    pts = [xyz0]
    for lvl in range(1, len(npts)):
        cur = pts[-1]
        pts.append(cur if npts[lvl] == cur.shape[1] else oracle.gather_point(cur, oracle.farthest_point_sample(npts[lvl], cur)))
    for (k, di, qi, c) in enc + dec:
        data, qrs = pts[di], pts[qi]
        _, idx = oracle.knn_point(k, data, qrs)
        oracle.group_point(data, idx)
        f = np.zeros((frames, data.shape[1], c), np.float32)
        g = oracle.group_point(f, idx)
        oracle.group_point_grad(f.shape, idx, g)
    if img_c:
        calib = np.repeat(KITTI_P2[None], frames, 0)
        out, pix = oracle.project_gather(xyz0, calib, img)
        oracle.project_gather_grad(img.shape, pix, out)
    return time.perf_counter() - t0, xyz0, pts


def _cpu_chain_worker(args):
    return (_cpu_chain_pointcnn if args[0] == "pointcnn" else _cpu_chain)(*args[1:])[0]


def cpu_baseline(frames, levels, fp_channels, pointcnn_img_c=None):
    """The CPU oracle's op chain for the same backbone (forward ops + the backward ops of the path), `frames` frames on ONE
    core (the reference's CPU code is single-threaded), and the same chain on ALL host cores (one process per core, two
    frames each).  The dense layers are not part of the custom-op path and are on neither side of this figure.
    pointcnn_img_c is not None: the chain of the rpn_multiclass step (_cpu_chain_pointcnn) instead of the SA / FP chain."""
    import multiprocessing as mp
    import oracle
    pointcnn = pointcnn_img_c is not None
    if pointcnn:
        dt, xyz0, xyzs = _cpu_chain_pointcnn(frames, 0, pointcnn_img_c)
        what = ("FPS + gather per encoder level, kNN (K=8) + group(+grad) of the coordinates and features at each of the 5 X-Conv and "
                "6 X-DeConv layers, image projection + gather(+grad)")
    else:
        dt, xyz0, xyzs = _cpu_chain(frames, 0, levels, fp_channels)
        what = ("FPS, gather, ball query, group(+grad) at every SA level / scale, three_nn, three_interpolate(+grad) at every FP level")
    res = {"value": round(frames / dt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frame(s) of the bench workload, custom-op chain only (%s), oracle/hf_oracle.c, 1 thread, %.1f s"
                     % (frames, what, dt)}
    # the cores this process may use, at most 16 (the GPU box's CPU share for one GPU)
    ncore = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    per = 2
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(ncore) as pool:
        jobs = [(("pointcnn", per, 100 + i, pointcnn_img_c) if pointcnn else ("sa_fp", per, 100 + i, levels, fp_channels))
                for i in range(ncore)]
        pool.map(_cpu_chain_worker, jobs)
    wall = time.perf_counter() - t0
    res["all_cores"] = {"value": round(per * ncore / wall, 3), "unit": "frames/s", "cores": ncore, "cpu_model": model,
                        "sample": "%d processes x %d frames, %.1f s wall (process start-up included)" % (ncore, per, wall)}
    # the reference's own CPU program for the headline pair, where oracle/_ref was built
    if oracle.ref_available("qbp"):
        x = kitti_uniform(np.random.default_rng(5), 1, N0)
        q = oracle.gather_point(x, oracle.farthest_point_sample(SA[0][0], x))
        t1 = time.perf_counter()
        ridx = oracle.ref_query_ball_point(0.5, KNN, x, q)
        oracle.ref_group_point(x, ridx)
        res["reference_qbp_group_s_per_frame"] = round(time.perf_counter() - t1, 4)
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default="auto",
                    help="strong: a global batch of 8 frames split over the ranks (BASELINE config 4: 'batch=8 KITTI frames, DP over "
                         "8 GPUs'; hf/experiments/mpi_run_training.sh); weak: 8 frames per GPU (a fixed per-rank batch, "
                         "rpn_multiclass.config:206 + trainer.py:71); auto (default) = strong as the headline and, with several "
                         "ranks, the weak figure measured right after it and reported under extra.weak_scaling")
    ap.add_argument("--workload", choices=("rpn_multiclass", "rpn", "stack"), default="rpn_multiclass",
                    help="rpn_multiclass: the RPN train step of rpn_multiclass.config (PointCNN backbone, image-feature fusion, three "
                         "classes; BASELINE.json configs[3]); rpn: the RPN train step of rpn_cars_pointnet_paper.config (PointNet++ MSG "
                         "backbone); stack: the round-1 workload, BASELINE.json configs[1] (single-scale SA+FP stack, mean loss)")
    ap.add_argument("--img-channels", type=int, default=IMG_C,
                    help="width of the image branch's feature map fed to the fusion of rpn_multiclass (0: point branch only)")
    ap.add_argument("--frames-per-gpu", type=int, default=0, help="override the per-GPU batch (default: 8, or 8 / N with --scaling strong)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the step kernel by kernel (DistributedDataParallel for N > 1) "
                                                            "instead of replaying the captured hipGraph")
    ap.add_argument("--graph", action="store_true", help="replay the captured hipGraph whatever the batch and workload (default: "
                                                         "rpn_multiclass always replays; the other workloads below 8 frames per "
                                                         "GPU, where launches bound the step)")
    ap.add_argument("--optimizer", choices=("hf", "torch"), default="hf",
                    help="hf: optim.MultiTensorAdam (every parameter tensor in one launch, csrc/optim.hip, torch.optim.Adam's "
                         "arithmetic); torch: torch.optim.Adam(fused=True)")
    ap.add_argument("--x-branch-stream", choices=("auto", "on", "off"), default="auto",
                    help="the X-transformation branch of every X-Conv on a side HIP stream (pointcnn.CONCURRENT_X_BRANCH); auto = on, except "
                         "for enqueued steps under DistributedDataParallel")
    ap.add_argument("--lift-branch-max-rows", type=int, default=-1,
                    help="with the X branch on a side stream: the lifting branch of X-Conv layers with at most this many rows (B*P*K) "
                         "goes on a second one (default: pointcnn.CONCURRENT_LIFT_BRANCH_MAX_ROWS)")
    ap.add_argument("--gemm-tuning", default="off",
                    help="off (default): the library's own heuristic; auto: load heterofusionrcnn_amd/tuned_gemms.csv if present "
                         "(library-GEMM selections per shape, PyTorch TunableOp); tune:<file>: time the candidates of every shape of "
                         "this run and record them in <file>.  Measured in round 4: 0 .. 6 %% at one frame per GPU depending on the box, "
                         "no numerical check of the candidates by default -- not worth shipping on")
    ap.add_argument("--blas", choices=("default", "rocblas", "hipblaslt"), default="default",
                    help="diagnostic: which vendor library the framework's matmuls go to (torch.backends.cuda.preferred_blas_library)")
    ap.add_argument("--with-vgg", action="store_true",
                    help="rpn_multiclass only: run the image branch too (inference.ImgVggPyr forward + backward on the vendor "
                         "library, trained with the RPN: hf/core/models/rpn_model.py:126-127,223) instead of feeding a resident "
                         "feature map; the default run reports this under extra.rpn_multiclass_with_vgg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-op-table", action="store_true")
    ap.add_argument("--no-side-runs", action="store_true", help="skip the child runs reported in `extra` (other workload, 1 frame per GPU)")
    ap.add_argument("--no-prefetch", action="store_true", help="run the geometry ops inline instead of one step ahead")
    ap.add_argument("--prefetch-depth", type=int, default=2, help="batches of geometry in flight (one HIP stream each)")
    ap.add_argument("--prefetch-group", type=int, default=0,
                    help="batches whose geometry is computed in one launch (0 = the largest divisor of --steps up to 16, "
                         "so that the timed steps contain exactly as many geometry launches as they consume)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the 1-core CPU leg (0: 4 for rpn_multiclass, else 24)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks that all use GPU 0 with gloo collectives: the whole multi-rank code path of this script on a "
                         "one-GPU box (RCCL needs one GPU per rank); the figure it prints is not a scaling measurement")
    ap.add_argument("--stub", action="store_true",
                    help="CPU test hook: the ranks join a gloo group and time a stand-in step (no HIP); exercises the "
                         "launcher / barrier / max-over-ranks / one-JSON-line plumbing without a GPU")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as a CHILD job
    (python -m torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) and relay its one JSON line.
    This process has not touched the GPU (no HIP call, no torch.cuda call) and never execs."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    out = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [l for l in out.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return out.returncode if (out.returncode != 0 or lines) else 1


def stub_main(args):
    """--stub: the measurement plumbing on CPU ranks (gloo): same barrier / max-over-ranks / rank-0 JSON line."""
    from heterofusionrcnn_amd import dp
    ctx = dp.init("gloo")
    scaling = args.scaling if args.scaling != "auto" else "strong"
    per_rank = B if scaling == "weak" else max(1, B // ctx.world)
    torch.manual_seed(0)
    w = torch.nn.Linear(16, 16)
    net = dp.wrap_model(w, ctx)
    x = torch.randn(per_rank, 16)
    for _ in range(args.warmup):
        net(x).sum().backward()
    dp.fence(ctx)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net(x).sum().backward()
    dp.fence(ctx)
    dt = dp.max_over_ranks(time.perf_counter() - t0, ctx)
    if ctx.rank == 0:
        print(json.dumps({"metric": "stub frames/sec (CPU plumbing test)", "value": round(ctx.world * per_rank * args.steps / dt, 3),
                          "unit": "frames/s", "n_gpus": ctx.world, "ranks": ctx.world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": scaling,
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "stub", "frames_per_gpu": per_rank, "global_batch": per_rank * ctx.world}}))
    dp.shutdown(ctx)


def _child_bench(extra_args, timeout=420):
    """a side measurement in a CHILD process after the timed region (same contract: barrier, K timed steps, one JSON line)"""
    import subprocess
    try:
        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-op-table", "--no-cpu-baseline", "--no-side-runs"] +
                               list(extra_args), capture_output=True, text=True, timeout=timeout)
        line = [l for l in child.stdout.splitlines() if l.startswith("{")][-1]
        o = json.loads(line)
        return {"frames_per_s": o["value"], "ms_per_step": o["ms_per_step"], "frames_per_gpu": o["config"]["frames_per_gpu"],
                "host_enqueue_ms_per_step": o["config"]["host_enqueue_ms_per_step"], "hip_graph": o["config"]["hip_graph"],
                "steps": o["steps"], "workload": o["config"]["workload"]}
    except Exception as e:   # the headline line must not depend on a side measurement
        return {"error": repr(e)[:200]}


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    # BEFORE anything touches the GPU: without a launcher around us, N > 1 ranks are started as a child job
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    if args.stub:
        return stub_main(args)
    # stdout carries exactly ONE line (the JSON result): library banners (RCCL prints its version table on
    # stdout when the communicator is created) and stray prints are routed to stderr at the fd level.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    from heterofusionrcnn_amd import dp
    ctx = dp.init("nccl", share_gpu=args.rehearse_on_one_gpu)   # RCCL; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher
    world, rank, local_rank = ctx.world, ctx.rank, ctx.local_rank
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started by a launcher with WORLD_SIZE=%d" % (args.gpus, world))

    import heterofusionrcnn_amd as hf
    from heterofusionrcnn_amd import modules
    from heterofusionrcnn_amd import gemm_tuning
    tuned = None
    if args.gemm_tuning.startswith("tune:"):
        tuned = gemm_tuning.enable(args.gemm_tuning[5:], tune=True)
    elif args.gemm_tuning == "auto":
        tuned = gemm_tuning.enable()

    if args.blas != "default":
        torch.backends.cuda.preferred_blas_library("cublas" if args.blas == "rocblas" else "cublaslt")

    timer = EventTimer()
    headline = lambda radius, nsample, xyz1, xyz2, center=True: (xyz1.shape[1] == N0 and xyz2.shape[1] == SA[0][0] and nsample == KNN)
    from heterofusionrcnn_amd import rpn as rpn_mod
    rpn_mod.query_ball_group = timer.wrap(rpn_mod.query_ball_group, headline)
    modules.query_ball_group = timer.wrap(modules.query_ball_group, headline)

    scaling = args.scaling if args.scaling != "auto" else "strong"   # BASELINE config 4: a GLOBAL batch of 8 frames over the ranks
    headline_per_gpu = args.frames_per_gpu or (B if scaling == "weak" else max(1, B // world))

    def measure(per_gpu):
        """K timed steps at `per_gpu` frames per rank under the contract of the file header -> the figures the JSON line needs"""
        import types
        torch.manual_seed(1234)  # same initial weights on every rank (the reference broadcasts from rank 0)
        rng = np.random.default_rng(1000 + rank)                   # rank-sharded synthetic frames
        multiclass = args.workload == "rpn_multiclass"
        img_c = args.img_channels if multiclass else 0
        # rpn_multiclass: frames filtered to the camera's field of view as the reference's loader does (every point has a pixel)
        xyz = torch.from_numpy(kitti_frustum(rng, per_gpu, N0) if multiclass else kitti_uniform(rng, per_gpu, N0)).cuda()
        intensity = torch.from_numpy(rng.uniform(-0.5, 0.5, (per_gpu, N0, 1)).astype(np.float32)).cuda()
        inputs = {"xyz": xyz, "intensity": intensity}
        levels = fp_channels = None
        if args.workload in ("rpn", "rpn_multiclass"):
            cfg = rpn_mod.rpn_cars_pointnet_paper() if args.workload == "rpn" else rpn_mod.rpn_multiclass(img_c)
            model = rpn_mod.RpnModel(cfg).cuda()
            # ground truth: 12 objects per frame on the road plane; the per-point class / box labels are made once, as the
            # reference's data loader makes them on the host (kitti_dataset.py:416-440)
            gt_boxes, gt_cls = rpn_mod.synthetic_ground_truth(rng, per_gpu, 12, cfg, ground_y=3.0)
            inputs["label_cls"], inputs["label_reg"] = rpn_mod.point_labels(xyz, torch.from_numpy(gt_boxes).cuda(), torch.from_numpy(gt_cls).cuda())
            if img_c:
                # the image branch's output (pyramid_fusion1 of img_vgg_pyramid.py: full resolution, vgg_conv1[1] channels): a resident
                # synthetic feature map that requires a gradient -- the step ends where the VGG pyramid's backward pass would start
                if args.with_vgg:
                    # the whole reference step: a synthetic 8-bit-range image, the VGG pyramid trained with the RPN
                    from heterofusionrcnn_amd.inference import ImgVggPyr
                    assert img_c == 32, "ImgVggPyr ends in vgg_conv1[1] = 32 channels"
                    inputs["img_fts"] = torch.rand(per_gpu, IMG_H, IMG_W, 3, device="cuda") * 255.0
                    model = rpn_mod.RpnWithImageBranch(model, ImgVggPyr().cuda()).cuda()
                else:
                    inputs["img_fts"] = torch.randn(per_gpu, IMG_H, IMG_W, img_c, device="cuda").requires_grad_(True)
                inputs["calib"] = torch.from_numpy(KITTI_P2).cuda().repeat(per_gpu, 1, 1).contiguous()
            if args.workload == "rpn":
                levels = [(l.npoint, [(sc.radius, sc.nsample, sc.mlp[-1]) for sc in l.scales]) for l in cfg.sa]
                nl = len(cfg.sa)   # an FP level interpolates the features of the coarser level: the deepest SA output, then FP outputs
                fp_channels = [cfg.fp[nl - 2 - lvl][-1] for lvl in range(nl - 1)] + [sum(sc.mlp[-1] for sc in cfg.sa[-1].scales)]
                workload = ("RPN train step, hf/configs/rpn_cars_pointnet_paper.config point branch: MSG set abstraction 16384->4096->1024->"
                            "512->64 (nsample 16/32), 4 FP levels, fc 256/256, seg head + bin-based box head (fc 512/512, 76 outputs), "
                            "targets + focal/softmax/smooth-L1 losses, fwd+bwd+Adam, fp32; image branch not part of the step")
            else:
                workload = ("RPN train step, hf/configs/rpn_multiclass.config: PointCNN 5 xconv (K=8; 16384/4096/1024/256/64 points; C 256.."
                            "1024, X-transformation, global branch) + 6 xdconv layers + fc 256/256, 3 classes, seg head, " +
                            ("projection of the points into the image + gather of the image feature map (B,%d,%d,%d) + path drop 0.9/0.9 + "
                             "'concat' fusion, " % (IMG_H, IMG_W, img_c) if img_c else "") +
                            "bin-based box head (fc 512/512, 3x76 outputs), targets + focal/softmax/smooth-L1 losses, fwd+bwd" +
                            (" (incl. the scatter gradient of the image feature map)" if img_c else "") + "+Adam, fp32; " +
                            (("WITH the image branch: VGG pyramid (conv 32/64/128/256 + 3 up-convolutions, vendor-library convolutions) "
                              "forward + backward on a synthetic (B,%d,%d,3) image, trained with the RPN" % (IMG_H, IMG_W)) if (img_c and args.with_vgg) else
                             "the VGG pyramid that produces the image feature map is NOT run: the map is a resident synthetic input and "
                             "its gradient the step's last output" if img_c else "image branch (VGG pyramid + concat fusion) not part of the step"))
            loss_fn = None
        else:
            model = modules.PointnetSAFPStack(in_channel=1, sa=SA, fp=FP).cuda()
            levels = [(npoint, [(radius, ns, mlp[-1])]) for (npoint, radius, ns, mlp) in SA]
            fp_channels = [FP[1][-1], FP[0][-1], SA[-1][3][-1]]
            workload = ("SA+FP stack 16384->4096->1024->256, K=32, radii 0.5/1.0/2.0, fwd+bwd+Adam, mean loss, fp32 "
                        "(BASELINE.json configs[1])")
            loss_fn = lambda m, inp, geo: m(inp["xyz"], inp["intensity"], geometry=geo).mean()

        from heterofusionrcnn_amd.graph_step import TrainStep, broadcast_parameters
        from heterofusionrcnn_amd.pipeline import GeometryPrefetcher, choose_group
        # one rank: the PointCNN RPN step is replayed from its captured hipGraphs whatever the batch (several ranks at 8 frames per
        # rank = the fixed-per-rank-batch mode stay on DistributedDataParallel): at 8 frames the device bounds the step
        # either way (27.0 ms replayed, 27.0-27.2 enqueued), but enqueueing costs the host 21.6 ms of those 27 -- a slower or busier
        # host would make the enqueued form launch-bound; the enqueued figure is reported under extra
        use_graph = (args.graph or per_gpu < B or (args.workload == "rpn_multiclass" and not args.with_vgg and world == 1)) and not args.no_graph
        group = args.prefetch_group or choose_group(args.steps)
        prefetch = None if args.no_prefetch else GeometryPrefetcher(model.geometry, depth=args.prefetch_depth, group=group)
        lr = dp.scaled_lr(1e-3, world)                                   # optimizer_builder.py:105
        from heterofusionrcnn_amd import pointcnn as pointcnn_mod
        # measured: -1.6 ms at 1 frame (replayed), -1.5 ms at 8 (enqueued).  Not under DistributedDataParallel (several ranks, enqueued steps):
        # its bucket hooks copy gradients as they appear, on whatever stream the producing node ran
        pointcnn_mod.CONCURRENT_X_BRANCH = args.x_branch_stream == "on" or (args.x_branch_stream == "auto" and (use_graph or world == 1))
        if args.lift_branch_max_rows >= 0:
            pointcnn_mod.CONCURRENT_LIFT_BRANCH_MAX_ROWS = args.lift_branch_max_rows
        from heterofusionrcnn_amd.optim import MultiTensorAdam
        make_opt = ((lambda ps, capturable: MultiTensorAdam(ps, lr=lr, tf_epsilon=False)) if args.optimizer == "hf" else
                    (lambda ps, capturable: torch.optim.Adam(ps, lr=lr, fused=True, capturable=capturable)))
        train = None
        if use_graph:
            # one hipGraph launch per step: forward, losses, backward into ONE flat gradient buffer (+ Adam when there is a single
            # rank); N > 1: one RCCL all-reduce of that buffer (hvd.DistributedOptimizer's average), then the fused Adam step
            broadcast_parameters(model)                                  # hvd.broadcast_global_variables(0)
            opt = make_opt(list(model.parameters()), True)
            train = TrainStep(model, opt, inputs, model.geometry(xyz), world=world, graph=True, loss_fn=loss_fn)
            net = model
        else:
            net = dp.wrap_model(model, ctx)                              # broadcast from rank 0 + bucketed gradient all-reduce (RCCL)
            opt = make_opt(list(net.parameters()), False)
            eager_loss = loss_fn or (lambda m, inp, geo: _eager_rpn_loss(m, model, inp, geo))

        def step():
            # the coordinate-only ops of the NEXT batches run on side streams while this batch trains;
            # each step consumes one geometry result and submits one: every step does the full work
            geo = None
            if prefetch is not None:
                geo = prefetch.get()
                prefetch.submit(xyz)
            if use_graph:
                return train(geometry=geo if geo is not None else model.geometry(xyz))
            opt.zero_grad(set_to_none=True)
            if "img_fts" in inputs:
                inputs["img_fts"].grad = None
            loss = eager_loss(net, inputs, geo)
            loss.backward()
            opt.step()
            return loss

        if prefetch is not None:
            for _ in range(prefetch.capacity):
                prefetch.submit(xyz)
        for _ in range(args.warmup):
            step()
        align = 0
        while prefetch is not None and prefetch.staged:  # start the timed steps on a group boundary: K steps then
            step()                                       # launch the geometry of exactly K batches
            align += 1

        dp.fence(ctx)
        timer.enabled = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        t_enqueued = time.perf_counter() - t0   # the host is done submitting; what remains until the fence is device time
        dp.fence(ctx)
        dt = dp.max_over_ranks(time.perf_counter() - t0, ctx)
        timer.enabled = False
        assert torch.isfinite(loss).item(), "loss is not finite"
        if "img_fts" in inputs and inputs["img_fts"].requires_grad:
            g = inputs["img_fts"].grad
            # (the last step's path drop may have switched the image branch off -- one step in ten at 0.9 / 0.9 -- so an all-zero
            # gradient is legitimate here; tests/test_graph_step.py checks that the gradient lands on the projected pixels)
            assert g is not None and torch.isfinite(g).all().item(), "the gradient of the image feature map is missing or not finite"

        out = types.SimpleNamespace(per_gpu=per_gpu, use_graph=use_graph, dt=dt, t_enqueued=t_enqueued, align=align, workload=workload,
                                    levels=levels, fp_channels=fp_channels, multiclass=multiclass, img_c=img_c, in_step_us=timer.mean_us(),
                                    in_step_launches=len(timer.pairs), prefetch_depth=prefetch.depth if prefetch is not None else 0,
                                    prefetch_group=prefetch.group if prefetch is not None else 0,
                                    x_branch=bool(pointcnn_mod.CONCURRENT_X_BRANCH), chunks=getattr(train, "chunks", 1) if train is not None else 0)
        train = prefetch = None                                      # release the graph's memory pool
        torch.cuda.empty_cache()
        return out

    M = measure(headline_per_gpu)
    per_gpu, use_graph, dt, t_enqueued, multiclass, img_c = M.per_gpu, M.use_graph, M.dt, M.t_enqueued, M.multiclass, M.img_c
    weak = None
    if world > 1 and args.scaling == "auto" and not args.frames_per_gpu and headline_per_gpu != B:
        timer.pairs.clear()
        weak = measure(B)                                            # the reference's own mode (a fixed per-rank batch): reported under extra

    result = None
    if rank == 0:
        frames = world * per_gpu * args.steps
        in_step_us = M.in_step_us  # inside the timed steps: shares the device with the overlapped MLP kernels
        xyz8 = torch.from_numpy(kitti_uniform(np.random.default_rng(1000), B, N0)).cuda()   # KITTI-extent uniform points
        k_us, n_burst = headline_kernel_burst(hf, xyz8)
        fused_bytes = ball_group_bytes(B, N0, SA[0][0], KNN)
        # algorithmic bytes of the two ops at their boundary = 24 641 536 B (the figure `frac` is computed from)
        achieved = SURVEY_TWO_OP_BYTES / (k_us * 1e-6) / 1e9 if k_us else None
        result = {
            "metric": "KITTI frames/sec RPN train step" if args.workload != "stack" else "KITTI frames/sec, PointNet++ SA+FP stack train step (fwd+bwd+Adam)",
            "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "ranks": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": M.workload, **({"rehearsal": "all ranks on GPU 0, gloo collectives: not a scaling measurement"}
                                                if args.rehearse_on_one_gpu else {}),
                       "frames_per_gpu": per_gpu, "global_batch": world * per_gpu, "parallelism": "dp%d" % world,
                       "hip_graph": use_graph, "optimizer": args.optimizer,
                       "x_branch_on_side_stream": M.x_branch, "scaling": scaling,
                       "library_gemm_selection": ("tuned per shape (%s)" % os.path.basename(tuned)) if tuned else "library default heuristic",
                       "gradient_exchange": ("none (1 rank)" if world == 1 else
                                             ("RCCL all-reduce of the flat gradient buffer in %d chunk(s); the first overlaps the rest of the "
                                              "replayed backward pass" % M.chunks if use_graph else
                                              "DistributedDataParallel buckets overlapped with backward")),
                       "geometry_prefetch_depth": M.prefetch_depth,
                       "geometry_prefetch_group": M.prefetch_group,
                       "extra_untimed_alignment_steps": M.align,
                       # host time to submit a step (geometry launches, input copies, the graph launch); the rest of ms_per_step is
                       # the host waiting for the device
                       "host_enqueue_ms_per_step": round(1e3 * t_enqueued / args.steps, 3)},
            "roofline": {"kernel": "query_ball_point+group_point fused (hf_query_ball_group_xyz), B=8 N=16384 M=4096 K=32",
                         "bound": "hbm", "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
                         "traffic": measured_traffic(), "algorithmic_bytes": SURVEY_TWO_OP_BYTES,
                         # the fused kernel never re-reads idx / xyz: its own compulsory traffic is smaller
                         "fused_kernel_compulsory_bytes": fused_bytes,
                         "frac_vs_fused_compulsory_bytes": round(fused_bytes / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if k_us else None,
                         "avg_launch_us": round(k_us, 3) if k_us else None,
                         "launches_timed": n_burst,
                         "in_step_avg_launch_us": round(in_step_us, 3) if in_step_us else None,
                         "in_step_launches": M.in_step_launches,
                         "in_step_clouds_per_launch": per_gpu * max(M.prefetch_group, 1),
                         # the geometry-group shape launched alone (in the step it shares the device with the main stream)
                         "batched_alone": batched_kernel_burst(hf) if world == 1 else None},
        }
    if rank == 0 and weak is not None:
        # the reference's own data-parallel mode (a fixed per-rank batch of 8), same contract, measured right after the headline
        result["extra"] = {"weak_scaling": {"frames_per_s": round(world * weak.per_gpu * args.steps / weak.dt, 3),
                                            "ms_per_step": round(1e3 * weak.dt / args.steps, 3), "frames_per_gpu": weak.per_gpu,
                                            "global_batch": world * weak.per_gpu, "hip_graph": weak.use_graph,
                                            "host_enqueue_ms_per_step": round(1e3 * weak.t_enqueued / args.steps, 3)}}
    if rank == 0 and world == 1:
        if not args.no_op_table:
            result["extra"] = per_op_table(hf, xyz8)
        if not args.no_side_runs:
            result.setdefault("extra", {})
            if multiclass:
                # the per-rank shape of BASELINE config 4 (a batch of 8 over 8 GPUs = 1 frame per GPU), and the step without the graph
                result["extra"]["rpn_multiclass_1_frame_per_gpu"] = _child_bench(["--frames-per-gpu", "1", "--steps", "32"])
                result["extra"]["rpn_multiclass_%s" % ("eager_launches" if use_graph else "hip_graph")] = _child_bench(
                    ["--no-graph" if use_graph else "--graph", "--steps", "20"])
                result["extra"]["rpn_multiclass_1_frame_per_gpu_eager_launches"] = _child_bench(["--frames-per-gpu", "1", "--no-graph", "--steps", "32"])
                result["extra"]["rpn_pointnet_train_step"] = _child_bench(["--workload", "rpn", "--steps", "16"])
                # the reference's whole step: the VGG pyramid forward + backward in front of the fusion (rpn_model.py:126-127,223)
                result["extra"]["rpn_multiclass_with_vgg"] = _child_bench(["--with-vgg", "--steps", "10"])
            elif args.workload == "rpn":
                result["extra"]["rpn_multiclass_train_step"] = _child_bench(["--workload", "rpn_multiclass", "--steps", "8"])
        if not args.no_cpu_baseline:
            if multiclass:
                result["cpu_baseline"] = cpu_baseline(args.cpu_frames or 4, None, None, pointcnn_img_c=img_c)
            else:
                result["cpu_baseline"] = cpu_baseline(args.cpu_frames or 24, M.levels, M.fp_channels)
    dp.shutdown(ctx)
    sys.stdout.flush()
    if rank == 0:
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)


def _eager_rpn_loss(net, model, inputs, geo):
    seg_logits, head = net(inputs["xyz"], inputs["intensity"], geometry=geo, img_fts=inputs.get("img_fts"), calib=inputs.get("calib"))
    loss, _ = model.loss(inputs["xyz"], seg_logits, head, inputs["label_cls"], inputs["label_reg"])
    return loss


if __name__ == "__main__":
    main()
