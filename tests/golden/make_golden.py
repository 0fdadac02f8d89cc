"""Generate the committed golden vectors in tests/golden/*.npz.

Run from the repo root:  python tests/golden/make_golden.py

Every fixture holds seeded inputs and the outputs of the CPU oracle (oracle/hf_oracle.c) for
them.  Before anything is written, the oracle is cross-checked here against the reference's own
standalone CPU programs compiled where they lie (oracle/_ref/libhfref_{qbp,sel,itp}.so, built
by oracle/Makefile from grouping/test/query_ball_point.cpp, grouping/test/selection_sort.cpp and
interpolate/interpolate.cpp) wherever such a twin exists, and against the known answers of the
reference demos (bev_iou/bev_iou.py:47-65).  The fixtures are data only (inputs + expected
outputs); they are what travels to the GPU box, where /root/reference does not exist.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def kitti_uniform(rng, b, n):
    """SURVEY 8d cloud 'kitti-uniform': x in [-40,40], y in [-5,3], z in [0,70] (rpn_multiclass.config:262)"""
    lo = np.array([-40.0, -5.0, 0.0], np.float32)
    hi = np.array([40.0, 3.0, 70.0], np.float32)
    return (lo + (hi - lo) * rng.random((b, n, 3), dtype=np.float32)).astype(np.float32)


def box_3d_to_8co(boxes_3d):
    """hf/core/box_8c_encoder.py:101-185 restated in numpy: (N,7)[x,y,z,l,w,h,ry] -> (N,3,8)"""
    boxes_3d = np.asarray(boxes_3d, np.float32)
    x, y, z, l, w, h, ry = [boxes_3d[:, i] for i in range(7)]
    s, c = np.sin(ry).astype(np.float32), np.cos(ry).astype(np.float32)
    hl, hw = l / 2, w / 2
    xc = np.stack([hl, hl, -hl, -hl, hl, hl, -hl, -hl], 1)
    zero = np.zeros_like(h)
    yc = np.stack([zero, zero, zero, zero, -h, -h, -h, -h], 1)
    zc = np.stack([hw, -hw, -hw, hw, hw, -hw, -hw, hw], 1)
    # rot_mats^T @ corners with rot = [[c,0,s],[0,1,0],[-s,0,c]] stacked on axis 2 (i.e. transposed once already)
    X = c[:, None] * xc + s[:, None] * zc + x[:, None]
    Y = yc + y[:, None]
    Z = -s[:, None] * xc + c[:, None] * zc + z[:, None]
    return np.stack([X, Y, Z], 1).astype(np.float32)


def random_boxes3d(rng, n, extent=((-40, 40), (0, 70))):
    """SURVEY 8d boxes: centre uniform in BEV extents, (l,w)~N((3.9,1.6),(0.4,0.1)) clipped, ry~U(-pi,pi)"""
    cx = rng.uniform(*extent[0], n)
    cz = rng.uniform(*extent[1], n)
    cy = rng.uniform(0.5, 2.5, n)
    l = np.clip(rng.normal(3.9, 0.4, n), 0.5, None)
    w = np.clip(rng.normal(1.6, 0.1, n), 0.5, None)
    h = np.clip(rng.normal(1.5, 0.1, n), 0.5, None)
    ry = rng.uniform(-np.pi, np.pi, n)
    return np.stack([cx, cy, cz, l, w, h, ry], 1).astype(np.float32)


def boxes3d_to_bev(b):
    """hf/core/compute_iou.py:7-20: [x - l/2, z - w/2, x + l/2, z + w/2, ry]"""
    b = np.asarray(b, np.float32)
    hl, hw = b[:, 3] / 2, b[:, 4] / 2
    return np.stack([b[:, 0] - hl, b[:, 2] - hw, b[:, 0] + hl, b[:, 2] + hw, b[:, 6]], 1).astype(np.float32)


def clustered_bev(rng, clusters, copies, sigma=0.3, sigma_ry=0.1):
    base = random_boxes3d(rng, clusters)
    boxes = np.repeat(base, copies, axis=0)
    boxes[:, 0] += rng.normal(0, sigma, len(boxes))
    boxes[:, 2] += rng.normal(0, sigma, len(boxes))
    boxes[:, 6] += rng.normal(0, sigma_ry, len(boxes))
    boxes = boxes[rng.permutation(len(boxes))]  # "score order"
    return boxes3d_to_bev(boxes)


def nms_margin_ok(bev, thresh, margin=1e-4):
    """no IoU within `margin` of the threshold: the keep vector then cannot depend on the last ulps of sinf/cosf/atan2f"""
    _, iou = oracle.compute_bev_iou(bev, bev)
    return not np.any(np.abs(iou - thresh) < margin)


def threenn_cases():
    """clouds that walk every branch of the three_nn cascade: random (all three branches, many times), exact duplicates and
    mirrored points (equal distances: tie order), fewer than three known points (sentinels), descending / ascending order"""
    rng = np.random.default_rng(21)
    cases = {"random": rng.standard_normal((4, 500, 3)).astype(np.float32),
             "large": rng.standard_normal((1, 6000, 3)).astype(np.float32) * 40}
    dup = rng.standard_normal((3, 60, 3)).astype(np.float32)
    dup[:, 20:40] = dup[:, :20]                               # exact duplicates at a higher index
    dup[:, 40:60] = -dup[:, :20]                              # mirrored: the same x*x+y*y+z*z bits
    cases["ties"] = dup
    cases["lattice"] = rng.integers(-3, 4, (2, 400, 3)).astype(np.float32)   # distances collide everywhere
    for m in (1, 2, 3):
        cases["m%d" % m] = rng.standard_normal((5, m, 3)).astype(np.float32)
    ramp = np.zeros((2, 64, 3), np.float32)
    ramp[0, :, 0] = np.arange(64, 0, -1)                      # strictly descending: always the first branch
    ramp[1, :, 1] = np.arange(1, 65)                          # strictly ascending: branches 1, 2, 3 once, then none
    cases["ramps"] = ramp
    cases["all_equal"] = np.ones((1, 17, 3), np.float32)
    return cases


def three_nn_origin():
    """three_nn_origin.npz: OUTPUTS OF THE REFERENCE's compiled threenn_cpu (interpolate/interpolate.cpp:21-64 through
    oracle/_ref/libhfref_itp.so) for unknown points at the origin, with the known clouds they belong to."""
    assert oracle.ref_available("itp"), "needs oracle/_ref/libhfref_itp.so (make -C oracle, with /root/reference present)"
    arrays = {}
    for name, known in threenn_cases().items():
        rd, ri = oracle.ref_threenn_origin(known)
        od, oi = oracle.three_nn(np.zeros((known.shape[0], 1, 3), np.float32), known)
        assert np.array_equal(oi, ri) and np.array_equal(od.view(np.uint32), rd.view(np.uint32)), name
        arrays[name + "_known"], arrays[name + "_dist"], arrays[name + "_idx"] = known, rd, ri
    save("three_nn_origin", **arrays)


def main():
    if sys.argv[1:] == ["three_nn_origin"]:                  # one fixture on its own (the others stay byte-identical)
        return three_nn_origin()
    have_ref = all(oracle.ref_available(k) for k in ("qbp", "sel", "itp"))
    print("reference CPU builds available:", have_ref)

    # ------------------------------------------------------------ ball query + group (plumbing config P)
    rng = np.random.default_rng(1)
    xyz = rng.random((2, 1024, 3), dtype=np.float32)
    fps = oracle.farthest_point_sample(256, xyz)
    new_xyz = oracle.gather_point(xyz, fps)
    feats = rng.standard_normal((2, 1024, 8)).astype(np.float32)
    cases = {}
    for tag, r, ns in (("r010", 0.1, 32), ("r030", 0.3, 32), ("r005k8", 0.05, 8)):
        idx, cnt = oracle.query_ball_point(r, ns, xyz, new_xyz)
        if have_ref:
            assert np.array_equal(idx, oracle.ref_query_ball_point(r, ns, xyz, new_xyz)), tag
        cases["idx_" + tag], cases["cnt_" + tag] = idx, cnt
    g_xyz = oracle.group_point(xyz, cases["idx_r010"])
    idx_f = np.ascontiguousarray(cases["idx_r010"][:, :64])  # feature grouping on the first 64 queries (fixture size)
    g_feat = oracle.group_point(feats, idx_f)
    go = rng.standard_normal(g_feat.shape).astype(np.float32)
    g_grad = oracle.group_point_grad(feats.shape, idx_f, go)
    if have_ref:
        assert np.array_equal(g_feat, oracle.ref_group_point(feats, idx_f))
        assert np.array_equal(g_grad, oracle.ref_group_point_grad(feats.shape, idx_f, go))
    # independent random queries, tiny radius: most rows have no hit -> the zero-row rule
    q_rand = rng.random((2, 64, 3), dtype=np.float32)
    idx0, cnt0 = oracle.query_ball_point(0.02, 16, xyz, q_rand)
    assert (cnt0 == 0).any() and (cnt0 > 0).any()
    if have_ref:
        assert np.array_equal(idx0, oracle.ref_query_ball_point(0.02, 16, xyz, q_rand))
    save("grouping_P", xyz=xyz, fps=fps, new_xyz=new_xyz, feats=feats, grouped_xyz_r010=g_xyz,
         grouped_feat_r010=g_feat, grad_out=go, grad_points=g_grad, q_rand=q_rand, idx_rand=idx0, cnt_rand=cnt0,
         **cases)

    # reference test shape: grouping/tf_grouping_op_test.py:12-17
    rng = np.random.default_rng(11)
    pts = rng.random((1, 128, 16), dtype=np.float32)
    x1 = rng.random((1, 128, 3), dtype=np.float32)
    x2 = rng.random((1, 8, 3), dtype=np.float32)
    idx, cnt = oracle.query_ball_point(0.3, 32, x1, x2)
    save("grouping_reftest", points=pts, xyz1=x1, xyz2=x2, idx=idx, cnt=cnt, grouped=oracle.group_point(pts, idx))

    # ------------------------------------------------------------ FPS (+ tie rule on duplicated points)
    rng = np.random.default_rng(2)
    base = kitti_uniform(rng, 2, 700)
    dup = base.copy()
    dup[:, 630:] = base[:, :70]          # last 10 % duplicate the first 10 % (kitti_dataset.py:358-364)
    big = kitti_uniform(rng, 1, 2048)
    big[:, 1536:] = big[:, :512]         # duplicates exactly 1536 apart: k mod 512 equal -> smallest k decides
    big2 = kitti_uniform(rng, 1, 1500)
    big2[:, 1000:1300] = big2[:, 100:400]  # duplicates 900 apart: the LATER copy has the smaller k mod 512
    save("fps", unit=xyz, unit_fps=fps, dup=dup, dup_fps=oracle.farthest_point_sample(256, dup),
         big=big, big_fps=oracle.farthest_point_sample(1700, big),
         big2=big2, big2_fps=oracle.farthest_point_sample(1400, big2),
         tiny=xyz[:, :5], tiny_fps=oracle.farthest_point_sample(5, xyz[:, :5]))
    gg = rng.standard_normal(new_xyz.shape).astype(np.float32)
    save("gather", xyz=xyz, idx=fps, out=new_xyz, out_g=gg, inp_g=oracle.gather_point_grad(xyz.shape, fps, gg))

    # ------------------------------------------------------------ three_nn / three_interpolate
    rng = np.random.default_rng(3)
    unknown = kitti_uniform(rng, 2, 512)
    known = oracle.gather_point(unknown, oracle.farthest_point_sample(128, unknown))
    dist, idx = oracle.three_nn(unknown, known)
    d_full = ((unknown[:, :, None, :].astype(np.float64) - known[:, None, :, :]) ** 2).sum(-1)
    assert np.array_equal(idx[:, :, 0], d_full.argmin(-1))  # hand check of the nearest one
    known2 = known[:, :2]
    dist2, idx2 = oracle.three_nn(unknown[:, :16], known2)   # m < 3: +inf / 0 slots
    assert np.isinf(dist2[:, :, 2]).all() and (idx2[:, :, 2] == 0).all()
    pf = rng.standard_normal((2, 128, 12)).astype(np.float32)
    d = np.maximum(dist, 1e-10)
    w = ((1.0 / d) / (1.0 / d).sum(-1, keepdims=True)).astype(np.float32)  # pointnet_util.py:304-307
    out = oracle.three_interpolate(pf, idx, w)
    go = rng.standard_normal(out.shape).astype(np.float32)
    gp = oracle.three_interpolate_grad(pf.shape, idx, w, go)
    if have_ref:
        assert np.array_equal(out, oracle.ref_three_interpolate(pf, idx, w))
        assert np.array_equal(gp, oracle.ref_three_interpolate_grad(pf.shape, idx, w, go))
    assert np.array_equal(out, oracle.three_interpolate_cf(pf.transpose(0, 2, 1), idx, w).transpose(0, 2, 1))
    save("interpolate", unknown=unknown, known=known, dist=dist, idx=idx, dist_m2=dist2, idx_m2=idx2, points=pf,
         weight=w, out=out, grad_out=go, grad_points=gp)

    # ------------------------------------------------------------ selection sort (unused op; tiny)
    dsel = rng.random((2, 3, 40), dtype=np.float32)
    oi, od = oracle.select_top_k(5, dsel)
    if have_ref:
        ri, rd = oracle.ref_select_top_k(5, dsel)
        assert np.array_equal(oi, ri) and np.array_equal(od, rd)
    save("select_top_k", dist=dsel, outi=oi, out=od)

    # ------------------------------------------------------------ bev_iou / nms
    prop = np.array([[0, 0, 1, 1, 0], [2, 2, 3, 3, 0]], np.float32)         # bev_iou/bev_iou.py:47-50
    gt = np.array([[0, 0, 1, 1, 0], [2, 2, 4, 4, 0], [5, 5, 6, 6, 0]], np.float32)
    ov, iou = oracle.compute_bev_iou(prop, gt)
    assert np.array_equal(ov, np.array([[1, 0, 0], [0, 1, 0]], np.float32))
    assert np.array_equal(iou, np.array([[1, 0, 0], [0, 0.25, 0]], np.float32))
    demo_nms = np.array([[0, 0, 1, 1, 0], [2, 2, 3, 3, 0], [0, 0, .75, .75, 0]], np.float32)  # bev_iou.py:62-65
    assert oracle.oriented_nms(demo_nms, 0.5).tolist() == [0, 1, 0]
    rng = np.random.default_rng(3)
    a = clustered_bev(rng, 12, 8)            # 96 boxes, heavy overlap inside clusters
    b = boxes3d_to_bev(random_boxes3d(rng, 16))
    b[:8] = a[:8]                            # exact duplicates: IoU 1
    ov_ab, iou_ab = oracle.compute_bev_iou(a, b)
    seed = 4
    while True:
        rng = np.random.default_rng(seed)
        nb = clustered_bev(rng, 40, 10)      # 400 boxes
        if all(nms_margin_ok(nb, t) for t in (0.85, 0.8, 0.01)):
            break
        seed += 100
    keeps = {"keep_%03d" % int(t * 100): oracle.oriented_nms(nb, t) for t in (0.85, 0.8, 0.01)}
    save("bev_iou", demo_prop=prop, demo_gt=gt, demo_overlap=ov, demo_iou=iou, demo_nms=demo_nms,
         demo_keep=np.array([0, 1, 0], np.int32), a=a, b=b, overlap_ab=ov_ab, iou_ab=iou_ab, nms_boxes=nb,
         nms_seed=np.array([seed]), **keeps)

    # ------------------------------------------------------------ crop
    rng = np.random.default_rng(5)
    pts = kitti_uniform(rng, 2, 2048)
    pts[:, :, 1] = rng.uniform(0.0, 2.0, (2, 2048)).astype(np.float32)
    fts = rng.standard_normal((2, 2048, 8)).astype(np.float32)
    inten = rng.uniform(-0.5, 0.5, (2, 2048, 1)).astype(np.float32)
    mask = rng.random((2, 2048)) < 0.1
    b3 = random_boxes3d(rng, 10)
    b3[:, 1] = 2.2
    b3[:, 3:6] += np.array([6.0, 6.0, 2.0], np.float32)      # big enough to catch points
    b3[0, 3:6] = 0.01                                         # an empty box
    b3[1, 3:6] = np.array([60, 60, 10], np.float32)          # a box with far more than R points
    b3[1, 0], b3[1, 2] = 0.0, 35.0
    b3[2, 3:6] = np.array([60, 60, 10], np.float32)          # a full box for the second frame
    b3[2, 0], b3[2, 2] = 0.0, 35.0
    boxes = box_3d_to_8co(b3)
    box_ind = np.array([0, 0, 1, 1, 0, 1, 0, 1, 0, 1], np.int32)
    res = oracle.pc_crop_and_sample(pts, fts, inten, mask, boxes, box_ind, 32)
    assert not res[5][0] and res[5][1:].all(), res[5]
    names = ("crop_pts", "crop_fts", "crop_int", "crop_mask", "crop_ind", "non_empty")
    gcf = rng.standard_normal(res[1].shape).astype(np.float32)
    gf = oracle.pc_crop_and_sample_grad_fts(fts.shape, box_ind, res[4], gcf)
    # demo of cropping/tf_cropping.py:57-77: two points, unit box rotated 3.14/4, resize 1
    dp = np.asarray([[[1.0, 0, 0.1], [-0.3, -0.5, -0.3]]], np.float32)
    dbox = box_3d_to_8co(np.asarray([[0, 0, 0, 1, 1, 1, 3.14 / 4]], np.float32))
    dres = oracle.pc_crop_and_sample(dp, np.ones((1, 2, 1), np.float32), np.arange(2, dtype=np.float32).reshape(1, 2, 1),
                                     np.array([[True, False]]), dbox, np.array([0], np.int32), 1)
    # unit cube inside/outside points of hf/core/obj_utils_test.py:10-60 (same slab predicate)
    cube = box_3d_to_8co(np.asarray([[0.0, 0.5, 0.0, 1, 1, 1, 0]], np.float32))  # y in (-0.5, 0.5)... see test
    cp = np.asarray([[[0.1, 0.2, 0.1], [-0.6, 0.0, 0.0], [0.0, 0.6, 0.0]]], np.float32)
    cres = oracle.pc_crop_and_sample(cp, np.ones((1, 3, 1), np.float32), np.zeros((1, 3, 1), np.float32),
                                     np.zeros((1, 3), bool), cube, np.array([0], np.int32), 4)
    assert cres[4][0].tolist() == [0, 0, 0, 0] and cres[5][0]
    save("crop", pts=pts, fts=fts, intensities=inten, mask=mask, boxes3d=b3, boxes=boxes, box_ind=box_ind,
         grad_crop_fts=gcf, grad_fts=gf, demo_pts=dp, demo_box=dbox, demo_crop_ind=dres[4], demo_non_empty=dres[5],
         **dict(zip(names, res)))
    if have_ref:
        three_nn_origin()


if __name__ == "__main__":
    main()
