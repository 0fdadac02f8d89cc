"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
  (1) the committed golden vectors,
  (2) the CPU oracle on seeded inputs up to BASELINE.json's full sizes,
  (3) the reference's own CUDA kernels compiled unmodified by hipcc (oracle/_ref/libhfref_gpu.so,
      when the snapshot carries it),
  (4) size-independent properties.
Bar: integer outputs bit-exact; fp32 copies bit-exact; IoU / interpolation / gradients within 1e-5."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5  # BASELINE.json north_star: "within 1e-5 fp32 for interpolated features and IoU scores"


@pytest.fixture(scope="module")
def hf():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import heterofusionrcnn_amd as m
    return m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def kitti_uniform(rng, b, n):
    lo = np.array([-40.0, -5.0, 0.0], np.float32)
    hi = np.array([40.0, 3.0, 70.0], np.float32)
    return (lo + (hi - lo) * rng.random((b, n, 3), dtype=np.float32)).astype(np.float32)


# ------------------------------------------------------------------ golden vectors
def test_golden_grouping(hf):
    g = load_golden("grouping_P")
    xyz, q = dev(g["xyz"]), dev(g["new_xyz"])
    for tag, r, ns in (("r010", 0.1, 32), ("r030", 0.3, 32), ("r005k8", 0.05, 8)):
        idx, cnt = hf.query_ball_point(r, ns, xyz, q)
        assert np.array_equal(host(idx), g["idx_" + tag]), tag
        assert np.array_equal(host(cnt), g["cnt_" + tag]), tag
    idx, cnt = hf.query_ball_point(0.02, 16, xyz, dev(g["q_rand"]))
    assert np.array_equal(host(idx), g["idx_rand"]) and np.array_equal(host(cnt), g["cnt_rand"])
    idx = dev(g["idx_r010"])
    assert np.array_equal(host(hf.group_point(xyz, idx)), g["grouped_xyz_r010"])
    idx_f = idx[:, :64].contiguous()
    feats = dev(g["feats"]).requires_grad_(True)
    out = hf.group_point(feats, idx_f)
    assert np.array_equal(host(out), g["grouped_feat_r010"])
    out.backward(dev(g["grad_out"]))
    np.testing.assert_allclose(host(feats.grad), g["grad_points"], rtol=0, atol=TOL)
    # fused launch == the two ops
    i2, c2, gx = hf.query_ball_group(0.1, 32, xyz, q, center=False)
    assert np.array_equal(host(i2), g["idx_r010"]) and np.array_equal(host(c2), g["cnt_r010"])
    assert np.array_equal(host(gx), g["grouped_xyz_r010"])
    _, _, gc = hf.query_ball_group(0.1, 32, xyz, q, center=True)
    assert np.array_equal(host(gc), g["grouped_xyz_r010"] - g["new_xyz"][:, :, None, :])
    r = load_golden("grouping_reftest")
    idx, cnt = hf.query_ball_point(0.3, 32, dev(r["xyz1"]), dev(r["xyz2"]))
    assert np.array_equal(host(idx), r["idx"]) and np.array_equal(host(cnt), r["cnt"])
    assert np.array_equal(host(hf.group_point(dev(r["points"]), idx)), r["grouped"])


def _fps_both_kernels(hf, m, xyz_dev):
    """run the plain and the bucketed FPS kernel (hf_farthest_point_sample_variant) and insist they agree"""
    outs = []
    for mode, nt in (("plain", 0), ("bucket", 512), ("bucket", 1024)):     # 8 waves x 32 buckets / 16 waves x 16 buckets
        outs.append(hf.farthest_point_sample(m, xyz_dev, kernel=mode, threads=nt))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "plain and bucketed FPS disagree"
    if xyz_dev.shape[1] <= 512:       # one wave per cloud (what "auto" picks at this size)
        assert torch.equal(outs[0], hf.farthest_point_sample(m, xyz_dev, kernel="wave")), "plain and one-wave FPS disagree"
    auto = hf.farthest_point_sample(m, xyz_dev)
    assert torch.equal(auto, outs[0])
    return auto


@pytest.mark.parametrize("b,n,m", [(37, 512, 128), (5, 128, 32), (9, 32, 8), (3, 500, 600), (2, 64, 64), (4, 65, 10), (1, 1, 3)])
def test_fps_one_wave_per_cloud_against_oracle(hf, oracle_mod, b, n, m):
    """fps_wave_kernel (clouds of at most 512 points: the second stage's RoI clouds, rcnn_multiclass.config:157-186) against the
    oracle, with the cyclic padding of pc_crop_and_sample in the data (exact duplicates -> exact distance ties) and more samples than
    points"""
    rng = np.random.default_rng(b * 1000 + n)
    x = (rng.normal(0, 1, (b, n, 3)) * np.array([1.5, 0.8, 2.5])).astype(np.float32)
    if n >= 8:
        x[:, n // 2:] = x[:, :n - n // 2]          # the second half repeats the first: every distance has a tie
        x[0, :4] = x[0, 4:8]
    want = oracle_mod.farthest_point_sample(m, x)
    got = hf.farthest_point_sample(m, dev(x), kernel="wave")
    assert np.array_equal(host(got), want)
    assert np.array_equal(host(hf.farthest_point_sample(m, dev(x))), want)                 # auto
    assert np.array_equal(host(hf.farthest_point_sample(m, dev(x), kernel="plain")), want)


def test_golden_fps_gather(hf):
    g = load_golden("fps")
    for name, m in (("unit", 256), ("dup", 256), ("big", 1700), ("big2", 1400), ("tiny", 5)):
        out = _fps_both_kernels(hf, m, dev(g[name]))
        assert out.dtype == torch.int32
        assert np.array_equal(host(out), g[name + "_fps"]), name
    g = load_golden("gather")
    xyz = dev(g["xyz"]).requires_grad_(True)
    out = hf.gather_point(xyz, dev(g["idx"]))
    assert np.array_equal(host(out), g["out"])
    out.backward(dev(g["out_g"]))
    np.testing.assert_allclose(host(xyz.grad), g["inp_g"], rtol=0, atol=TOL)


def test_golden_interpolate(hf):
    g = load_golden("interpolate")
    dist, idx = hf.three_nn(dev(g["unknown"]), dev(g["known"]))
    assert np.array_equal(host(idx), g["idx"])
    assert np.array_equal(host(dist), g["dist"])  # same fp32 expression, no contraction: bit-exact
    d2, i2 = hf.three_nn(dev(g["unknown"][:, :16]), dev(g["known"][:, :2]))
    assert np.array_equal(host(i2), g["idx_m2"]) and np.array_equal(host(d2), g["dist_m2"])
    pts = dev(g["points"]).requires_grad_(True)
    out = hf.three_interpolate(pts, idx, dev(g["weight"]))
    np.testing.assert_allclose(host(out), g["out"], rtol=0, atol=TOL)
    assert np.array_equal(host(out), g["out"])
    out.backward(dev(g["grad_out"]))
    np.testing.assert_allclose(host(pts.grad), g["grad_points"], rtol=0, atol=TOL)
    from heterofusionrcnn_amd.interpolate import (three_interpolate_channel_first,
                                                  three_interpolate_channel_first_grad)
    cf = three_interpolate_channel_first(dev(g["points"].transpose(0, 2, 1)), idx, dev(g["weight"]))
    assert np.array_equal(host(cf).transpose(0, 2, 1), g["out"])
    gcf = three_interpolate_channel_first_grad((2, 12, 128), idx, dev(g["weight"]),
                                               dev(g["grad_out"].transpose(0, 2, 1)))
    np.testing.assert_allclose(host(gcf).transpose(0, 2, 1), g["grad_points"], rtol=0, atol=TOL)


def test_golden_select_top_k(hf):
    g = load_golden("select_top_k")
    oi, od = hf.select_top_k(5, dev(g["dist"]))
    assert np.array_equal(host(oi), g["outi"]) and np.array_equal(host(od), g["out"])


def test_golden_bev_iou_nms(hf):
    g = load_golden("bev_iou")
    ov, iou = hf.compute_bev_iou(dev(g["demo_prop"]), dev(g["demo_gt"]))
    assert np.array_equal(host(ov), g["demo_overlap"]) and np.array_equal(host(iou), g["demo_iou"])
    assert host(hf.oriented_nms(dev(g["demo_nms"]), 0.5)).tolist() == [0, 1, 0]
    ov, iou = hf.compute_bev_iou(dev(g["a"]), dev(g["b"]))
    np.testing.assert_allclose(host(ov), g["overlap_ab"], rtol=0, atol=2e-5)   # areas up to ~8 m^2
    np.testing.assert_allclose(host(iou), g["iou_ab"], rtol=0, atol=TOL)
    assert ((host(iou) == 0) == (g["iou_ab"] == 0)).all()                       # the early-out is exact
    nb = dev(g["nms_boxes"])
    for t in (0.85, 0.8, 0.01):
        keep, num = hf.oriented_nms(nb, t, return_count=True)
        want = g["keep_%03d" % int(t * 100)]
        assert np.array_equal(host(keep), want), t
        d = np.diff(want) <= 0
        assert int(host(num)[0]) == (1 + int(np.argmax(d)) if d.any() else len(want))


def test_golden_crop(hf):
    g = load_golden("crop")
    res = hf.pc_crop_and_sample(dev(g["pts"]), dev(g["fts"]), dev(g["intensities"]), dev(g["mask"]), dev(g["boxes"]),
                                dev(g["box_ind"]), 32)
    for got, name in zip(res, ("crop_pts", "crop_fts", "crop_int", "crop_mask", "crop_ind", "non_empty")):
        assert np.array_equal(host(got), g[name]), name
    from heterofusionrcnn_amd.cropping import pc_crop_and_sample_grad_fts
    gf = pc_crop_and_sample_grad_fts(dev(g["fts"]), dev(g["box_ind"]), res[4], dev(g["grad_crop_fts"]))
    np.testing.assert_allclose(host(gf), g["grad_fts"], rtol=0, atol=TOL)
    d = hf.pc_crop_and_sample(dev(g["demo_pts"]), torch.ones(1, 2, 1).cuda(),
                              torch.arange(2, dtype=torch.float32).reshape(1, 2, 1).cuda(),
                              torch.tensor([[True, False]]).cuda(), dev(g["demo_box"]),
                              torch.zeros(1, dtype=torch.int32).cuda(), 1)
    assert np.array_equal(host(d[4]), g["demo_crop_ind"]) and np.array_equal(host(d[5]), g["demo_non_empty"])


def _tf_running_var(torch_bn, rows):
    """moving variance after ONE step from the initial value 1 under the reference's rule (TensorFlow accumulates the
    BIASED batch variance, tf_util.py:571-581), derived from a torch BatchNorm that accumulated the unbiased one"""
    m = torch_bn.momentum
    return (1.0 - m) + (torch_bn.running_var - (1.0 - m)) * (rows - 1) / rows


# ------------------------------------------------------------------ oracle, seeded inputs, ragged shapes
@pytest.mark.parametrize("b,n,m,r,ns", [(2, 1024, 256, 0.12, 32), (3, 1000, 77, 0.2, 16), (1, 5000, 300, 0.08, 64),
                                        (2, 300, 513, 0.5, 7), (1, 64, 1, 10.0, 128), (1, 3000, 5000, 0.15, 16), (1, 2000, 9000, 0.2, 8),
                                        (40, 2048, 2000, 0.06, 32)])   # the last: 256 queries per workgroup (batched launches)
def test_oracle_ball_query(hf, oracle_mod, b, n, m, r, ns):
    rng = np.random.default_rng(b * 1000 + n)
    x1 = rng.random((b, n, 3), dtype=np.float32)
    x2 = rng.random((b, m, 3), dtype=np.float32)
    idx, cnt = hf.query_ball_point(r, ns, dev(x1), dev(x2))
    oi, oc = oracle_mod.query_ball_point(r, ns, x1, x2)
    assert np.array_equal(host(idx), oi) and np.array_equal(host(cnt), oc)
    i2, c2, gx = hf.query_ball_group(r, ns, dev(x1), dev(x2), center=True)
    assert np.array_equal(host(i2), oi) and np.array_equal(host(c2), oc)
    assert np.array_equal(host(gx), oracle_mod.group_point(x1, oi) - x2[:, :, None, :])


def test_ball_query_both_kernels_and_stress_paths(hf, oracle_mod):
    """cell kernel (ballquery.hip): clustered queries, dense data (candidate buffer overflow -> chunked walk with early
    stop, more than two hits per list -> serial hand-off), hits >> nsample, several register segments (n > 16384), every
    LDS geometry (nsample 32/64/128), the exhaustive mode (coordinates ~1e6 radii from the origin) and the brute-force
    kernel (nsample > 128, n > 2^19, variant="bruteforce"), and the cell-sorted pair of kernels (variant="sorted": dense buckets ->
    more than two hits per lane, the exhaustive walk, every table size) -- all must give the oracle's rows"""
    rng = np.random.default_rng(77)
    cases = []
    # clustered queries: ~2900 of 3000 queries sit in a 1 cm cube -> one slab owns them all (rounds)
    x1 = kitti_uniform(rng, 1, 6000)
    q = kitti_uniform(rng, 1, 3000)
    q[0, :2900] = np.array([3.0, -1.0, 20.0], np.float32) + rng.uniform(0, 0.01, (2900, 3)).astype(np.float32)
    x1[0, :500] = np.array([3.0, -1.0, 20.0], np.float32) + rng.uniform(-0.4, 0.4, (500, 3)).astype(np.float32)
    cases.append((x1, q, 0.5, 32))
    # dense data: 20000 points inside a 4 m cube, radius 1.0 -> candidate buffer flushes, hundreds of hits per ball
    x1 = rng.uniform(0, 4, (2, 20000, 3)).astype(np.float32)
    q = rng.uniform(0, 4, (2, 500, 3)).astype(np.float32)
    cases.append((x1, q, 1.0, 32))
    cases.append((x1[:, :5000], q, 0.7, 64))
    cases.append((x1[:, :3000], q[:, :100], 1.5, 128))
    cases.append((x1[:, :3000], q[:, :100], 1.5, 140))     # brute-force geometry
    # all queries identical on every axis (zero extent), duplicates in the data
    x1 = kitti_uniform(rng, 1, 4096)
    x1[0, 2000:2100] = x1[0, 0]
    q = np.repeat(x1[:, :1], 300, axis=1)
    cases.append((x1, q, 0.05, 16))
    # huge radius: every point is a hit for every query
    cases.append((kitti_uniform(rng, 2, 700), kitti_uniform(rng, 2, 130), 500.0, 32))
    # coordinates a million radii away from the origin: the padded box of a query may span three cells -> exhaustive mode
    x1 = (kitti_uniform(rng, 1, 3000) + np.float32(1.0e6)).astype(np.float32)
    cases.append((x1, x1[:, ::7].copy(), 0.5, 16))
    # tiny radius against the coordinates (same switch), hits are the duplicates only
    x1 = kitti_uniform(rng, 1, 2500)
    x1[0, 100:110] = x1[0, 5]
    cases.append((x1, x1[:, :300].copy(), 1e-5, 8))
    # more than 2^19 points per cloud: outside the cell kernel's index packing -> brute force
    x1 = kitti_uniform(rng, 1, (1 << 19) + 5)
    cases.append((x1, x1[:, -12:].copy(), 0.3, 16))
    for (a, b, r, ns) in cases:
        oi, oc = oracle_mod.query_ball_point(r, ns, a, b)
        for mode in ("auto", "bruteforce") + (("sorted",) if ns <= 128 else ()):
            idx, cnt, gx = hf.query_ball_group(r, ns, dev(a), dev(b), center=False, variant=mode)
            assert np.array_equal(host(idx), oi), (mode, r, ns)
            assert np.array_equal(host(cnt), oc), (mode, r, ns)
            assert np.array_equal(host(gx), oracle_mod.group_point(a, oi)), (mode, r, ns)
        if ns <= 128:
            i2, c2 = hf.query_ball_point(r, ns, dev(a), dev(b), variant="sorted")
            assert np.array_equal(host(i2), oi) and np.array_equal(host(c2), oc), ("sorted, no grouping", r, ns)


def test_ball_threshold_boundary(hf, oracle_mod):
    """points exactly at / one ulp around distance == radius: the sqrt-free test must agree with
    max(sqrtf(s),1e-20f) < radius"""
    rs = [0.1, 0.3, 0.5, 1.0, 2.0, 4.0, 0.37, 1e-3, 123.456]
    for r in rs:
        r32 = np.float32(r)
        base = np.zeros((1, 64, 3), np.float32)
        d = r32
        vals = [d]
        up, dn = d, d
        for _ in range(20):
            up = np.nextafter(up, np.float32(np.inf)); dn = np.nextafter(dn, np.float32(0))
            vals += [up, dn]
        base[0, :len(vals), 0] = np.array(vals, np.float32)
        base[0, len(vals):, 0] = 1e6
        q = np.zeros((1, 1, 3), np.float32)
        idx, cnt = hf.query_ball_point(float(r32), 64, dev(base), dev(q))
        oi, oc = oracle_mod.query_ball_point(float(r32), 64, base, q)
        assert np.array_equal(host(idx), oi) and np.array_equal(host(cnt), oc), r


@pytest.mark.parametrize("b,n,m", [(2, 1024, 300), (1, 1500, 1500), (3, 700, 64), (1, 4096, 700), (2, 3000, 500),
                                   (1, 16384, 600), (1, 20000, 128), (2, 5, 5), (1, 1, 1), (2, 10, 25), (1, 3000, 3100)])
def test_oracle_fps(hf, oracle_mod, b, n, m):
    rng = np.random.default_rng(n + m)
    xyz = kitti_uniform(rng, b, n)
    if n >= 1000:  # duplicated tail as in kitti_dataset.py:358-364 -> exact distance ties
        xyz[:, n - n // 10:] = xyz[:, :n // 10]
    out = _fps_both_kernels(hf, m, dev(xyz)) if n <= 16384 else hf.farthest_point_sample(m, dev(xyz))
    assert np.array_equal(host(out), oracle_mod.farthest_point_sample(m, xyz))


def test_oracle_fps_clustered(hf, oracle_mod):
    """non-uniform cloud (dense blobs + sparse background + a flat ground plane): stresses the bucket skip test"""
    rng = np.random.default_rng(12)
    n = 8192
    xyz = kitti_uniform(rng, 2, n)
    xyz[:, :3000] = np.array([5.0, -1.0, 12.0], np.float32) + rng.normal(0, 0.5, (2, 3000, 3)).astype(np.float32)
    xyz[:, 3000:5000, 1] = 1.7                      # ground plane
    xyz[:, 5000:5200] = xyz[:, 0:200]               # duplicates
    out = _fps_both_kernels(hf, 2048, dev(xyz))
    assert np.array_equal(host(out), oracle_mod.farthest_point_sample(2048, xyz))


def test_oracle_fps_degenerate(hf, oracle_mod):
    """all points identical / more samples than distinct points: every distance ties at 0"""
    xyz = np.ones((2, 2000, 3), np.float32)
    xyz[1, :7] = np.arange(21, dtype=np.float32).reshape(7, 3)
    out = _fps_both_kernels(hf, 40, dev(xyz))
    assert np.array_equal(host(out), oracle_mod.farthest_point_sample(40, xyz))


@pytest.mark.parametrize("b,n,m", [(2, 2048, 512), (1, 1000, 3), (3, 333, 1), (1, 100, 5000)])
def test_oracle_three_nn(hf, oracle_mod, b, n, m):
    rng = np.random.default_rng(n * 7 + m)
    u = kitti_uniform(rng, b, n)
    k = kitti_uniform(rng, b, m)
    if m >= 8:
        k[:, m // 2:m // 2 + 4] = k[:, :4]  # exact duplicates among the known points: earlier index first
        u[:, :4] = k[:, :4]
    dist, idx = hf.three_nn(dev(u), dev(k))
    od, oi = oracle_mod.three_nn(u, k)
    assert np.array_equal(host(idx), oi)
    assert np.array_equal(host(dist), od)


def test_three_nn_against_reference_cpu_outputs_at_the_origin(hf):
    """tests/golden/three_nn_origin.npz = outputs of the reference's compiled threenn_cpu (interpolate/interpolate.cpp:21-64;
    made by tests/golden/make_golden.py) for unknown points at the origin, where its x2*x2+y2*y2+z2*z2 is the fp32 expression
    of tf_interpolate_g.cu:47.  Both HIP kernels (grid ring search and all-pairs scan) must give those bits: cascade, tie
    order among duplicated / mirrored / lattice points, +inf / index 0 when fewer than three points are known."""
    from heterofusionrcnn_amd.interpolate import three_nn_all_pairs
    g = load_golden("three_nn_origin")
    names = [k[:-6] for k in g if k.endswith("_known")]
    assert len(names) == 9
    for name in names:
        known = g[name + "_known"]
        u = np.zeros((known.shape[0], 5, 3), np.float32)
        for fn in (hf.three_nn, three_nn_all_pairs):
            dist, idx = fn(dev(u), dev(known))
            for row in range(5):
                assert np.array_equal(host(idx)[:, row], g[name + "_idx"][:, 0]), (name, fn.__name__)
                assert np.array_equal(host(dist)[:, row].view(np.uint32), g[name + "_dist"][:, 0].view(np.uint32)), (name, fn.__name__)


def test_three_nn_translated_lattice_against_reference_cpu(hf, oracle_mod):
    """where u - x is exact (dyadic lattice) the reference expression on (u, known) has the bits of threenn_cpu on the cloud
    translated by -u: the HIP kernels against reference-compiled code with a NON-zero query (tests/test_oracle.py has the
    oracle's twin).  Needs oracle/_ref (travels with the snapshot)."""
    from heterofusionrcnn_amd.interpolate import three_nn_all_pairs
    if not oracle_mod.ref_available("itp"):
        pytest.skip("oracle/_ref/libhfref_itp.so not built")
    rng = np.random.default_rng(23)
    known = (rng.integers(-2048, 2048, (700, 3)) / 64.0).astype(np.float32)
    unknown = (rng.integers(-2048, 2048, (512, 3)) / 64.0).astype(np.float32)
    rd, ri = oracle_mod.ref_threenn_origin((unknown[:, None, :] - known[None, :, :]).astype(np.float32))
    for fn in (hf.three_nn, three_nn_all_pairs):
        dist, idx = fn(dev(unknown[None]), dev(known[None]))
        assert np.array_equal(host(idx)[0], ri[:, 0]) and np.array_equal(host(dist)[0].view(np.uint32), rd[:, 0].view(np.uint32))


@pytest.mark.parametrize("kind", ["lattice", "same_x", "clustered", "two_known", "nonfinite", "max_lds", "beyond_lds"])
def test_three_nn_sorted_sweep_matches_all_pairs_scan(hf, oracle_mod, kind):
    """hf.three_nn bins the known points into a 2-D grid and searches rings of cells (the k = 3 case of the kNN
    kernel); it must give the all-pairs scan's answer bit for bit (distances, and ties to the lower index) on inputs
    built to stress the search's stop rule"""
    from heterofusionrcnn_amd.interpolate import three_nn_all_pairs
    rng = np.random.default_rng(len(kind))
    b, n, m = 2, 3000, 700
    if kind == "lattice":        # integer lattice: masses of exactly equal distances and equal x
        k = rng.integers(0, 6, (b, m, 3)).astype(np.float32)
        u = (rng.integers(0, 6, (b, n, 3)) + 0.5 * rng.integers(0, 2, (b, n, 3))).astype(np.float32)
    elif kind == "same_x":       # no pruning possible along x
        k = kitti_uniform(rng, b, m); k[..., 0] = 3.25
        u = kitti_uniform(rng, b, n)
    elif kind == "clustered":
        k = (rng.standard_normal((b, m, 3)) * 0.05).astype(np.float32) + np.float32(10)
        k[:, ::7] += 30
        u = kitti_uniform(rng, b, n); u[:, ::2] = (k[:, rng.integers(0, m, n // 2)] + np.float32(1e-3)).astype(np.float32)
    elif kind == "two_known":
        m = 2
        k = kitti_uniform(rng, b, m); u = kitti_uniform(rng, b, n)
    elif kind == "nonfinite":    # NaN / inf coordinates never win a `<` in the reference either
        k = kitti_uniform(rng, b, m); u = kitti_uniform(rng, b, n)
        k[0, 5, 0] = np.nan; k[0, 9, 1] = np.inf; k[1, 3, 0] = -np.inf; u[0, 7, 0] = np.nan; u[1, 8, 2] = np.inf
    elif kind == "max_lds":
        b, n, m = 1, 2000, 8192
        k = kitti_uniform(rng, b, m); u = kitti_uniform(rng, b, n)
    else:                        # a cloud larger than the former LDS limit: the grid path handles it too
        b, n, m = 1, 500, 8193
        k = kitti_uniform(rng, b, m); u = kitti_uniform(rng, b, n)
    dist, idx = hf.three_nn(dev(u), dev(k))
    d2, i2 = three_nn_all_pairs(dev(u), dev(k))
    assert torch.equal(idx, i2)
    assert np.array_equal(host(dist), host(d2), equal_nan=True)
    if kind != "nonfinite":
        od, oi = oracle_mod.three_nn(u, k)
        assert np.array_equal(host(idx), oi) and np.array_equal(host(dist), od)


@pytest.mark.parametrize("b,n,m,c", [(2, 400, 90, 16), (1, 1, 1, 4), (3, 5000, 7, 30), (2, 16384, 4096, 128), (1, 300, 8192, 1)])
def test_three_nn_inverse_and_gather_gradient(hf, oracle_mod, b, n, m, c):
    """CSR inverse of a three_nn index and the gather form of ThreeInterpolateGrad: buckets hold ascending
    (unknown*3+slot) positions, so the sum order is the reference's sequential loop -> bit-exact with the oracle"""
    from heterofusionrcnn_amd.interpolate import three_nn_inverse
    rng = np.random.default_rng(n + m)
    idx3 = rng.integers(0, m, (b, n, 3)).astype(np.int32)
    if n > 100:
        idx3[0, : n // 2, 1] = m // 2          # one heavily referenced known point
    w = rng.random((b, n, 3), dtype=np.float32)
    off, ent = three_nn_inverse(dev(idx3), m)
    off, ent = host(off), host(ent)
    for i in range(b):
        flat = idx3[i].reshape(-1)
        order = np.argsort(flat, kind="stable")
        assert np.array_equal(off[i], np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=m))]))
        assert np.array_equal(ent[i], order)
    pts = rng.standard_normal((b, m, c)).astype(np.float32)
    go = rng.standard_normal((b, n, c)).astype(np.float32)
    p = dev(pts).requires_grad_(True)
    out = hf.three_interpolate(p, dev(idx3), dev(w), inverse=three_nn_inverse(dev(idx3), m))
    assert np.array_equal(host(out), oracle_mod.three_interpolate(pts, idx3, w))
    out.backward(dev(go))
    assert np.array_equal(host(p.grad), oracle_mod.three_interpolate_grad(pts.shape, idx3, w, go))


@pytest.mark.parametrize("c,c1", [(128, 1), (16, 16), (30, 0), (7, 5)])
def test_three_interpolate_concat_against_interpolate_and_cat(hf, c, c1):
    """interpolate.three_interpolate_concat = cat([three_interpolate(points2), points1]) plus zero columns, bit for
    bit; gradients w.r.t. both feature tensors equal the two-op form's (gather form: same summation order)"""
    from heterofusionrcnn_amd.interpolate import three_interpolate_concat, three_nn_inverse
    rng = np.random.default_rng(c + c1)
    b, n, m = 2, 700, 90
    idx3 = dev(rng.integers(0, m, (b, n, 3)).astype(np.int32))
    w = dev(rng.random((b, n, 3), dtype=np.float32))
    inv = three_nn_inverse(idx3, m)
    p2 = dev(rng.standard_normal((b, m, c)).astype(np.float32))
    p1 = dev(rng.standard_normal((b, n, c1)).astype(np.float32)) if c1 else None
    a2 = p2.clone().requires_grad_(True); b2 = p2.clone().requires_grad_(True)
    a1 = p1.clone().requires_grad_(True) if c1 else None
    b1 = p1.clone().requires_grad_(True) if c1 else None
    out = three_interpolate_concat(a2, a1, idx3, w, inv)
    width = (c + c1 + 3) // 4 * 4
    assert out.shape == (b, n, width)
    interp = hf.three_interpolate(b2, idx3, w, inverse=inv)
    ref = torch.cat([interp, b1], dim=2) if c1 else interp
    assert torch.equal(out[..., :c + c1], ref) and bool((out[..., c + c1:] == 0).all())
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g[..., :c + c1].contiguous())
    assert torch.equal(a2.grad, b2.grad)
    if c1:
        assert torch.equal(a1.grad, b1.grad)


def test_three_nn_inverse_drops_out_of_range(hf):
    from heterofusionrcnn_amd.interpolate import three_nn_inverse
    idx3 = np.array([[[0, 5, 1], [-1, 1, 2], [1, 1, 99]]], np.int32)
    off, ent = three_nn_inverse(dev(idx3), 3)
    assert host(off).tolist() == [[0, 1, 5, 6]]
    assert host(ent)[0, :6].tolist() == [0, 2, 4, 6, 7, 5]
    with pytest.raises(ValueError):
        three_nn_inverse(dev(idx3), 8193)


@pytest.mark.parametrize("c", [1, 3, 16, 30, 256])
def test_oracle_interpolate_group_channels(hf, oracle_mod, c):
    rng = np.random.default_rng(c)
    b, n, m, ns = 2, 400, 90, 8
    pts = rng.standard_normal((b, m, c)).astype(np.float32)
    idx3 = rng.integers(0, m, (b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3), dtype=np.float32)
    p = dev(pts).requires_grad_(True)
    out = hf.three_interpolate(p, dev(idx3), dev(w))
    assert np.array_equal(host(out), oracle_mod.three_interpolate(pts, idx3, w))
    go = rng.standard_normal((b, n, c)).astype(np.float32)
    out.backward(dev(go))
    np.testing.assert_allclose(host(p.grad), oracle_mod.three_interpolate_grad(pts.shape, idx3, w, go), rtol=0,
                               atol=5e-5)
    idx = rng.integers(0, m, (b, n, ns)).astype(np.int32)
    p2 = dev(pts).requires_grad_(True)
    g = hf.group_point(p2, dev(idx))
    assert np.array_equal(host(g), oracle_mod.group_point(pts, idx))
    go = rng.standard_normal((b, n, ns, c)).astype(np.float32)
    g.backward(dev(go))
    np.testing.assert_allclose(host(p2.grad), oracle_mod.group_point_grad(pts.shape, idx, go), rtol=0, atol=2e-4)


def test_group_point_gradient_check_reference_criterion(hf):
    """grouping/tf_grouping_op_test.py:12-28: numeric-vs-analytic gradient error < 1e-4 for group_point
    w.r.t. points through query_ball_point indices, shapes (1,128,16)/(1,128,3)/(1,8,3), r=0.3, ns=32"""
    r = load_golden("grouping_reftest")
    pts = dev(r["points"]).double().float()
    idx, _ = hf.query_ball_point(0.3, 32, dev(r["xyz1"]), dev(r["xyz2"]))
    p = pts.clone().requires_grad_(True)
    out = hf.group_point(p, idx)
    gen = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(4):
        gvec = torch.randn(out.shape, device="cuda", generator=gen)
        v = torch.randn(pts.shape, device="cuda", generator=gen)
        (analytic,) = torch.autograd.grad(out, p, gvec, retain_graph=True)
        eps = 0.5
        numeric = (hf.group_point(pts + eps * v, idx) - hf.group_point(pts - eps * v, idx)) / (2 * eps)
        lhs = (analytic * v).sum().item()
        rhs = (gvec * numeric).sum().item()
        assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def _rand_bev(rng, n):
    cx, cz = rng.uniform(-40, 40, n), rng.uniform(0, 70, n)
    l, w = np.clip(rng.normal(3.9, 0.4, n), 0.5, None), np.clip(rng.normal(1.6, 0.1, n), 0.5, None)
    ry = rng.uniform(-np.pi, np.pi, n)
    return np.stack([cx - l / 2, cz - w / 2, cx + l / 2, cz + w / 2, ry], 1).astype(np.float32)


def test_oracle_bev_iou(hf, oracle_mod):
    rng = np.random.default_rng(3)
    a = _rand_bev(rng, 3000)
    b = _rand_bev(rng, 64)
    a[:500, :4] = np.repeat(b[:50, :4], 10, 0) + rng.normal(0, 0.3, (500, 4)).astype(np.float32)  # overlapping ones
    a[500:600] = np.repeat(b[:10], 10, 0)                                                        # identical
    a[600:610, 4] = 0.0; a[610:620, 4] = np.float32(np.pi / 2)                                    # axis aligned
    ov, iou = hf.compute_bev_iou(dev(a), dev(b))
    oo, oi = oracle_mod.compute_bev_iou(a, b)
    np.testing.assert_allclose(host(iou), oi, rtol=0, atol=TOL)
    np.testing.assert_allclose(host(ov), oo, rtol=0, atol=3e-5)
    assert ((host(iou) == 0) == (oi == 0)).all()
    assert (oi > 0.3).sum() > 100  # the case is not trivially all-zero
    # odd shapes: single row / single column / non multiple of the chunk
    for na, nb in ((1, 1), (1, 77), (130, 1), (257, 9)):
        o2, i2 = hf.compute_bev_iou(dev(a[:na]), dev(a[500:500 + nb]))
        e2 = oracle_mod.compute_bev_iou(a[:na], a[500:500 + nb])
        np.testing.assert_allclose(host(i2), e2[1], rtol=0, atol=TOL)


def _clustered(rng, clusters, copies):
    base = _rand_bev(rng, clusters)
    boxes = np.repeat(base, copies, 0)
    shift = rng.normal(0, 0.3, (len(boxes), 2)).astype(np.float32)
    boxes[:, [0, 2]] += shift[:, :1]
    boxes[:, [1, 3]] += shift[:, 1:]
    boxes[:, 4] += rng.normal(0, 0.1, len(boxes)).astype(np.float32)
    return boxes[rng.permutation(len(boxes))].astype(np.float32)


def _drop_borderline(hf, boxes, thresh, margin=3e-5):
    """fixture with a margin: one box of every pair whose IoU lies within `margin` of the threshold goes (device and host IoUs
    agree to 1e-5), so a keep vector can be compared with the oracle's unconditionally"""
    import oracle
    _, iou = hf.compute_bev_iou(dev(boxes), dev(boxes))
    iou = host(iou)
    _, oiou = oracle.compute_bev_iou(boxes, boxes)
    near = (np.abs(iou - thresh) < margin) | (np.abs(oiou - thresh) < margin)
    near &= ~((iou == 0) & (oiou == 0))                          # disjoint on both sides: exact zeros, nothing borderline
    drop = set()
    for i, j in np.argwhere(np.triu(near, 1)):
        if i not in drop and j not in drop:
            drop.add(int(j))
    return np.ascontiguousarray(np.delete(boxes, sorted(drop), axis=0))


@pytest.mark.parametrize("n,thresh", [(1, 0.5), (64, 0.7), (65, 0.0), (1000, 0.8), (1500, 0.01)])
def test_oracle_nms(hf, oracle_mod, n, thresh):
    rng = np.random.default_rng(n)
    boxes = _clustered(rng, max(1, n // 10), 10)[:n]
    if n in (64, 65):                                            # keep the word-boundary sizes exact after the margin pass
        boxes = _drop_borderline(hf, _clustered(rng, 10, 10), thresh)[:n]
        assert len(boxes) == n
    else:
        boxes = _drop_borderline(hf, boxes, thresh)
        assert len(boxes) >= n - n // 20
    keep, num = hf.oriented_nms(dev(boxes), thresh, return_count=True)
    ok, okept = oracle_mod.oriented_nms(boxes, thresh, return_count=True)
    assert np.array_equal(host(keep), ok), "keep differs (no IoU of this fixture is within 3e-5 of the threshold)"
    assert int(host(num)[0]) == okept
    # the raw mask, every tile (drop-in for oriented_nms_gpu)
    mask = host(hf.nms_mask(dev(boxes), thresh)).view(np.uint64)
    assert np.array_equal(mask, oracle_mod.nms_mask(boxes, thresh))


def test_oracle_crop(hf, oracle_mod):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import box_3d_to_8co, random_boxes3d
    rng = np.random.default_rng(9)
    bsz, p, c, nb, r = 3, 5000, 20, 40, 100
    pts = kitti_uniform(rng, bsz, p)
    pts[:, :, 1] = rng.uniform(0, 2, (bsz, p)).astype(np.float32)
    fts = rng.standard_normal((bsz, p, c)).astype(np.float32)
    inten = rng.uniform(-.5, .5, (bsz, p, 1)).astype(np.float32)
    mask = rng.random((bsz, p)) < 0.2
    b3 = random_boxes3d(rng, nb)
    b3[:, 1] = 2.2
    b3[:, 3:6] += rng.uniform(0, 25, (nb, 1)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
    b3[:3, 3:6] = 0.01  # empty boxes
    boxes = box_3d_to_8co(b3)
    box_ind = rng.integers(0, bsz, nb).astype(np.int32)
    for cc, ff in ((c, fts), (3, fts[:, :, :3].copy())):  # float4 path and scalar path
        res = hf.pc_crop_and_sample(dev(pts), dev(ff), dev(inten), dev(mask), dev(boxes), dev(box_ind), r)
        want = oracle_mod.pc_crop_and_sample(pts, ff, inten, mask, boxes, box_ind, r)
        for got, w, name in zip(res, want, ("pts", "fts", "int", "mask", "ind", "non_empty")):
            assert np.array_equal(host(got), w), name
    ne = want[5]
    assert ne.any() and (~ne).any()


# ------------------------------------------------------------------ full BASELINE sizes
def test_full_size_headline_ball_group(hf, oracle_mod):
    """B=8, N=16384, M=4096, K=32, kitti-uniform cloud, queries = gather(fps)"""
    rng = np.random.default_rng(0)
    xyz = kitti_uniform(rng, 8, 16384)
    x = dev(xyz)
    fps = hf.farthest_point_sample(4096, x)
    assert np.array_equal(host(fps), oracle_mod.farthest_point_sample(4096, xyz))
    new_xyz = hf.gather_point(x, fps)
    q = host(new_xyz)
    assert np.array_equal(q, oracle_mod.gather_point(xyz, host(fps)))
    for r in (0.5, 2.0):
        idx, cnt, gx = hf.query_ball_group(r, 32, x, new_xyz, center=True)
        oi, oc = oracle_mod.query_ball_point(r, 32, xyz, q)
        assert np.array_equal(host(idx), oi) and np.array_equal(host(cnt), oc)
        assert np.array_equal(host(gx), oracle_mod.group_point(xyz, oi) - q[:, :, None, :])
        i2, c2 = hf.query_ball_point(r, 32, x, new_xyz)
        assert torch.equal(i2, idx) and torch.equal(c2, cnt)
        assert (host(cnt) >= 1).all()  # every query is a data point
    dist, i3 = hf.three_nn(x, new_xyz)
    od, oi3 = oracle_mod.three_nn(xyz, q)
    assert np.array_equal(host(i3), oi3) and np.array_equal(host(dist), od)
    # levels 2 and 3 of the config-2 stack at full size: 4096 -> 1024 (r = 1.0), 1024 -> 256 (r = 2.0), K = 32
    lvl_x, lvl_np = new_xyz, q
    for npoint, r in ((1024, 1.0), (256, 2.0)):
        f = hf.farthest_point_sample(npoint, lvl_x)
        assert np.array_equal(host(f), oracle_mod.farthest_point_sample(npoint, lvl_np))
        nx = hf.gather_point(lvl_x, f)
        nq = host(nx)
        idx, cnt, gx = hf.query_ball_group(r, 32, lvl_x, nx, center=True)
        oi, oc = oracle_mod.query_ball_point(r, 32, lvl_np, nq)
        assert np.array_equal(host(idx), oi) and np.array_equal(host(cnt), oc), (npoint, r)
        assert np.array_equal(host(gx), oracle_mod.group_point(lvl_np, oi) - nq[:, :, None, :])
        d3, j3 = hf.three_nn(lvl_x, nx)
        od3, oj3 = oracle_mod.three_nn(lvl_np, nq)
        assert np.array_equal(host(j3), oj3) and np.array_equal(host(d3), od3)
        lvl_x, lvl_np = nx, nq


def test_full_size_bev_iou_properties(hf):
    """70 000 x 64 (BASELINE config 3): symmetry, range, self-IoU, zero pattern vs a bounding-circle bound"""
    rng = np.random.default_rng(3)
    a = _rand_bev(rng, 70000)
    b = _rand_bev(rng, 64)
    ov, iou = hf.compute_bev_iou(dev(a), dev(b))
    ov_t, iou_t = hf.compute_bev_iou(dev(b), dev(a))
    np.testing.assert_allclose(host(iou), host(iou_t).T, rtol=0, atol=TOL)
    i = host(iou)
    assert (i >= 0).all() and (i <= 1 + TOL).all() and np.isfinite(i).all()
    ca = np.stack([(a[:, 0] + a[:, 2]) / 2, (a[:, 1] + a[:, 3]) / 2], 1)
    cb = np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2], 1)
    ra = np.hypot(a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]) / 2
    rb = np.hypot(b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]) / 2
    far = np.linalg.norm(ca[:, None] - cb[None], axis=-1) > (ra[:, None] + rb[None]) + 1e-3
    assert (i[far] == 0).all()
    _, self_iou = hf.compute_bev_iou(dev(b), dev(b))
    np.testing.assert_allclose(np.diag(host(self_iou)), 1.0, atol=TOL)


@pytest.mark.parametrize("thresh", [0.5, 0.9])
def test_oracle_nms_dense_columns(hf, oracle_mod, thresh):
    """> 2048 jittered copies of ONE box: every mask word is nonzero, so the per-column-block lists of the sweep
    (2048 entries) overflow and the last columns are gathered from the dense mask instead.  The fixture is built with a
    margin: boxes that take part in a pair whose IoU lies within 3e-5 of the threshold are dropped first, so the keep list
    must equal the oracle's unconditionally (device and host IoUs agree to 1e-5)."""
    rng = np.random.default_rng(11)
    boxes = _clustered(rng, 1, 2700)
    _, iou = hf.compute_bev_iou(dev(boxes), dev(boxes))
    near = np.abs(host(iou) - thresh) < 3e-5
    drop = set()
    for i, j in np.argwhere(np.triu(near, 1)):                 # one box of every borderline pair goes
        if i not in drop and j not in drop:
            drop.add(int(j))
    boxes = np.ascontiguousarray(np.delete(boxes, sorted(drop), axis=0))
    assert len(boxes) > 2200, len(boxes)                       # still overflows the 2048-entry lists
    keep, num = hf.oriented_nms(dev(boxes), thresh, return_count=True)
    ok, okept = oracle_mod.oriented_nms(boxes, thresh, return_count=True)
    assert np.array_equal(host(keep), ok)
    assert int(host(num)[0]) == okept


def test_full_size_nms_properties(hf):
    """N=9000 (rpn_multiclass.config:25): kept set is an independent set, every dropped box is covered"""
    rng = np.random.default_rng(4)
    boxes = _clustered(rng, 300, 30)
    bd = dev(boxes)
    for thresh in (0.8, 0.01):
        keep, num = hf.oriented_nms(bd, thresh, return_count=True)
        kept = int(host(num)[0])
        k = host(keep)
        assert k[0] == 0 and (np.diff(k[:kept]) > 0).all() and (k[kept:] == k[0]).all()
        kb = bd[keep[:kept].long()]
        _, iou = hf.compute_bev_iou(kb, kb)
        iou = host(iou)
        np.fill_diagonal(iou, 0)
        assert (iou <= thresh + TOL).all()                       # no kept pair above the threshold
        dropped = np.setdiff1d(np.arange(len(boxes)), k[:kept])
        _, cover = hf.compute_bev_iou(bd[torch.from_numpy(dropped).cuda()], kb)
        cover = host(cover)
        earlier = k[:kept][None, :] < dropped[:, None]
        assert ((cover > thresh - TOL) & earlier).any(1).all()   # each dropped box has an earlier kept suppressor


# ------------------------------------------------------------------ reference CUDA kernels compiled by hipcc
def _ref_gpu():
    path = os.path.join(ROOT, "oracle", "_ref", "libhfref_gpu.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libhfref_gpu.so not in this snapshot")
    return ctypes.CDLL(path)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_reference_kernels_grouping_sampling(hf):
    R = _ref_gpu()
    rng = np.random.default_rng(21)
    b, n, m, ns, c = 4, 4096, 1024, 32, 16
    xyz = kitti_uniform(rng, b, n)
    xyz[:, n - 400:] = xyz[:, :400]
    x = dev(xyz)
    # FPS: the real 512-thread tree reduction with its tie behaviour
    temp = torch.empty((32, n), dtype=torch.float32, device="cuda")
    ref_fps = torch.zeros((b, m), dtype=torch.int32, device="cuda")
    assert R.hfref_farthest_point_sample(b, n, m, _p(x), _p(temp), _p(ref_fps)) == 0
    fps = hf.farthest_point_sample(m, x)
    assert torch.equal(fps, ref_fps)
    ref_q = torch.empty((b, m, 3), dtype=torch.float32, device="cuda")
    assert R.hfref_gather_point(b, n, m, _p(x), _p(fps), _p(ref_q)) == 0
    q = hf.gather_point(x, fps)
    assert torch.equal(q, ref_q)
    for r in (0.5, 1.5, 4.0):
        ref_idx = torch.zeros((b, m, ns), dtype=torch.int32, device="cuda")
        ref_cnt = torch.zeros((b, m), dtype=torch.int32, device="cuda")
        assert R.hfref_query_ball_point(b, n, m, ctypes.c_float(r), ns, _p(x), _p(q), _p(ref_idx), _p(ref_cnt)) == 0
        idx, cnt = hf.query_ball_point(r, ns, x, q)
        assert torch.equal(idx, ref_idx) and torch.equal(cnt, ref_cnt), r
    feats = dev(rng.standard_normal((b, n, c)).astype(np.float32))
    ref_g = torch.empty((b, m, ns, c), dtype=torch.float32, device="cuda")
    assert R.hfref_group_point(b, n, c, m, ns, _p(feats), _p(idx), _p(ref_g)) == 0
    assert torch.equal(hf.group_point(feats, idx), ref_g)
    go = dev(rng.standard_normal((b, m, ns, c)).astype(np.float32))
    ref_gp = torch.zeros((b, n, c), dtype=torch.float32, device="cuda")
    assert R.hfref_group_point_grad(b, n, c, m, ns, _p(go), _p(idx), _p(ref_gp)) == 0
    f2 = feats.clone().requires_grad_(True)
    hf.group_point(f2, idx).backward(go)
    torch.testing.assert_close(f2.grad, ref_gp, rtol=0, atol=2e-4)   # atomics: order differs, sums of <= ~100 terms


def test_reference_kernels_bev_iou(hf, oracle_mod):
    R = _ref_gpu()
    # a fixture whose pairwise a x a IoUs (the ones NMS compares) all keep a margin from both thresholds: the keep
    # vector then cannot depend on the last ulps of sinf / cosf / atan2f, and equality is asserted unconditionally
    for seed in range(22, 60):
        rng = np.random.default_rng(seed)
        a = _clustered(rng, 60, 10)
        self_iou = oracle_mod.compute_bev_iou(a, a)[1]
        if all(not np.any(np.abs(self_iou - t) < 1e-4) for t in (0.7, 0.05)):
            break
    else:
        raise AssertionError("no margin-safe fixture found")
    b = a[:64].copy()
    ad, bd = dev(a), dev(b)
    ref_ov = torch.zeros((len(a), len(b)), device="cuda")
    ref_iou = torch.zeros((len(a), len(b)), device="cuda")
    assert R.hfref_compute_bev_iou(len(a), _p(ad), len(b), _p(bd), _p(ref_ov), _p(ref_iou)) == 0
    ov, iou = hf.compute_bev_iou(ad, bd)
    torch.testing.assert_close(iou, ref_iou, rtol=0, atol=TOL)
    torch.testing.assert_close(ov, ref_ov, rtol=0, atol=3e-5)
    assert torch.equal(iou == 0, ref_iou == 0)
    # NMS: reference mask kernel + the host sweep of bev_iou.cpp:87-112 (restated in the oracle)
    n = len(a)
    cb = (n + 63) // 64
    for thresh in (0.7, 0.05):
        ref_mask = torch.zeros((n, cb), dtype=torch.int64, device="cuda")
        assert R.hfref_nms_mask(_p(ad), _p(ref_mask), n, ctypes.c_float(thresh)) == 0
        want, kept = oracle_mod.nms_sweep(host(ref_mask).view(np.uint64))
        keep, num = hf.oriented_nms(ad, thresh, return_count=True)
        assert np.array_equal(host(keep), want), "keep differs from the reference kernels"
        assert int(host(num)[0]) == kept
        assert torch.equal(hf.nms_mask(ad, thresh), ref_mask)


def test_reference_kernels_crop(hf):
    """the reference appends in atomic order: compare as sets (and exactly where cnt <= R leaves no choice)"""
    R = _ref_gpu()
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import box_3d_to_8co, random_boxes3d
    rng = np.random.default_rng(23)
    bsz, p, c, nb, r = 2, 4096, 8, 24, 64
    pts = kitti_uniform(rng, bsz, p)
    pts[:, :, 1] = rng.uniform(0, 2, (bsz, p)).astype(np.float32)
    b3 = random_boxes3d(rng, nb)
    b3[:, 1] = 2.2
    b3[:, 3:6] += rng.uniform(0, 12, (nb, 1)).astype(np.float32) * np.array([1, 1, 0.3], np.float32)
    t = dict(pts=dev(pts), fts=dev(rng.standard_normal((bsz, p, c)).astype(np.float32)),
             inten=dev(rng.uniform(-.5, .5, (bsz, p, 1)).astype(np.float32)), mask=dev(rng.random((bsz, p)) < 0.2),
             boxes=dev(box_3d_to_8co(b3)), box_ind=dev(rng.integers(0, bsz, nb).astype(np.int32)))
    mine = hf.pc_crop_and_sample(t["pts"], t["fts"], t["inten"], t["mask"], t["boxes"], t["box_ind"], r)
    ref = [torch.zeros_like(x) for x in mine]
    ref[5].fill_(True)
    assert R.hfref_pc_crop_and_sample(_p(t["pts"]), _p(t["fts"]), _p(t["inten"]), _p(t["mask"]), _p(t["boxes"]),
                                      _p(t["box_ind"]), nb, bsz, p, r, c, 1, *[_p(x) for x in ref]) == 0
    assert torch.equal(mine[5], ref[5])
    mi, ri = host(mine[4]), host(ref[4])
    checked = 0
    for bx in range(nb):
        if not host(mine[5])[bx]:
            continue
        row = mi[bx]
        cnt = next((s for s in range(1, r) if row[s] <= row[s - 1]), r)
        if cnt < r:  # all inside points taken: same set, ours ascending; pad slots are copies of taken ones
            assert sorted(set(ri[bx].tolist())) == row[:cnt].tolist()
            checked += 1
    assert checked >= 3


# ------------------------------------------------------------------ streams / re-entrancy
def test_runs_on_the_callers_stream(hf, oracle_mod):
    rng = np.random.default_rng(5)
    x1 = rng.random((2, 2048, 3), dtype=np.float32)
    x2 = rng.random((2, 256, 3), dtype=np.float32)
    s = torch.cuda.Stream()
    a, b = dev(x1), dev(x2)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        idx, cnt = hf.query_ball_point(0.1, 16, a, b)
        fps = hf.farthest_point_sample(64, a)
    s.synchronize()
    oi, oc = oracle_mod.query_ball_point(0.1, 16, x1, x2)
    assert np.array_equal(host(idx), oi) and np.array_equal(host(fps), oracle_mod.farthest_point_sample(64, x1))


# ------------------------------------------------------------------ callers: SA/FP modules, prefetch pipeline
def test_stack_with_prefetched_geometry_matches_inline(hf):
    from heterofusionrcnn_amd import modules
    from heterofusionrcnn_amd.pipeline import GeometryPrefetcher
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    sa = ((512, 2.0, 16, (16, 32)), (128, 4.0, 16, (32, 64)))
    net = modules.PointnetSAFPStack(in_channel=1, sa=sa, fp=((64, 64), (32, 32))).cuda()
    xyz = dev(kitti_uniform(rng, 2, 2048))
    inten = dev(rng.uniform(-.5, .5, (2, 2048, 1)).astype(np.float32))
    out_inline = net(xyz, inten)
    pf = GeometryPrefetcher(net.geometry)
    pf.submit(xyz)
    geo = pf.get()
    out_pre = net(xyz, inten, geometry=geo)
    assert out_inline.shape == (2, 2048, 32)
    assert torch.equal(out_inline, out_pre)
    out_pre.mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    # two batches in flight, one HIP stream each: results come back in submission order and match inline
    xyz_b = dev(kitti_uniform(rng, 2, 2048))
    pf2 = GeometryPrefetcher(net.geometry, depth=2)
    pf2.submit(xyz); pf2.submit(xyz_b)
    assert len(pf2) == 2
    with torch.no_grad():
        for pts in (xyz, xyz_b, xyz):
            geo = pf2.get()
            pf2.submit(pts)
            assert torch.equal(net(pts, inten, geometry=geo), net(pts, inten))
    # groups: the clouds of 3 batches sampled in one launch, results handed back batch by batch (a partial
    # group at the end is flushed by get)
    pf3 = GeometryPrefetcher(net.geometry, depth=2, group=3)
    order = [xyz, xyz_b, xyz_b, xyz, xyz_b]
    for pts in order[:4]:
        pf3.submit(pts)
    assert len(pf3) == 4 and pf3.capacity == 6
    with torch.no_grad():
        for i, pts in enumerate(order):
            geo = pf3.get()
            if i == 0:
                pf3.submit(order[4])
            assert torch.equal(net(pts, inten, geometry=geo), net(pts, inten))
    assert len(pf3) == 0


@pytest.mark.parametrize("c", [1, 5, 64, 127])
def test_group_concat_against_group_point_and_cat(hf, c):
    """grouping.group_concat = cat([grouped_xyz, group_point(points, idx)]) plus zero columns up to a multiple of 4,
    bit for bit, and the same gradient w.r.t. points as the two-op form"""
    from heterofusionrcnn_amd.grouping import group_concat
    rng = np.random.default_rng(c)
    b, n, m, k = 2, 300, 40, 8
    pts = dev(rng.standard_normal((b, n, c)).astype(np.float32))
    idx = dev(rng.integers(0, n, (b, m, k)).astype(np.int32))
    gxyz = dev(rng.standard_normal((b, m, k, 3)).astype(np.float32))
    p1 = pts.clone().requires_grad_(True); p2 = pts.clone().requires_grad_(True)
    out = group_concat(p1, idx, gxyz)
    width = (3 + c + 3) // 4 * 4
    assert out.shape == (b, m, k, width)
    ref = torch.cat([gxyz, hf.group_point(p2, idx)], dim=-1)
    assert torch.equal(out[..., :3 + c], ref) and bool((out[..., 3 + c:] == 0).all())
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g[..., :3 + c].contiguous())
    torch.testing.assert_close(p1.grad, p2.grad, rtol=1e-5, atol=1e-5)  # atomics: summation order differs
    with pytest.raises(ValueError):
        group_concat(pts, idx, gxyz, width=3 + c + 1 if (3 + c + 1) % 4 else 3 + c - 1)
    # the multi-scale module's order: [features, xyz, padding]
    p3 = pts.clone().requires_grad_(True); p4 = pts.clone().requires_grad_(True)
    out2 = group_concat(p3, idx, gxyz, xyz_last=True)
    ref2 = torch.cat([hf.group_point(p4, idx), gxyz], dim=-1)
    assert torch.equal(out2[..., :3 + c], ref2) and bool((out2[..., 3 + c:] == 0).all())
    out2.backward(g); ref2.backward(g[..., :3 + c].contiguous())
    torch.testing.assert_close(p3.grad, p4.grad, rtol=1e-5, atol=1e-5)


def test_sa_module_composition_against_oracle(hf, oracle_mod):
    """sample_and_group restated from pointnet_util.py:24-66: op order, centring, concat order [xyz, feats]"""
    from heterofusionrcnn_amd import modules
    rng = np.random.default_rng(1)
    xyz = rng.random((2, 1024, 3), dtype=np.float32)
    feats = rng.standard_normal((2, 1024, 5)).astype(np.float32)
    new_xyz, new_points, idx, gxyz = modules.sample_and_group(128, 0.2, 16, dev(xyz), dev(feats))
    o_fps = oracle_mod.farthest_point_sample(128, xyz)
    o_new = oracle_mod.gather_point(xyz, o_fps)
    o_idx, _ = oracle_mod.query_ball_point(0.2, 16, xyz, o_new)
    o_gxyz = oracle_mod.group_point(xyz, o_idx) - o_new[:, :, None, :]
    o_np = np.concatenate([o_gxyz, oracle_mod.group_point(feats, o_idx)], -1)
    assert np.array_equal(host(new_xyz), o_new) and np.array_equal(host(idx), o_idx)
    assert np.array_equal(host(gxyz), o_gxyz) and np.array_equal(host(new_points), o_np)
    # MSG order is [feats, xyz] (pointnet_util.py:264)
    msg = modules.PointnetSAModuleMSG(128, [0.2], [16], 5, [[8]], bn=False).cuda()
    with torch.no_grad():
        _, out = msg(dev(xyz), dev(feats))
        w, b = msg.mlps[0][0].fc.weight, msg.mlps[0][0].fc.bias
        want = torch.relu(dev(np.concatenate([oracle_mod.group_point(feats, o_idx), o_gxyz], -1)) @ w.t() + b).max(dim=2).values
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    # two scales with BatchNorm (training): fused [features, xyz, pad] rows + one node per scale, against the
    # op-by-op form (group_point, cat, Linear, BatchNorm1d, relu, max) with the same weights; channel count 6 + 3 = 9
    # is padded to 12 inside
    torch.manual_seed(1)
    feats6 = dev(np.random.default_rng(3).standard_normal((2, 1024, 6)).astype(np.float32))
    msg2 = modules.PointnetSAModuleMSG(128, [0.2, 0.4], [16, 32], 6, [[8, 16], [8, 12]]).cuda()
    _, out2 = msg2(dev(xyz), feats6)
    assert out2.shape == (2, 128, 28)
    new2 = hf.gather_point(dev(xyz), hf.farthest_point_sample(128, dev(xyz)))
    refs = []
    for radius, ns, mlp in zip([0.2, 0.4], [16, 32], msg2.mlps):
        idx2, _, gx2 = hf.query_ball_group(radius, ns, dev(xyz), new2, center=True)
        h = torch.cat([hf.group_point(feats6, idx2), gx2], dim=-1).reshape(-1, 9)
        for layer in mlp:
            bn = torch.nn.BatchNorm1d(layer.fc.out_features, eps=1e-3).cuda()
            h = torch.relu(bn(torch.nn.functional.linear(h, layer.fc.weight, layer.fc.bias)))
        refs.append(h.view(2, 128, ns, -1).max(dim=2).values)
    torch.testing.assert_close(out2, torch.cat(refs, dim=-1), rtol=2e-4, atol=5e-5)


def test_fp_module_composition_against_oracle(hf, oracle_mod):
    """pointnet_fp_module restated from pointnet_util.py:289-330: three_nn -> dist = max(dist, 1e-10) -> w = (1/d) / sum(1/d)
    -> three_interpolate -> concat [interpolated, points1] -> conv2d 1x1 + BN + ReLU.  The oracle runs the ops, numpy the
    weights / concat / MLP (batch statistics, eps 1e-3)."""
    from heterofusionrcnn_amd import modules
    rng = np.random.default_rng(9)
    for (n, m, c1, c2, widths) in ((1024, 256, 5, 16, (24, 8)), (700, 3, 0, 7, (6,))):
        xyz1 = rng.random((2, n, 3), dtype=np.float32)
        xyz2 = xyz1[:, rng.permutation(n)[:m]].copy()          # known points are a subset: exact zeros among the distances
        p1 = rng.standard_normal((2, n, c1)).astype(np.float32) if c1 else None
        p2 = rng.standard_normal((2, m, c2)).astype(np.float32)
        torch.manual_seed(n)
        fp = modules.PointnetFPModule(c1 + c2, list(widths)).cuda().train()
        out = fp(dev(xyz1), dev(xyz2), dev(p1) if c1 else None, dev(p2))
        dist, idx = oracle_mod.three_nn(xyz1, xyz2)
        d = np.maximum(dist, np.float32(1e-10))
        w = (np.float32(1.0) / d) / (np.float32(1.0) / d).sum(axis=2, keepdims=True)
        interp = oracle_mod.three_interpolate(p2, idx, w.astype(np.float32))
        x = np.concatenate([interp, p1], axis=2) if c1 else interp           # [interpolated, points1] (:311-313)
        # the composition up to the MLP, bit for bit: the module's own pieces on the same inputs
        gi, gw, ginv = modules.PointnetFPModule.geometry(dev(xyz1), dev(xyz2))
        assert np.array_equal(host(gi), idx)
        np.testing.assert_allclose(host(gw), w, rtol=0, atol=1e-6)
        pre = hf.three_interpolate(dev(p2), gi, gw)
        np.testing.assert_allclose(host(pre), interp, rtol=0, atol=1e-5)
        h = x.reshape(-1, x.shape[-1]).astype(np.float64)
        for layer in fp.mlp:
            z = h @ host(layer.fc.weight).T.astype(np.float64) + host(layer.fc.bias)
            z = (z - z.mean(0)) / np.sqrt(z.var(0) + 1e-3) * host(layer.bn.weight) + host(layer.bn.bias)
            h = np.maximum(z, 0.0)
        np.testing.assert_allclose(host(out).reshape(-1, widths[-1]), h, rtol=2e-4, atol=2e-4)


def test_full_size_config2_stack_fwd_bwd_against_op_by_op_form(hf):
    """BASELINE config 2 at full size (B=8, 16384 -> 4096 -> 1024 -> 256, K=32, fp32): the fused path (grouped geometry,
    group_concat, MFMA linear/BN/ReLU(+max-pool) nodes, interpolate+concat, gather-form gradients) against the op-by-op
    torch form that materialises what the reference materialises (group_point -> concat -> 1x1 conv -> BN -> ReLU -> max;
    three_interpolate -> concat -> ...), same weights: output and every parameter gradient"""
    from heterofusionrcnn_amd import modules
    from heterofusionrcnn_amd.modules import three_nn_weights
    F = torch.nn.functional
    torch.manual_seed(5)
    rng = np.random.default_rng(5)
    xyz = dev(kitti_uniform(rng, 8, 16384))
    inten = dev(rng.uniform(-0.5, 0.5, (8, 16384, 1)).astype(np.float32))
    model = modules.PointnetSAFPStack(in_channel=1).cuda().train()

    def layer_ref(layer, x):
        z = F.linear(x, layer.fc.weight, layer.fc.bias)
        mu, var = z.mean(0), z.var(0, unbiased=False)
        return torch.relu((z - mu) / torch.sqrt(var + 1e-3) * layer.bn.weight + layer.bn.bias)

    def stack_ref(xyz, pts):
        xyzs, feats = [xyz], [pts]
        for m in model.sa:
            new_xyz = hf.gather_point(xyzs[-1], hf.farthest_point_sample(m.npoint, xyzs[-1]))
            idx, _ = hf.query_ball_point(m.radius, m.nsample, xyzs[-1], new_xyz)
            g = torch.cat([hf.group_point(xyzs[-1], idx) - new_xyz.unsqueeze(2), hf.group_point(feats[-1], idx)], -1)
            b_, n_, k_, c_ = g.shape
            x = g.reshape(-1, c_)
            for layer in m.mlp:
                x = layer_ref(layer, x)
            xyzs.append(new_xyz)
            feats.append(x.reshape(b_, n_, k_, -1).max(2).values)
        up = feats[-1]
        for i, m in enumerate(model.fp):
            d = len(model.sa) - 1 - i
            dist, idx = hf.three_nn(xyzs[d], xyzs[d + 1])
            x = torch.cat([hf.three_interpolate(up, idx, three_nn_weights(dist)), feats[d]], 2)
            b_, n_, c_ = x.shape
            x = x.reshape(-1, c_)
            for layer in m.mlp:
                x = layer_ref(layer, x)
            up = x.reshape(b_, n_, -1)
        return up

    # the loss: mean squared distance to a fixed per-channel target.  (A random cotangent makes every column sum of the
    # backward pass a heavily cancelling sum in which ONE flipped ReLU mask -- either fp32 path flips a few against an fp64
    # evaluation -- is a percent-level change; profiles/r03_grad_noise.md has both variants, per node and end to end.)
    tgt = torch.linspace(-1.0, 1.0, model.out_channel, device="cuda")
    lossf = lambda o: ((o - tgt) ** 2).mean()
    params = list(model.parameters())
    out = model(xyz, inten, geometry=model.geometry(xyz))
    grads = torch.autograd.grad(lossf(out), params, allow_unused=True)
    ref = stack_ref(xyz, inten)
    grads_ref = torch.autograd.grad(lossf(ref), params, allow_unused=True)
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-3)
    # Measured against an fp64 evaluation of the same graph (scripts/probes/grad_noise_probe.py, LOSS=smooth): the op-by-op fp32
    # form is within 0.19 % of a parameter's largest gradient component, the fused path within 0.61 % (one parameter; the rest
    # within 0.31 %); node by node the HIP passes are as accurate as the framework's (scripts/probes/bn_layer_noise.py,
    # pool_node_noise.py).  Bound between the two fp32 paths: 1 % of the largest component (round 2: 3 %).
    for (name, _), a, b in zip(model.named_parameters(), grads, grads_ref):
        if a is None or b is None or name.endswith("fc.bias"):
            # a bias in front of a BatchNorm has an exactly-zero gradient analytically: the fused node returns zeros,
            # autograd through the op-by-op form returns rounding noise
            assert (a is None or float(a.abs().max()) < 1e-3) and (b is None or float(b.abs().max()) < 1e-3), name
            continue
        scale = float(b.abs().max()) + 1e-9
        assert float((a - b).abs().max()) <= 1e-2 * scale + 1e-8, (name, float((a - b).abs().max()), scale)


def test_iou3d_and_nms_adapters(hf, oracle_mod):
    """box3d_iou / oriented_nms_3d / sb_nms restated from compute_iou.py:23-80, model_util.py:101-142"""
    from heterofusionrcnn_amd import modules
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import random_boxes3d, boxes3d_to_bev
    rng = np.random.default_rng(2)
    a = random_boxes3d(rng, 60)
    a[30:] = a[:30] + rng.normal(0, 0.2, (30, 7)).astype(np.float32)
    b = a[:20].copy()
    i3, i2 = modules.box3d_iou(dev(a), dev(b))
    ov, o2 = oracle_mod.compute_bev_iou(boxes3d_to_bev(a), boxes3d_to_bev(b))
    np.testing.assert_allclose(host(i2), o2, atol=TOL)
    h = np.clip(np.minimum(a[:, None, 1], b[None, :, 1]) - np.maximum(a[:, None, 1] - a[:, None, 5], b[None, :, 1] - b[None, :, 5]), 0, None)
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None]
    np.testing.assert_allclose(host(i3), ov * h / np.clip(va + vb - ov * h, 1e-7, None), atol=2e-5)
    np.testing.assert_allclose(np.diag(host(i3)[:20]), 1.0, atol=1e-4)
    scores = rng.random(60).astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    want = order[oracle_mod.oriented_nms(boxes3d_to_bev(a)[order], 0.1)]
    got = modules.oriented_nms_3d(dev(a), dev(scores), 0.1)
    assert np.array_equal(host(got), want)
    ind, n = modules.sb_nms(dev(a), dev(scores), 0.1, 100, fixed_num_proposal_nms=False)
    kept = oracle_mod.oriented_nms(boxes3d_to_bev(a)[order], 0.1, return_count=True)[1]
    assert n == kept and (host(ind)[n:] == -1).all() and np.array_equal(host(ind)[:n], want[:n])
    c8 = modules.box_3d_to_box_8co(dev(a[:4]))
    from make_golden import box_3d_to_8co
    np.testing.assert_allclose(host(c8), box_3d_to_8co(a[:4]), atol=1e-5)


@pytest.mark.parametrize("rows,c,relu", [(5000, 32, True), (777, 96, True), (4096, 196, False), (300, 7, True),
                                         (70000, 64, True), (33, 256, True), (1024, 64, True), (1025, 64, True), (2, 8, True),
                                         (64, 1280, False), (1500, 20, True)])
def test_fused_bn_relu_against_torch(hf, rows, c, relu):
    """csrc/mlp.hip vs nn.BatchNorm1d(eps=1e-3, momentum=0.1) [+ relu]: forward, backward, running stats, eval"""
    from heterofusionrcnn_amd.mlp import BatchNormReLU
    torch.manual_seed(rows + c)
    x = (torch.randn(rows, c, device="cuda") * 2.0 + 0.5)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.1).cuda()
    mine = BatchNormReLU(c, eps=1e-3, momentum=0.1, relu=relu).cuda()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-0.5, 0.5)
        mine.weight.copy_(ref.weight); mine.bias.copy_(ref.bias)
    x1 = x.clone().requires_grad_(True)
    x2 = x.clone().requires_grad_(True)
    y1 = ref(x1)
    if relu:
        y1 = torch.relu(y1)
    y2 = mine(x2)
    torch.testing.assert_close(y2, y1, rtol=1e-4, atol=1e-5)
    g = torch.randn_like(y1)
    y1.backward(g)
    y2.backward(g)
    torch.testing.assert_close(x2.grad, x1.grad, rtol=1e-3, atol=2e-5)
    torch.testing.assert_close(mine.weight.grad, ref.weight.grad, rtol=1e-3, atol=1e-3 * (rows ** 0.5) * 1e-1)
    torch.testing.assert_close(mine.bias.grad, ref.bias.grad, rtol=1e-3, atol=1e-3 * (rows ** 0.5) * 1e-1)
    torch.testing.assert_close(mine.running_mean, ref.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mine.running_var, _tf_running_var(ref, rows), rtol=1e-4, atol=1e-5)
    ref.running_var.copy_(mine.running_var)
    ref.eval(); mine.eval()
    with torch.no_grad():
        e1 = ref(x)
        if relu:
            e1 = torch.relu(e1)
        torch.testing.assert_close(mine(x), e1, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("b,n,m,k", [(2, 2048, 512, 8), (1, 300, 1000, 32), (3, 64, 5, 64), (1, 5000, 257, 1)])
def test_oracle_knn(hf, oracle_mod, b, n, m, k):
    """hf_knn_point vs the oracle's k smallest (distance, index) pairs; duplicated data points exercise the
    lower-index-first tie rule.  (The reference's tf.nn.top_k is third-party: parity unpinned beyond this rule.)"""
    rng = np.random.default_rng(k * 31 + n)
    x1 = kitti_uniform(rng, b, n)
    x2 = kitti_uniform(rng, b, m)
    if n >= 64:
        x1[:, n // 2:n // 2 + 8] = x1[:, :8]
        x2[:, :4] = x1[:, :4]
    val, idx = hf.knn_point(k, dev(x1), dev(x2), all_pairs=False)
    ov, oi = oracle_mod.knn_point(k, x1, x2)
    assert np.array_equal(host(idx), oi)
    assert np.array_equal(host(val), ov)
    assert (np.diff(host(val), axis=-1) >= 0).all()
    v2, i2 = hf.knn_point(k, dev(x1), dev(x2), all_pairs=True)      # the tiled all-pairs kernel: same answer
    assert torch.equal(i2, idx) and torch.equal(v2, val)


@pytest.mark.parametrize("kind", ["lattice", "same_x", "clustered", "k_equals_n", "max_sorted", "beyond_sorted"])
def test_knn_sorted_sweep_matches_all_pairs_scan(hf, oracle_mod, kind):
    """hf.knn_point bins the data into a 2-D grid and searches rings of cells; it must give the tiled all-pairs kernel's
    answer bit for bit (distances, and ties to the lower index) on inputs built to stress the search's stop rule"""
    rng = np.random.default_rng(len(kind) + 3)
    b, n, m, k = 2, 900, 2500, 8
    if kind == "lattice":        # integer lattice: masses of exactly equal distances and equal x
        x1 = rng.integers(0, 6, (b, n, 3)).astype(np.float32)
        x2 = (rng.integers(0, 6, (b, m, 3)) + 0.5 * rng.integers(0, 2, (b, m, 3))).astype(np.float32)
    elif kind == "same_x":       # no pruning possible along x
        x1 = kitti_uniform(rng, b, n); x1[..., 0] = -7.5
        x2 = kitti_uniform(rng, b, m)
    elif kind == "clustered":
        x1 = (rng.standard_normal((b, n, 3)) * 0.05).astype(np.float32) + np.float32(10)
        x1[:, ::7] += 30
        x2 = kitti_uniform(rng, b, m); x2[:, ::2] = (x1[:, rng.integers(0, n, m // 2)] + np.float32(1e-3)).astype(np.float32)
    elif kind == "k_equals_n":
        n, k = 40, 40
        x1 = kitti_uniform(rng, b, n); x2 = kitti_uniform(rng, b, m)
    elif kind == "max_sorted":
        b, n, m, k = 1, 16384, 1500, 16
        x1 = kitti_uniform(rng, b, n); x2 = kitti_uniform(rng, b, m)
    else:                        # one more data point than the binned path takes: the all-pairs kernel answers
        b, n, m, k = 1, 65537, 300, 4
        x1 = kitti_uniform(rng, b, n); x2 = kitti_uniform(rng, b, m)
        x2[0, :5] = [[1e6, 0, 0], [-1e6, 0, 0], [np.nan, 0, 0], [0, np.inf, 0], [0, 0, 0]]   # queries far outside / non-finite
    if kind == "max_sorted":
        x2[0, :3] = [[1e6, 0, 0], [-1e6, 0, 0], [0, 1e30, 0]]           # queries far outside the data's x range
        x1[0, :4, 0] = [np.inf, -np.inf, np.nan, 3e38]                   # non-finite / huge data coordinates
    val, idx = hf.knn_point(k, dev(x1), dev(x2), all_pairs=False)
    v2, i2 = hf.knn_point(k, dev(x1), dev(x2), all_pairs=True)
    assert torch.equal(idx, i2) and np.array_equal(host(val), host(v2), equal_nan=True)
    if n <= 1000:
        ov, oi = oracle_mod.knn_point(k, x1, x2)
        assert np.array_equal(host(idx), oi) and np.array_equal(host(val), ov)


def test_sample_and_group_knn_mode(hf, oracle_mod):
    """sample_and_group(knn=True), the mode both shipped PointNet++ configs use (rpn_cars_pointnet.config:61)"""
    from heterofusionrcnn_amd import modules
    rng = np.random.default_rng(4)
    xyz = kitti_uniform(rng, 2, 1024)
    new_xyz, new_points, idx, gxyz = modules.sample_and_group(128, 0.0, 8, dev(xyz), None, knn=True)
    o_new = oracle_mod.gather_point(xyz, oracle_mod.farthest_point_sample(128, xyz))
    _, o_idx = oracle_mod.knn_point(8, xyz, o_new)
    assert np.array_equal(host(idx), o_idx)
    assert np.array_equal(host(gxyz), oracle_mod.group_point(xyz, o_idx) - o_new[:, :, None, :])
    assert (host(idx)[:, :, 0] == oracle_mod.farthest_point_sample(128, xyz)).all()  # nearest neighbour = itself


def test_caller_side_ops_on_empty_and_ragged_shapes(hf, oracle_mod):
    """zero-size and non-tile-multiple shapes through the entry points added for the SA/FP callers"""
    from heterofusionrcnn_amd import modules
    from heterofusionrcnn_amd.grouping import group_concat
    from heterofusionrcnn_amd.interpolate import three_interpolate_concat, three_nn_inverse
    from heterofusionrcnn_amd.mlp import linear_bn_fwd, linear_wgrad, shared_mlp, BatchNormReLU
    # no queries: an empty (b, 0, k, width) result and a zero gradient
    pts = torch.randn(2, 50, 6, device="cuda", requires_grad=True)
    out = group_concat(pts, torch.zeros((2, 0, 4), dtype=torch.int32, device="cuda"), torch.zeros((2, 0, 4, 3), device="cuda"))
    assert out.shape == (2, 0, 4, 12)
    out.sum().backward()
    assert torch.equal(pts.grad, torch.zeros_like(pts))
    # no unknown points: empty interpolation, empty inverse, zero gradient
    idx0 = torch.zeros((2, 0, 3), dtype=torch.int32, device="cuda")
    off, ent = three_nn_inverse(idx0, 7)
    assert off.shape == (2, 8) and not off.any() and ent.shape == (2, 0)
    p2 = torch.randn(2, 7, 8, device="cuda", requires_grad=True)
    o = three_interpolate_concat(p2, None, idx0, torch.zeros((2, 0, 3), device="cuda"), (off, ent))
    assert o.shape == (2, 0, 8)
    o.sum().backward()
    assert torch.equal(p2.grad, torch.zeros_like(p2))
    # three_nn with fewer than three known points, one unknown point
    d, i = hf.three_nn(dev(np.zeros((1, 1, 3), np.float32)), dev(np.ones((1, 2, 3), np.float32)))
    od, oi = oracle_mod.three_nn(np.zeros((1, 1, 3), np.float32), np.ones((1, 2, 3), np.float32))
    assert np.array_equal(host(i), oi) and np.array_equal(host(d), od)
    # MFMA kernels on shapes that are not multiples of any tile: 1 row, 129 rows, odd channel counts
    for rows, cin, cout in [(1, 3, 2), (129, 33, 65), (257, 130, 31)]:
        x = torch.randn(rows, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda"); b = torch.randn(cout, device="cuda")
        bn = BatchNormReLU(cout).cuda()
        z, mean, invstd, _ = linear_bn_fwd(x, w, b, bn)
        ref = x.double() @ w.double().t() + b.double()
        torch.testing.assert_close(z.double(), ref, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(mean.double(), ref.mean(0), rtol=1e-5, atol=1e-5)
        g = torch.randn(rows, cout, device="cuda")
        torch.testing.assert_close(linear_wgrad(g, x).double(), g.double().t() @ x.double(), rtol=1e-5, atol=1e-4)
    # a one-row shared MLP (batch statistics of a single row: variance 0) stays finite
    layers = torch.nn.ModuleList([modules.SharedMLPLayer(3, 4), modules.SharedMLPLayer(4, 5)]).cuda()
    y = shared_mlp(layers, torch.randn(1, 3, device="cuda"))
    assert y.shape == (1, 5) and torch.isfinite(y).all()


# ------------------------------------------------------------------ glue either side of the ops
@pytest.mark.parametrize("c", [1, 6, 32])
def test_project_gather_against_oracle(hf, oracle_mod, c):
    """fusion.project_gather (projection + int cast + gather_nd of image features, one kernel): pixels and gathered
    rows bit-exact against the oracle, zeros outside the image, gradient = scatter-add onto the pixels"""
    from heterofusionrcnn_amd.fusion import project_gather, rect_to_image
    rng = np.random.default_rng(c)
    b, p, h, w = 2, 5000, 37, 123
    pts = np.stack([rng.uniform(-40, 40, (b, p)), rng.uniform(-3, 2, (b, p)), rng.uniform(0.5, 70, (b, p))], -1).astype(np.float32)
    pts[0, :5, 2] = [-1.0, 0.0, 1e-30, np.inf, np.nan]          # behind the camera, zero / tiny / non-finite depth
    calib = np.tile(np.array([[70.0, 0, 61.0, 4.0], [0, 70.0, 18.0, -0.2], [0, 0, 1, 0.003]], np.float32), (b, 1, 1))
    calib[1, 0, 0] = 55.0
    img = rng.standard_normal((b, h, w, c)).astype(np.float32)
    im = dev(img).requires_grad_(True)
    out, pix = project_gather(dev(pts), dev(calib), im, return_pixels=True)
    o_out, o_pix = oracle_mod.project_gather(pts, calib, img)
    assert np.array_equal(host(pix), o_pix) and np.array_equal(host(out), o_out)
    inside = (o_pix[..., 0] >= 0) & (o_pix[..., 0] < w) & (o_pix[..., 1] >= 0) & (o_pix[..., 1] < h)
    assert 0.2 < inside.mean() < 0.95 and not host(out)[~inside].any()
    # the float pixels of the unfused form truncate to the same integers away from pixel edges
    fp = host(rect_to_image(dev(pts), dev(calib)))
    ok = inside & np.isfinite(fp).all(-1) & (np.abs(fp - np.round(fp)) > 1e-3).all(-1)
    assert np.array_equal(np.trunc(fp[ok]).astype(np.int32), o_pix[ok])
    go = rng.standard_normal((b, p, c)).astype(np.float32)
    out.backward(dev(go))
    np.testing.assert_allclose(host(im.grad), oracle_mod.project_gather_grad(img.shape, o_pix, go), rtol=0, atol=1e-4)
    empty = project_gather(dev(np.zeros((2, 0, 3), np.float32)), dev(calib), dev(img))
    assert empty.shape == (2, 0, c)
    with pytest.raises(ValueError):
        project_gather(dev(pts), dev(calib[:, :2]), dev(img))


@pytest.mark.parametrize("rank,with_theta", [(3, False), (3, True), (2, True)])
def test_bin_box_codec_against_oracle(hf, oracle_mod, rank, with_theta):
    """box_codec.encode / decode (bin_based_box3d_encoder.py as one kernel each) against the oracle: bit-exact without
    a reference heading (no trigonometry), 1e-5 with one (sin/cos differ by ulps between ocml and glibc); bins equal
    away from bin edges; encode -> decode reproduces the boxes"""
    from heterofusionrcnn_amd import box_codec
    rng = np.random.default_rng(rank * 2 + with_theta)
    ss, deltas, r, nbt, k = [3.0, 1.5], [0.5, 0.25], 0.25 * np.pi, 12, 2
    dt = 2 * r / nbt
    lead = (3, 700) if rank == 3 else (2100,)
    rows = int(np.prod(lead))
    ref = rng.uniform(-30, 30, lead + (3,)).astype(np.float32)
    th = rng.uniform(-3, 3, lead).astype(np.float32) if with_theta else None
    boxes = np.empty(lead + (7,), np.float32)
    boxes[..., :3] = ref + rng.uniform(-1.0, 1.0, lead + (3,))
    boxes[..., 3:6] = rng.uniform(0.5, 4.0, lead + (3,))
    boxes[..., 6] = rng.uniform(-0.9 * r, 0.9 * r, lead) + (th if with_theta else 0)
    mean = rng.uniform(1, 3, lead + (3,)).astype(np.float32)
    got = box_codec.encode(dev(ref), dev(th) if with_theta else 0, dev(boxes), dev(mean), ss, deltas, r, dt, k)
    want = oracle_mod.bin_box_encode(ref.reshape(rows, 3), th.reshape(rows) if with_theta else None, boxes.reshape(rows, 7),
                                     mean.reshape(rows, 3), ss, deltas, r, dt, k, rcnn=rank == 2)
    shapes = [lead + (k,)] * 4 + [lead, lead, lead, lead + (3,)]
    for g, wv, shp in zip(got, want, shapes):
        assert tuple(g.shape) == shp
        g = host(g).reshape(wv.shape)
        if not with_theta:
            assert np.array_equal(g, wv)
        elif g.dtype == np.int32:
            assert (g != wv).mean() < 2e-3      # a residual within an ulp of a bin edge may fall either side
        else:
            close = np.isclose(g, wv, rtol=0, atol=2e-5)
            assert close.mean() > 0.998         # the same rows: their residual jumps by one bin
    if rank == 3:
        rep = lambda t, tail=(): t.unsqueeze(len(lead)).expand(*lead, k, *tail).contiguous()
        dec = box_codec.decode(dev(ref), dev(th) if with_theta else 0, got[0], got[1], got[2], got[3], rep(got[4]), rep(got[5]),
                               rep(got[6]), rep(got[7], (3,)), rep(dev(mean), (3,)), ss, deltas, r, dt)
        assert dec.shape == lead + (k, 7)
        for j in range(k):
            np.testing.assert_allclose(host(dec[..., j, :]), boxes, rtol=0, atol=3e-4)
        o_dec = oracle_mod.bin_box_decode(ref.reshape(rows, 3), th.reshape(rows) if with_theta else None,
                                          *[host(t).reshape(rows, k) for t in (got[0], got[1], got[2], got[3], rep(got[4]),
                                                                               rep(got[5]), rep(got[6]))],
                                          host(rep(got[7], (3,))).reshape(rows, k, 3), host(rep(dev(mean), (3,))).reshape(rows, k, 3),
                                          ss, deltas, r, dt)
        if with_theta:
            np.testing.assert_allclose(host(dec).reshape(rows, k, 7), o_dec, rtol=0, atol=1e-5)
        else:
            assert np.array_equal(host(dec).reshape(rows, k, 7), o_dec)


@pytest.mark.parametrize("with_theta,with_cls", [(False, False), (True, False), (False, True), (True, True)])
def test_bin_head_decode_against_oracle(hf, oracle_mod, with_theta, with_cls):
    """box_codec.decode_head (slice + argmax + residual gather + decode + class gather, one kernel) against the oracle:
    bit-exact without a reference heading, 1e-5 with one; equal to the op-by-op torch route"""
    from heterofusionrcnn_amd import box_codec
    rng = np.random.default_rng(11 + 2 * with_theta + with_cls)
    b, p, k, nbx, nbz, nbt = 2, 3000, 3, 12, 12, 12
    ss, deltas, r = [3.0, 1.5, 1.5], [0.5, 0.25, 0.25], 0.25 * np.pi
    dt = 2 * r / nbt
    d = 2 * nbx + 2 * nbz + 2 * nbt + 4
    head = rng.standard_normal((b, p, k * d)).astype(np.float32)
    head.reshape(b, p, k, d)[0, :40, :, 2] = head.reshape(b, p, k, d)[0, :40, :, 9] = 7.5     # ties
    ref = rng.uniform(-30, 30, (b, p, 3)).astype(np.float32)
    th = rng.uniform(-3, 3, (b, p)).astype(np.float32) if with_theta else None
    ms = np.array([[3.9, 1.6, 1.5], [0.8, 0.6, 1.7], [1.8, 0.6, 1.7]], np.float32)
    cls = rng.integers(0, k, (b, p)).astype(np.int32) if with_cls else None
    got = box_codec.decode_head(dev(head), dev(ref), dev(th) if with_theta else 0, ms, nbx, nbz, nbt, ss, deltas, r, dt,
                                cls=dev(cls) if with_cls else None)
    want = oracle_mod.bin_head_decode(head.reshape(b * p, k, d), ref.reshape(-1, 3), th.reshape(-1) if with_theta else None,
                                      ms, nbx, nbz, nbt, ss, deltas, r, dt, cls=cls.reshape(-1) if with_cls else None)
    assert tuple(got.shape) == ((b, p, 7) if with_cls else (b, p, k, 7))
    if with_theta:
        np.testing.assert_allclose(host(got).reshape(want.shape), want, rtol=0, atol=1e-5)
    else:
        assert np.array_equal(host(got).reshape(want.shape), want)
    # the op-by-op route on the device
    hv = dev(head).view(b, p, k, d)
    sl = torch.split(hv, [nbx, nbx, nbz, nbz, nbt, nbt, 1, 3], dim=-1)
    bx, bz, bt = sl[0].argmax(-1), sl[2].argmax(-1), sl[4].argmax(-1)
    take = lambda a, i: torch.gather(a, -1, i.unsqueeze(-1)).squeeze(-1)
    route = box_codec.decode(dev(ref), dev(th) if with_theta else 0, bx, take(sl[1], bx), bz, take(sl[3], bz), bt,
                             take(sl[5], bt), sl[6].squeeze(-1), sl[7].contiguous(),
                             dev(ms).expand(b, p, k, 3).contiguous(), ss, deltas, r, dt)
    if with_cls:
        route = torch.gather(route, 2, dev(cls).long().view(b, p, 1, 1).expand(b, p, 1, 7)).squeeze(2)
    assert torch.equal(got, route)


def test_mlp_entry_points_reject_bad_arguments(hf):
    """argument checks of the caller-side entry points: HF_EINVAL -> ValueError, never a launch"""
    import ctypes
    from heterofusionrcnn_amd import _lib
    from heterofusionrcnn_amd.mlp import BatchNormReLU, linear_bn_fwd, linear_wgrad
    L = _lib.lib()
    x = torch.randn(64, 8, device="cuda")
    bn = BatchNormReLU(300).cuda()
    with pytest.raises(ValueError):   # more output channels than the forward kernel holds accumulator tiles for
        linear_bn_fwd(x, torch.randn(300, 8, device="cuda"), torch.zeros(300, device="cuda"), bn)
    z = torch.empty(64, 4, device="cuda"); m = torch.empty(4, device="cuda")
    ws = torch.empty(L.hf_linear_bn_fwd_workspace(4) // 4, device="cuda")
    args = [64, 8, 4, _lib.ptr(x), None, None, None, None, _lib.ptr(torch.empty_like(x)), _lib.ptr(torch.randn(4, 8, device="cuda")),
            None, _lib.ptr(z), 1e-3, 0.1, None, None, _lib.ptr(m), _lib.ptr(m.clone()), _lib.ptr(ws), ws.numel() * 4, None]
    assert L.hf_linear_bn_fwd(*args) == _lib.HF_EINVAL          # x_act without an activation to apply
    args[8] = None; args[19] = 16
    assert L.hf_linear_bn_fwd(*args) == _lib.HF_EWORKSPACE      # workspace too small
    assert L.hf_linear_bn_bwd(64, 4, 300, _lib.ptr(z), *([None] * 8), _lib.ptr(x), _lib.ptr(x), *([None] * 7), None, 0,
                              None) == _lib.HF_EINVAL           # cin beyond the input-gradient kernel
    assert L.hf_bn_stats(0, 4, _lib.ptr(z), 1e-3, 0.1, None, None, _lib.ptr(m), _lib.ptr(m), None, 0, None) == _lib.HF_EINVAL
    from heterofusionrcnn_amd import modules
    from heterofusionrcnn_amd.mlp import shared_mlp
    with pytest.raises(RuntimeError):  # no CPU implementation anywhere on the path: host tensors raise
        linear_wgrad(torch.randn(8, 4), torch.randn(8, 3))
    with pytest.raises(RuntimeError):
        shared_mlp([modules.SharedMLPLayer(3, 4)], torch.randn(8, 3))
    with pytest.raises(RuntimeError):
        modules.SharedMLPLayer(3, 4)(torch.randn(8, 3))
    assert L.hf_linear_wgrad_workspace(0, 4, 4) == 0


@pytest.mark.parametrize("workload", ["stack", "rpn_multiclass", "rpn_multiclass_graph"])
def test_two_ddp_ranks_on_one_gpu(hf, workload):
    """scripts/ddp_two_ranks_one_gpu.py: two DistributedDataParallel ranks (gloo) sharing this GPU train the SA/FP
    stack with its fused nodes -- or the RPN of rpn_multiclass.config (PointCNN backbone, heads, losses: BASELINE
    config 4) -- for one step; gradients equal the mean of the per-shard gradients of a single process.
    rpn_multiclass_graph: the same through graph_step.TrainStep (captured step, flat gradient buffer, one all-reduce)"""
    import socket, subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "scripts", "ddp_two_ranks_one_gpu.py"), workload],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0 and "TWO_RANK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_batched_nms_equals_per_frame(hf, oracle_mod):
    rng = np.random.default_rng(8)
    frames = np.stack([_clustered(rng, 30, 10) for _ in range(3)])
    keep, num = hf.oriented_nms_batched(dev(frames), 0.7)
    for f in range(3):
        want, kept = oracle_mod.oriented_nms(frames[f], 0.7, return_count=True)
        assert np.array_equal(host(keep[f]), want) and int(host(num)[f]) == kept


def test_fuzz_all_ops_against_oracle(hf, oracle_mod):
    """seeded sweep over ragged shapes, radii and K for every op of the path; the ball query runs on both of its
    kernels and FPS on all of its kernels (plain 256/512/1024 threads, bucketed).  Everything integer is bit-exact."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import box_3d_to_8co, random_boxes3d
    rng = np.random.default_rng(2026)
    for trial in range(24):
        b = int(rng.integers(1, 4))
        n = int(rng.choice([1, 7, 63, 64, 65, 500, 1023, 1025, 2500, 4097, 9000]))
        m = int(rng.choice([1, 5, 64, 255, 257, 1000, 3000]))
        scale = float(rng.choice([1.0, 10.0, 80.0]))
        x1 = (rng.random((b, n, 3), dtype=np.float32) * scale - scale / 3).astype(np.float32)
        x2 = (rng.random((b, m, 3), dtype=np.float32) * scale - scale / 3).astype(np.float32)
        if n > 20:
            x1[:, -5:] = x1[:, :5]                         # duplicates
            x2[:, :min(m, 3)] = x1[:, :min(m, 3)]          # queries that coincide with data points
        r = float(rng.choice([0.02, 0.1, 0.5, 2.0])) * scale / 4
        ns = int(rng.choice([1, 3, 16, 32, 64, 100, 140]))
        oi, oc = oracle_mod.query_ball_point(r, ns, x1, x2)
        for mode in ("auto", "bruteforce") + (("sorted", "cell") if ns <= 128 else ()):
            idx, cnt, gx = hf.query_ball_group(r, ns, dev(x1), dev(x2), center=True, variant=mode)
            assert np.array_equal(host(idx), oi) and np.array_equal(host(cnt), oc), (trial, mode, b, n, m, r, ns)
            assert np.array_equal(host(gx), oracle_mod.group_point(x1, oi) - x2[:, :, None, :]), (trial, mode)
        # FPS on every kernel
        mm = int(min(n + 3, rng.choice([1, 2, 17, 128, 700])))
        want = oracle_mod.farthest_point_sample(mm, x1)
        variants = ([("plain", 1024), ("plain", 512), ("bucket", 0), ("bucket", 512), ("bucket", 1024)] + ([("plain", 256)] if n <= 4096 else []) +
                    ([("wave", 0)] if n <= 512 else []))
        for mode, nt in variants:
            got = hf.farthest_point_sample(mm, dev(x1), kernel=mode, threads=nt)
            assert np.array_equal(host(got), want), (trial, mode, nt, b, n, mm)
        # three_nn / kNN (x1 as known / data, x2 as unknown / queries)
        d3, i3 = hf.three_nn(dev(x2), dev(x1))
        od3, oi3 = oracle_mod.three_nn(x2, x1)
        assert np.array_equal(host(i3), oi3) and np.array_equal(host(d3), od3), (trial, "three_nn")
        k = int(min(n, rng.choice([1, 4, 8, 33])))
        kv, ki = hf.knn_point(k, dev(x1), dev(x2))
        okv, oki = oracle_mod.knn_point(k, x1, x2)
        assert np.array_equal(host(ki), oki) and np.array_equal(host(kv), okv), (trial, "knn", n, m, k)
        # group / interpolate with random channel counts
        c = int(rng.choice([1, 2, 4, 12, 33]))
        feats = rng.standard_normal((b, n, c)).astype(np.float32)
        assert np.array_equal(host(hf.group_point(dev(feats), dev(oi))), oracle_mod.group_point(feats, oi))
        w = rng.random((b, m, 3), dtype=np.float32)
        assert np.array_equal(host(hf.three_interpolate(dev(feats), dev(oi3), dev(w))),
                              oracle_mod.three_interpolate(feats, oi3, w))
    # crop + NMS + IoU on random box sets
    for trial in range(6):
        bsz, p = int(rng.integers(1, 4)), int(rng.choice([100, 1000, 4000]))
        pts = kitti_uniform(rng, bsz, p)
        pts[:, :, 1] = rng.uniform(0, 2, (bsz, p)).astype(np.float32)
        c = int(rng.choice([1, 4, 7]))
        fts = rng.standard_normal((bsz, p, c)).astype(np.float32)
        inten = rng.random((bsz, p, 1), dtype=np.float32)
        msk = rng.random((bsz, p)) < 0.3
        nb = int(rng.integers(1, 20))
        b3 = random_boxes3d(rng, nb)
        b3[:, 1] = 2.2
        b3[:, 3:6] += rng.uniform(0, 30, (nb, 1)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
        boxes = box_3d_to_8co(b3)
        bi = rng.integers(0, bsz, nb).astype(np.int32)
        rsz = int(rng.choice([1, 8, 100]))
        got = hf.pc_crop_and_sample(dev(pts), dev(fts), dev(inten), dev(msk), dev(boxes), dev(bi), rsz)
        want = oracle_mod.pc_crop_and_sample(pts, fts, inten, msk, boxes, bi, rsz)
        for g, wv in zip(got, want):
            assert np.array_equal(host(g), wv), ("crop", trial)
        bev = _clustered(rng, max(1, nb), 7)
        _, iou = hf.compute_bev_iou(dev(bev), dev(bev[:5]))
        np.testing.assert_allclose(host(iou), oracle_mod.compute_bev_iou(bev, bev[:5])[1], rtol=0, atol=TOL)


@pytest.mark.parametrize("rows,cin,cout", [(140000, 32, 64), (5000, 7, 32), (300, 131, 196)])
def test_fused_linear_bn_relu_node(hf, rows, cin, cout):
    """modules.SharedMLPLayer (one fused Linear+BN+ReLU autograd node, split-K dW, bias gradient from the BN
    backward pass) against nn.Linear + nn.BatchNorm1d + relu"""
    from heterofusionrcnn_amd import modules
    torch.manual_seed(rows)
    layer = modules.SharedMLPLayer(cin, cout).cuda()
    ref_fc = torch.nn.Linear(cin, cout).cuda()
    ref_bn = torch.nn.BatchNorm1d(cout, eps=1e-3, momentum=0.1).cuda()
    with torch.no_grad():
        ref_fc.weight.copy_(layer.fc.weight); ref_fc.bias.copy_(layer.fc.bias)
        layer.bn.weight.uniform_(0.5, 1.5); layer.bn.bias.uniform_(-.5, .5)
        ref_bn.weight.copy_(layer.bn.weight); ref_bn.bias.copy_(layer.bn.bias)
    x1 = torch.randn(rows, cin, device="cuda", requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    y1 = torch.relu(ref_bn(ref_fc(x1)))
    y2 = layer(x2)
    torch.testing.assert_close(y2, y1, rtol=1e-4, atol=2e-5)
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    torch.testing.assert_close(x2.grad, x1.grad, rtol=1e-3, atol=2e-5)
    scale = float(ref_fc.weight.grad.abs().max())
    torch.testing.assert_close(layer.fc.weight.grad, ref_fc.weight.grad, rtol=1e-3, atol=2e-4 * max(scale, 1.0))
    torch.testing.assert_close(layer.bn.weight.grad, ref_bn.weight.grad, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(layer.bn.bias.grad, ref_bn.bias.grad, rtol=1e-3, atol=1e-2)
    # the bias of a Linear feeding a batch norm has an analytically zero gradient: both are rounding noise
    assert layer.fc.bias.grad.abs().max() < 1e-2 and ref_fc.bias.grad.abs().max() < 1e-2


@pytest.mark.parametrize("rows,pool_k,widths", [(32768, 32, (4, 32, 32, 64)), (40000, 0, (129, 128, 128)),
                                                 (65536, 16, (67, 64, 96, 128)), (33000, 0, (5, 196, 256)),
                                                 (4096, 32, (7, 16, 8))])
def test_shared_mlp_chain_node(hf, rows, pool_k, widths):
    """mlp.shared_mlp: a whole Linear+BN+ReLU stack (+ max over K rows) as one autograd node on the MFMA forward
    kernel (statistics from the accumulators, BN+ReLU applied on load by the next layer), against
    nn.Linear / nn.BatchNorm1d / relu / max.  The last case is below the row threshold: layer-by-layer path."""
    from heterofusionrcnn_amd import modules
    from heterofusionrcnn_amd.mlp import shared_mlp
    torch.manual_seed(rows + pool_k)
    layers = torch.nn.ModuleList([modules.SharedMLPLayer(a, b) for a, b in zip(widths[:-1], widths[1:])]).cuda()
    ref = []
    for l in layers:
        fc = torch.nn.Linear(l.fc.in_features, l.fc.out_features).cuda()
        bn = torch.nn.BatchNorm1d(l.fc.out_features, eps=1e-3, momentum=0.1).cuda()
        with torch.no_grad():
            l.fc.bias.uniform_(-.2, .2); l.bn.weight.uniform_(0.5, 1.5); l.bn.bias.uniform_(-.5, .5)
            fc.weight.copy_(l.fc.weight); fc.bias.copy_(l.fc.bias); bn.weight.copy_(l.bn.weight); bn.bias.copy_(l.bn.bias)
        ref.append((fc, bn))
    x1 = torch.randn(rows, widths[0], device="cuda", requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    y1 = x1
    for fc, bn in ref:
        y1 = torch.relu(bn(fc(y1)))
    if pool_k:
        y1 = y1.view(rows // pool_k, pool_k, -1).max(dim=1).values
    y2 = shared_mlp(layers, x2, pool_k)
    torch.testing.assert_close(y2, y1, rtol=2e-4, atol=5e-5)
    for l, (fc, bn) in zip(layers, ref):
        torch.testing.assert_close(l.bn.running_mean, bn.running_mean, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(l.bn.running_var, _tf_running_var(bn, rows), rtol=1e-4, atol=1e-5)
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    # a pre-activation within rounding of zero may fall on the other side of the ReLU in the two implementations;
    # that flips one unit for one row (expected a handful of times in ~1e7 activations): tolerate 1 row in 10^4
    tol = 1e-3 * float(x1.grad.abs().max()) + 2e-5
    bad_rows = ((x2.grad - x1.grad).abs() > tol + 2e-3 * x1.grad.abs()).any(dim=1).float().mean().item()
    assert bad_rows <= 1e-4, bad_rows
    for l, (fc, bn) in zip(layers, ref):
        # sums over all rows: a flipped unit moves them by one row's contribution, bounded relative to the largest entry
        # by about one row's contribution, in this layer and every layer below it: everything within 2% of the
        # largest entry, and the mean error two orders of magnitude below that
        for got, want in ((l.fc.weight.grad, fc.weight.grad), (l.bn.weight.grad, bn.weight.grad), (l.bn.bias.grad, bn.bias.grad)):
            big = max(float(want.abs().max()), 1.0)
            torch.testing.assert_close(got, want, rtol=2e-3, atol=2e-2 * big)
            assert (got - want).abs().mean().item() <= 5e-4 * big
    # inference: running statistics, MFMA forward with the normalisation applied on load where it pays
    layers.eval()
    for fc, bn in ref:
        bn.eval()
    with torch.no_grad():
        e1 = x1.detach()
        for fc, bn in ref:
            e1 = torch.relu(bn(fc(e1)))
        if pool_k:
            e1 = e1.view(rows // pool_k, pool_k, -1).max(dim=1).values
        before = [l.bn.running_mean.clone() for l in layers]
        e2 = shared_mlp(layers, x2.detach(), pool_k)
    torch.testing.assert_close(e2, e1, rtol=2e-4, atol=5e-5)
    assert all(torch.equal(l.bn.running_mean, b) for l, b in zip(layers, before))  # inference leaves them alone


@pytest.mark.parametrize("groups,k,cin,cout", [(4096, 32, 32, 64), (333, 17, 7, 5), (1024, 64, 128, 256), (50, 255, 16, 24)])
def test_fused_linear_bn_relu_maxpool_node(hf, groups, k, cin, cout):
    """mlp.linear_bn_relu_maxpool (Linear + BN + ReLU + max over the K grouped rows as one node; the dense
    activation is never written) against nn.Linear + nn.BatchNorm1d + relu + max(dim=1); train and eval"""
    from heterofusionrcnn_amd import modules
    from heterofusionrcnn_amd.mlp import linear_bn_relu_maxpool
    torch.manual_seed(groups + k)
    layer = modules.SharedMLPLayer(cin, cout).cuda()
    ref_fc = torch.nn.Linear(cin, cout).cuda()
    ref_bn = torch.nn.BatchNorm1d(cout, eps=1e-3, momentum=0.1).cuda()
    with torch.no_grad():
        ref_fc.weight.copy_(layer.fc.weight); ref_fc.bias.copy_(layer.fc.bias)
        layer.bn.weight.uniform_(0.5, 1.5); layer.bn.bias.uniform_(-.5, .5)
        ref_bn.weight.copy_(layer.bn.weight); ref_bn.bias.copy_(layer.bn.bias)
    x1 = torch.randn(groups * k, cin, device="cuda")
    # ball-query style padding: the tail rows of some groups repeat the group's first row (exact ties)
    xv = x1.view(groups, k, cin)
    xv[::3, k // 2:] = xv[::3, :1]
    x1.requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    y1 = torch.relu(ref_bn(ref_fc(x1))).view(groups, k, cout).max(dim=1).values
    y2 = linear_bn_relu_maxpool(x2, layer.fc.weight, layer.fc.bias, layer.bn, k)
    torch.testing.assert_close(y2, y1, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(layer.bn.running_mean, ref_bn.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(layer.bn.running_var, _tf_running_var(ref_bn, groups * k), rtol=1e-4, atol=1e-5)
    g = torch.randn_like(y1)
    y1.backward(g); y2.backward(g)
    # which of several identical rows receives the gradient is a free choice: compare per-group sums over ties
    # through quantities that do not depend on it (dW, dgamma, dbeta) and dx summed over identical rows
    scale = float(ref_fc.weight.grad.abs().max())
    torch.testing.assert_close(layer.fc.weight.grad, ref_fc.weight.grad, rtol=1e-3, atol=2e-4 * max(scale, 1.0))
    torch.testing.assert_close(layer.bn.weight.grad, ref_bn.weight.grad, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(layer.bn.bias.grad, ref_bn.bias.grad, rtol=1e-3, atol=1e-2)
    d1 = x1.grad.view(groups, k, cin); d2 = x2.grad.view(groups, k, cin)
    torch.testing.assert_close(d2[1::3], d1[1::3], rtol=1e-3, atol=2e-5)       # groups without ties: row by row
    torch.testing.assert_close(d2[2::3], d1[2::3], rtol=1e-3, atol=2e-5)
    tied = lambda d: torch.cat([d[::3, 1:k // 2], d[::3, :1] + d[::3, k // 2:].sum(1, keepdim=True)], dim=1)
    torch.testing.assert_close(tied(d2), tied(d1), rtol=1e-3, atol=5e-5)
    assert layer.fc.bias.grad.abs().max() < 1e-2
    # eval: running statistics
    layer.eval(); ref_bn.eval()
    with torch.no_grad():
        e1 = torch.relu(ref_bn(ref_fc(x1))).view(groups, k, cout).max(dim=1).values
        e2 = linear_bn_relu_maxpool(x2.detach(), layer.fc.weight, layer.fc.bias, layer.bn, k)
    torch.testing.assert_close(e2, e1, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("rows,c", [(1000, 64), (4099, 76), (300, 7), (64, 1024), (2048, 8)])
def test_bn_with_elu_on_load(hf, rows, c):
    """BatchNormReLU(elu_in=True) = batch_norm(elu(x)) of pointfly.dense (pointfly.py:371-497), forward, running statistics
    and all gradients, against torch in fp64"""
    from heterofusionrcnn_amd.mlp import BatchNormReLU
    g = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, c, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(rows, c, generator=g).cuda()
    bn = BatchNormReLU(c, eps=1e-3, momentum=0.01, relu=False, elu_in=True).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5); bn.bias.copy_(torch.randn(c, generator=g))
    y = bn(x); y.backward(dy)
    xr = x.detach().double().requires_grad_(True)
    w, b = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    e = torch.nn.functional.elu(xr)
    mean, var = e.mean(0), e.var(0, unbiased=False)
    ref = (e - mean) / torch.sqrt(var + 1e-3) * w + b
    ref.backward(dy.double())
    assert torch.allclose(y.double(), ref, rtol=1e-4, atol=1e-4)
    assert torch.allclose(x.grad.double(), xr.grad, rtol=1e-3, atol=1e-4)
    assert torch.allclose(bn.weight.grad.double(), w.grad, rtol=1e-3, atol=1e-3)
    assert torch.allclose(bn.bias.grad.double(), b.grad, rtol=1e-3, atol=1e-3)
    assert torch.allclose(bn.running_mean.double(), 0.01 * mean.detach(), rtol=1e-3, atol=1e-5)
    bn.eval()
    ye = bn(x.detach())
    refe = (e.detach() - bn.running_mean.double()) / torch.sqrt(bn.running_var.double() + 1e-3) * w.detach() + b.detach()
    assert torch.allclose(ye.double(), refe, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,c,mode,rate", [(20000, 64, 2, 0.5), (5000, 256, 2, 0.3), (9000, 76, 1, 0.5), (3001, 7, 2, 0.25)])
def test_bn_with_fused_dropout(hf, rows, c, mode, rate):
    """BatchNormReLU(x, dropout=rate) = tf.layers.dropout(rate)(bn(act(x))) of pointfly's dense -> dropout (pointcnn.py:371-384,
    rpn_model.py:556-568) as one node.  The mask of a call depends on (layer seed, call number, element index) only, so it is read
    back by repeating the call from the same state with gamma = 0, beta = 1 (output = mask / (1 - rate)); then: output and all
    gradients against torch in fp64 with THAT mask, keep fraction, a second call draws another mask; inference: the plain eval form."""
    from heterofusionrcnn_amd.mlp import BatchNormReLU
    g = torch.Generator().manual_seed(rows + c)
    x = torch.randn(rows, c, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(rows, c, generator=g).cuda()
    bn = BatchNormReLU(c, eps=1e-3, momentum=0.01, relu=bool(mode & 1), elu_in=bool(mode & 2)).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5); bn.bias.copy_(torch.randn(c, generator=g))
    state0 = bn.drop_state.clone()
    y = bn(x, dropout=rate)
    y.backward(dy)
    running_mean = bn.running_mean.clone()
    assert int(bn.drop_state[1]) == int(state0[1]) + 1 and int(bn.drop_state[0]) == int(state0[0])
    with torch.no_grad():
        w0, b0 = bn.weight.clone(), bn.bias.clone()
        bn.weight.zero_(); bn.bias.fill_(1.0)
        bn.drop_state.copy_(state0)
        mask = bn(x.detach(), dropout=rate).double()          # 0 or 1 / (1 - rate)
        bn.weight.copy_(w0); bn.bias.copy_(b0)
    keep = mask > 0
    assert torch.allclose(mask[keep], torch.full_like(mask[keep], 1.0 / (1.0 - rate)), rtol=1e-6)
    frac, n = keep.double().mean().item(), keep.numel()
    assert abs(frac - (1.0 - rate)) < 5.0 * (rate * (1.0 - rate) / n) ** 0.5 + 2.0 / 65536, frac
    # rows and columns are both mixed: no column or row is all kept / all dropped
    assert 0 < keep.double().mean(0).min() and keep.double().mean(0).max() < 1
    xr = x.detach().double().requires_grad_(True)
    w, b = w0.double().requires_grad_(True), b0.double().requires_grad_(True)
    e = torch.nn.functional.elu(xr) if mode & 2 else xr
    mean, var = e.mean(0), e.var(0, unbiased=False)
    full = (e - mean) / torch.sqrt(var + 1e-3) * w + b
    if mode & 1:
        full = torch.relu(full)
    ref = full * mask
    ref.backward(dy.double())
    assert torch.allclose(y.double(), ref.detach(), rtol=1e-4, atol=2e-4)
    assert (y.detach()[~keep] == 0).all()
    assert torch.allclose(x.grad.double(), xr.grad, rtol=1e-3, atol=1e-3 * xr.grad.abs().max().item())
    assert torch.allclose(bn.weight.grad.double(), w.grad, rtol=1e-3, atol=1e-3 * (1 + w.grad.abs().max().item()))
    assert torch.allclose(bn.bias.grad.double(), b.grad, rtol=1e-3, atol=1e-3 * (1 + b.grad.abs().max().item()))
    # the statistics are those of the undropped activation
    assert torch.allclose(running_mean.double(), 0.01 * mean.detach(), rtol=1e-3, atol=1e-5)
    # another call, another mask (the call counter lives on the device)
    y2 = bn(x.detach(), dropout=rate)
    differ = ((y2 != 0) != keep).double().mean().item()
    assert abs(differ - 2 * rate * (1 - rate)) < 0.05, differ
    bn.eval()
    ye = bn(x.detach())          # callers pass no rate at inference
    bn_ref = (e.detach() - bn.running_mean.double()) / torch.sqrt(bn.running_var.double() + 1e-3) * w.detach() + b.detach()
    assert torch.allclose(ye.double(), torch.relu(bn_ref) if mode & 1 else bn_ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,cin,cout", [(20000, 256, 4), (4099, 76, 2), (3000, 7, 3), (131072, 256, 4)])
def test_narrow_linear_input_gradient(hf, rows, cin, cout):
    """hf_narrow_linear_dx (the segmentation head's input gradient, one pass) against g @ W; through mlp.linear_narrow's autograd node"""
    from heterofusionrcnn_amd.mlp import linear_narrow
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g).cuda().requires_grad_(True)
    w = torch.randn(cout, cin, generator=g).cuda().requires_grad_(True)
    b = torch.randn(cout, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(rows, cout, generator=g).cuda()
    y = linear_narrow(x, w, b)
    y.backward(dy)
    xr, wr, br = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    torch.nn.functional.linear(xr, wr, br).backward(dy.double())
    assert torch.allclose(y.double(), torch.nn.functional.linear(xr, wr, br), rtol=1e-4, atol=1e-4)
    assert torch.allclose(x.grad.double(), xr.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(w.grad.double(), wr.grad, rtol=1e-3, atol=1e-2)
    assert torch.allclose(b.grad.double(), br.grad, rtol=1e-4, atol=1e-2)


def test_knn_point_against_three_term_formula(hf, oracle_mod):
    """knn_point (exact (q-p)^2, ties to the lower index) against the reference's expression restated as written
    (tf_grouping.py:80-92: |q|^2 - 2 q.p^T + |p|^2, top_k): the neighbour SETS must agree wherever the gap between the
    k-th and the (k+1)-th distance exceeds the rounding of the expansion; near-ties are parity unpinned and only counted"""
    rng = np.random.default_rng(21)
    k = 8
    x1 = kitti_uniform(rng, 2, 4096)
    x2 = x1[:, rng.permutation(4096)[:1024]]
    val, idx = hf.knn_point(k, dev(x1), dev(x2))
    idx = host(idx)
    tv, ti = oracle_mod.knn_point_three_term(k, x1, x2)
    d = ((x2[:, :, None, :].astype(np.float64) - x1[:, None, :, :].astype(np.float64)) ** 2).sum(-1)
    ds = np.sort(d, axis=2)
    # rounding of the expansion: a few ulp of |q|^2 + |p|^2 (coordinates up to 70 m -> magnitudes ~ 1e4, ulp ~ 1e-3)
    scale = (x2.astype(np.float64) ** 2).sum(-1)[:, :, None] * 2 + 1.0
    clear = (ds[:, :, k] - ds[:, :, k - 1]) > 4e-6 * scale[:, :, 0]
    same = np.array([[set(idx[b, j]) == set(ti[b, j]) for j in range(idx.shape[1])] for b in range(idx.shape[0])])
    assert clear.mean() > 0.5, "the test must check most queries"
    assert same[clear].all(), "neighbour sets differ where the k-th gap exceeds the rounding of the three-term formula"
    # the kernel's own distances are the exact ones
    np.testing.assert_allclose(host(val), np.take_along_axis(d, idx.astype(np.int64), axis=2), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("c,ch", [(256, 64), (1, 64), (7, 5), (320, 128)])
def test_concat_group_equals_cat_of_group_point(hf, c, ch):
    """grouping.concat_group (group_point into a column slice of the concat buffer, gradient read in place) against
    torch.cat([head, group_point(points, idx)], -1) and its autograd"""
    from heterofusionrcnn_amd import grouping
    g = torch.Generator().manual_seed(c * 31 + ch)
    b, n, m, k = 2, 500, 130, 8
    pts = torch.randn(b, n, c, generator=g).cuda().requires_grad_(True)
    head = torch.randn(b, m, k, ch, generator=g).cuda().requires_grad_(True)
    idx = torch.randint(0, n, (b, m, k), generator=g, dtype=torch.int32).cuda()
    go = torch.randn(b, m, k, ch + c, generator=g).cuda()
    out = grouping.concat_group(head, pts, idx)
    out.backward(go)
    p2, h2 = pts.detach().clone().requires_grad_(True), head.detach().clone().requires_grad_(True)
    ref = torch.cat([h2, hf.group_point(p2, idx)], dim=-1)
    ref.backward(go)
    assert torch.equal(out, ref)
    assert torch.equal(head.grad, h2.grad)
    assert torch.allclose(pts.grad, p2.grad, rtol=1e-5, atol=1e-5)


def test_group_point_gradient_in_gather_form(hf, oracle_mod):
    """hf_index_inverse + hf_group_point_grad_gather: the gradient of the feature gather of an X-Conv written once per data point
    over the CSR inverse of the neighbour table -- bit for bit the oracle's sequential accumulation (ascending (query, slot)
    order), and equal to rounding to the atomic scatter it replaces; hubs (one point named by every query), unreferenced
    points, a non-multiple-of-4 width"""
    from heterofusionrcnn_amd.grouping import concat_group, index_inverse
    rng = np.random.default_rng(12)
    for (b, n, m, k, c, ch) in ((2, 300, 500, 8, 32, 16), (1, 4096, 16384, 8, 64, 64), (3, 70, 40, 4, 5, 3), (1, 16384, 4096, 8, 256, 64)):
        idx = rng.integers(0, n, (b, m, k)).astype(np.int32)
        idx[:, :, 0] = 7 % n                                    # a hub
        idx[idx == (n - 1)] = 0                                 # point n-1 is never referenced
        pts = dev(rng.standard_normal((b, n, c)).astype(np.float32)).requires_grad_(True)
        head = dev(rng.standard_normal((b, m, k, ch)).astype(np.float32)).requires_grad_(True)
        go = rng.standard_normal((b, m, k, ch + c)).astype(np.float32)
        i_d = dev(idx)
        off, ent = index_inverse(i_d, n)
        # the inverse is a permutation of the flat positions, grouped by target, ascending inside a group
        o, e = host(off), host(ent)
        for bb in range(b):
            assert o[bb, 0] == 0 and o[bb, -1] == m * k and np.array_equal(np.sort(e[bb]), np.arange(m * k))
            flat = idx[bb].reshape(-1)
            assert np.array_equal(flat[e[bb]], np.repeat(np.arange(n), np.diff(o[bb])))
            starts = o[bb, :-1]
            inner = np.ones(m * k, bool)
            inner[starts[starts < m * k]] = False
            assert (np.diff(e[bb])[inner[1:]] > 0).all()
        out = concat_group(head, pts, i_d, (off, ent))
        g_pts, g_head = torch.autograd.grad(out, (pts, head), dev(go))
        want = oracle_mod.group_point_grad((b, n, c), idx, np.ascontiguousarray(go[..., ch:]))
        assert np.array_equal(host(g_pts), want), (b, n, m, k, c)
        assert np.array_equal(host(g_head), go[..., :ch])
        out2 = concat_group(head, pts, i_d)
        g2 = torch.autograd.grad(out2, pts, dev(go))[0]
        np.testing.assert_allclose(host(g2), want, rtol=0, atol=2e-4 * max(1.0, np.abs(want).max()))
        assert not host(g_pts)[:, n - 1].any()


@pytest.mark.gpu
@pytest.mark.parametrize("c0,c1,m", [(64, 256, 1), (64, 1, 4), (128, 96, 2)])
def test_xconv_depthwise_gather_equals_materialised_concat(c0, c1, m):
    """hf_xconv_depthwise_gather(+grad) (F_* read in place through the neighbour table) against the route it replaces: concat_group
    (hf_group_point_into) followed by hf_xconv_depthwise on the materialised (B,P,K,C0+C1) tensor.  Same arithmetic in the same
    order -> bit-identical output and gradients (the depthwise weights' gradient ends in atomics: 1e-5)."""
    from heterofusionrcnn_amd import pointcnn as pc
    from heterofusionrcnn_amd.grouping import concat_group, index_inverse
    torch.manual_seed(3)
    b, n, p, k = 3, 500, 200, 8
    dev = "cuda"
    x = torch.randn(b, p, k, k, device=dev, requires_grad=True)
    fd = torch.randn(b, p, k, c0, device=dev, requires_grad=True)
    fts = torch.randn(b, n, c1, device=dev, requires_grad=True)
    wd = torch.randn(k, c0 + c1, m, device=dev, requires_grad=True)
    idx = torch.randint(0, n, (b, p, k), device=dev, dtype=torch.int32)
    idx[0, :, :] = 7                                      # one table row named by every slot of a cloud, most rows by none
    inv = index_inverse(idx, n)
    go = torch.randn(b, p, (c0 + c1) * m, device=dev)
    ref = pc.xconv_depthwise(x, concat_group(fd, fts, idx, inv), wd)
    g_ref = torch.autograd.grad(ref, (x, fd, fts, wd), go)
    for use_ws in (True, False):      # the table's gradient staged in a workspace / rebuilt per table row: the same values
        out = pc.xconv_depthwise_gather(x, fd, fts, idx, wd, inv, use_workspace=use_ws)
        g = torch.autograd.grad(out, (x, fd, fts, wd), go)
        assert torch.equal(out, ref)
        for a, r, name in zip(g[:3], g_ref[:3], ("x", "f_delta", "fts")):
            assert torch.equal(a, r), (name, use_ws)
        assert float((g[3] - g_ref[3]).abs().max()) <= 1e-5 * float(g_ref[3].abs().max())
    # no gradient asked of the table: no inverse needed
    out2 = pc.xconv_depthwise_gather(x, fd, fts.detach(), idx, wd)
    assert torch.equal(out2, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,c", [(3, 64), (3, 128), (24, 32)])
def test_dense_chain_elu_node_equals_layer_by_layer(cin, c):
    """pointcnn.dense_chain (two pf.dense layers as one node: statistics from the GEMM accumulators, first normalisation on
    operand load, first layer's BatchNorm-backward sums from the second layer's input-gradient GEMM) against the same two
    Dense modules run layer by layer (library / MFMA GEMM, hf_bn_* passes), and both against torch ops in fp64."""
    from heterofusionrcnn_amd import pointcnn as pc
    torch.manual_seed(5)
    rows = 40000
    d0, d1 = pc.Dense(cin, c).cuda().train(), pc.Dense(c, c).cuda().train()
    with torch.no_grad():
        for d in (d0, d1):
            d.post.bn.weight.uniform_(0.5, 1.5)
            d.post.bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(8, rows // 8, cin, device="cuda")
    go = torch.randn(8, rows // 8, c, device="cuda") / rows
    params = [d0.linear.weight, d0.post.bn.weight, d0.post.bn.bias, d1.linear.weight, d1.post.bn.weight, d1.post.bn.bias]
    rm = [d.post.bn.running_mean.clone() for d in (d0, d1)]
    out_a = pc.dense_chain(d0, d1, x)          # cin = 3: the recompute form (_LiftChain), else the stored form (_DenseChainElu)
    node = out_a.grad_fn.next_functions[0][0].__class__.__name__          # below the reshape
    assert node.startswith("_LiftChain" if cin == 3 else "_DenseChainElu"), node
    g_a = torch.autograd.grad(out_a, params, go)
    rm_a = [d.post.bn.running_mean.clone() for d in (d0, d1)]
    for d, r in zip((d0, d1), rm):
        d.post.bn.running_mean.copy_(r)
    out_b = d1(d0(x))
    g_b = torch.autograd.grad(out_b, params, go)
    rm_b = [d.post.bn.running_mean.clone() for d in (d0, d1)]

    def ref64():
        h = x.double().reshape(rows, cin)
        for d in (d0, d1):
            e = torch.nn.functional.elu(h @ d.linear.weight.double().t())
            mu, var = e.mean(0), e.var(0, unbiased=False)
            h = (e - mu) / torch.sqrt(var + d.post.bn.eps) * d.post.bn.weight.double() + d.post.bn.bias.double()
        return h.reshape(8, rows // 8, c)
    out_r = ref64()
    g_r = torch.autograd.grad(out_r, params, go.double())
    assert float((out_a.detach() - out_r.detach()).abs().max()) <= 2e-4 and float((out_b.detach() - out_r.detach()).abs().max()) <= 2e-4
    for a, b, r, name in zip(g_a, g_b, g_r, ("w0", "gamma0", "beta0", "w1", "gamma1", "beta1")):
        scale = float(r.abs().max())
        assert float((a - r.float()).abs().max()) <= 2e-3 * scale + 1e-9, (name, float((a - r.float()).abs().max()), scale)
        assert float((b - r.float()).abs().max()) <= 2e-3 * scale + 1e-9, name
    for a, b in zip(rm_a, rm_b):
        assert float((a - b).abs().max()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("k,m,c0,c1", [(4, 4, 128, 33), (12, 2, 128, 64), (12, 1, 256, 40), (4, 1, 64, 64)])
def test_xconv_depthwise_kernels_at_the_rcnn_neighbourhood_sizes(k, m, c0, c1):
    """the fused X-apply + depthwise kernels (dense and in-place-gather form) at K = 4 and K = 12 (rcnn_multiclass.config:157-186)
    against torch.matmul followed by the einsum form of the depthwise step, forward and all four gradients"""
    from heterofusionrcnn_amd import pointcnn as pc
    from heterofusionrcnn_amd.grouping import index_inverse
    torch.manual_seed(k * 10 + m)
    b, n, p = 2, 200, 90
    dev = "cuda"
    x = torch.randn(b, p, k, k, device=dev, requires_grad=True)
    fd = torch.randn(b, p, k, c0, device=dev, requires_grad=True)
    fts = torch.randn(b, n, c1, device=dev, requires_grad=True)
    wd = torch.randn(k, c0 + c1, m, device=dev, requires_grad=True)
    idx = torch.randint(0, n, (b, p, k), device=dev, dtype=torch.int32)
    inv = index_inverse(idx, n)
    go = torch.randn(b, p, (c0 + c1) * m, device=dev)
    gathered = torch.gather(fts, 1, idx.long().reshape(b, -1, 1).expand(-1, -1, c1)).reshape(b, p, k, c1)
    fstar = torch.cat([fd, gathered], -1)
    ref = torch.einsum("bpkc,kcm->bpcm", torch.matmul(x, fstar), wd).reshape(b, p, -1)
    g_ref = torch.autograd.grad(ref, (x, fd, fts, wd), go, retain_graph=True)          # fstar's graph is used again below
    for name, out in (("gather", pc.xconv_depthwise_gather(x, fd, fts, idx, wd, inv)), ("dense", pc.xconv_depthwise(x, fstar, wd))):
        assert "XConvDepthwise" in out.grad_fn.next_functions[0][0].__class__.__name__ + out.grad_fn.__class__.__name__     # the HIP node ran
        g = torch.autograd.grad(out, (x, fd, fts, wd), go, retain_graph=True)
        assert float((out.detach() - ref.detach()).abs().max()) <= 1e-4 * float(ref.detach().abs().max()), name
        for a, r, what in zip(g, g_ref, ("x", "f_delta", "fts", "wd")):
            assert float((a - r).abs().max()) <= 1e-4 * float(r.abs().max()), (name, what)


@pytest.mark.gpu
@pytest.mark.parametrize("c", [64, 128, 256])
def test_dense_chain_inference_form_equals_layer_by_layer(c):
    """eval mode: pointcnn.dense_chain on a 3-channel input rebuilds the first layer inside the second GEMM with the RUNNING
    statistics (hf_lift_elu_fwd_eval); against the two Dense modules run one after the other in eval mode"""
    from heterofusionrcnn_amd import pointcnn as pc
    torch.manual_seed(c)
    rows = 40000
    d0, d1 = pc.Dense(3, c).cuda(), pc.Dense(c, c).cuda()
    with torch.no_grad():
        for d in (d0, d1):
            d.post.bn.weight.uniform_(0.5, 1.5)
            d.post.bn.bias.uniform_(-0.5, 0.5)
            d.post.bn.running_mean.uniform_(-0.3, 0.3)
            d.post.bn.running_var.uniform_(0.5, 2.0)
    d0.eval(); d1.eval()
    x = torch.randn(4, rows // 4, 3, device="cuda")
    with torch.no_grad():
        a = pc.dense_chain(d0, d1, x)
        b = d1(d0(x))
    assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.parametrize("rows,c,mode", [(64, 64, 1), (4096, 64, 0), (513, 228, 1), (4096, 1024, 2), (3000, 12, 2), (1, 4, 1),
                                         (4097, 64, 1)])
def test_bn_short_tensors_strided_and_column_sums(hf, rows, c, mode):
    """hf_bn_relu_fwd_train_ld / hf_bn_relu_bwd_ld through the C-ABI on tensors of at most 4096 rows (the single-launch kernels
    of csrc/mlp.hip; 4097 rows = the three-launch form beside them): output rows inside a wider buffer, gradient rows inside a
    wider buffer, the column sums of dx requested; against fp64 torch.  mode: 0 = BN, 1 = BN + ReLU, 2 = BN of elu(x)"""
    from heterofusionrcnn_amd import _lib
    from heterofusionrcnn_amd._lib import ptr, stream_ptr, check
    L = _lib.lib()
    g = torch.Generator().manual_seed(rows * 7 + c)
    x = (torch.randn(rows, c, generator=g) * 1.5 + 0.3).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    ld = c + 8
    ybuf = torch.full((rows, ld), 7.0, device="cuda")
    dybuf = torch.randn(rows, ld, generator=g).cuda()
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    mean, invstd = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    nbytes = L.hf_bn_workspace(rows, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    check(L.hf_bn_relu_fwd_train_ld(rows, c, ptr(x), ptr(gamma), ptr(beta), 1e-3, 0.1, ptr(rm), ptr(rv), mode, ptr(ybuf), ld,
                                    ptr(mean), ptr(invstd), ptr(ws), nbytes, stream_ptr()), "fwd")
    dx, dg, db, cs = torch.empty_like(x), torch.empty(c, device="cuda"), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    check(L.hf_bn_relu_bwd_ld(rows, c, ptr(x), ptr(dybuf), ld, ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), mode, ptr(dx), ptr(dg),
                              ptr(db), ptr(cs), ptr(ws), nbytes, stream_ptr()), "bwd")
    xr = x.double().requires_grad_(True)
    w, b = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    e = torch.nn.functional.elu(xr) if mode == 2 else xr
    mu, var = e.mean(0), e.var(0, unbiased=False)
    ref = (e - mu) / torch.sqrt(var + 1e-3) * w + b
    if mode == 1:
        ref = torch.relu(ref)
    ref.backward(dybuf[:, :c].double())
    assert (ybuf[:, c:] == 7.0).all(), "columns beyond c of the strided output were written"
    y = ybuf[:, :c].double()
    if mode == 1:       # an element within rounding of the ReLU threshold may land on either side
        near = ref.detach().abs() < 1e-5
        assert torch.allclose(y[~near], ref.detach()[~near], rtol=1e-4, atol=1e-4)
    else:
        assert torch.allclose(y, ref.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(mean.double(), mu.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(invstd.double(), 1.0 / torch.sqrt(var.detach() + 1e-3), rtol=1e-4, atol=1e-5)
    assert torch.allclose(rm.double(), 0.1 * mu.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(rv.double(), 0.9 + 0.1 * var.detach(), rtol=1e-4, atol=1e-5)
    tol = 1e-3 * max(1.0, rows ** 0.5 * 0.1)
    assert torch.allclose(dg.double(), w.grad, rtol=1e-3, atol=tol) and torch.allclose(db.double(), b.grad, rtol=1e-3, atol=tol)
    if rows > 1:
        assert torch.allclose(dx.double(), xr.grad, rtol=1e-3, atol=1e-4)
    assert torch.allclose(cs.double(), xr.grad.sum(0), rtol=1e-3, atol=tol)
