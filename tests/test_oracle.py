"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors, the reference's
own CPU programs (oracle/_ref, present only where /root/reference was available at build time)
and the known answers recoverable from the reference demos."""
import numpy as np
import pytest

from conftest import load_golden


def test_query_ball_point_golden(oracle_mod):
    g = load_golden("grouping_P")
    for tag, r, ns in (("r010", 0.1, 32), ("r030", 0.3, 32), ("r005k8", 0.05, 8)):
        idx, cnt = oracle_mod.query_ball_point(r, ns, g["xyz"], g["new_xyz"])
        assert np.array_equal(idx, g["idx_" + tag]) and np.array_equal(cnt, g["cnt_" + tag])
    idx, cnt = oracle_mod.query_ball_point(0.02, 16, g["xyz"], g["q_rand"])
    assert np.array_equal(idx, g["idx_rand"]) and np.array_equal(cnt, g["cnt_rand"])
    assert (idx[cnt == 0] == 0).all()  # zero-hit rows are zeros (defined here; undefined in the reference)


def test_query_ball_point_semantics(oracle_mod):
    """first nsample hits ascending, padded with the first hit (tf_grouping_g.cu:16-31)"""
    g = load_golden("grouping_P")
    xyz, q = g["xyz"], g["new_xyz"]
    idx, cnt = g["idx_r030"], g["cnt_r030"]
    d = np.sqrt(((q[0, :8, None, :] - xyz[0, None, :, :]) ** 2).sum(-1, dtype=np.float32))
    for j in range(8):
        hits = np.nonzero(d[j] < np.float32(0.3))[0][:32]
        assert cnt[0, j] == len(hits)
        assert idx[0, j, :len(hits)].tolist() == hits.tolist()
        assert (idx[0, j, len(hits):] == hits[0]).all()


def test_group_point_golden(oracle_mod):
    g = load_golden("grouping_P")
    assert np.array_equal(oracle_mod.group_point(g["xyz"], g["idx_r010"]), g["grouped_xyz_r010"])
    idx_f = np.ascontiguousarray(g["idx_r010"][:, :64])
    assert np.array_equal(oracle_mod.group_point(g["feats"], idx_f), g["grouped_feat_r010"])
    assert np.array_equal(oracle_mod.group_point_grad(g["feats"].shape, idx_f, g["grad_out"]), g["grad_points"])
    r = load_golden("grouping_reftest")
    idx, cnt = oracle_mod.query_ball_point(0.3, 32, r["xyz1"], r["xyz2"])
    assert np.array_equal(idx, r["idx"]) and np.array_equal(cnt, r["cnt"])
    assert np.array_equal(oracle_mod.group_point(r["points"], idx), r["grouped"])


def test_fps_golden_and_tie_rule(oracle_mod):
    g = load_golden("fps")
    for name, m in (("unit", 256), ("dup", 256), ("big", 1700), ("big2", 1400), ("tiny", 5)):
        out = oracle_mod.farthest_point_sample(m, g[name])
        assert np.array_equal(out, g[name + "_fps"]), name
        assert (out[:, 0] == 0).all()
    # duplicates 1536 apart share k mod 512: the smaller k wins, so no index >= 1536 before the
    # distinct points (1536 of them) are exhausted
    assert (g["big_fps"][0, :1536] < 1536).all()
    # duplicates 900 apart: copy k+900 has (k+900) mod 512 = (k+388) mod 512, smaller than k mod 512
    # whenever k mod 512 >= 124, so some LATER copies are picked first -- the rule is not "smallest k"
    first = g["big2_fps"][0, :1200]
    assert ((first >= 1000) & (first < 1300)).any()


def test_fps_is_greedy_maxmin(oracle_mod):
    g = load_golden("fps")
    xyz, out = g["unit"][0], g["unit_fps"][0]
    td = np.full(len(xyz), np.inf, np.float32)
    for j in range(1, 32):
        td = np.minimum(td, ((xyz - xyz[out[j - 1]]) ** 2).sum(-1, dtype=np.float32))
        assert td[out[j]] == td.max()


def test_gather_golden(oracle_mod):
    g = load_golden("gather")
    assert np.array_equal(oracle_mod.gather_point(g["xyz"], g["idx"]), g["out"])
    assert np.array_equal(oracle_mod.gather_point_grad(g["xyz"].shape, g["idx"], g["out_g"]), g["inp_g"])


def test_interpolate_golden(oracle_mod):
    g = load_golden("interpolate")
    dist, idx = oracle_mod.three_nn(g["unknown"], g["known"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])
    # hand-derivable: sorted, squared distances, ascending
    d_full = ((g["unknown"][:, :, None, :].astype(np.float64) - g["known"][:, None, :, :]) ** 2).sum(-1)
    order = np.argsort(d_full, axis=-1, kind="stable")[:, :, :3]
    assert np.array_equal(idx, order)
    np.testing.assert_allclose(dist, np.take_along_axis(d_full, order, -1), rtol=1e-5)
    d2, i2 = oracle_mod.three_nn(g["unknown"][:, :16], g["known"][:, :2])
    assert np.array_equal(i2, g["idx_m2"]) and np.array_equal(d2, g["dist_m2"])
    out = oracle_mod.three_interpolate(g["points"], g["idx"], g["weight"])
    assert np.array_equal(out, g["out"])
    assert np.array_equal(oracle_mod.three_interpolate_grad(g["points"].shape, g["idx"], g["weight"], g["grad_out"]),
                          g["grad_points"])
    cf = oracle_mod.three_interpolate_cf(g["points"].transpose(0, 2, 1), g["idx"], g["weight"])
    assert np.array_equal(cf.transpose(0, 2, 1), out)
    gcf = oracle_mod.three_interpolate_cf_grad(g["points"].transpose(0, 2, 1).shape, g["idx"], g["weight"],
                                               g["grad_out"].transpose(0, 2, 1))
    np.testing.assert_allclose(gcf.transpose(0, 2, 1), g["grad_points"], rtol=1e-5, atol=1e-6)


def test_select_top_k_golden(oracle_mod):
    g = load_golden("select_top_k")
    oi, od = oracle_mod.select_top_k(5, g["dist"])
    assert np.array_equal(oi, g["outi"]) and np.array_equal(od, g["out"])
    assert np.array_equal(od[:, :, :5], np.sort(g["dist"], -1)[:, :, :5])
    # the reference's own demo input, grouping/test/selection_sort.cpp:68-78: dist[i] = 10 - i
    d = (10 - np.arange(2 * 2 * 4, dtype=np.float32)).reshape(2, 2, 4)
    oi, od = oracle_mod.select_top_k(3, d)
    assert oi[0, 0, :3].tolist() == [3, 2, 1] and od[0, 0, :3].tolist() == [7.0, 8.0, 9.0]


def test_bev_iou_known_answers(oracle_mod):
    g = load_golden("bev_iou")
    ov, iou = oracle_mod.compute_bev_iou(g["demo_prop"], g["demo_gt"])          # bev_iou/bev_iou.py:47-50
    assert np.array_equal(ov, [[1, 0, 0], [0, 1, 0]]) and np.array_equal(iou, [[1, 0, 0], [0, 0.25, 0]])
    assert oracle_mod.oriented_nms(g["demo_nms"], 0.5).tolist() == [0, 1, 0]     # bev_iou/bev_iou.py:62-65
    _, iou = oracle_mod.compute_bev_iou(g["demo_nms"], g["demo_nms"])
    assert iou[0, 2] == np.float32(0.5625)
    # a rotated square against itself and against its 90-degree rotation: IoU 1
    sq = np.array([[0, 0, 2, 2, 0.3], [0, 0, 2, 2, 0.3 + np.pi / 2]], np.float32)
    _, iou = oracle_mod.compute_bev_iou(sq, sq)
    np.testing.assert_allclose(iou, 1.0, atol=2e-6)
    # 2x1 box crossed with its 90-degree rotation: overlap = 1x1 square, IoU = 1/3
    cr = np.array([[-1, -0.5, 1, 0.5, 0.0], [-1, -0.5, 1, 0.5, np.pi / 2]], np.float32)
    ov, iou = oracle_mod.compute_bev_iou(cr[:1], cr[1:])
    np.testing.assert_allclose(ov, 1.0, atol=1e-5)
    np.testing.assert_allclose(iou, 1.0 / 3.0, atol=1e-5)


def test_bev_iou_golden(oracle_mod):
    g = load_golden("bev_iou")
    ov, iou = oracle_mod.compute_bev_iou(g["a"], g["b"])
    assert np.array_equal(ov, g["overlap_ab"]) and np.array_equal(iou, g["iou_ab"])
    np.testing.assert_allclose(iou[np.arange(8), np.arange(8)], 1.0, atol=1e-5)  # exact duplicates
    assert (iou >= 0).all() and (iou <= 1 + 1e-5).all()
    for t in (0.85, 0.8, 0.01):
        keep, kept = oracle_mod.oriented_nms(g["nms_boxes"], t, return_count=True)
        assert np.array_equal(keep, g["keep_%03d" % int(t * 100)])
        assert keep[0] == 0 and (keep[kept:] == keep[0]).all() and (np.diff(keep[:kept]) > 0).all()
    # the margin the fixture was built with: no decision sits within 1e-4 of a threshold
    _, iou = oracle_mod.compute_bev_iou(g["nms_boxes"], g["nms_boxes"])
    for t in (0.85, 0.8, 0.01):
        assert not np.any(np.abs(iou - t) < 1e-4)


def test_nms_sweep_matches_greedy(oracle_mod):
    g = load_golden("bev_iou")
    boxes = g["nms_boxes"][:130]
    _, iou = oracle_mod.compute_bev_iou(boxes, boxes)
    alive, keep = np.ones(len(boxes), bool), []
    for i in range(len(boxes)):
        if alive[i]:
            keep.append(i)
            alive[i + 1:] &= ~(iou[i, i + 1:] > 0.8)
    got, kept = oracle_mod.oriented_nms(boxes, 0.8, return_count=True)
    assert kept == len(keep) and got[:kept].tolist() == keep
    mask = oracle_mod.nms_mask(boxes, 0.8)
    assert mask.shape == (130, 3)
    assert np.array_equal(oracle_mod.nms_sweep(mask)[0], got)


def test_crop_golden(oracle_mod):
    g = load_golden("crop")
    res = oracle_mod.pc_crop_and_sample(g["pts"], g["fts"], g["intensities"], g["mask"], g["boxes"], g["box_ind"], 32)
    for got, name in zip(res, ("crop_pts", "crop_fts", "crop_int", "crop_mask", "crop_ind", "non_empty")):
        assert np.array_equal(got, g[name]), name
    assert not res[5][0] and (res[0][0] == 0).all() and (res[4][0] == 0).all()      # empty box: zeros
    ind = res[4]
    assert (np.diff(ind[1]) > 0).all()                                                # full box: ascending, no pad
    gf = oracle_mod.pc_crop_and_sample_grad_fts(g["fts"].shape, g["box_ind"], g["crop_ind"], g["grad_crop_fts"])
    assert np.array_equal(gf, g["grad_fts"])
    # demo of cropping/tf_cropping.py:57-77
    d = oracle_mod.pc_crop_and_sample(g["demo_pts"], np.ones((1, 2, 1), np.float32),
                                      np.arange(2, dtype=np.float32).reshape(1, 2, 1), np.array([[True, False]]),
                                      g["demo_box"], np.array([0], np.int32), 1)
    assert np.array_equal(d[4], g["demo_crop_ind"]) and np.array_equal(d[5], g["demo_non_empty"])


def test_crop_padding_rule(oracle_mod):
    """slot s >= cnt copies slot (s - cnt) mod cnt (tf_cropping_g.cu:108-126)"""
    g = load_golden("crop")
    ind, ne = g["crop_ind"], g["non_empty"]
    pts, boxes, box_ind = g["pts"], g["boxes"], g["box_ind"]
    for bx in range(len(boxes)):
        if not ne[bx]:
            continue
        row = ind[bx]
        cnt = next((s for s in range(1, 32) if row[s] <= row[s - 1]), 32)
        assert (np.diff(row[:cnt]) > 0).all()
        for s in range(cnt, 32):
            assert row[s] == row[(s - cnt) % cnt]


# ---- against the reference's own CPU programs (only where oracle/_ref was built) ----
def _need_ref(oracle_mod, kind):
    if not oracle_mod.ref_available(kind):
        pytest.skip("oracle/_ref/libhfref_%s.so not built (no /root/reference on this box)" % kind)


def test_oracle_vs_reference_cpu_grouping(oracle_mod):
    _need_ref(oracle_mod, "qbp")
    rng = np.random.default_rng(7)
    for (b, n, m, r, ns, c) in ((2, 1024, 256, 0.1, 32, 16), (3, 777, 130, 0.25, 64, 5), (1, 128, 8, 0.3, 32, 16)):
        x1 = rng.random((b, n, 3), dtype=np.float32)
        x2 = rng.random((b, m, 3), dtype=np.float32)
        idx, _ = oracle_mod.query_ball_point(r, ns, x1, x2)
        assert np.array_equal(idx, oracle_mod.ref_query_ball_point(r, ns, x1, x2))
        pts = rng.standard_normal((b, n, c)).astype(np.float32)
        assert np.array_equal(oracle_mod.group_point(pts, idx), oracle_mod.ref_group_point(pts, idx))
        go = rng.standard_normal((b, m, ns, c)).astype(np.float32)
        assert np.array_equal(oracle_mod.group_point_grad(pts.shape, idx, go),
                              oracle_mod.ref_group_point_grad(pts.shape, idx, go))


def test_oracle_vs_reference_cpu_interpolate(oracle_mod):
    _need_ref(oracle_mod, "itp")
    rng = np.random.default_rng(8)
    b, n, m, c = 2, 300, 77, 9
    idx = rng.integers(0, m, (b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3), dtype=np.float32)
    pts = rng.standard_normal((b, m, c)).astype(np.float32)
    assert np.array_equal(oracle_mod.three_interpolate(pts, idx, w), oracle_mod.ref_three_interpolate(pts, idx, w))
    go = rng.standard_normal((b, n, c)).astype(np.float32)
    assert np.array_equal(oracle_mod.three_interpolate_grad(pts.shape, idx, w, go),
                          oracle_mod.ref_three_interpolate_grad(pts.shape, idx, w, go))


def _threenn_cases():
    g = load_golden("three_nn_origin")
    return {k[:-6]: g[k] for k in g if k.endswith("_known")}, g


def test_oracle_three_nn_vs_reference_cpu_at_the_origin(oracle_mod):
    """a5 pin: oracle.three_nn (restating tf_interpolate_g.cu:22-65, unbuildable here) against the reference's compiled
    threenn_cpu for unknown points at the origin -- bit-for-bit, distances and indices (see oracle.ref_threenn_origin)."""
    _need_ref(oracle_mod, "itp")
    cases, _ = _threenn_cases()
    assert set(cases) == {"random", "large", "ties", "lattice", "m1", "m2", "m3", "ramps", "all_equal"}
    for name, known in cases.items():
        b, m, _ = known.shape
        rd, ri = oracle_mod.ref_threenn_origin(known, n=2)
        od, oi = oracle_mod.three_nn(np.zeros((b, 2, 3), np.float32), known)
        assert np.array_equal(oi, ri), name
        assert np.array_equal(od.view(np.uint32), rd.view(np.uint32)), name
        if m < 3:                                             # sentinels: double 1e40 -> +inf in fp32, index 0
            assert np.isinf(rd[:, :, m:]).all() and (ri[:, :, m:] == 0).all(), name


def test_oracle_three_nn_golden_from_reference_cpu(oracle_mod):
    """the same pin as committed data (tests/golden/three_nn_origin.npz holds threenn_cpu's outputs): runs where the
    reference build is absent too"""
    cases, g = _threenn_cases()
    for name, known in cases.items():
        od, oi = oracle_mod.three_nn(np.zeros((known.shape[0], 1, 3), np.float32), known)
        assert np.array_equal(oi, g[name + "_idx"]), name
        assert np.array_equal(od.view(np.uint32), g[name + "_dist"].view(np.uint32)), name
    # duplicates: among equal distances the LOWER index must come first (strict '<')
    d, i = g["ties_dist"][:, 0], g["ties_idx"][:, 0]
    for r in range(len(d)):
        for s_ in range(2):
            if d[r, s_] == d[r, s_ + 1]:
                assert i[r, s_] < i[r, s_ + 1]


def test_oracle_three_nn_translated_lattice_vs_reference_cpu(oracle_mod):
    """The subtraction of tf_interpolate_g.cu:47 against reference-compiled code where it is exact: on a dyadic lattice
    (multiples of 2^-6 below 64) ux - x is representable, so the reference's d for (u, known) has the bits of threenn_cpu's
    x2*x2+y2*y2+z2*z2 on the cloud translated by -u.  One batch element per query (threenn_cpu advances both clouds per batch)."""
    _need_ref(oracle_mod, "itp")
    rng = np.random.default_rng(22)
    known = (rng.integers(-2048, 2048, (300, 3)) / 64.0).astype(np.float32)
    unknown = (rng.integers(-2048, 2048, (128, 3)) / 64.0).astype(np.float32)
    shifted = (known[None, :, :] - unknown[:, None, :]).astype(np.float32)         # exact
    assert np.array_equal(shifted.astype(np.float64), known[None].astype(np.float64) - unknown[:, None].astype(np.float64))
    rd, ri = oracle_mod.ref_threenn_origin(-shifted)                               # (u - x): the sign the kernel squares
    od, oi = oracle_mod.three_nn(unknown[None], known[None])
    assert np.array_equal(oi[0], ri[:, 0]) and np.array_equal(od[0].view(np.uint32), rd[:, 0].view(np.uint32))
