"""The second stage (heterofusionrcnn_amd/rcnn.py) and the two-stage flow (two_stage.py) of BASELINE config 5.

CPU (-m "not gpu"): the glue restated from hf/core/models/rcnn_model.py against loops -- crop_and_resize (the documented
formula of tf.image.crop_and_resize), the projection of 3-D boxes into the image (projection.py:35-96), the canonical
transform, the proposal expansion, the numbers rcnn_multiclass.config implies (bins, head width, layer widths).
GPU (-m gpu): the two stages at the config's OWN sizes -- RPN = rpn_multiclass.config (PointCNN, image fusion), 9000 boxes
into the first NMS, 100 proposals per frame, RoI crops of 512 points x 288 channels, the RCNN's PointCNN (K = 4 / 8 / 12 / 12),
flat_concat fusion with 7 x 7 image crops -- with the crop and both NMS hand-offs checked against the oracle.
Weights are random (no checkpoint ships with the reference); TensorFlow is not importable: parity unpinned against reference
outputs."""
import numpy as np
import pytest
import torch

from heterofusionrcnn_amd import fusion
from heterofusionrcnn_amd import rcnn as RC


def test_rcnn_config_numbers_and_wiring():
    cfg = RC.RcnnConfig()
    assert cfg.num_bin_xz == 6 and cfg.num_bin_theta == 9 and cfg.head_width == 46          # rcnn_model.py:109-116
    assert abs(cfg.r_theta - 0.25 * np.pi) < 1e-12 and abs(cfg.delta_theta - np.pi / 18) < 1e-12
    with torch.device("meta"):
        m = RC.RcnnModel(cfg)
    # pointcnn.py:257-266 on rcnn_multiclass.config:157-186: C_pts_fts, depth multipliers, K, with_global on the last layer
    assert [x.k for x in m.encoder.enc] == [4, 8, 12, 12]
    assert [x.lift0.linear.out_features for x in m.encoder.enc] == [128, 128, 128, 256]
    assert [x.conv.depthwise.shape[2] for x in m.encoder.enc] == [4, 1, 2, 1]
    assert m.encoder.enc[0].conv.depthwise.shape[1] == 128 + 288 + 256                      # lifted | crop_fts | mlp
    assert m.encoder.out_channel == 1280 and m.roi_points == 8
    assert m.cls_fc.layers[0].linear.in_features == 8 * 1280 + 7 * 7 * 32                   # flat_concat
    assert m.cls_logits.out_features == 4 and m.reg_out.linear.out_features == 3 * 46
    assert m.mlp.layers[0].linear.in_features == 6


def test_image_crop_and_resize_against_a_loop():
    rng = np.random.default_rng(0)
    img = rng.standard_normal((2, 9, 13, 3)).astype(np.float32)
    boxes = np.array([[0.1, 0.2, 0.8, 0.9], [0.0, 0.0, 1.0, 1.0], [-0.2, 0.3, 0.5, 1.3], [0.4, 0.4, 0.4, 0.4], [0.9, 0.1, 0.2, 0.7]], np.float32)
    ind = np.array([0, 1, 1, 0, 1], np.int32)
    got = fusion.image_crop_and_resize(torch.from_numpy(img), torch.from_numpy(boxes), torch.from_numpy(ind), 4).numpy()
    h, w, crop = 9, 13, 4
    want = np.zeros((5, crop, crop, 3), np.float32)
    for n, (y1, x1, y2, x2) in enumerate(boxes):
        for i in range(crop):
            y = y1 * (h - 1) + i * (y2 - y1) * (h - 1) / (crop - 1)
            for j in range(crop):
                x = x1 * (w - 1) + j * (x2 - x1) * (w - 1) / (crop - 1)
                if y < 0 or y > h - 1 or x < 0 or x > w - 1:
                    continue
                y0, x0 = int(np.floor(y)), int(np.floor(x))
                yb, xb = int(np.ceil(y)), int(np.ceil(x))
                ly, lx = y - y0, x - x0
                top = img[ind[n], y0, x0] + (img[ind[n], y0, xb] - img[ind[n], y0, x0]) * lx
                bot = img[ind[n], yb, x0] + (img[ind[n], yb, xb] - img[ind[n], yb, x0]) * lx
                want[n, i, j] = top + (bot - top) * ly
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    assert np.count_nonzero(want[2] == 0) > 0          # the box that leaves the image has extrapolated samples


def test_project_boxes_and_geometry_helpers():
    from heterofusionrcnn_amd.modules import box_3d_to_box_8co
    rng = np.random.default_rng(1)
    boxes = np.concatenate([rng.uniform(-10, 10, (2, 5, 1)), rng.uniform(0, 2, (2, 5, 1)), rng.uniform(8, 40, (2, 5, 1)),
                            rng.uniform(1, 4, (2, 5, 3)), rng.uniform(-3, 3, (2, 5, 1))], -1).astype(np.float32)
    p2 = np.array([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.2], [0, 0, 1, 0.003]], np.float32)
    calib = np.stack([p2, p2 * np.float32(1.0)])
    box, norm = fusion.project_boxes_to_image(torch.from_numpy(boxes), torch.from_numpy(calib), (360, 1200))
    c8 = box_3d_to_box_8co(torch.from_numpy(boxes.reshape(-1, 7))).numpy().reshape(2, 5, 3, 8)
    for b in range(2):
        for n in range(5):
            pts = np.concatenate([c8[b, n], np.ones((1, 8), np.float32)], 0)                 # (4, 8)
            pix = calib[b] @ pts
            pix = pix[:2] / pix[2]
            want = np.array([pix[0].min(), pix[1].min(), pix[0].max(), pix[1].max()])
            np.testing.assert_allclose(box[b, n].numpy(), want, rtol=1e-4, atol=1e-2)
            np.testing.assert_allclose(norm[b, n].numpy(), want / np.array([1200, 360, 1200, 360]), rtol=1e-4, atol=1e-4)
    prop = torch.from_numpy(boxes.reshape(-1, 7))
    e = RC.expand_proposals(prop, 1.0)
    assert torch.allclose(e[:, 3:6], prop[:, 3:6] + 2.0) and torch.allclose(e[:, 1], prop[:, 1] + 1.0) and torch.equal(e[:, [0, 2, 6]], prop[:, [0, 2, 6]])
    pts = torch.randn(10, 7, 3)
    ct = RC.canonical_transform(pts, prop)
    assert torch.allclose(torch.norm(ct, dim=2), torch.norm(pts - prop[:, None, :3], dim=2), atol=1e-4)   # a rotation about y
    assert torch.allclose(ct[..., 1], pts[..., 1] - prop[:, None, 1])
    # a point ahead of the box along its heading lands on +x
    ahead = prop[:, None, :3] + torch.stack([torch.cos(prop[:, 6]), torch.zeros(10), -torch.sin(prop[:, 6])], -1)[:, None]
    ca = RC.canonical_transform(ahead, prop)
    assert torch.allclose(ca[:, 0], torch.tensor([1.0, 0.0, 0.0]).expand(10, 3), atol=1e-4)


# ------------------------------------------------------------------------------------------------ GPU
def _frames(b, seed):
    import bench
    rng = np.random.default_rng(seed)
    xyz = torch.from_numpy(bench.kitti_frustum(rng, b, bench.N0)).cuda()
    inten = torch.from_numpy(rng.uniform(-0.5, 0.5, (b, bench.N0, 1)).astype(np.float32)).cuda()
    img = torch.randn(b, bench.IMG_H, bench.IMG_W, bench.IMG_C, device="cuda")
    calib = torch.from_numpy(bench.KITTI_P2).cuda().repeat(b, 1, 1).contiguous()
    return xyz, inten, img, calib


@pytest.mark.gpu
def test_config5_two_stage_inference_at_its_own_sizes():
    import oracle
    from heterofusionrcnn_amd import dp, modules
    from heterofusionrcnn_amd.two_stage import TwoStageDetector, run_sharded
    torch.manual_seed(3)
    det = TwoStageDetector().cuda().eval()
    assert det.pre_nms_size == 9000 and det.rpn_post_nms_size == 100 and det.rcnn_cfg.roi_crop_size == 512
    b = 2
    xyz, inten, img, calib = _frames(b, 3)
    dets, dbg = det(xyz, inten, img, calib, return_debug=True)
    rpn, rc = dbg["rpn"], dbg["rcnn"]
    host = lambda t: t.detach().cpu().numpy()
    # ---- first hand-off: 9000 score-sorted boxes -> oriented NMS 0.8 -> 100 proposals (padded with the first kept box)
    assert rpn["pre_nms_boxes"].shape == (b, 9000, 7) and rpn["proposals"].shape == (b, 100, 7) and rpn["rpn_fts"].shape == (b, 16384, 288)
    assert bool((rpn["pre_nms_scores"][:, :-1] >= rpn["pre_nms_scores"][:, 1:]).all())
    bev0 = host(modules.boxes3d_to_bev(rpn["pre_nms_boxes"][0]))
    o_keep, o_num = oracle.oriented_nms(bev0, 0.8, return_count=True)
    num0 = int(rpn["num_kept"][0])
    assert num0 == o_num and np.array_equal(host(rpn["keep"][0])[:num0], o_keep[:o_num])
    take = host(rpn["keep"][0])[:100]
    assert np.array_equal(host(rpn["proposals"][0]), host(rpn["pre_nms_boxes"][0])[take])
    # ---- RoI pooling: the expanded boxes' crops equal the oracle's (512 points, 288 channels, 200 RoIs)
    pool = rc["pool"]
    assert pool["crop_fts"].shape == (b * 100, 512, 288) and pool["img_rois"].shape == (b * 100, 7, 7, 32)
    want = oracle.pc_crop_and_sample(host(xyz), host(rpn["rpn_fts"]), host(inten), host(rpn["fg_mask"]), host(pool["boxes8"]),
                                     host(pool["box_ind"]), 512)
    for name, w in zip(("crop_pts", "crop_fts", "crop_int", "crop_mask", "crop_ind", "non_empty"), want):
        assert np.array_equal(host(pool[name]), w), name
    # ---- second hand-off: per frame, non-empty RoIs in score order -> oriented NMS 0.01 -> at most 100 boxes
    for f in range(b):
        ne = host(pool["non_empty"]).reshape(b, 100)[f]
        order = host(rc["order"][f])
        n_real = int(ne.sum())
        assert ne[order[:n_real]].all() and not ne[order[n_real:]].any()
        bev = host(modules.boxes3d_to_bev(rc["boxes_all"][f]))[order[:n_real]]
        if n_real:
            o_keep, o_num = oracle.oriented_nms(bev, 0.01, return_count=True)
            sel = order[o_keep[:min(o_num, 100)]]
            assert np.array_equal(host(dets[f]["boxes"]), host(rc["boxes_all"][f])[sel]), f
            assert len(dets[f]["boxes"]) == min(o_num, 100)
        else:
            assert len(dets[f]["boxes"]) == 0
        assert torch.isfinite(dets[f]["boxes"]).all() and bool(((dets[f]["classes"] >= 1) & (dets[f]["classes"] <= 3)).all())
        assert len(dets[f]["scores"]) == len(dets[f]["boxes"]) == len(dets[f]["classes"])
    # ---- sharded run with the RPN geometry computed ahead: the same detections frame by frame
    ctx = dp.DPContext(0, 1, 0, torch.device("cuda", 0))
    frames = [{"xyz": xyz[i], "intensity": inten[i], "img_fts": img[i], "calib": calib[i]} for i in (0, 1, 1)]
    merged = run_sharded(det, frames, ctx, frames_per_batch=1)
    assert sorted(merged) == [0, 1, 2]
    for j, i in enumerate((0, 1, 1)):
        alone = det(xyz[i:i + 1], inten[i:i + 1], img[i:i + 1], calib[i:i + 1])[0]
        assert torch.equal(merged[j]["boxes"], alone["boxes"].cpu()) and torch.equal(merged[j]["scores"], alone["scores"].cpu())


@pytest.mark.gpu
def test_rcnn_empty_rois_never_reach_the_output():
    """proposals parked in empty space: their RoIs are empty (rcnn_model.py:731-733 removes them before the NMS)"""
    torch.manual_seed(4)
    m = RC.RcnnModel().cuda().eval()
    xyz, inten, img, calib = _frames(1, 4)
    fts = torch.randn(1, 16384, 288, device="cuda")
    fg = torch.rand(1, 16384, device="cuda") < 0.3
    prop = torch.zeros(1, 12, 7, device="cuda")
    prop[0, :, 0] = torch.linspace(-20, 20, 12, device="cuda")
    prop[0, :, 1] = 1.5
    prop[0, :, 2] = torch.linspace(10, 60, 12, device="cuda")
    prop[0, :, 3:6] = torch.tensor([3.9, 1.6, 1.5], device="cuda")
    prop[0, 6:, 1] = -500.0                                   # far above everything: empty RoIs
    out, dbg = m.detect(xyz, fts, inten, fg, prop, img, calib)
    ne = dbg["pool"]["non_empty"]
    assert int(ne.sum()) <= 6 and not ne[6:].any()
    assert len(out[0]["boxes"]) <= int(ne.sum())
    kept = dbg["order"][0, :int(ne.sum())]
    assert bool(ne[kept.long()].all())
