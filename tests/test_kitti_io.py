"""host-side data formats: synthetic round trips everywhere, and -- only where the reference
checkout is present (this container, never the GPU box) -- its own 13 KITTI test frames read in place."""
import os

import numpy as np
import pytest

from heterofusionrcnn_amd import kitti_io

REF = "/root/reference/hf/tests/datasets/Kitti/object/training"


def _write_calib(path, p2, r0, tr):
    with open(path, "w") as f:
        for name in ("P0", "P1"):
            f.write(name + ": " + " ".join("%.12e" % v for v in np.eye(3, 4).ravel()) + "\n")
        f.write("P2: " + " ".join("%.12e" % v for v in p2.ravel()) + "\n")
        f.write("P3: " + " ".join("%.12e" % v for v in np.eye(3, 4).ravel()) + "\n")
        f.write("R0_rect: " + " ".join("%.12e" % v for v in r0.ravel()) + "\n")
        f.write("Tr_velo_to_cam: " + " ".join("%.12e" % v for v in tr.ravel()) + "\n")
        f.write("Tr_imu_to_velo: " + " ".join("%.12e" % v for v in np.eye(3, 4).ravel()) + "\n")


def test_velodyne_calib_and_camera_filter(tmp_path):
    rng = np.random.default_rng(0)
    cloud = np.concatenate([rng.uniform(-60, 60, (5000, 3)), rng.uniform(0, 1, (5000, 1))], 1).astype(np.float32)
    cloud.tofile(tmp_path / "000007.bin")
    p2 = np.array([[700.0, 0, 600, 45], [0, 700, 180, -0.3], [0, 0, 1, 0.004]])
    r0 = np.eye(3)
    tr = np.array([[0.0, -1, 0, 0.01], [0, 0, -1, -0.07], [1, 0, 0, -0.27]])    # velodyne x forward -> camera z forward
    _write_calib(tmp_path / "000007.txt", p2, r0, tr)
    calib = kitti_io.read_calib(tmp_path / "000007.txt")
    assert np.allclose(calib["p2"], p2) and np.allclose(calib["tr_velo_to_cam"], tr) and calib["r0_rect"].shape == (3, 3)
    assert np.array_equal(kitti_io.read_velodyne(tmp_path / "000007.bin"), cloud)
    rect = kitti_io.lidar_to_rect(cloud[:, :3], calib)
    assert np.allclose(rect[:, 2], cloud[:, 0] - 0.27) and np.allclose(rect[:, 0], -cloud[:, 1] + 0.01)
    pts, inten = kitti_io.load_point_cloud(tmp_path / "000007.bin", calib, (370, 1224))
    uv = kitti_io.project_to_image(pts, p2)
    assert len(pts) and (pts[:, 2] > 0).all() and (uv > 0).all() and (uv[:, 0] < 1224).all() and (uv[:, 1] < 370).all()
    assert inten.shape == (len(pts), 1)
    with pytest.raises(ValueError):
        np.zeros(7, np.float32).tofile(tmp_path / "bad.bin"); kitti_io.read_velodyne(tmp_path / "bad.bin")


def test_sampling_rule_invariants():
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(-40, 40, 30000), rng.uniform(-3, 2, 30000), rng.uniform(0.1, 70, 30000)], 1)
    inten = rng.uniform(0, 1, (30000, 1))
    out, choice = kitti_io.sample_points(pts, inten, 16384, np.random.default_rng(2))
    far = np.where(pts[:, 2] >= 40.0)[0]
    assert out.shape == (16384, 4) and out.dtype == np.float32 and len(np.unique(choice)) == 16384
    assert np.isin(far, choice).all()                                   # every far point survives
    assert np.allclose(out[:, :3], pts[choice]) and np.allclose(out[:, 3], inten[choice, 0] - 0.5, atol=1e-6)
    again, choice2 = kitti_io.sample_points(pts, inten, 16384, np.random.default_rng(2))
    assert np.array_equal(choice, choice2)                              # reproducible with a seeded generator
    # a short cloud: every point at least once, padded by re-draws (without replacement up to 2x)
    out, choice = kitti_io.sample_points(pts[:10000], inten[:10000], 16384, np.random.default_rng(3))
    counts = np.bincount(choice, minlength=10000)
    assert counts.min() == 1 and counts.max() == 2 and counts.sum() == 16384
    out, choice = kitti_io.sample_points(pts[:3000], inten[:3000], 16384, np.random.default_rng(4))
    assert np.bincount(choice, minlength=3000).min() >= 1 and len(choice) == 16384
    with pytest.raises(ValueError):                                     # more far points than the budget
        kitti_io.sample_points(np.array([[0, 0, 50.0]] * 100), np.zeros((100, 1)), 10)


def test_handoff_and_result_files_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    props, scores = rng.uniform(-30, 30, (17, 7)), rng.uniform(0, 1, 17)
    kitti_io.save_proposals_and_scores(tmp_path / "p.txt", props, scores)
    p2, s2 = kitti_io.load_proposals_and_scores(tmp_path / "p.txt")
    assert np.allclose(p2, props, atol=5e-4) and np.allclose(s2, scores, atol=5e-4)        # "%.3f"
    assert all(len(line.split()) == 8 for line in open(tmp_path / "p.txt"))
    n = 64
    feats = dict(pts=rng.standard_normal((n, 3)), intensity=rng.uniform(-.5, .5, (n, 1)), fg_mask=rng.integers(0, 2, n),
                 pts_fts=rng.standard_normal((n, 128)), img_fts=rng.standard_normal((n, 32)))
    kitti_io.save_rpn_features(tmp_path / "f.npy", **feats)
    back = kitti_io.load_rpn_features(tmp_path / "f.npy", 128)
    assert np.load(tmp_path / "f.npy").shape == (n, 3 + 1 + 1 + 128 + 32)
    for k, v in feats.items():
        assert np.allclose(back[k].reshape(np.asarray(v).shape), v)
    kitti_io.write_kitti_results(tmp_path / "r.txt", ["Car", "Pedestrian"], [[1, 2, 3, 4], [5, 6, 7, 8]],
                                 [[1, 1.5, 20, 3.9, 1.6, 1.5, 0.1], [2, 1.4, 10, 0.8, 0.6, 1.7, -1.0]], [0.9, 0.5])
    types, b3, b2, misc = kitti_io.read_labels(tmp_path / "r.txt")          # a result file is a label file + score
    assert types == ["Car", "Pedestrian"] and np.allclose(b3[0], [1, 1.5, 20, 3.9, 1.6, 1.5, 0.1]) and np.allclose(b2[1], [5, 6, 7, 8])
    assert kitti_io.read_labels(tmp_path / "missing.txt")[1].shape == (0, 7)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference KITTI test frames not present on this machine")
def test_reference_kitti_frames_read_in_place():
    names = sorted(f[:-4] for f in os.listdir(os.path.join(REF, "velodyne")) if f.endswith(".bin"))
    assert len(names) >= 10
    for name in names[:4]:
        calib = kitti_io.read_calib(os.path.join(REF, "calib", name + ".txt"))
        assert calib["p2"][2, 2] == 1.0 and abs(np.linalg.det(calib["r0_rect"]) - 1) < 1e-3
        cloud = kitti_io.read_velodyne(os.path.join(REF, "velodyne", name + ".bin"))
        assert cloud.shape[1] == 4 and len(cloud) > 50000 and 0 <= cloud[:, 3].min() and cloud[:, 3].max() <= 1
        pts, inten = kitti_io.load_point_cloud(os.path.join(REF, "velodyne", name + ".bin"), calib, (370, 1224))
        assert 10000 < len(pts) < len(cloud) and (pts[:, 2] > 0).all()
        out, _ = kitti_io.sample_points(pts, inten, 16384, np.random.default_rng(0))
        assert out.shape == (16384, 4) and -0.5 <= out[:, 3].min() and out[:, 3].max() <= 0.5
        types, b3, b2, _ = kitti_io.read_labels(os.path.join(REF, "label_2", name + ".txt"), classes=("Car", "Pedestrian", "Cyclist"))
        if len(types):
            # a labelled object's centre projects inside its own 2D box (ties the label, calib and projection readers)
            uv = kitti_io.project_to_image(b3[:, :3] - np.array([0, 1, 0]) * b3[:, 5:6] / 2, calib["p2"])
            inside = (uv[:, 0] > b2[:, 0] - 2) & (uv[:, 0] < b2[:, 2] + 2) & (uv[:, 1] > b2[:, 1] - 2) & (uv[:, 1] < b2[:, 3] + 2)
            assert inside.mean() > 0.7
