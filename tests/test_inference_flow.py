"""The file-to-file flow of BASELINE config 5 (heterofusionrcnn_amd/inference.py): KITTI files -> tensors -> two stages ->
KITTI result files.  CPU: the image branch's shapes, the result-box projection rules of box_3d_projector.py:88-163, the frame
loader on synthetic files.  GPU: three synthetic frames through the whole flow (random weights: plumbing, not accuracy)."""
import os

import numpy as np
import pytest
import torch

from heterofusionrcnn_amd import inference as INF
from heterofusionrcnn_amd import kitti_io

P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]])


def _write_frame(root, name, rng, n=30000):
    from PIL import Image
    for d in ("velodyne", "calib", "image_2"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    # velodyne frame: x forward, y left, z up; the calibration below maps it to the camera frame (x right, y down, z forward)
    pts = np.stack([rng.uniform(2, 70, n), rng.uniform(-30, 30, n), rng.uniform(-2.5, 1.0, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
    pts.tofile(os.path.join(root, "velodyne", name + ".bin"))
    tr = np.array([[0, -1, 0, 0.0], [0, 0, -1, -0.08], [1, 0, 0, -0.27]])
    with open(os.path.join(root, "calib", name + ".txt"), "w") as f:
        for i in range(4):
            f.write("P%d: %s\n" % (i, " ".join("%.6e" % v for v in P2.reshape(-1))))
        f.write("R0_rect: %s\n" % " ".join("%.6e" % v for v in np.eye(3).reshape(-1)))
        f.write("Tr_velo_to_cam: %s\n" % " ".join("%.6e" % v for v in tr.reshape(-1)))
        f.write("Tr_imu_to_velo: %s\n" % " ".join("0" for _ in range(12)))
    Image.fromarray(rng.integers(0, 255, (375, 1242, 3), dtype=np.uint8)).save(os.path.join(root, "image_2", name + ".png"))


def test_img_vgg_pyr_shapes_and_preprocessing():
    torch.manual_seed(0)
    net = INF.ImgVggPyr().eval()
    img = torch.rand(2, 16, 24, 3) * 255
    with torch.no_grad():
        out = net(img)
    assert out.shape == (2, 16, 24, 32) and torch.isfinite(out).all()
    # vgg_conv1..4 of rpn_multiclass.config:120-128: 2 + 2 + 3 + 3 encoder convolutions, 3 up-convolutions, 3 fusion convolutions
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    ups = [m for m in net.modules() if isinstance(m, torch.nn.ConvTranspose2d)]
    assert len(convs) == 13 and len(ups) == 3
    assert [m.out_channels for m in convs[:10]] == [32, 32, 64, 64, 128, 128, 128, 256, 256, 256]
    # the mean image is subtracted before the first convolution
    with torch.no_grad():
        a = net(torch.tensor(INF.IMAGE_MEAN).view(1, 1, 1, 3).expand(1, 16, 24, 3).contiguous())
        b = net.conv1[0][0](torch.zeros(1, 3, 16, 24))
    assert float(b.abs().max()) == 0.0 and torch.isfinite(a).all()


def test_result_box_projection_rules():
    size = (1242, 375)
    car = np.array([2.0, 1.6, 20.0, 3.9, 1.6, 1.5, 0.3])
    box = INF.project_box3d_to_image(car, P2, size)
    assert box is not None and 0 <= box[0] < box[2] <= 1242 and 0 <= box[1] < box[3] <= 375
    assert INF.project_box3d_to_image(np.array([60.0, 1.6, 20.0, 3.9, 1.6, 1.5, 0.0]), P2, size) is None          # right of the image
    assert INF.project_box3d_to_image(np.array([0.0, 1.6, 2.2, 3.9, 1.6, 1.5, 1.57]), P2, size) is None           # > 80 % of the image
    cut = INF.project_box3d_to_image(np.array([-12.5, 1.6, 15.0, 3.9, 1.6, 1.5, 0.0]), P2, size)                  # crosses the left border
    assert cut is not None and cut[0] == 0.0


def test_frame_loader_and_result_writer_on_synthetic_files(tmp_path):
    rng = np.random.default_rng(3)
    root = str(tmp_path / "kitti")
    _write_frame(root, "000007", rng)
    f = INF.load_kitti_frame(root, "000007", np.random.default_rng(0))
    assert f["xyz"].shape == (16384, 3) and f["intensity"].shape == (16384, 1) and f["image"].shape == (360, 1200, 3)
    assert f["calib"].shape == (3, 4) and f["image_size"] == (1242, 375)
    assert -0.5 <= f["intensity"].min() and f["intensity"].max() <= 0.5
    uv = kitti_io.project_to_image(f["xyz"], f["calib_orig"])
    assert (uv[:, 0] > 0).all() and (uv[:, 0] < 1242).all() and (uv[:, 1] > 0).all() and (uv[:, 1] < 375).all() and (f["xyz"][:, 2] > 0).all()
    # the detector's matrix is P2 rescaled with the image (kitti_dataset.py:399-400): a point whose original pixel is (u, v)
    # lands on (u * 1200/1242, v * 360/375) of the resized map
    uv_r = kitti_io.project_to_image(f["xyz"], f["calib"])
    assert np.allclose(uv_r[:, 0], uv[:, 0] * (1200 / 1242), atol=2e-2) and np.allclose(uv_r[:, 1], uv[:, 1] * (360 / 375), atol=2e-2)
    assert np.array_equal(f["calib"][2], f["calib_orig"][2]) and not np.array_equal(f["calib"][0], f["calib_orig"][0])
    det = {"boxes": np.array([[2.0, 1.6, 20.0, 3.9, 1.6, 1.5, 0.3], [60.0, 1.6, 20.0, 3.9, 1.6, 1.5, 0.0], [1.0, 1.6, 30.0, 0.8, 0.6, 1.7, 0.1]]),
           "scores": np.array([0.9, 0.8, 0.05]), "classes": np.array([1, 1, 2])}
    out = str(tmp_path / "000007.txt")
    assert INF.write_frame_results(out, det, f["calib_orig"], f["image_size"]) == 1           # one outside the image, one below the threshold
    line = open(out).read().split()
    assert line[0] == "Car" and len(line) == 16 and float(line[-1]) == 0.9 and float(line[3]) == -10.0


@pytest.mark.gpu
def test_loaded_frame_gathers_image_features_at_the_rescaled_pixel(tmp_path):
    """with the resized 360 x 1200 image the fusion must read pixel (u * sx, v * sy); the feature map here
    holds its own pixel coordinates, so the gathered rows say which pixel every point read"""
    from heterofusionrcnn_amd import fusion
    rng = np.random.default_rng(9)
    root = str(tmp_path / "kitti")
    _write_frame(root, "000001", rng)
    f = INF.load_kitti_frame(root, "000001", np.random.default_rng(0))
    vv, uu = np.meshgrid(np.arange(360, dtype=np.float32), np.arange(1200, dtype=np.float32), indexing="ij")
    fmap = torch.from_numpy(np.stack([uu, vv], -1)[None]).cuda()                       # (1,360,1200,2): [u, v] of the pixel
    got = fusion.project_gather(torch.from_numpy(f["xyz"][None]).cuda(), torch.from_numpy(f["calib"][None]).cuda(), fmap)[0].cpu().numpy()
    uv0 = kitti_io.project_to_image(f["xyz"], f["calib_orig"])                          # original-image pixels
    want = np.stack([uv0[:, 0] * (1200 / 1242), uv0[:, 1] * (360 / 375)], 1)
    inside = (want[:, 0] > 1) & (want[:, 0] < 1198) & (want[:, 1] > 1) & (want[:, 1] < 358)
    assert inside.sum() > 10000
    d = want[inside] - got[inside]                                                      # the integer cast of the scaled pixel
    assert (d > -2e-2).all() and (d < 1.0 + 2e-2).all()                                 # (fp32 projection on the device, fp64 here)
    wrong = np.abs(got[inside, 0] - uv0[inside, 0])                                     # what the unscaled matrix would read
    assert wrong.max() > 10


@pytest.mark.gpu
def test_kitti_files_to_kitti_results_through_both_stages(tmp_path):
    from heterofusionrcnn_amd import dp
    from heterofusionrcnn_amd.two_stage import TwoStageDetector
    rng = np.random.default_rng(5)
    root = str(tmp_path / "kitti")
    names = ["%06d" % i for i in (3, 11, 42)]
    for n in names:
        _write_frame(root, n, rng)
    torch.manual_seed(1)
    det = TwoStageDetector().cuda().eval()
    img_net = INF.ImgVggPyr().cuda().eval()
    ctx = dp.DPContext(0, 1, 0, torch.device("cuda", 0))
    out_dir = str(tmp_path / "results")
    written = INF.run_kitti_inference(det, img_net, root, names, out_dir, ctx, frames_per_batch=2)
    assert sorted(written) == names
    for n in names:
        path = os.path.join(out_dir, n + ".txt")
        assert os.path.exists(path)
        lines = [l.split() for l in open(path).read().splitlines() if l.strip()]
        assert len(lines) == written[n] <= 100
        for p in lines:
            assert len(p) == 16 and p[0] in INF.CLASSES and float(p[15]) >= 0.1
            x1, y1, x2, y2 = (float(v) for v in p[4:8])
            assert 0 <= x1 <= x2 <= 1242 and 0 <= y1 <= y2 <= 375
    # a second rank's shard is the complement of the first's
    assert dp.shard_frames(3, 0, 2) == [0, 2] and dp.shard_frames(3, 1, 2) == [1]
