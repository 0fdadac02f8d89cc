"""On-disk formats either side of the path, host side, numpy only.

  KITTI object files     velodyne/*.bin (N x 4 float32 [x, y, z, reflectance]), calib/*.txt (P0..P3 3x4, R0_rect 3x3,
                         Tr_velo_to_cam 3x4), label_2/*.txt (15 columns)      -- readers as hf/core/calib_utils.py:55-112,
                         327-360, obj_utils.read_labels
  rect-frame point cloud velodyne -> rectified camera frame, kept if in front of the camera and strictly inside the
                         image (hf/core/obj_utils.py:221-277, calib_utils.py:370-407, 280-296)
  16384-point sampling   every far point (depth >= 40 m) kept, near points drawn without replacement, short clouds
                         padded by re-drawing, shuffled, reflectance shifted to [-0.5, 0.5]
                         (hf/datasets/kitti/kitti_dataset.py:341-371)
  RPN -> RCNN hand-off   proposals_and_scores/*.txt (7 box columns + score, "%.3f"), rpn_feature/*.npy rows
                         [xyz(3), intensity, fg_mask, point features, image features]  (hf/core/evaluator.py:934-983)
  KITTI result txt       one detection per line in the label format + score

The reference draws with the global numpy RNG; here a numpy Generator is passed in, so the SAME rule gives a
reproducible sample: parity is the rule and its invariants, not the random stream.
"""
import os

import numpy as np

FAR_DEPTH = 40.0


def read_velodyne(path):
    """(N, 4) float32 [x, y, z, reflectance in 0..1]"""
    data = np.fromfile(path, dtype=np.float32)
    if data.size % 4:
        raise ValueError("%s: size is not a multiple of 4 floats" % path)
    return data.reshape(-1, 4)


def read_calib(path):
    """dict: p0..p3 (3,4), r0_rect (3,3), tr_velo_to_cam (3,4), float64 -- the six lines the reference reads"""
    rows = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if parts:
                rows[parts[0].rstrip(":")] = np.array([float(v) for v in parts[1:]], dtype=np.float64)
    try:
        out = {"p%d" % i: rows["P%d" % i].reshape(3, 4) for i in range(4)}
        out["r0_rect"] = rows["R0_rect"].reshape(3, 3)
        out["tr_velo_to_cam"] = rows["Tr_velo_to_cam"].reshape(3, 4)
    except KeyError as e:
        raise ValueError("%s: calibration line %s missing" % (path, e))
    return out


def lidar_to_rect(xyz_lidar, calib):
    """(N,3) velodyne frame -> (N,3) rectified camera frame: R0_rect (padded 4x4) . Tr_velo_to_cam (padded) . [x y z 1]"""
    r0 = np.eye(4)
    r0[:3, :3] = calib["r0_rect"]
    tr = np.eye(4)
    tr[:3, :4] = calib["tr_velo_to_cam"]
    hom = np.concatenate([np.asarray(xyz_lidar, dtype=np.float64), np.ones((len(xyz_lidar), 1))], axis=1)
    return (r0 @ tr @ hom.T)[:3].T


def project_to_image(pts_rect, p):
    """(N,3) rect points, p (3,4) -> (N,2) float pixel coordinates"""
    hom = np.concatenate([np.asarray(pts_rect, dtype=np.float64), np.ones((len(pts_rect), 1))], axis=1)
    uvw = hom @ np.asarray(p, dtype=np.float64).T
    return uvw[:, :2] / uvw[:, 2:3]


def load_point_cloud(velo_path, calib, image_shape):
    """rect-frame points of one frame that the camera sees: (pts_rect (N,3), intensity (N,1)); image_shape = (h, w)"""
    cloud = read_velodyne(velo_path)
    pts = lidar_to_rect(cloud[:, :3], calib)
    front = pts[:, 2] > 0
    pts, inten = pts[front], cloud[front, 3]
    uv = project_to_image(pts, calib["p2"])
    h, w = image_shape
    inside = (uv[:, 0] > 0) & (uv[:, 0] < w) & (uv[:, 1] > 0) & (uv[:, 1] < h)
    return pts[inside], inten[inside].reshape(-1, 1)


def sample_points(pts_rect, intensity, num_points=16384, rng=None):
    """The network's fixed-size input: (num_points, 4) float32 [x, y, z, intensity - 0.5] and the chosen indices."""
    rng = np.random.default_rng() if rng is None else rng
    n = len(pts_rect)
    if n == 0:
        raise ValueError("empty point cloud")
    if num_points < n:
        near = pts_rect[:, 2] < FAR_DEPTH
        far_idx, near_idx = np.where(~near)[0], np.where(near)[0]
        if len(far_idx) > num_points or num_points - len(far_idx) > len(near_idx):
            raise ValueError("cannot keep every far point and fill the rest from the near ones")  # np.random.choice raises too
        near_choice = rng.choice(near_idx, num_points - len(far_idx), replace=False)
        choice = np.concatenate([near_choice, far_idx]) if len(far_idx) else near_choice
    else:
        choice = np.arange(n)
        if num_points > n:
            extra = rng.choice(choice, num_points - n, replace=num_points > 2 * n)
            choice = np.concatenate([choice, extra])
    choice = rng.permutation(choice)
    out = np.concatenate([pts_rect[choice], np.asarray(intensity).reshape(n, 1)[choice] - 0.5], axis=1).astype(np.float32)
    return out, choice


LABEL_COLUMNS = ("type", "truncation", "occlusion", "alpha", "x1", "y1", "x2", "y2", "h", "w", "l", "tx", "ty", "tz", "ry")


def read_labels(path, classes=None):
    """label_2 file -> (types list, boxes_3d (N,7) [x, y, z, l, w, h, ry] as box_3d_encoder.object_label_to_box_3d,
    boxes_2d (N,4), misc (N,3) [truncation, occlusion, alpha]); `classes` filters by type name"""
    types, b3, b2, misc = [], [], [], []
    if os.path.exists(path):
        with open(path) as f:
            for line in f:
                p = line.split()
                if len(p) < 15 or (classes is not None and p[0] not in classes):
                    continue
                v = [float(x) for x in p[1:15]]
                types.append(p[0])
                misc.append(v[0:3])
                b2.append(v[3:7])
                h, w, l = v[7:10]
                b3.append([v[10], v[11], v[12], l, w, h, v[13]])
    return (types, np.asarray(b3, dtype=np.float64).reshape(-1, 7), np.asarray(b2, dtype=np.float64).reshape(-1, 4),
            np.asarray(misc, dtype=np.float64).reshape(-1, 3))


def save_proposals_and_scores(path, proposals, scores):
    """(N,7) boxes + (N,) scores -> text, 8 columns, "%.3f" (evaluator.py:934-961)"""
    arr = np.hstack([np.asarray(proposals, dtype=np.float64).reshape(-1, 7), np.asarray(scores, dtype=np.float64).reshape(-1, 1)])
    np.savetxt(path, arr, fmt="%.3f")


def load_proposals_and_scores(path):
    arr = np.loadtxt(path, ndmin=2) if os.path.getsize(path) else np.zeros((0, 8))
    return arr[:, :7], arr[:, 7]


def save_rpn_features(path, pts, intensity, fg_mask, pts_fts, img_fts=None):
    """rows [xyz(3), intensity, fg_mask, point features, image features] as one .npy (evaluator.py:963-983)"""
    n = len(pts)
    cols = [np.asarray(pts).reshape(n, 3), np.asarray(intensity).reshape(n, 1), np.asarray(fg_mask).reshape(n, 1),
            np.asarray(pts_fts).reshape(n, -1)]
    if img_fts is not None:
        cols.append(np.asarray(img_fts).reshape(n, -1))
    np.save(path, np.hstack(cols))


def load_rpn_features(path, num_pts_fts):
    """-> dict(pts, intensity, fg_mask, pts_fts, img_fts)"""
    a = np.load(path, allow_pickle=False)
    return {"pts": a[:, 0:3], "intensity": a[:, 3:4], "fg_mask": a[:, 4], "pts_fts": a[:, 5:5 + num_pts_fts],
            "img_fts": a[:, 5 + num_pts_fts:]}


def write_kitti_results(path, types, boxes_2d, boxes_3d, scores, alphas=None):
    """KITTI detection file: type -1 -1 alpha x1 y1 x2 y2 h w l x y z ry score, one line per detection"""
    boxes_3d = np.asarray(boxes_3d, dtype=np.float64).reshape(-1, 7)
    boxes_2d = np.asarray(boxes_2d, dtype=np.float64).reshape(-1, 4)
    with open(path, "w") as f:
        for i, t in enumerate(types):
            x, y, z, l, w, h, ry = boxes_3d[i]
            alpha = -10.0 if alphas is None else float(alphas[i])
            f.write("%s -1 -1 %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.4f\n" %
                    ((t, alpha) + tuple(boxes_2d[i]) + (h, w, l, x, y, z, ry, float(scores[i]))))
