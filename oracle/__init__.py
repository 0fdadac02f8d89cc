"""oracle -- CPU checker for the hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (heterofusionrcnn_amd) never does.  It wraps

  * libhforacle.so           our C restatement (oracle/hf_oracle.c), numpy in / numpy out
  * _ref/libhfref_{qbp,sel,itp}.so   the reference's own standalone CPU programs (g++ build)
  * _ref/libhfref_gpu.so     the reference's CUDA kernels compiled unmodified by hipcc
                             (device pointers; only usable on a GPU box)

See oracle/README.md for how each function is pinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_c_float_p = ctypes.POINTER(ctypes.c_float)
_c_int_p = ctypes.POINTER(ctypes.c_int)


def build(quiet=True):
    """(Re)build libhforacle.so and, when /root/reference exists, oracle/_ref/."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libhforacle.so")
        if not os.path.exists(path):
            build()
        _lib = ctypes.CDLL(path)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------- sampling
def farthest_point_sample(npoint, xyz):
    xyz = _f(xyz)
    b, n, _ = xyz.shape
    out = np.zeros((b, npoint), np.int32)
    lib().hfo_farthest_point_sample(b, n, npoint, _p(xyz), _p(out))
    return out


def gather_point(inp, idx):
    inp, idx = _f(inp), _i(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.empty((b, m, 3), np.float32)
    lib().hfo_gather_point(b, n, m, _p(inp), _p(idx), _p(out))
    return out


def gather_point_grad(inp_shape, idx, out_g):
    idx, out_g = _i(idx), _f(out_g)
    b, n, _ = inp_shape
    m = idx.shape[1]
    g = np.empty((b, n, 3), np.float32)
    lib().hfo_gather_point_grad(b, n, m, _p(out_g), _p(idx), _p(g))
    return g


# ---------------------------------------------------------------- grouping
def query_ball_point(radius, nsample, xyz1, xyz2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    cnt = np.empty((b, m), np.int32)
    lib().hfo_query_ball_point(b, n, m, ctypes.c_float(radius), nsample, _p(xyz1), _p(xyz2), _p(idx), _p(cnt))
    return idx, cnt


def group_point(points, idx):
    points, idx = _f(points), _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    lib().hfo_group_point(b, n, c, m, ns, _p(points), _p(idx), _p(out))
    return out


def group_point_grad(points_shape, idx, grad_out):
    idx, grad_out = _i(idx), _f(grad_out)
    b, n, c = points_shape
    _, m, ns = idx.shape
    g = np.empty((b, n, c), np.float32)
    lib().hfo_group_point_grad(b, n, c, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def select_top_k(k, dist):
    dist = _f(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), np.int32)
    out = np.empty((b, m, n), np.float32)
    lib().hfo_select_top_k(b, n, m, k, _p(dist), _p(outi), _p(out))
    return outi, out


def knn_point(k, xyz1, xyz2):
    """tf_grouping.py:62-95 restated with (q - p)^2 per axis and ties to the lower index: PARITY UNPINNED against the
    reference's three-term expansion + tf.nn.top_k for exact ties and near-ties (no fixture, no TensorFlow here)."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    val = np.empty((b, m, k), np.float32)
    idx = np.empty((b, m, k), np.int32)
    lib().hfo_knn_point(b, n, m, k, _p(xyz1), _p(xyz2), _p(val), _p(idx))
    return val, idx


# ------------------------------------------------------------- interpolate
def three_nn(xyz1, xyz2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().hfo_three_nn(b, n, m, _p(xyz1), _p(xyz2), _p(dist), _p(idx))
    return dist, idx


def three_interpolate(points, idx, weight):
    """channel-last API of interpolate/tf_interpolate.py:26-37: points (b,m,c) -> (b,n,c)"""
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().hfo_three_interpolate_cl(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(points_shape, idx, weight, grad_out):
    idx, weight, grad_out = _i(idx), _f(weight), _f(grad_out)
    b, m, c = points_shape
    n = idx.shape[1]
    g = np.empty((b, m, c), np.float32)
    lib().hfo_three_interpolate_cl_grad(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


def three_interpolate_cf(points, idx, weight):
    """channel-first op layout (tf_interpolate.cpp:25-37): points (b,c,m) -> (b,c,n)"""
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), np.float32)
    lib().hfo_three_interpolate(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_cf_grad(points_shape, idx, weight, grad_out):
    idx, weight, grad_out = _i(idx), _f(weight), _f(grad_out)
    b, c, m = points_shape
    n = idx.shape[1]
    g = np.empty((b, c, m), np.float32)
    lib().hfo_three_interpolate_grad(b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


# ----------------------------------------------------------------- bev_iou
def compute_bev_iou(a, b):
    a, b = _f(a), _f(b)
    na, nb = a.shape[0], b.shape[0]
    ov = np.empty((na, nb), np.float32)
    iou = np.empty((na, nb), np.float32)
    lib().hfo_compute_bev_iou(na, _p(a), nb, _p(b), _p(ov), _p(iou))
    return ov, iou


def nms_mask(boxes, thresh):
    boxes = _f(boxes)
    n = boxes.shape[0]
    mask = np.empty((n, (n + 63) // 64), np.uint64)
    lib().hfo_nms_mask(_p(boxes), _p(mask), n, ctypes.c_float(thresh))
    return mask


def nms_sweep(mask):
    mask = np.ascontiguousarray(mask, dtype=np.uint64)
    n = mask.shape[0]
    keep = np.empty((n,), np.int32)
    kept = lib().hfo_nms_sweep(_p(mask), n, _p(keep))
    return keep, kept


def oriented_nms(boxes, thresh, return_count=False):
    boxes = _f(boxes)
    n = boxes.shape[0]
    keep = np.empty((n,), np.int32)
    kept = lib().hfo_oriented_nms(_p(boxes), n, ctypes.c_float(thresh), _p(keep))
    return (keep, kept) if return_count else keep


# ---------------------------------------------------------------- cropping
def pc_crop_and_sample(pts, fts, intensities, mask, boxes, box_ind, resize):
    pts, fts, intensities, boxes = _f(pts), _f(fts), _f(intensities), _f(boxes)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    box_ind = _i(box_ind)
    bsz, npts, _ = pts.shape
    c, ic = fts.shape[2], intensities.shape[2]
    nb = boxes.shape[0]
    cp = np.empty((nb, resize, 3), np.float32)
    cf = np.empty((nb, resize, c), np.float32)
    ci = np.empty((nb, resize, ic), np.float32)
    cm = np.empty((nb, resize), np.uint8)
    cn = np.empty((nb, resize), np.int32)
    ne = np.empty((nb,), np.uint8)
    lib().hfo_pc_crop_and_sample(_p(pts), _p(fts), _p(intensities), _p(mask), _p(boxes), _p(box_ind), nb, bsz,
                                 npts, resize, c, ic, _p(cp), _p(cf), _p(ci), _p(cm), _p(cn), _p(ne))
    return cp, cf, ci, cm.astype(bool), cn, ne.astype(bool)


def pc_crop_and_sample_grad_fts(fts_shape, box_ind, crop_ind, grad_crop_fts):
    box_ind, crop_ind, grad_crop_fts = _i(box_ind), _i(crop_ind), _f(grad_crop_fts)
    bsz, npts, c = fts_shape
    nb, resize = crop_ind.shape
    g = np.empty((bsz, npts, c), np.float32)
    lib().hfo_pc_crop_and_sample_grad_fts(_p(box_ind), _p(crop_ind), _p(grad_crop_fts), nb, bsz, npts, resize, c,
                                          _p(g))
    return g


# ------------------------------------------------- reference builds (_ref/)
def ref_available(kind):
    """kind in {'qbp','sel','itp','gpu'}"""
    return os.path.exists(os.path.join(_HERE, "_ref", "libhfref_%s.so" % kind))


_ref_libs = {}


def ref_lib(kind):
    if kind not in _ref_libs:
        _ref_libs[kind] = ctypes.CDLL(os.path.join(_HERE, "_ref", "libhfref_%s.so" % kind))
    return _ref_libs[kind]


# mangled names of the reference's standalone CPU functions (its C++ files have no extern "C"):
#   grouping/test/query_ball_point.cpp:19,52,70   grouping/test/selection_sort.cpp   interpolate/interpolate.cpp:84,109
_REF_QBP = "_Z20query_ball_point_cpuiiifiPKfS0_Pi"
_REF_GROUP = "_Z15group_point_cpuiiiiiPKfPKiPf"
_REF_GROUP_GRAD = "_Z20group_point_grad_cpuiiiiiPKfPKiPf"
_REF_SELSORT = "_Z18selection_sort_cpuiiiiPKfPiPf"
_REF_THREENN = "_Z11threenn_cpuiiiPKfS0_PfPi"           # interpolate/interpolate.cpp:21
_REF_ITP = "_Z15interpolate_cpuiiiiPKfPKiS0_Pf"
_REF_ITP_GRAD = "_Z20interpolate_grad_cpuiiiiPKfPKiS0_Pf"


def ref_query_ball_point(radius, nsample, xyz1, xyz2):
    """the reference's query_ball_point_cpu; idx pre-zeroed like its own main() does (:94)"""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    getattr(ref_lib("qbp"), _REF_QBP)(b, n, m, ctypes.c_float(radius), nsample, _p(xyz1), _p(xyz2), _p(idx))
    return idx


def ref_group_point(points, idx):
    points, idx = _f(points), _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    getattr(ref_lib("qbp"), _REF_GROUP)(b, n, c, m, ns, _p(points), _p(idx), _p(out))
    return out


def ref_group_point_grad(points_shape, idx, grad_out):
    idx, grad_out = _i(idx), _f(grad_out)
    b, n, c = points_shape
    _, m, ns = idx.shape
    g = np.zeros((b, n, c), np.float32)
    getattr(ref_lib("qbp"), _REF_GROUP_GRAD)(b, n, c, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def ref_select_top_k(k, dist):
    dist = _f(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), np.int32)
    out = np.empty((b, m, n), np.float32)
    getattr(ref_lib("sel"), _REF_SELSORT)(b, n, m, k, _p(dist), _p(outi), _p(out))
    return outi, out


def ref_threenn_origin(known, n=1):
    """the reference's threenn_cpu (interpolate/interpolate.cpp:21-64) on known (b,m,3) -> (dist (b,n,3), idx (b,n,3)).
    That function ranks x2*x2 + y2*y2 + z2*z2 -- the squared distance to the ORIGIN, whatever xyz1 holds (its subtraction is
    commented out, :34-35) -- so it is no general three_nn, but for an unknown point at the origin it is the same fp32
    expression as tf_interpolate_g.cu:47 ((0-x)*(0-x) == x*x exactly) walked by the same double-1e40 strict-'<' cascade:
    it pins the cascade, the tie order, the sentinels and the m < 3 case to reference-compiled code.  Every one of the n
    rows of a batch element is that one answer."""
    known = _f(known)
    b, m, _ = known.shape
    xyz1 = np.zeros((b, n, 3), np.float32)
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    getattr(ref_lib("itp"), _REF_THREENN)(b, n, m, _p(xyz1), _p(known), _p(dist), _p(idx))
    return dist, idx


def ref_three_interpolate(points, idx, weight):
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    getattr(ref_lib("itp"), _REF_ITP)(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def ref_three_interpolate_grad(points_shape, idx, weight, grad_out):
    idx, weight, grad_out = _i(idx), _f(weight), _f(grad_out)
    b, m, c = points_shape
    n = idx.shape[1]
    g = np.zeros((b, m, c), np.float32)
    getattr(ref_lib("itp"), _REF_ITP_GRAD)(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


# ---- fusion gather and the bin-based box codec ----
def project_gather(pts, calib, img):
    pts, calib, img = _f(pts), _f(calib), _f(img)
    b, p, _ = pts.shape
    _, h, w, c = img.shape
    out = np.empty((b, p, c), np.float32)
    pix = np.empty((b, p, 2), np.int32)
    lib().hfo_project_gather(b, p, h, w, c, _p(pts), _p(calib), _p(img), _p(out), _p(pix))
    return out, pix


def project_gather_grad(img_shape, pix, grad_out):
    pix, grad_out = _i(pix), _f(grad_out)
    b, h, w, c = img_shape
    g = np.empty((b, h, w, c), np.float32)
    lib().hfo_project_gather_grad(b, pix.shape[1], h, w, c, _p(grad_out), _p(pix), _p(g))
    return g


def bin_box_decode(ref_pts, ref_theta, bin_x, res_x_norm, bin_z, res_z_norm, bin_theta, res_theta_norm, res_y,
                   res_size_norm, mean_sizes, ss, deltas, r, delta_theta):
    """flattened form: ref_pts (rows,3), per (row, class) arrays (rows,k[,3]); ref_theta (rows,) or None"""
    rows, k = bin_x.shape
    boxes = np.empty((rows, k, 7), np.float32)
    th = _f(ref_theta) if ref_theta is not None else None
    lib().hfo_bin_box_decode(ctypes.c_longlong(rows), k, _p(_f(ref_pts)), _p(th) if th is not None else None, _p(_i(bin_x)),
                             _p(_f(res_x_norm)), _p(_i(bin_z)), _p(_f(res_z_norm)), _p(_i(bin_theta)),
                             _p(_f(res_theta_norm)), _p(_f(res_y)), _p(_f(res_size_norm)), _p(_f(mean_sizes)),
                             _p(_f(np.asarray(ss, np.float64).astype(np.float32))),
                             _p(_f(np.asarray(deltas, np.float64).astype(np.float32))), ctypes.c_float(np.float32(r)),
                             ctypes.c_float(np.float32(delta_theta)), _p(boxes))
    return boxes


def bin_box_encode(ref_pts, ref_theta, boxes, mean_sizes, ss, deltas, r, delta_theta, k, rcnn):
    rows = ref_pts.shape[0]
    ss64, d64 = np.asarray(ss, np.float64), np.asarray(deltas, np.float64)
    outs = [np.empty((rows, k), np.int32), np.empty((rows, k), np.float32), np.empty((rows, k), np.int32),
            np.empty((rows, k), np.float32), np.empty((rows,), np.int32), np.empty((rows,), np.float32),
            np.empty((rows,), np.float32), np.empty((rows, 3), np.float32)]
    th = _f(ref_theta) if ref_theta is not None else None
    lib().hfo_bin_box_encode(ctypes.c_longlong(rows), k, int(rcnn), _p(_f(ref_pts)), _p(th) if th is not None else None,
                             _p(_f(boxes)), _p(_f(mean_sizes)), _p(_f(ss64.astype(np.float32))),
                             _p(_f(d64.astype(np.float32))), _p(_f((2.0 * ss64 - 1e-3).astype(np.float32))),
                             ctypes.c_float(np.float32(r)), ctypes.c_float(np.float32(2.0 * float(r) - 1e-3)),
                             ctypes.c_float(np.float32(delta_theta)), ctypes.c_float(np.float32(0.5 * float(delta_theta))),
                             *[_p(o) for o in outs])
    return tuple(outs)


def bin_head_decode(head, ref_pts, ref_theta, mean_sizes_k, nbx, nbz, nbt, ss, deltas, r, delta_theta, cls=None):
    """head (rows, k, d); -> boxes (rows, k, 7) or (rows, 7) with cls"""
    rows, k, _ = head.shape
    boxes = np.zeros((rows, 7), np.float32) if cls is not None else np.empty((rows, k, 7), np.float32)
    th = _f(ref_theta) if ref_theta is not None else None
    c = _i(cls) if cls is not None else None
    lib().hfo_bin_head_decode(ctypes.c_longlong(rows), k, nbx, nbz, nbt, _p(_f(head)), _p(_f(ref_pts)),
                              _p(th) if th is not None else None, _p(_f(mean_sizes_k)),
                              _p(_f(np.asarray(ss, np.float64).astype(np.float32))),
                              _p(_f(np.asarray(deltas, np.float64).astype(np.float32))), ctypes.c_float(np.float32(r)),
                              ctypes.c_float(np.float32(delta_theta)), _p(c) if c is not None else None, _p(boxes))
    return boxes


# ---- the per-point products of PointCNN's X-Conv (numpy, fp32 accumulation in index order) ----
def xconv_apply(x, f):
    """hf/core/feature_extractors/pointcnn.py:133  fts_X = tf.matmul(X, nn_fts_input): x (rows,K,K), f (rows,K,C) ->
    (rows,K,C), out[r,i,:] = sum_j x[r,i,j] * f[r,j,:], the j in ascending order (the kernel's order)"""
    x, f = _f(x), _f(f)
    out = np.zeros_like(f)
    for j in range(x.shape[2]):
        out = out + x[:, :, j:j + 1] * f[:, j:j + 1, :] if j else x[:, :, 0:1] * f[:, 0:1, :]
    return out


def depthwise_k(x, w):
    """hf/core/pointfly.py:437-457 depthwise_conv2d(.., (1,K)) on a width-K input (TensorFlow filter layout (1,K,C,M)):
    x (rows,K,C), w (K,C,M) -> (rows, C*M), y[r, c*M + m] = sum_k x[r,k,c] * w[k,c,m], k ascending"""
    x, w = _f(x), _f(w)
    rows, k, c = x.shape
    m = w.shape[2]
    y = np.zeros((rows, c, m), np.float32)
    for kk in range(k):
        y = y + x[:, kk, :, None] * w[kk][None]
    return y.reshape(rows, c * m)


def knn_point_three_term(k, xyz1, xyz2):
    """grouping/tf_grouping.py:80-92 as written: dist = |q|^2 - 2 q.p^T + |p|^2 in fp32 (may go slightly negative),
    then tf.nn.top_k(-dist, k) (equal values: the lower index first).  Used by the tests to assert that the (q - p)^2
    kernel returns the same neighbour sets wherever the k-th gap exceeds the rounding of this expansion; the matmul's
    fp32 summation order inside TensorFlow is not reproducible here, so near-ties stay parity unpinned."""
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    r1 = np.sum(xyz1 * xyz1, axis=2, keepdims=True, dtype=np.float32)                  # (b, n, 1)
    r2 = np.sum(xyz2 * xyz2, axis=2, keepdims=True, dtype=np.float32)                  # (b, m, 1)
    mul = np.matmul(xyz2, np.transpose(xyz1, (0, 2, 1))).astype(np.float32)             # (b, m, n)
    dist = (r2 - np.float32(2) * mul + np.transpose(r1, (0, 2, 1))).astype(np.float32)
    idx = np.argsort(dist, axis=2, kind="stable")[:, :, :k].astype(np.int32)            # stable: lower index first on ties
    return np.take_along_axis(dist, idx, axis=2), idx
