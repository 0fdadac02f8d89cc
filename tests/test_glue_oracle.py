"""CPU checks of the oracle's restatement of the glue steps: hand-derived cases and
size-independent properties.  The reference's Python (TensorFlow) cannot be imported here and its tests hold no
fixtures for these functions: parity unpinned, the restatement follows hf/core/projection.py:5-32,
hf/core/models/rpn_model.py:227-235 and hf/core/bin_based_box3d_encoder.py:9-269 as text."""
import numpy as np
import pytest

import oracle


def test_project_gather_known_pixels():
    # P2 = [[2,0,10,0],[0,3,20,0],[0,0,1,0]]: u = 2x/z + 10, v = 3y/z + 20
    calib = np.array([[[2, 0, 10, 0], [0, 3, 20, 0], [0, 0, 1, 0]]], np.float32)
    pts = np.array([[[1, 1, 1], [4, -2, 2], [-11, 0, 2], [0, 0, 1], [100, 0, 1], [0.2, 0, 1]]], np.float32)
    img = np.arange(1 * 30 * 40 * 2, dtype=np.float32).reshape(1, 30, 40, 2)
    out, pix = oracle.project_gather(pts, calib, img)
    assert pix[0].tolist() == [[12, 23], [14, 17], [-1, 20], [10, 20], [210, 20], [10, 20]]  # 10.4 -> 10 (truncation)
    assert np.array_equal(out[0, 0], img[0, 23, 12]) and np.array_equal(out[0, 1], img[0, 17, 14])
    assert not out[0, 2].any() and not out[0, 4].any()          # outside the image: zeros
    # truncation toward zero: -0.5 -> 0, which is INSIDE the image (tf.cast semantics)
    out2, pix2 = oracle.project_gather(np.array([[[-5.25, 0, 1]]], np.float32), calib, img)
    assert pix2[0, 0].tolist() == [0, 20] and np.array_equal(out2[0, 0], img[0, 20, 0])
    # gradient: rows land on their pixels, points sharing a pixel add up
    g = oracle.project_gather_grad(img.shape, pix, np.ones((1, 6, 2), np.float32))
    assert g[0, 20, 10].tolist() == [2, 2] and g[0, 23, 12].tolist() == [1, 1] and g.sum() == 8


SS, DELTAS = [3.0, 1.5], [0.5, 0.25]
R, NBT = 0.25 * np.pi, 12
DT = 2 * R / NBT


def _boxes(rng, rows, ref_pts, spread):
    b = np.empty((rows, 7), np.float32)
    b[:, :3] = ref_pts + rng.uniform(-spread, spread, (rows, 3))
    b[:, 3:6] = rng.uniform(0.5, 4.0, (rows, 3))
    b[:, 6] = rng.uniform(-0.9 * R, 0.9 * R, rows)
    return b


@pytest.mark.parametrize("with_theta", [False, True])
def test_bin_codec_round_trip_rpn(with_theta):
    rng = np.random.default_rng(5)
    rows, k = 500, 2
    ref = rng.uniform(-30, 30, (rows, 3)).astype(np.float32)
    th = rng.uniform(-3, 3, rows).astype(np.float32) if with_theta else None
    boxes = _boxes(rng, rows, ref, 1.0)            # |(dx, dz)| <= 1.42: inside the smaller class's +-1.5 range at any heading
    if with_theta:
        boxes[:, 6] += th
    mean = rng.uniform(1, 3, (rows, 3)).astype(np.float32)
    bx, rx, bz, rz, bt, rt, ry, rs = oracle.bin_box_encode(ref, th, boxes, mean, SS, DELTAS, R, DT, k, rcnn=0)
    for j in range(k):
        assert bx[:, j].min() >= 0 and bx[:, j].max() < int(2 * SS[j] / DELTAS[j])
    assert np.abs(rx).max() <= 0.5 + 1e-5 and np.abs(rt).max() <= 1 + 1e-5 and bt.min() >= 0 and bt.max() < NBT
    dec = oracle.bin_box_decode(ref, th, bx, rx, bz, rz, np.repeat(bt[:, None], k, 1), np.repeat(rt[:, None], k, 1),
                                np.repeat(ry[:, None], k, 1), np.repeat(rs[:, None], k, 1),
                                np.repeat(mean[:, None], k, 1), SS, DELTAS, R, DT)
    for j in range(k):  # every class decodes back to the encoded box
        np.testing.assert_allclose(dec[:, j], boxes, rtol=0, atol=2e-4)


def test_bin_codec_matches_float64_formulas():
    rng = np.random.default_rng(6)
    rows, k = 300, 2
    ref = rng.uniform(-30, 30, (rows, 3)).astype(np.float32)
    boxes = _boxes(rng, rows, ref, 5.0)            # some outside the search range: clipped into the last bins
    mean = rng.uniform(1, 3, (rows, 3)).astype(np.float32)
    bx, rx, bz, rz, bt, rt, ry, rs = oracle.bin_box_encode(ref, None, boxes, mean, SS, DELTAS, R, DT, k, rcnn=0)
    d = boxes.astype(np.float64)
    for j in range(k):
        xs = np.clip(d[:, 0] - ref[:, 0] + SS[j], 0, 2 * SS[j] - 1e-3)
        want = np.floor(xs / DELTAS[j])
        near = np.abs(xs / DELTAS[j] - np.round(xs / DELTAS[j])) < 1e-4     # fp32 may fall on the other side of an edge
        assert np.array_equal(bx[~near, j], want[~near].astype(np.int32))
        np.testing.assert_allclose(rx[~near, j], ((xs - (want + 0.5) * DELTAS[j]) / DELTAS[j])[~near], atol=2e-5)
    ts = np.clip(d[:, 6] + R, 0, 2 * R - 1e-3)
    np.testing.assert_array_equal(bt, np.floor(ts / DT).astype(np.int32))
    np.testing.assert_allclose(ry, d[:, 1] - ref[:, 1], atol=1e-5)
    np.testing.assert_allclose(rs, (d[:, 3:6] - mean) / mean, atol=1e-5)


def test_bin_codec_rcnn_orientation_folding():
    # the RCNN rule folds the heading difference into (-pi/2, pi/2] around the proposal's heading (flipped boxes
    # encode like unflipped ones), shifts by pi/2 and clips to [1e-3, 2R - 1e-3] after subtracting R
    ref = np.zeros((4, 3), np.float32)
    th = np.array([0.0, 0.0, 1.0, 7.0], np.float32)
    boxes = np.zeros((4, 7), np.float32); boxes[:, 3:6] = 1
    boxes[:, 6] = [0.1, 0.1 + np.pi, 1.0 - 0.2, 7.0 + 0.3 - 2 * np.pi]
    Rr = 0.5 * np.pi
    dt = 2 * Rr / 12
    out = oracle.bin_box_encode(ref, th, boxes, np.ones((4, 3), np.float32), [3.0], [0.5], Rr, dt, 1, rcnn=1)
    bt, rt = out[4], out[5]
    shift = bt * dt + dt / 2 + rt * dt / 2            # decoded dtheta_shift
    folded = np.array([0.1, 0.1, -0.2, 0.3])          # heading differences after folding
    np.testing.assert_allclose(shift, np.clip(folded + 0.5 * np.pi - Rr, 1e-3, 2 * Rr - 1e-3), atol=2e-4)


def test_bin_head_decode_is_parse_argmax_gather_decode():
    """the fused head decoding equals the reference's op-by-op route: slice, argmax (first maximum), take the winning
    bin's residual, tile the class mean sizes, tf_decode, pick the row's class"""
    rng = np.random.default_rng(9)
    rows, k, nbx, nbz, nbt = 400, 2, 12, 12, 9
    d = 2 * nbx + 2 * nbz + 2 * nbt + 4
    head = rng.standard_normal((rows, k, d)).astype(np.float32)
    head[:50, :, 3] = head[:50, :, 7] = 9.0          # tied maxima: the first one wins
    ref = rng.uniform(-30, 30, (rows, 3)).astype(np.float32)
    th = rng.uniform(-3, 3, rows).astype(np.float32)
    ms = np.array([[3.9, 1.6, 1.5], [0.8, 0.6, 1.7]], np.float32)
    dt = 2 * R / nbt
    parts = np.split(head, np.cumsum([nbx, nbx, nbz, nbz, nbt, nbt, 1]), axis=-1)
    bx, bz, bt = parts[0].argmax(-1), parts[2].argmax(-1), parts[4].argmax(-1)
    assert (bx[:50] == 3).all()
    take = lambda a, i: np.take_along_axis(a, i[..., None], -1)[..., 0]
    want = oracle.bin_box_decode(ref, th, bx, take(parts[1], bx), bz, take(parts[3], bz), bt, take(parts[5], bt),
                                 parts[6][..., 0], parts[7], np.broadcast_to(ms, (rows, k, 3)), SS, DELTAS, R, dt)
    got = oracle.bin_head_decode(head, ref, th, ms, nbx, nbz, nbt, SS, DELTAS, R, dt)
    assert np.array_equal(got, want)
    cls = rng.integers(0, k, rows).astype(np.int32)
    one = oracle.bin_head_decode(head, ref, th, ms, nbx, nbz, nbt, SS, DELTAS, R, dt, cls=cls)
    assert np.array_equal(one, want[np.arange(rows), cls])
    assert np.array_equal(oracle.bin_head_decode(head, ref, None, ms, nbx, nbz, nbt, SS, DELTAS, R, dt),
                          oracle.bin_box_decode(ref, None, bx, take(parts[1], bx), bz, take(parts[3], bz), bt,
                                                take(parts[5], bt), parts[6][..., 0], parts[7],
                                                np.broadcast_to(ms, (rows, k, 3)), SS, DELTAS, R, dt))
