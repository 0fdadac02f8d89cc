"""host logic of the geometry prefetch pipeline (no GPU): group size choice, per-batch slicing of a grouped result"""
import torch

from heterofusionrcnn_amd.pipeline import _slice_frames, _walk, choose_group


def test_choose_group_divides_the_run():
    for steps in range(1, 200):
        g = choose_group(steps)
        assert 1 <= g <= 16 and steps % g == 0
        assert all(steps % h for h in range(g + 1, 17))
    assert choose_group(10) == 10 and choose_group(16) == 16 and choose_group(7) == 7 and choose_group(17) == 1
    assert choose_group(20) == 10 and choose_group(50) == 10 and choose_group(100) == 10
    assert choose_group(0) == 1


def test_slice_frames_keeps_structure_and_shares_storage():
    geo = {"sa": [(torch.arange(24.).view(6, 4), torch.arange(6).view(6, 1), None)],
           "fp": [(torch.arange(12).view(6, 2), (torch.ones(6, 3), torch.zeros(6, 5)))]}
    part = _slice_frames(geo, 2, 4)
    assert set(part) == {"sa", "fp"} and isinstance(part["sa"][0], tuple) and part["sa"][0][2] is None
    assert torch.equal(part["sa"][0][0], geo["sa"][0][0][2:4]) and part["sa"][0][0].is_contiguous()
    assert part["sa"][0][0].data_ptr() == geo["sa"][0][0][2:4].data_ptr()  # a view, not a copy
    assert part["fp"][0][1][1].shape == (2, 5)
    seen = []
    _walk(part, lambda t: seen.append(tuple(t.shape)))
    assert seen == [(2, 4), (2, 1), (2, 2), (2, 3), (2, 5)]
