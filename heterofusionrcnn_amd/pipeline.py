"""Run the coordinate-only half of a pass one step ahead, on its own HIP stream.

FPS is m-1 strictly sequential rounds on ONE compute unit per cloud: at B = 8 it keeps 8 of the
MI355X's 256 CUs busy for milliseconds.  Everything the SA/FP stack derives from coordinates
(FPS, gather, ball query + grouping of xyz, three_nn and its weights) is independent of features
and of the weights, exactly like the host-side sampling the reference does in its data loader
(hf/datasets/kitti/kitti_dataset.py:341-371).  GeometryPrefetcher therefore computes it for the
NEXT batch on a side stream while the current batch runs its MLP GEMMs / BatchNorm / backward on
the main stream.  Every step still does every piece of work; only the order on the device changes.
"""
import torch


def _walk(obj, fn):
    if isinstance(obj, torch.Tensor):
        fn(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _walk(v, fn)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _walk(v, fn)


class GeometryPrefetcher:
    def __init__(self, geometry_fn, device=None):
        self.fn = geometry_fn
        self.stream = torch.cuda.Stream(device=device)
        self._pending = None

    def submit(self, xyz):
        """enqueue geometry_fn(xyz) on the side stream (xyz must already be produced on the current stream)"""
        assert self._pending is None, "one batch in flight at a time"
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)  # xyz ready
        with torch.cuda.stream(self.stream):
            geo = self.fn(xyz)
            done = torch.cuda.Event()
            done.record(self.stream)
        xyz.record_stream(self.stream)
        self._pending = (geo, done)

    def get(self):
        """geometry of the submitted batch; the current stream waits for it (no host sync)"""
        geo, done = self._pending
        self._pending = None
        main = torch.cuda.current_stream()
        main.wait_event(done)
        _walk(geo, lambda t: t.record_stream(main))  # allocated on the side stream, consumed on main
        return geo
