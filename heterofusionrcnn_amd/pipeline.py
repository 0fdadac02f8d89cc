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
    """`depth` batches in flight, each on its own HIP stream (round robin), consumed in submission order.

    One FPS chain (16384 -> 4096 -> 1024 -> 256) is ~5.6 ms of strictly sequential rounds on 8 CUs; once the
    feature half of a step is shorter than that, a single side stream bounds the step.  With depth 2 the
    geometry of batches i+1 and i+2 overlap each other as well as the training of batch i (16 of 256 CUs)."""

    def __init__(self, geometry_fn, device=None, depth=1):
        assert depth >= 1
        self.fn = geometry_fn
        self.depth = depth
        self.streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
        self._pending = []
        self._turn = 0

    @property
    def stream(self):
        return self.streams[0]

    def __len__(self):
        return len(self._pending)

    def submit(self, xyz):
        """enqueue geometry_fn(xyz) on the next side stream (xyz must already be produced on the current stream)"""
        assert len(self._pending) < self.depth, "at most `depth` batches in flight"
        side = self.streams[self._turn % self.depth]
        self._turn += 1
        main = torch.cuda.current_stream()
        side.wait_stream(main)  # xyz ready
        with torch.cuda.stream(side):
            geo = self.fn(xyz)
            done = torch.cuda.Event()
            done.record(side)
        xyz.record_stream(side)
        self._pending.append((geo, done))

    def get(self):
        """geometry of the oldest submitted batch; the current stream waits for it (no host sync)"""
        geo, done = self._pending.pop(0)
        main = torch.cuda.current_stream()
        main.wait_event(done)
        _walk(geo, lambda t: t.record_stream(main))  # allocated on the side stream, consumed on main
        return geo
