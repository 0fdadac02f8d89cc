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


def choose_group(steps, largest=16):
    """batches per geometry launch for a run of `steps` timed steps: the largest divisor of `steps` up to `largest`,
    so that the run launches the geometry of exactly as many batches as it consumes"""
    steps = int(steps)
    return max(g for g in range(1, largest + 1) if steps % g == 0) if steps > 0 else 1


def _slice_frames(obj, lo, hi):
    """frames [lo, hi) of every tensor in a nested geometry result (all of them are batch-major)"""
    if isinstance(obj, torch.Tensor):
        return obj[lo:hi]
    if isinstance(obj, dict):
        return {k: _slice_frames(v, lo, hi) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_slice_frames(v, lo, hi) for v in obj)
    return obj


class GeometryPrefetcher:
    """`depth` launches in flight, each on its own HIP stream (round robin) and each covering `group` batches,
    consumed in submission order.

    One FPS chain (16384 -> 4096 -> 1024 -> 256) is ~5.6 ms of strictly sequential rounds on one CU per cloud;
    once the feature half of a step is shorter than that, a single side stream bounds the step, hence depth 2.
    `group` > 1 samples the clouds of several batches in ONE launch (the kernel takes the same time for 8 or 32
    clouds, it only uses more CUs): the sampling kernel is then resident for a fraction of the time, and the
    kernels of the feature half that are sized to one workgroup per CU (the library GEMMs) run their tail round
    less often (scripts/contention_step_probe.py: +0.85 ms of GEMM time per step while an FPS kernel is resident)."""

    def __init__(self, geometry_fn, device=None, depth=1, group=1):
        assert depth >= 1 and group >= 1
        self.fn = geometry_fn
        self.depth, self.group = depth, group
        self.streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
        self._staged = []    # batches waiting for their group to fill
        self._pending = []   # (geometry of one batch, event of its launch)
        self._turn = 0

    @property
    def stream(self):
        return self.streams[0]

    def __len__(self):
        return len(self._pending) + len(self._staged)

    @property
    def capacity(self):
        return self.depth * self.group

    @property
    def staged(self):
        """batches submitted but not yet launched (their group is not full)"""
        return len(self._staged)

    def _launch(self):
        batches, self._staged = self._staged, []
        side = self.streams[self._turn % self.depth]
        self._turn += 1
        main = torch.cuda.current_stream()
        side.wait_stream(main)  # every staged xyz is ready
        with torch.cuda.stream(side):
            xyz = batches[0] if len(batches) == 1 else torch.cat(batches, dim=0)
            geo = self.fn(xyz)
            done = torch.cuda.Event()
            done.record(side)
        for t in batches:
            t.record_stream(side)
        lo = 0
        for t in batches:
            hi = lo + t.shape[0]
            self._pending.append((geo if len(batches) == 1 else _slice_frames(geo, lo, hi), done))
            lo = hi

    def submit(self, xyz):
        """stage one batch (xyz must already be produced on the current stream); a full group is launched at once"""
        assert len(self) < self.capacity, "at most depth * group batches in flight"
        self._staged.append(xyz)
        if len(self._staged) == self.group:
            self._launch()

    def get(self):
        """geometry of the oldest submitted batch; the current stream waits for it (no host sync)"""
        if not self._pending:
            assert self._staged, "nothing submitted"
            self._launch()  # a partial group at the end of a run
        geo, done = self._pending.pop(0)
        main = torch.cuda.current_stream()
        main.wait_event(done)
        _walk(geo, lambda t: t.record_stream(main))  # allocated on the side stream, consumed on main
        return geo
