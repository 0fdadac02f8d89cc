"""Adam for the train step: every parameter tensor in ONE launch (csrc/optim.hip, hf_adam_multi).

The reference's step ends in tf.train.AdamOptimizer.apply_gradients (hf/builders/optimizer_builder.py:59-64, wrapped by
hvd.DistributedOptimizer in hf/core/trainer.py:71).  torch.optim.Adam(fused=True) needs seven launches for this model's 260
tensors (the tensor list travels in kernel arguments) and moves 1.2 TB/s; here the list is a table in device memory and a
workgroup looks its chunk up.  The gradients are read where autograd left them: when their addresses change (eager steps
allocate fresh gradients; a captured graph does not) the table is rewritten and uploaded, otherwise a step is two launches
(the step counter, the update).

`tf_epsilon=True` (default) is TensorFlow's update  p -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps); False is
torch.optim.Adam's placement of epsilon (tests/test_optim.py pins the arithmetic of that mode to torch's own optimizer).
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class _Entry(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("numel", ctypes.c_longlong)]


class MultiTensorAdam:
    """step() semantics of torch.optim.Adam(params, lr, betas, eps) without weight decay / amsgrad.  Parameters whose .grad is
    None at a step are skipped for that step (their moments stay), as the framework optimizers do."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, tf_epsilon=True, grad_scale=1.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("MultiTensorAdam: no parameter requires a gradient")
        dev = self.params[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 or not p.is_contiguous() for p in self.params):
            raise RuntimeError("MultiTensorAdam: heterofusionrcnn_amd has no CPU implementation (contiguous fp32 parameters on one GPU)")
        self.lr, self.betas, self.eps, self.tf_epsilon, self.grad_scale = float(lr), betas, float(eps), tf_epsilon, float(grad_scale)
        total = sum(p.numel() for p in self.params)
        # both moments of every tensor in two flat buffers, 16-byte aligned slices
        offs, o = [], 0
        for p in self.params:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4
        self._exp_avg = torch.zeros(o, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(o, dtype=torch.float32, device=dev)
        self.exp_avg = [self._exp_avg[a:a + p.numel()] for a, p in zip(offs, self.params)]
        self.exp_avg_sq = [self._exp_avg_sq[a:a + p.numel()] for a, p in zip(offs, self.params)]
        self.step_count = torch.zeros((), dtype=torch.float32, device=dev)     # on the device: a captured step advances it
        self.chunk = _lib.lib().hf_adam_chunk()
        n = len(self.params)
        self._table_bytes = n * ctypes.sizeof(_Entry)
        self._map_rows = sum((p.numel() + self.chunk - 1) // self.chunk for p in self.params)
        # pinned staging for the table upload.  The upload is asynchronous and the host may be several steps ahead of the device,
        # so a rewrite never touches a buffer whose upload may still be pending: eager rewrites rotate through a ring (an event
        # per slot, waited on before the slot is reused); a capture takes a slot of its own for good (the graph's upload nodes
        # re-read it at every replay).  All pinned allocations happen outside captures (hipHostMalloc is not capturable).
        self._ring = [self._new_staging() for _ in range(4)]
        self._ring_at = 0
        self._for_captures = [self._new_staging() for _ in range(2)]
        self._captured_staging = []
        self._dev_table = torch.empty(self._table_bytes, dtype=torch.uint8, device=dev)
        self._dev_map = torch.empty((self._map_rows, 2), dtype=torch.int32, device=dev)
        self._key, self._chunks = None, 0
        self.total_elements = total

    def _new_staging(self):
        return [torch.empty(self._table_bytes, dtype=torch.uint8).pin_memory(),
                torch.empty((self._map_rows, 2), dtype=torch.int32).pin_memory(), None]

    # the framework optimizers' surface that callers here use
    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _refresh(self):
        """rewrite + upload the table when a gradient moved (or appeared / vanished)"""
        key = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in self.params)
        if key == self._key:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            if not self._for_captures:
                raise RuntimeError("MultiTensorAdam: no pinned staging left for another capture (run one eager step between captures)")
            slot = self._for_captures.pop()
            self._captured_staging.append(slot)
        else:
            while len(self._for_captures) < 2:
                self._for_captures.append(self._new_staging())
            slot = self._ring[self._ring_at]
            self._ring_at = (self._ring_at + 1) % len(self._ring)
            if slot[2] is not None:
                slot[2].synchronize()           # the upload that last read this slot has executed (four rewrites ago: rarely waits)
        host_table, host_map = slot[0], slot[1]
        entries = (_Entry * len(self.params)).from_buffer(host_table.numpy())
        rows, live = [], 0
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                raise RuntimeError("MultiTensorAdam: gradients must be contiguous fp32 on the parameters' device")
            entries[live] = _Entry(p.data_ptr(), g.data_ptr(), self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr(), p.numel())
            rows += [(live, c) for c in range((p.numel() + self.chunk - 1) // self.chunk)]
            live += 1
        self._chunks = len(rows)
        if rows:
            host_map[:len(rows)] = torch.tensor(rows, dtype=torch.int32)
        # pinned staging -> device on the current stream (inside a capture this becomes a node of the graph: it re-uploads the
        # same bytes at every replay, 12 KB)
        self._dev_table.copy_(host_table, non_blocking=True)
        self._dev_map.copy_(host_map, non_blocking=True)
        if not capturing:
            slot[2] = torch.cuda.Event()
            slot[2].record()
        self._key = key

    @torch.no_grad()
    def step(self):
        self._refresh()
        self.step_count.add_(1.0)
        check(_lib.lib().hf_adam_multi(self._chunks, ptr(self._dev_table), ptr(self._dev_map), ptr(self.step_count), self.lr,
                                       self.betas[0], self.betas[1], self.eps, self.grad_scale, 0 if self.tf_epsilon else 1,
                                       stream_ptr()), "adam_multi")

    # ---- state hand-over (graph_step.TrainStep undoes its warm-up steps with these) ----
    def snapshot(self):
        return (self._exp_avg.clone(), self._exp_avg_sq.clone(), self.step_count.clone())

    def restore(self, snap):
        self._exp_avg.copy_(snap[0])
        self._exp_avg_sq.copy_(snap[1])
        self.step_count.copy_(snap[2])
