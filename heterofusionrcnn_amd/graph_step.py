"""The RPN train step as ONE HIP graph launch (forward + losses + backward [+ Adam]).

The step of hf/core/trainer.py:71-73,150-190 (session.run of train_op: forward, losses, gradients, the Horovod-averaged
gradient, Adam) is ~1400 kernel launches for the PointCNN configuration.  Enqueued one by one from Python they cost the host
27 ms per step -- more than the device needs at the reference's own per-rank batch of 2 (rpn_multiclass.config:206) or at one
frame per GPU (BASELINE config 4: a batch of 8 over 8 GPUs).  Here the step is captured once into a hipGraph and replayed:

  static slots     the frames (xyz, intensity, image feature map, calibration, labels) and the coordinate-only geometry
                   (sampled points, neighbour tables: pipeline.GeometryPrefetcher computes it ahead on side streams) live in
                   fixed buffers; a step copies its inputs in (one foreach copy) and replays the graph;
  flat gradients   with several ranks the gradients are gathered into ONE buffer by a foreach copy at the end of the captured
                   backward pass; the exchange is one RCCL all-reduce of that buffer (hvd.DistributedOptimizer averages
                   gradients, trainer.py:71), followed by the fused Adam step on views of it; with one rank Adam is part of
                   the graph and reads the gradients where autograd left them.  (Pre-set .grad views would make autograd ADD
                   every gradient into the buffer: 250 extra kernels per step for this model.)
  random numbers   dropout and the path-drop coin flips draw from the device generator, whose Philox offset PyTorch
                   advances per replay: every step sees fresh masks, as in eager mode (tests/test_graph_step.py).

  two graphs       with several ranks and a model whose backward pass can be cut (the PointCNN RPN: everything after the encoder
                   | the encoder), the step is captured as TWO graphs: the first ends when the gradients of the late half of
                   the parameters (decoder, fc, heads: produced first) sit in their chunk of the flat buffer, whose all-reduce
                   then runs on the collective's stream WHILE the second graph replays the encoder's backward pass; the second
                   chunk follows.  One exchange of 54 MB after the graph was ~10 % of a 9 ms step at one frame per GPU.

No work is skipped or cached: a replay launches exactly the kernels the eager step launches.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, stream_ptr


def copy_many(dst, src):
    """dst[i] <- src[i] for lists of device tensors of ANY mix of dtypes in one launch per 64 pairs (csrc/optim.hip,
    hf_copy_multi): torch._foreach_copy_ falls back to one copy launch per tensor as soon as the list mixes dtypes (int32
    neighbour tables, fp32 points, int64 labels ...), ~40 launches of 5 us each in front of every replayed step.  Pairs that are
    not plain byte copies (a dtype change, a non-contiguous side) go to the framework."""
    L = _lib.lib()
    fast, slow = [], []
    for d, s_ in zip(dst, src):
        ok = (d.is_cuda and s_.is_cuda and d.device == s_.device and d.dtype == s_.dtype and d.shape == s_.shape and
              d.is_contiguous() and s_.is_contiguous())
        (fast if ok else slow).append((d, s_))
    cap = L.hf_copy_multi_max()
    for i in range(0, len(fast), cap):
        part = fast[i:i + cap]
        n = len(part)
        d_arr = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in part])
        s_arr = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in part])
        b_arr = (ctypes.c_longlong * n)(*[d.numel() * d.element_size() for d, _ in part])
        check(L.hf_copy_multi(n, d_arr, s_arr, b_arr, stream_ptr()), "copy_multi")
    if slow:
        torch._foreach_copy_([d for d, _ in slow], [s_ for _, s_ in slow])



def tree_tensors(obj, out=None):
    """the tensors of a nested dict / list / tuple structure, in a fixed order"""
    out = [] if out is None else out
    if isinstance(obj, torch.Tensor):
        out.append(obj)
    elif isinstance(obj, dict):
        for k in sorted(obj):
            tree_tensors(obj[k], out)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            tree_tensors(v, out)
    return out


def tree_map(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: tree_map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(tree_map(v, fn) for v in obj)
    return obj


class FlatGrads:
    """One contiguous buffer with a view per parameter (gradient_as_bucket_view with a single bucket).  `adopt()` makes the views
    the parameters' .grad (eager steps: autograd then accumulates into the zeroed buffer); `gather(grads)` copies freshly
    produced gradients in with one foreach copy (captured steps)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def adopt(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def release(self):
        for p in self.params:
            p.grad = None

    def zero(self):
        self.flat.zero_()

    def gather(self, grads):
        """grads: one tensor (or None = zero gradient, False = leave the slot alone) per parameter"""
        dst = [v for v, g in zip(self.views, grads) if g is not None and g is not False]
        src = [g for g in grads if g is not None and g is not False]
        for v, g in zip(self.views, grads):
            if g is None:                                              # False: not this call's business (the other chunk)
                v.zero_()
        if dst:
            torch._foreach_copy_(dst, src)

    def all_reduce_mean(self, world, scale=True):
        """hvd.DistributedOptimizer: the average over the replicas, one collective for the whole model (scale=False: the sum;
        the optimizer then applies 1 / world itself while it loads the gradients)"""
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if scale:
                self.flat.mul_(1.0 / world)

    def segment(self, first, count):
        """the contiguous slice of the flat buffer that holds parameters first .. first + count - 1"""
        o0 = sum(p.numel() for p in self.params[:first])
        return self.flat[o0:o0 + sum(p.numel() for p in self.params[first:first + count])]


def broadcast_parameters(module, src=0):
    """hvd.broadcast_global_variables(0) (trainer.py:73,144): rank 0's parameters and buffers to everyone"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


class TrainStep:
    """One RPN train step.  graph=False: the eager step (flat gradients, explicit all-reduce); graph=True: the same step
    replayed from a captured hipGraph.

      step = TrainStep(model, opt, inputs, geometry, world=..., graph=True)
      loss = step(geometry=geo)          # inputs are the static tensors given at construction unless passed again

    `inputs`: dict with xyz, intensity, label_cls, label_reg and optionally img_fts, calib.  `loss_fn(model, inputs, geometry)`
    returns the scalar loss (default: RpnModel forward + RpnModel.loss).  Capturing needs `warmup` eager steps first (lazy
    initialisations, optimizer state); their effect on parameters, buffers and optimizer state is undone before the capture, so
    the first replay starts from the state the caller handed in -- the same trajectory as graph=False."""

    def __init__(self, model, optimizer, inputs, geometry, world=1, graph=True, loss_fn=None, warmup=3, overlap_exchange=True):
        self.model, self.opt, self.world = model, optimizer, world
        self.loss_fn = loss_fn or _rpn_loss
        self.flat_mode = world > 1                 # one rank: the optimizer reads the gradients where autograd leaves them
        # the cut of the backward pass: parameters of the encoder ("early": their gradients come last) | everything else
        enc = getattr(getattr(model, "backbone", None), "enc", None)
        self.split = bool(overlap_exchange and self.flat_mode and loss_fn is None and enc is not None)
        early_ids = {id(p) for p in enc.parameters()} if self.split else set()
        params = [p for p in model.parameters() if p.requires_grad]
        self.late = [p for p in params if id(p) not in early_ids]
        self.early = [p for p in params if id(p) in early_ids]
        self.grads = FlatGrads(self.late + self.early)        # [late | early]: each chunk one contiguous slice
        self.chunks = 2 if self.split else 1
        # optim.MultiTensorAdam multiplies the gradients by 1 / world on load: the flat buffer then carries the SUM
        self.opt_scales = self.flat_mode and hasattr(optimizer, "grad_scale")
        if self.opt_scales:
            optimizer.grad_scale = 1.0 / world
        self.inputs = dict(inputs)
        self.geometry = tree_map(geometry, lambda t: t.clone())       # static slots
        self._geo_slots = tree_tensors(self.geometry)
        self.graph = None
        self.loss = None
        self.opt_in_graph = world == 1
        if graph:
            self._capture(warmup)

    # ------------------------------------------------------------------ pieces
    def _forward_backward(self):
        # fresh gradients every step (.grad = None: autograd assigns, it does not add); with several ranks they are gathered
        # into the flat buffer by one foreach copy
        self.grads.release()
        img = self.inputs.get("img_fts")
        if img is not None and img.requires_grad:
            img.grad = None                                           # a leaf outside the model: backward stores, never adds
        loss = self.loss_fn(self.model, self.inputs, self.geometry)
        loss.backward()
        if self.flat_mode:
            self.grads.gather([p.grad for p in self.grads.params])
        return loss.detach()

    def _finish(self):
        if self.flat_mode:
            self.grads.all_reduce_mean(self.world, scale=not self.opt_scales)
            self.grads.adopt()                                        # the optimizer steps on the averaged views
        self.opt.step()

    # ---- the step cut in two (several ranks): [forward, loss, backward down to the encoder's outputs] | [the encoder's backward]
    def _first_half(self):
        self.grads.release()
        img = self.inputs.get("img_fts")
        img_leaf = img is not None and img.requires_grad
        if img_leaf:
            img.grad = None
        taps = []
        loss = _rpn_loss(self.model, self.inputs, self.geometry, taps=taps)
        # taps = (encoder output, detached copy read by everything downstream): the copies are leaves, so their gradients hold
        # the downstream paths only and the encoder's graph is untouched by this call
        wanted = self.late + [c for _, c in taps] + ([img] if img_leaf else [])
        got = torch.autograd.grad(loss, wanted, allow_unused=True)
        n = len(self.late)
        self._taps, self._tap_grads = [t for t, _ in taps], list(got[n:n + len(taps)])
        if img_leaf:
            img.grad = got[-1]
        self.grads.gather(list(got[:n]) + [False] * len(self.early))
        return loss.detach()

    def _second_half(self):
        live = [(t, g) for t, g in zip(self._taps, self._tap_grads) if g is not None]
        got = torch.autograd.grad([t for t, _ in live], self.early, grad_outputs=[g for _, g in live], allow_unused=True)
        self.grads.gather([False] * len(self.late) + list(got))

    def _exchange_overlapped(self, second):
        """chunk 0 (late parameters) is reduced while `second` (the encoder's backward pass) runs; then chunk 1"""
        n = len(self.late)
        c0, c1 = self.grads.segment(0, n), self.grads.segment(n, len(self.early))
        work = dist.all_reduce(c0, op=dist.ReduceOp.SUM, async_op=True)    # on the collective's stream, after what is enqueued here
        second()
        dist.all_reduce(c1, op=dist.ReduceOp.SUM)
        work.wait()                                                         # this stream waits for chunk 0 as well
        if not self.opt_scales:
            self.grads.flat.mul_(1.0 / self.world)

    def _capture(self, warmup):
        # PyTorch's recipe: a few eager iterations on a side stream (lazy initialisations, allocator warm-up, optimizer
        # state), then capture on that stream
        # The warm-up steps are real steps (Adam, BatchNorm running statistics, with several ranks the all-reduce) on the batch
        # given at construction; building a TrainStep must not train, so parameters, buffers and optimizer state are put back
        # IN PLACE afterwards (same storage: the capture below records these addresses).  Only the device generator's offset
        # stays advanced (dropout / path-drop draws of the warm-up).
        snap = self._snapshot()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._forward_backward()
                self._finish()
        torch.cuda.current_stream().wait_stream(s)
        self._restore(snap)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        if self.split:
            with torch.cuda.graph(self.graph):
                self.loss = self._first_half()
            self.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph2, pool=self.graph.pool()):
                self._second_half()
            self._taps = self._tap_grads = None                       # the graphs own what they need; drop the autograd graph
        else:
            with torch.cuda.graph(self.graph):
                self.loss = self._forward_backward()
                if self.opt_in_graph:
                    self.opt.step()
        # the captured backward writes its gradients into the graph's own pool at every replay, whatever .grad points to
        # afterwards: with several ranks the parameters adopt the flat views (filled by the captured foreach copy)
        if self.flat_mode:
            self.grads.adopt()

    def _snapshot(self):
        with torch.no_grad():
            tensors = [t for t in list(self.model.parameters()) + list(self.model.buffers())]
            if hasattr(self.opt, "snapshot"):                           # optim.MultiTensorAdam: two flat moment buffers + the counter
                state = self.opt.snapshot()
            else:
                state = {id(t): {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                         for t, st in self.opt.state.items()}
            return [(t, t.clone()) for t in tensors], state

    def _restore(self, snap):
        tensors, state = snap
        with torch.no_grad():
            if tensors:
                torch._foreach_copy_([t for t, _ in tensors], [c for _, c in tensors])
            if hasattr(self.opt, "restore"):
                self.opt.restore(state)
                return
            for t, st in self.opt.state.items():                       # state the warm-up created: back to its initial zeros
                old = state.get(id(t), {})
                for k, v in st.items():
                    if isinstance(v, torch.Tensor):
                        v.copy_(old[k]) if k in old else v.zero_()
                    elif k in old:
                        st[k] = old[k]

    # ------------------------------------------------------------------ a step
    def load(self, geometry=None, **inputs):
        """copy this step's inputs into the static slots (device-to-device, no synchronisation)"""
        dst, src = [], []
        if geometry is not None:
            new = tree_tensors(geometry)
            assert len(new) == len(self._geo_slots), "geometry structure changed"
            dst += self._geo_slots
            src += new
        for k, v in inputs.items():
            if v is not None and v is not self.inputs[k]:
                dst.append(self.inputs[k] if not self.inputs[k].requires_grad else self.inputs[k].detach())
                src.append(v)
        if dst:
            with torch.no_grad():
                copy_many(dst, src)

    def __call__(self, geometry=None, **inputs):
        self.load(geometry, **inputs)
        if self.graph is None and self.split:
            self.loss = self._first_half()
            self._exchange_overlapped(self._second_half)
            self.grads.adopt()
            self.opt.step()
        elif self.graph is None:
            self.loss = self._forward_backward()
            self._finish()
        elif self.split:
            self.graph.replay()
            self._exchange_overlapped(self.graph2.replay)
            self.opt.step()
        else:
            self.graph.replay()
            if not self.opt_in_graph:
                self.grads.all_reduce_mean(self.world, scale=not self.opt_scales)
                self.opt.step()
        return self.loss


def _rpn_loss(model, inputs, geometry, taps=None):
    seg_logits, head = model(inputs["xyz"], inputs["intensity"], geometry=geometry, img_fts=inputs.get("img_fts"),
                             calib=inputs.get("calib"), **({"taps": taps} if taps is not None else {}))
    loss, _ = model.loss(inputs["xyz"], seg_logits, head, inputs["label_cls"], inputs["label_reg"])
    return loss
