"""Diagnostic: the X-transformation's depthwise layers (K = c = M = 8) at the step's row counts: forward, input gradient, weight
gradient; device time per call.  HFOPS_LIBRARY=<other build> for an A/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd import _lib
from heterofusionrcnn_amd._lib import ptr, stream_ptr, check
from bench import time_op
L = _lib.lib()
k = c = m = 8
for rows in (131072, 32768, 16384, 2048):
    x = torch.randn(rows, k, c, device="cuda"); w = torch.randn(k, c, m, device="cuda"); gy = torch.randn(rows, c * m, device="cuda")
    y = torch.empty(rows, c * m, device="cuda"); gx = torch.empty_like(x); gw = torch.empty_like(w)
    nb = L.hf_depthwise_k_grad_workspace(rows, k, c, m); ws = torch.empty(max(nb, 4), dtype=torch.uint8, device="cuda")
    st = stream_ptr()
    f = time_op(lambda: check(L.hf_depthwise_k(rows, k, c, m, ptr(x), ptr(w), ptr(y), st), "f"), iters=50, warm=5)
    dx = time_op(lambda: check(L.hf_depthwise_k_grad_ws(rows, k, c, m, ptr(x), ptr(w), ptr(gy), ptr(gx), None, None, 0, st), "dx"), iters=50, warm=5)
    dw = time_op(lambda: check(L.hf_depthwise_k_grad_ws(rows, k, c, m, ptr(x), ptr(w), ptr(gy), None, ptr(gw), ptr(ws), nb, st), "dw"), iters=50, warm=5)
    print("rows %7d: forward %6.1f us, input gradient %6.1f us, weight gradient %6.1f us" % (rows, f, dx, dw), flush=True)
