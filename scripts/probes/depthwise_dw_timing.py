"""Diagnostic: hf_depthwise_k_grad (the X-transform's depthwise layers, K = M = C = 8) -- weight gradient alone, input gradient alone."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import _lib
from heterofusionrcnn_amd._lib import ptr, stream_ptr, check
from bench import time_op
L = _lib.lib()
res = {}
for rows in (131072, 32768, 8192, 2048):
    k = c = m = 8
    x = torch.randn(rows, k, c, device="cuda")
    w = torch.randn(k, c, m, device="cuda")
    gy = torch.randn(rows, c * m, device="cuda")
    gx = torch.empty_like(x)
    gw = torch.empty_like(w)
    nbytes = L.hf_depthwise_k_grad_workspace(rows, k, c, m)
    ws = torch.empty(nbytes // 4, device="cuda")
    f_a = lambda: check(L.hf_depthwise_k_grad(rows, k, c, m, ptr(x), ptr(w), ptr(gy), None, ptr(gw), stream_ptr()), "dw")
    f_w = lambda: check(L.hf_depthwise_k_grad_ws(rows, k, c, m, ptr(x), ptr(w), ptr(gy), None, ptr(gw), ptr(ws), nbytes, stream_ptr()), "dw")
    f_x = lambda: check(L.hf_depthwise_k_grad(rows, k, c, m, ptr(x), ptr(w), ptr(gy), ptr(gx), None, stream_ptr()), "dx")
    f_w()
    ref = torch.einsum("rwc,rcm->wcm", x.double(), gy.view(rows, c, m).double())
    err = float((gw.double() - ref).abs().max() / ref.abs().max())
    res[rows] = {"grad_w_atomics_us": round(time_op(f_a, iters=20, warm=3), 2), "grad_w_us": round(time_op(f_w, iters=20, warm=3), 2), "grad_x_us": round(time_op(f_x, iters=20, warm=3), 2), "grad_w_rel_err": err}
print(json.dumps(res, indent=1))
