"""Diagnostic (diag build: HFOPS_LIBRARY=.../libhfops_diag.so): BatchNorm forward / backward entry points on the one-frame step's
short shapes, three-launch form (HF_BN_SMALL_ROWS=0) against the single-launch kernels under a few geometries; device time per
call from 200 back-to-back calls between two events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd import _lib
from heterofusionrcnn_amd._lib import ptr, stream_ptr, check
L = _lib.lib()
shapes = [(64, 64), (256, 64), (1024, 64), (4096, 64), (512, 256), (1024, 512), (64, 1024), (256, 1024), (2048, 128), (2048, 256), (4096, 256)]
variants = [("three launches", {"HF_BN_SMALL_ROWS": "0"}), ("single 1024", {}), ("single 512", {"HF_BN_SMALL_THREADS": "512"}),
            ("single 256", {"HF_BN_SMALL_THREADS": "256"}), ("single cvw4", {"HF_BN_SMALL_CVW": "4"}), ("single cvw2", {"HF_BN_SMALL_CVW": "2"}),
            ("single cvw1", {"HF_BN_SMALL_CVW": "1"})]
def timed(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("%-14s" % "shape" + "".join("%22s" % v[0] for v in variants) + "   (fwd / bwd us)")
for rows, c in shapes:
    x = torch.randn(rows, c, device="cuda"); dy = torch.randn(rows, c, device="cuda")
    gamma, beta = torch.rand(c, device="cuda") + .5, torch.randn(c, device="cuda")
    y, dx = torch.empty_like(x), torch.empty_like(x)
    rm, rv, mean, invstd, dg, db = [torch.zeros(c, device="cuda") for _ in range(6)]
    nbytes = L.hf_bn_workspace(rows, c); ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = stream_ptr()
    fwd = lambda: check(L.hf_bn_relu_fwd_train(rows, c, ptr(x), ptr(gamma), ptr(beta), 1e-3, 0.1, ptr(rm), ptr(rv), 1, ptr(y), ptr(mean), ptr(invstd), ptr(ws), nbytes, st), "f")
    bwd = lambda: check(L.hf_bn_relu_bwd(rows, c, ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), 1, ptr(dx), ptr(dg), ptr(db), None, ptr(ws), nbytes, st), "b")
    line = "%-14s" % ("%dx%d" % (rows, c))
    for name, env in variants:
        for k in ("HF_BN_SMALL_ROWS", "HF_BN_SMALL_THREADS", "HF_BN_SMALL_CVW"):
            os.environ.pop(k, None)
        os.environ.update(env)
        line += "%22s" % ("%.1f / %.1f" % (timed(fwd), timed(bwd)))
    print(line, flush=True)
