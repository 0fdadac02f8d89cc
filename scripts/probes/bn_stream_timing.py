"""Diagnostic: the streaming BatchNorm passes (hf_bn_relu_fwd_train = statistics + finalize + apply; hf_bn_relu_bwd = reduce + finalize
+ dx) on the tall tensors of the RPN step, device time per call from bursts of back-to-back calls between two events, and the
bandwidth that corresponds to the passes' compulsory traffic (forward: 2 reads + 1 write of the tensor; backward: 4 reads + 1 write).
A/B against another build: HFOPS_LIBRARY=<path> python scripts/probes/bn_stream_timing.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd import _lib
from heterofusionrcnn_amd._lib import ptr, stream_ptr, check
L = _lib.lib()
shapes = [(1048576, 64), (131072, 256), (131072, 64), (262144, 64), (262144, 128), (32768, 256), (65536, 64), (16384, 256), (16384, 64), (8192, 512)]
def timed(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("library", os.environ.get("HFOPS_LIBRARY", "(default)"))
print("%-16s %-5s %12s %12s %12s %12s" % ("shape", "mode", "fwd us", "fwd TB/s", "bwd us", "bwd TB/s"))
for rows, c in shapes:
    x = torch.randn(rows, c, device="cuda"); dy = torch.randn(rows, c, device="cuda")
    gamma, beta = torch.rand(c, device="cuda") + .5, torch.randn(c, device="cuda")
    y, dx = torch.empty_like(x), torch.empty_like(x)
    rm, rv, mean, invstd, dg, db = [torch.zeros(c, device="cuda") for _ in range(6)]
    nbytes = L.hf_bn_workspace(rows, c); ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = stream_ptr()
    for mode in (2, 1):
        fwd = lambda: check(L.hf_bn_relu_fwd_train(rows, c, ptr(x), ptr(gamma), ptr(beta), 1e-3, 0.1, ptr(rm), ptr(rv), mode, ptr(y), ptr(mean), ptr(invstd), ptr(ws), nbytes, st), "f")
        bwd = lambda: check(L.hf_bn_relu_bwd(rows, c, ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), mode, ptr(dx), ptr(dg), ptr(db), None, ptr(ws), nbytes, st), "b")
        tf, tb = timed(fwd), timed(bwd)
        nb = rows * c * 4
        print("%-16s %-5s %12.1f %12.2f %12.1f %12.2f" % ("%dx%d" % (rows, c), "elu" if mode == 2 else "relu", tf, 3 * nb / tf / 1e6, tb, 5 * nb / tb / 1e6), flush=True)
