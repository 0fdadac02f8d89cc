"""rocprofv3 target: ONE forward and ONE backward BatchNorm call (ELU mode) on a 1 048 576 x 64 and a 131 072 x 256 tensor, a few
times each, for the FETCH_SIZE / WRITE_SIZE passes behind profiles/r04_bn_stream_pmc.txt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd import _lib
from heterofusionrcnn_amd._lib import ptr, stream_ptr, check
L = _lib.lib()
for rows, c in ((1048576, 64), (131072, 256)):
    x = torch.randn(rows, c, device="cuda"); dy = torch.randn(rows, c, device="cuda")
    gamma, beta = torch.rand(c, device="cuda") + .5, torch.randn(c, device="cuda")
    y, dx = torch.empty_like(x), torch.empty_like(x)
    rm, rv, mean, invstd, dg, db = [torch.zeros(c, device="cuda") for _ in range(6)]
    nbytes = L.hf_bn_workspace(rows, c); ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    st = stream_ptr()
    for _ in range(4):
        check(L.hf_bn_relu_fwd_train(rows, c, ptr(x), ptr(gamma), ptr(beta), 1e-3, 0.1, ptr(rm), ptr(rv), 2, ptr(y), ptr(mean), ptr(invstd), ptr(ws), nbytes, st), "f")
        check(L.hf_bn_relu_bwd(rows, c, ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), 2, ptr(dx), ptr(dg), ptr(db), None, ptr(ws), nbytes, st), "b")
    torch.cuda.synchronize()
print("done")
