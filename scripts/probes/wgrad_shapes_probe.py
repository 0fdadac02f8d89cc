"""weight-gradient routes for the tall-skinny shapes of the one-frame-per-GPU step: dW = g^T x with r rows.
library GEMM, batched split-K with several chunk sizes (+ the sum over chunks), the MFMA kernel (hf_linear_wgrad)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from heterofusionrcnn_amd import mlp
from bench import time_op
shapes = [(131072, 64, 3), (131072, 64, 64), (131072, 256, 320), (32768, 64, 3), (32768, 64, 64), (32768, 256, 320), (16384, 64, 24), (16384, 256, 256),
          (16384, 512, 288), (8192, 128, 128), (1048576, 64, 3), (1048576, 64, 64)]
for (r, co, ci) in shapes:
    g = torch.randn(r, co, device="cuda"); x = torch.randn(r, ci, device="cuda")
    res = {"plain": time_op(lambda: g.t() @ x, iters=20, warm=3)}
    for chunk in (256, 512, 1024, 2048, 4096):
        if r // chunk >= 8:
            res["bmm%d" % chunk] = time_op(lambda: torch.bmm(g.view(-1, chunk, co).transpose(1, 2), x.view(-1, chunk, ci)).sum(dim=0), iters=20, warm=3)
    if co * ci <= 512 * 512:
        res["mfma"] = time_op(lambda: mlp.linear_wgrad(g, x), iters=20, warm=3)
    best = min(res, key=res.get)
    print((r, co, ci), {k: round(v, 1) for k, v in res.items()}, "best", best, flush=True)
