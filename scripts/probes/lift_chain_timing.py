"""Diagnostic: the lifting chain (pointcnn.dense_chain on the 3 local coordinates) forward + backward at a million rows -- the
recompute form (hf_lift_elu_bn_*) -- as a target for rocprofv3 --kernel-trace / --pmc."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
from heterofusionrcnn_amd import pointcnn as pc
from bench import time_op
torch.manual_seed(0)
rows, c = int(os.environ.get("ROWS", 1048576)), int(os.environ.get("C", 64))
d0, d1 = pc.Dense(3, c).cuda().train(), pc.Dense(c, c).cuda().train()
x = torch.randn(8, rows // 8, 3, device="cuda")
go = torch.randn(8, rows // 8, c, device="cuda")
params = [p for d in (d0, d1) for p in d.parameters()]
def step():
    out = pc.dense_chain(d0, d1, x)
    torch.autograd.grad(out, params, go)
print(json.dumps({"rows": rows, "c": c, "fwd_bwd_us": round(time_op(step, iters=int(os.environ.get("ITERS", 10)), warm=2), 1)}))
