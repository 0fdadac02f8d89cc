"""correctness (vs fp64) and time of the MFMA GEMM kernels against the hipBLASLt paths they replace"""
import sys, torch
sys.path.insert(0, '.')
from heterofusionrcnn_amd.mlp import _splitk_wgrad, linear_wgrad

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

shapes = [(1048576, 4, 32), (1048576, 32, 32), (1048576, 32, 64), (262144, 67, 64), (262144, 64, 96), (262144, 96, 128),
          (65536, 131, 128), (65536, 128, 196), (65536, 196, 256), (8192, 384, 256), (8192, 256, 256), (32768, 320, 256),
          (32768, 256, 256), (131072, 129, 128), (131072, 128, 128), (1000, 5, 3)]
for rows, cin, cout in shapes:
    torch.manual_seed(0)
    x = torch.randn(rows, cin, device='cuda'); dz = torch.randn(rows, cout, device='cuda')
    ref = (dz.double().t() @ x.double())
    a = linear_wgrad(dz, x); b = _splitk_wgrad(dz, x)
    ea = ((a.double() - ref).abs().max() / ref.abs().max()).item(); eb = ((b.double() - ref).abs().max() / ref.abs().max()).item()
    g = torch.rand(cin, device='cuda') + .5; be = torch.randn(cin, device='cuda'); mu = torch.randn(cin, device='cuda'); iv = torch.rand(cin, device='cuda') + .5
    h = torch.relu((x - mu) * iv * g + be)
    c = linear_wgrad(dz, x, (g, be, mu, iv)); refh = dz.double().t() @ h.double()
    ec = ((c.double() - refh).abs().max() / refh.abs().max()).item()
    ta = t(lambda: linear_wgrad(dz, x)); tb = t(lambda: _splitk_wgrad(dz, x)); tc = t(lambda: linear_wgrad(dz, x, (g, be, mu, iv)))
    bytes_ = rows * (cin + cout) * 4
    print("rows %8d cin %4d cout %4d  mfma %7.1f us (%5.2f TB/s, err %.1e)  act-on-load %7.1f us (err %.1e)  hipblaslt %7.1f us (err %.1e)" %
          (rows, cin, cout, ta, bytes_ / ta / 1e6, ea, tc, ec, tb, eb), flush=True)
