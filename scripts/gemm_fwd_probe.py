"""correctness (vs fp64) and time of the fused forward kernel (GEMM + bias + BN statistics, activation on load)
against the three passes it replaces (hipBLASLt addmm, BN statistics, BN apply)"""
import sys, torch
sys.path.insert(0, '.')
from heterofusionrcnn_amd.mlp import linear_bn_fwd, BatchNormReLU

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

shapes = [(1048576, 4, 32), (1048576, 32, 32), (1048576, 32, 64), (262144, 67, 64), (262144, 68, 64), (262144, 64, 96), (262144, 96, 128),
          (65536, 131, 128), (65536, 128, 196), (65536, 196, 256), (8192, 384, 256), (8192, 256, 256), (32768, 320, 256),
          (32768, 256, 256), (131072, 129, 128), (131072, 128, 128), (1000, 5, 3)]
for rows, cin, cout in shapes:
    torch.manual_seed(0)
    x = torch.randn(rows, cin, device='cuda'); w = torch.randn(cout, cin, device='cuda') / cin ** .5; b = torch.randn(cout, device='cuda')
    bn = BatchNormReLU(cout).cuda(); bn_in = BatchNormReLU(cin).cuda()
    g = torch.rand(cin, device='cuda') + .5; be = torch.randn(cin, device='cuda'); mu = torch.randn(cin, device='cuda'); iv = torch.rand(cin, device='cuda') + .5
    z, mean, invstd, _ = linear_bn_fwd(x, w, b, bn)
    ref = x.double() @ w.double().t() + b.double()
    ez = ((z.double() - ref).abs().max() / ref.abs().max()).item()
    em = (mean.double() - ref.mean(0)).abs().max().item()
    ei = ((invstd.double() - 1 / (ref.var(0, unbiased=False) + bn.eps).sqrt()).abs().max() * ref.std(0).max()).item()
    h = torch.relu((x - mu) * iv * g + be)
    z2, m2, i2, _ = linear_bn_fwd(x, w, b, bn, (g, be, mu, iv))
    ref2 = h.double() @ w.double().t() + b.double()
    ez2 = ((z2.double() - ref2).abs().max() / ref2.abs().max()).item()
    em2 = (m2.double() - ref2.mean(0)).abs().max().item()
    ta = t(lambda: linear_bn_fwd(x, w, b, bn)); tc = t(lambda: linear_bn_fwd(x, w, b, bn, (g, be, mu, iv)))
    def three():
        zz = torch.addmm(b, x, w.t()); return bn(zz)
    tb = t(three); tg = t(lambda: torch.addmm(b, x, w.t()))
    print("rows %8d cin %4d cout %4d  fused %7.1f us (err z %.1e mean %.1e invstd %.1e)  act-on-load %7.1f us (err %.1e %.1e)   addmm %6.1f  addmm+stats+apply %7.1f us" %
          (rows, cin, cout, ta, ez, em, ei, tc, ez2, em2, tg, tb), flush=True)
