import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import heterofusionrcnn_amd as hf
import oracle
rng = np.random.default_rng(0)
for (b, n, m, r, ns) in [(2, 1024, 256, 0.1, 32), (1, 1024, 128, 0.2, 32), (1, 4096, 2048, 0.05, 16)]:
    x1 = rng.random((b, n, 3), dtype=np.float32); x2 = rng.random((b, m, 3), dtype=np.float32)
    idx, cnt = hf.query_ball_point(r, ns, torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda())
    oi, oc = oracle.query_ball_point(r, ns, x1, x2)
    idx = idx.cpu().numpy(); cnt = cnt.cpu().numpy()
    bad = np.argwhere((idx != oi).any(-1) | (cnt != oc))
    print((b, n, m, r, ns), "bad rows:", len(bad), "of", b * m)
    for (bi, j) in bad[:4]:
        print("  row", bi, j, "cnt", cnt[bi, j], "oracle", oc[bi, j]); print("   got", idx[bi, j][:12]); print("   exp", oi[bi, j][:12])
