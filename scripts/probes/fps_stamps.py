"""Diagnostic: where a round of the bucketed FPS kernel spends its cycles (16384 -> 4096, one workgroup per cloud).
Uses the SEPARATE -DHF_FPS_STAMPS build of the library (scripts/probes/build_stamps.sh -> scripts/probes/libhfops_stamps.so),
started with HFOPS_LIBRARY pointing at it; wave 0 of every workgroup sums s_memtime differences per section over all
rounds.  Read the SHARES, not the total (the stamps fence the schedule)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
STAMPED = os.path.join(ROOT, "scripts", "probes", "libhfops_stamps.so")
assert os.environ.get("HFOPS_LIBRARY") == STAMPED, "start with HFOPS_LIBRARY=%s" % STAMPED
import heterofusionrcnn_amd as hf
from bench import kitti_frustum
rng = np.random.default_rng(0)
B, N, M = 8, 16384, 4096
xyz = torch.from_numpy(kitti_frustum(rng, B, N)).cuda()
for nt in (1024, 512):
    for _ in range(2):
        hf.farthest_point_sample(M, xyz, kernel="bucket", threads=nt)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * B))()
    assert ctypes.CDLL(STAMPED).hf_debug_fps_stamps(buf, B) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(B, 8).astype(np.float64) / (M - 1)
    names = ["loop head", "bucket test + ballot", "update touched buckets", "candidate of the wave", "slot write", "barrier", "block pick"]
    print("%d threads: shader-clock ticks per round (wave 0), median over %d clouds" % (nt, B))
    for i, nm in enumerate(names):
        print("  %-26s %8.1f" % (nm, np.median(a[:, i])))
    print("  %-26s %8.1f" % ("sum", np.median(a[:, :7].sum(1))))
