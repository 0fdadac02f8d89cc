"""Diagnostic: burst time of the fused ball-query kernel at the headline shape for one or more builds of the library
(scripts/probes/build_variant.sh), each checked against the brute-force kernel.  usage: qbp_variant.py lib1.so lib2.so ..."""
import ctypes, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, __file__, lib], capture_output=True, text=True)
        print(lib, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
    sys.exit(0)
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("heterofusionrcnn_amd._lib", os.path.join(ROOT, "heterofusionrcnn_amd", "_lib.py"))
_lib = importlib.util.module_from_spec(spec)
sys.modules["heterofusionrcnn_amd._lib"] = _lib
spec.loader.exec_module(_lib)
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
import heterofusionrcnn_amd as hf
from bench import kitti_uniform
L = _lib.lib()
rng = np.random.default_rng(0)
out = {}
for B in (8, 80):
    N, M, K, R = 16384, 4096, 32, 0.5
    xyz = torch.from_numpy(kitti_uniform(rng, B, N)).cuda()
    new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(M, xyz))
    idx = torch.empty((B, M, K), dtype=torch.int32, device="cuda"); cnt = torch.empty((B, M), dtype=torch.int32, device="cuda")
    grouped = torch.empty((B, M, K, 3), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    args = (B, N, M, R, K, xyz.data_ptr(), new_xyz.data_ptr(), 1, idx.data_ptr(), cnt.data_ptr(), grouped.data_ptr(), st)
    os.environ["HF_BALL_QUERY"] = "bruteforce"
    assert L.hf_query_ball_group_xyz(*args) == 0
    torch.cuda.synchronize()
    ref = (idx.clone(), cnt.clone(), grouped.clone())
    os.environ.pop("HF_BALL_QUERY")
    idx.fill_(-7); cnt.fill_(-7); grouped.fill_(-7.0)
    assert L.hf_query_ball_group_xyz(*args) == 0
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(ref, (idx, cnt, grouped)))
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            L.hf_query_ball_group_xyz(*args)
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / 200)
    out["B%d" % B] = round(best, 2); out["B%d_ok" % B] = ok
print(json.dumps(out))
