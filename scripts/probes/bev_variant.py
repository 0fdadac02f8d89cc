"""Diagnostic: bev_iou 70 000 x 64 burst time for several builds of the library in ONE run (boxes of the pool differ by
several per cent): usage bev_variant.py lib1.so lib2.so ... ; each library is timed in its own child process, three rounds."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    for rnd in range(3):
        for lib in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, lib], capture_output=True, text=True)
            print(rnd, os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
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
from bench import rand_bev
L = _lib.lib()
rng = np.random.default_rng(3)
a = torch.from_numpy(rand_bev(rng, 70000)).cuda(); g = torch.from_numpy(rand_bev(rng, 64)).cuda()
ov = torch.empty((70000, 64), dtype=torch.float32, device="cuda"); io = torch.empty_like(ov)
st = torch.cuda.current_stream().cuda_stream
args = (70000, a.data_ptr(), 64, g.data_ptr(), ov.data_ptr(), io.data_ptr(), st)
best = []
for rep in range(5):
    for _ in range(20): L.hf_compute_bev_iou(*args)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): L.hf_compute_bev_iou(*args)
    e1.record(); torch.cuda.synchronize()
    best.append(round(5 * e0.elapsed_time(e1), 2))
print(json.dumps(best))
