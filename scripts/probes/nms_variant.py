"""Diagnostic: oriented_nms on 9000 boxes (uniform / 300 clusters x 30) for several builds of the library in ONE run:
usage nms_variant.py lib1.so lib2.so ...; each library is timed in its own child process, two rounds."""
import os, sys, subprocess, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    for rnd in range(2):
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
rng = np.random.default_rng(3)
nb = torch.from_numpy(rand_bev(rng, 9000)).cuda()
r2 = np.random.default_rng(4)
cl = np.repeat(rand_bev(r2, 300), 30, 0)
cl[:, [0, 2]] += r2.normal(0, 0.3, (9000, 1)).astype(np.float32)
cl[:, [1, 3]] += r2.normal(0, 0.3, (9000, 1)).astype(np.float32)
cl[:, 4] += r2.normal(0, 0.1, 9000).astype(np.float32)
nbc = torch.from_numpy(cl.astype(np.float32)).cuda()
def wall(fn, iters=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return round(1e6 * (time.perf_counter() - t0) / iters, 1)
out = {"uniform": wall(lambda: hf.oriented_nms(nb, 0.8)), "clustered": wall(lambda: hf.oriented_nms(nbc, 0.8))}
os.environ["HF_NMS_STOP"] = "1"
out["mask_uniform"] = wall(lambda: hf.oriented_nms(nb, 0.8)); out["mask_clustered"] = wall(lambda: hf.oriented_nms(nbc, 0.8))
print(json.dumps(out))
