"""Diagnostic: where a workgroup of the cell ball-query kernel spends its cycles (headline shape).
Uses a SEPARATE build of the library with -DHF_QBP_STAMPS (scripts/probes/build_stamps.sh -> gpurun_out/libhfops_stamps.so);
wave 0 of every workgroup stamps s_memtime at the phase boundaries.  Read the SHARES, not the total (stamps fence the schedule)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
STAMPED = os.path.join(ROOT, "scripts", "probes", "libhfops_stamps.so")
import importlib.util
spec = importlib.util.spec_from_file_location("heterofusionrcnn_amd._lib", os.path.join(ROOT, "heterofusionrcnn_amd", "_lib.py"))
_lib = importlib.util.module_from_spec(spec)
sys.modules["heterofusionrcnn_amd._lib"] = _lib     # the package then finds this instance, pointed at the stamped build
spec.loader.exec_module(_lib)
_lib.LIB_PATH = STAMPED
import heterofusionrcnn_amd as hf
from bench import kitti_uniform
rng = np.random.default_rng(0)
B, N, M, K = 8, 16384, 4096, 32
xyz = torch.from_numpy(kitti_uniform(rng, B, N)).cuda()
new_xyz = hf.gather_point(xyz, hf.farthest_point_sample(M, xyz))
for _ in range(5):
    out = hf.query_ball_group(0.5, K, xyz, new_xyz, True)
torch.cuda.synchronize()
nwg = 256
buf = (ctypes.c_ulonglong * (16 * nwg))()
dbg = ctypes.CDLL(STAMPED).hf_debug_qbp_stamps
assert dbg(buf, nwg) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 16).astype(np.int64)
names = ["start", "issue loads+init+barrier0", "wait query+mark", "barrierA", "pass1+2", "barrierB", "link+barrierC", "search", "write"]
NS = len(names)
d = np.diff(a[:, :NS], axis=1)
print("phase (wave 0 of each workgroup)  median / p10 / p90 shader cycles")
for i, nm in enumerate(names[1:]):
    print("%-27s %8.0f %8.0f %8.0f" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("inside the first phase: start -> loads issued %.0f, -> LDS initialised %.0f, -> barrier0 passed %.0f" % (
    np.median(a[:, 9] - a[:, 0]), np.median(a[:, 10] - a[:, 9]), np.median(a[:, 1] - a[:, 10])))
print("inside pass1+2: barrierA -> masks done %.0f, -> slots reserved %.0f, -> candidates re-read and linked %.0f" % (
    np.median(a[:, 14] - a[:, 3]), np.median(a[:, 15] - a[:, 14]), np.median(a[:, 4] - a[:, 15])))
print("total                       %8.0f" % np.median(a[:, NS - 1] - a[:, 0]))
# dispatch skew and tail on the constant 100 MHz clock (s_memrealtime, the same for the whole device; 10 ns ticks)
t0 = a[:, 12].min()
st = (a[:, 12] - t0) * 10
en = (a[:, 13] - t0) * 10
print("workgroup start after the first one, ns: median %d  p90 %d  max %d" % (np.median(st), np.percentile(st, 90), st.max()))
print("last wave of a workgroup ends, ns:      min %d  median %d  p90 %d  max %d" % (en.min(), np.median(en), np.percentile(en, 90), en.max()))
print("workgroup lifetime, ns:                 median %d  p10 %d  p90 %d" % (np.median(en - st), np.percentile(en - st, 10), np.percentile(en - st, 90)))
