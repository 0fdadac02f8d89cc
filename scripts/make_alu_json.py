"""profiles/bev_iou_alu.json: the ALU-bound fraction of compute_bev_iou at 70 000 x 64 (BASELINE.md section 3 asks for the HBM- and the
ALU-bound fraction) from one rocprofv3 --pmc pass + one --kernel-trace pass of scripts/bev_nms_kernels.py, stamped with the hash of
bev_iou.hip: bench.py prints `bev_iou_frac_alu_bound` only while the stamp matches the source in the tree.
  alu fraction = SQ_INSTS_VALU x 4 cycles (a wave64 vector instruction occupies its SIMD16 for 4 cycles) / (1024 SIMDs x kernel cycles),
  kernel cycles = median kernel duration (kernel trace) x 2.0 GHz -- the shader clock the in-kernel stamps of round 2 measured under
  load (s_memtime against s_memrealtime: 17 368 cycles in 8.76 us, profiles/r02_qbp_cell_phase_stamps.txt).  GRBM_GUI_ACTIVE is
  collected too but sums several instances (434 981 for a 14.8 us kernel), so it is recorded, not used.
usage: python scripts/make_alu_json.py <pmc_dir> <trace_dir> <out.json>"""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stamp():
    with open(os.path.join(ROOT, "heterofusionrcnn_amd", "csrc", "bev_iou.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


pm, tr, out = sys.argv[1:4]
ctr = {}
for f in glob.glob(os.path.join(pm, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "bev_iou_kernel" in r["Kernel_Name"]:
            ctr.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            ctr[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
dur = []
for f in glob.glob(os.path.join(tr, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "bev_iou_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
c = {k: med(list(v.values())) for k, v in ctr.items()}
us = med(dur)
CLOCK_GHZ = 2.0
cycles = us * 1e3 * CLOCK_GHZ
res = {"kernel": "hf::bev_iou_kernel (hf_compute_bev_iou), 70000 x 64", "kernel_source_stamp": stamp(), "counters_median_per_launch": c,
       "kernel_us_median_isolated": round(us, 2), "launches": len(dur),
       "assumed_shader_clock_ghz": CLOCK_GHZ,
       "alu_bound_frac": round(c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles), 4),
       "wave_cycles_parked_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
       "formula": "SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * kernel_us * 2.0 GHz)",
       "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE + --kernel-trace of scripts/bev_nms_kernels.py"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
