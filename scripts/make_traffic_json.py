"""profiles/roofline_traffic.json from the two rocprofv3 --pmc passes of scripts/roofline_kernel.py (FETCH_SIZE and
WRITE_SIZE cannot share a pass on gfx950), stamped with the hash of the kernel source it was measured on: bench.py
reports `traffic` only while that stamp matches the source in the tree.
usage: python scripts/make_traffic_json.py <pmc_fetch_dir> <pmc_write_dir> <out.json> [copy-csv-prefix]"""
import csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_stamp, SURVEY_TWO_OP_BYTES, ball_group_bytes, B, N0, KNN, SA


def avg(d, counter, kernel="qbp_cell"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v), f


fetch, nf, ff = avg(sys.argv[1], "FETCH_SIZE")
write, nw, fw = avg(sys.argv[2], "WRITE_SIZE")
traffic = int(round((2.0 * fetch + write) * 1024))
out = {"kernel": "hf::qbp_cell_kernel<true, 2, 1024, true> (hf_query_ball_group_xyz), B=8 N=16384 M=4096 K=32",
       "kernel_source_stamp": kernel_source_stamp(),
       "FETCH_SIZE_KiB_per_launch": round(fetch, 1), "WRITE_SIZE_KiB_per_launch": round(write, 1),
       "launches": [nf, nw],
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
       "traffic_bytes_per_launch": traffic,
       "algorithmic_bytes_per_launch_survey_8d": SURVEY_TWO_OP_BYTES,
       "fused_kernel_compulsory_bytes": ball_group_bytes(B, N0, SA[0][0], KNN),
       "source": "separate rocprofv3 --pmc passes of scripts/roofline_kernel.py"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
if len(sys.argv) > 4:
    shutil.copy(ff, sys.argv[4] + "_pmc_FETCH_SIZE.csv")
    shutil.copy(fw, sys.argv[4] + "_pmc_WRITE_SIZE.csv")
print(json.dumps(out))
