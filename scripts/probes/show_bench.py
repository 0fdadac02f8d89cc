"""print the interesting fields of a bench.py JSON line"""
import json, sys
d = json.load(sys.stdin if sys.argv[1] == "-" else open(sys.argv[1]))
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "host", d["config"].get("host_enqueue_ms_per_step"), "graph", d["config"].get("hip_graph"))
if "--short" not in sys.argv:
    print("roofline", json.dumps(d.get("roofline")))
if "cpu_baseline" in d:
    print("cpu", json.dumps(d["cpu_baseline"])[:700])
for k, v in (d.get("extra") or {}).items():
    print("  ", k, v)
