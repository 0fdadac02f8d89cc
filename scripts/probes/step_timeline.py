"""Diagnostic: one replayed train step out of a rocprofv3 kernel trace (steps end at the Adam kernel): wall time, device-busy union,
idle gaps, per-queue busy time, and the step's launches in start order (written to the second argument)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])
idx = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
spans = [(S(rows[b]) - S(rows[a]), a, b) for a, b in zip(idx[:-1], idx[1:])]
spans.sort()
print("steps %d; Adam-to-Adam spans (ms): min %.2f median %.2f max %.2f" % (len(spans), spans[0][0] / 1e6, spans[len(spans) // 2][0] / 1e6, spans[-1][0] / 1e6))
_, a, b = spans[len(spans) // 4]            # a typical replayed step (the shorter half)
step = rows[a + 1:b + 1]
t0, t1 = E(rows[a]), E(step[-1])
print("chosen step: %d launches, wall %.0f us, sum of durations %.0f us" % (len(step), (t1 - t0) / 1e3, sum(E(r) - S(r) for r in step) / 1e3))
iv = sorted((S(r), E(r)) for r in step)
cs, ce, union, gaps = iv[0][0], iv[0][1], 0, [iv[0][0] - t0]
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs; gaps.append(s - ce); cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
g = sorted(gaps)
print("device busy (union) %.0f us; idle %.0f us in %d gaps: median %.2f us, mean %.2f us, > 5 us: %d totalling %.0f us" % (
    union / 1e3, sum(gaps) / 1e3, len(gaps), g[len(g) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3, sum(x > 5000 for x in gaps), sum(x for x in gaps if x > 5000) / 1e3))
qs = {}
for r in step:
    q = r.get("Queue_Id", "?"); qs.setdefault(q, [0, 0]); qs[q][0] += 1; qs[q][1] += E(r) - S(r)
for q, v in qs.items():
    print("queue %s: %d launches, %.0f us" % (q, v[0], v[1] / 1e3))
short = [r for r in step if E(r) - S(r) < 8000]
print("launches under 8 us: %d, totalling %.0f us" % (len(short), sum(E(r) - S(r) for r in short) / 1e3))
agg = {}
for r in step:
    k = r["Kernel_Name"][:70]; agg.setdefault(k, [0, 0]); agg[k][0] += 1; agg[k][1] += E(r) - S(r)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-70s x%-3d %7.0f us" % (k, v[0], v[1] / 1e3))
with open(sys.argv[2], "w") as f:
    prev = t0
    for r in step:
        f.write("%9.1f %8.2f gap %6.2f q%s %s\n" % ((S(r) - t0) / 1e3, (E(r) - S(r)) / 1e3, (S(r) - prev) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:100]))
        prev = max(prev, E(r))
