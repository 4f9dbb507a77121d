"""Reads a rocprofv3 kernel_trace.csv of one `coverm-amd contig` run over the device ingest and says how busy the device was:
per kernel the sum of its durations, the union of all kernels' busy intervals, the span from the first kernel's start to the last
one's end, and the share of the span during which two or more kernels ran at once.

    python tools/r06/ingest_timeline.py gpurun_out/prof_<tag>/trace/**/kernel_trace.csv [out.json]
"""
import collections
import csv
import glob
import json
import sys

paths = glob.glob(sys.argv[1], recursive=True)
assert paths, "no trace at " + sys.argv[1]
rows = list(csv.DictReader(open(paths[0])))
ev = []
per = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((a, b, name, r.get("Queue_Id", "")))
    per[name][0] += 1
    per[name][1] += (b - a) / 1e6
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for a, b, _, _ in ev:
    pts.append((a, 1))
    pts.append((b, -1))
pts.sort()
busy = multi = 0
depth, last = 0, pts[0][0]
for t, d in pts:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        multi += t - last
    depth += d
    last = t
ingest = [e for e in ev if e[2].startswith("covi::")]
i0, i1 = ingest[0][0], max(e[1] for e in ingest)
out = {"kernels": len(ev), "span_ms": (t1 - t0) / 1e6, "busy_union_ms": busy / 1e6, "two_or_more_at_once_ms": multi / 1e6,
       "ingest_span_ms": (i1 - i0) / 1e6, "sum_of_durations_ms": sum(v[1] for v in per.values()),
       "per_kernel_ms": {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
