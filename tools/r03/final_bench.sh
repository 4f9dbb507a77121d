#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_final; mkdir -p $OUT
cd $R
( time timeout 600 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python - <<PY
import json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms_per_step %.3f parity %s" % (j["value"], j["ms_per_step"], j.get("parity_checked", {}).get("equal")))
print("roofline", {k: j["roofline"][k] for k in ("bound", "kernel", "achieved", "frac", "traffic")})
e = j["bases"].get("end_to_end", {})
print("e2e gpu", e.get("gpu", {}).get("seconds"), e.get("gpu", {}).get("rep_seconds"), "cpu", e.get("cpu", {}).get("decode_runs"), e.get("cpu", {}).get("scan_runs"), "x", e.get("speedup_vs_cpu_overlapped"), e.get("speedup_vs_cpu_serial"), "equal", e.get("tables_equal"), e.get("error"))
l6 = e.get("level6", {}); print("level6", l6.get("gpu_seconds"), l6.get("rep_seconds"), l6.get("speedup_vs_cpu_overlapped"), l6.get("tables_equal"), l6.get("bam_bytes"))
b = j.get("binary_configs", {}); print("binary", {k: (v.get("seconds"), v.get("tables_equal")) for k, v in b.items() if isinstance(v, dict)}, b.get("error"))
print("push", j["bases"]["push_inclusive"]["ms"])
PY
