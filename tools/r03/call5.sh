#!/bin/bash
# round 3, GPU call 5: full GPU suite, the whole bench (timed), PC sampling exploration on the headline step
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call5; mkdir -p $OUT
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
tail -8 $OUT/pytest_gpu.log
( time timeout 1500 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<PY
import json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value %.3g %s, ms_per_step %.3f" % (j["value"], j["unit"], j["ms_per_step"]))
print("parity", j.get("parity_checked", {}).get("equal"))
e = j["bases"].get("end_to_end", {})
print("e2e gpu", e.get("gpu", {}).get("seconds"), e.get("gpu", {}).get("rep_seconds"), "cpu", e.get("cpu", {}).get("decode_runs"), e.get("cpu", {}).get("scan_runs"), "speedup", e.get("speedup_vs_cpu_overlapped"), "equal", e.get("tables_equal"), e.get("error"))
print("level6", e.get("level6"))
print("binary", json.dumps(j.get("binary_configs"))[:1500])
for l in e.get("gpu", {}).get("stderr_timing", []): print("   ", l)
PY
cd /tmp && export TMPDIR=/tmp
for m in stochastic host_trap; do
  timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method $m --pc-sampling-interval 65536 --kernel-trace --output-format csv json -d $OUT/pcs_$m -- \
     python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pcs_$m.log 2>&1
  echo "pc sampling $m rc $?"; tail -3 $OUT/pcs_$m.log
  find $OUT/pcs_$m -type f | head -20; du -sh $OUT/pcs_$m
done
# keep the merge small: heads of big files only
find $OUT -type f -size +8M -exec sh -c 'head -c 4000000 "$1" > "$1.head"; rm "$1"' _ {} \;
