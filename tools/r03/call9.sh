#!/bin/bash
# round 3, GPU call 9 (every step time-boxed): full GPU suite, valu_rate table, e2e at 50 M reads (one / two upload queues), headline step
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call9; mkdir -p $OUT
cd $R
( time timeout 330 python -m pytest tests -m gpu -x -q --timeout 100 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
timeout 120 python tools/make_bam.py /dev/shm/p.bam 50000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/p.bam -m mean trimmed_mean covered_fraction covered_bases variance length count reads_per_base rpkm tpm anir --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/p.tsv"
for q in 1 2 1 2; do
  sleep 2
  s=$(date +%s.%N)
  COVERM_INGEST_COPY_QUEUES=$q COVERM_CLI_TIMING=1 timeout 25 $CMD 2> $OUT/e2e_q$q.err; rc=$?
  e=$(date +%s.%N)
  echo "queues $q rc $rc wall $(echo "$e - $s" | bc) | $(grep -h 'device ingest: buffers' $OUT/e2e_q$q.err | sed 's/.*device ingest: //')" >> $OUT/e2e.log
done
cat $OUT/e2e.log
rm -f /dev/shm/p.bam /dev/shm/p.tsv
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "bench rc $?"
python - <<PY
import json
try:
    j = json.loads(open("$OUT/bench_headline.json").read().strip().splitlines()[-1])
    print("value %.4g ms_per_step %.3f kernels %s" % (j["value"], j["ms_per_step"], j["roofline"]["all_kernels_ms"]))
except Exception as ex:
    print("no bench line", ex)
PY
