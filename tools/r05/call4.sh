#!/bin/bash
# round 5, fourth GPU call: k_lz_stage (matches of a batch resolved in LDS) against k_lz_resolve: the ingest tests over both, kernel times of
# a 20 M-read file, end-to-end runs of a 100 M-read file alternating; the coverage step after the event / copy / wait changes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call4; mkdir -p $OUT; rm -f $OUT/*; cd $R
( timeout 600 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu --timeout 300 2>&1 | tail -15 ) > $OUT/pytest_ingest.log 2>&1
cat $OUT/pytest_ingest.log
python tools/make_bam.py /dev/shm/lz20.bam 20000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/lz20.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/lz.tsv"
cd /tmp && export TMPDIR=/tmp
for v in 2 1; do
  COVERM_LZ_V=$v COVERM_NO_FAST_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lz$v -- $CMD > $OUT/trace_lz$v.log 2>&1
  f=$(find $OUT/trace_lz$v -name "*kernel_stats.csv" | head -1); echo "== COVERM_LZ_V=$v"; grep -E "k_lz|k_inflate_wave|k_crc32|k_bam_extract" $f | cut -d, -f1-7 | sed 's/(.*)//' 
done 2>&1 | tee $OUT/lz_kernel_times.log
cd $R; rm -f /dev/shm/lz20.bam
find $OUT -name "*kernel_trace.csv" -size +8M -delete
python tools/make_bam.py /dev/shm/lz100.bam 100000000 16 >> $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/lz100.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/lz.tsv"
for rep in 1 2 3 4; do
  for v in 2 1; do
    sleep 3; s=$(date +%s.%N); COVERM_LZ_V=$v COVERM_CLI_TIMING=1 $CMD 2> $OUT/e2e_err.log; e=$(date +%s.%N)
    echo "COVERM_LZ_V=$v wall $(echo "$e - $s" | bc) $(grep -o 'ingest (decode+push) [0-9.]*s' $OUT/e2e_err.log) $(md5sum /dev/shm/lz.tsv | cut -c1-8)"
  done
done 2>&1 | tee $OUT/lz_e2e_100M.log
rm -f /dev/shm/lz100.bam /dev/shm/lz.tsv
export COVERM_BENCH_CACHE=/dev/shm
for rep in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('step', 'ms_per_step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in r['all_kernels_ms'].items()})"
done 2>&1 | tee $OUT/step.log
