#!/bin/bash
# round 5, sixth GPU call: k_lz_stage with the next batch's token fields fetched ahead: ingest tests, kernel times (LZ_V=2 then 1, after one
# untimed run that warms the page cache), end-to-end runs of a 100 M-read file alternating
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call6; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 600 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu --timeout 300 2>&1 | tail -6 ) > $OUT/pytest_ingest.log 2>&1
cat $OUT/pytest_ingest.log
python tools/make_bam.py /dev/shm/lz100.bam 100000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/lz100.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/lz.tsv"
$CMD 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for v in 2 1; do
  COVERM_LZ_V=$v COVERM_NO_FAST_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lz$v -- $CMD > $OUT/trace_lz$v.log 2>&1
  f=$(find $OUT/trace_lz$v -name "*kernel_stats.csv" | head -1); echo "== COVERM_LZ_V=$v"; grep -E "k_lz|k_inflate_wave|k_crc32|k_bam_extract" $f | sed 's/(.*)"//' | cut -d, -f1-8
done 2>&1 | tee $OUT/lz_kernel_times.log
cd $R
for rep in 1 2 3 4 5; do
  for v in 2 1; do
    sleep 2; COVERM_LZ_V=$v COVERM_CLI_TIMING=1 $CMD 2> $OUT/e2e_err.log
    echo "COVERM_LZ_V=$v $(grep -o 'ingest (decode+push) [0-9.]*s' $OUT/e2e_err.log) $(grep -o 'main: .*' $OUT/e2e_err.log | cut -c1-90) $(md5sum /dev/shm/lz.tsv | cut -c1-8)"
  done
done 2>&1 | tee $OUT/lz_e2e_100M.log
rm -f /dev/shm/lz100.bam /dev/shm/lz.tsv
find $OUT -name "*kernel_trace.csv" -size +8M -delete
