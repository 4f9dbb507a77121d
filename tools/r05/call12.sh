#!/bin/bash
# round 5: a literal behind a match in the same lock-step of k_inflate_wave (16am) against the default
# kernel times of a 20 M-read file (one full round), end-to-end runs of a 100 M-read file alternating
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call12; mkdir -p $OUT; rm -rf $OUT/*; cd $R


python tools/make_bam.py /dev/shm/lz100.bam 100000000 16 > $OUT/make.log 2>&1
python tools/make_bam.py /dev/shm/lz20.bam 20000000 16 >> $OUT/make.log 2>&1
CMD20="$R/coverm_amd/coverm-amd contig -b /dev/shm/lz20.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/lz.tsv"
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/lz100.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/lz.tsv"
$CMD 2>/dev/null; $CMD20 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for v in 16 16am; do
  COVERM_INFLATE_SINK=$v COVERM_NO_FAST_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sink$v -- $CMD20 > $OUT/trace_sink$v.log 2>&1
  f=$(find $OUT/trace_sink$v -name "*kernel_stats.csv" | head -1); echo "== COVERM_INFLATE_SINK=$v"; grep -E "k_lz|k_inflate_wave|k_crc32|k_bam_extract" $f | sed 's/(.*)"//' | cut -d, -f1-8
done 2>&1 | tee $OUT/sink_kernel_times.log
cd $R
for rep in 1 2 3 4 5; do
  for v in 16 16am; do
    sleep 2; COVERM_INFLATE_SINK=$v COVERM_CLI_TIMING=1 $CMD 2> $OUT/e2e_err.log
    echo "COVERM_INFLATE_SINK=$v $(grep -o 'ingest (decode+push) [0-9.]*s' $OUT/e2e_err.log) $(grep -o 'main: .*' $OUT/e2e_err.log | cut -c1-90) $(md5sum /dev/shm/lz.tsv | cut -c1-8)"
  done
done 2>&1 | tee $OUT/sink_e2e_100M.log
rm -f /dev/shm/lz100.bam /dev/shm/lz20.bam /dev/shm/lz.tsv
find $OUT -name "*kernel_trace.csv" -size +8M -delete
