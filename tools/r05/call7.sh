#!/bin/bash
# round 5, seventh GPU call: per instruction or per lane? (store probe); slicing-by-8 CRC and the mapped-file mode under the ingest tests; kernel times
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call7; mkdir -p $OUT; rm -rf $OUT/*; cd $R
timeout 120 tools/ubench/store_probe > $OUT/store_probe.log 2>&1; tail -9 $OUT/store_probe.log
( timeout 700 python -m pytest tests/test_gpu_ingest.py tests/test_cli_binary.py -x -q -m gpu --timeout 300 2>&1 | tail -6 ) > $OUT/pytest_ingest.log 2>&1
cat $OUT/pytest_ingest.log
python tools/make_bam.py /dev/shm/lz20.bam 20000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/lz20.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/lz.tsv"
$CMD 2>/dev/null
cd /tmp && export TMPDIR=/tmp
COVERM_NO_FAST_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); grep -E "k_lz|k_inflate_wave|k_crc32|k_bam" $f | sed 's/(.*)"//' | cut -d, -f1-8 | tee $OUT/kernel_times.log
rm -f /dev/shm/lz20.bam /dev/shm/lz.tsv
find $OUT -name "*kernel_trace.csv" -size +8M -delete
