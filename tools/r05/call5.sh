#!/bin/bash
# round 5, fifth GPU call: timeline of a 100 M-read ingest (kernels + copies) with k_lz_stage and with k_lz_resolve
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call5; mkdir -p $OUT; rm -rf $OUT/*; cd $R
python tools/make_bam.py /dev/shm/lz100.bam 100000000 16 > $OUT/make.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/lz100.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/lz.tsv"
cd /tmp && export TMPDIR=/tmp
for v in 2 1; do
  COVERM_LZ_V=$v COVERM_NO_FAST_EXIT=1 COVERM_CLI_TIMING=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tl_lz$v -- $CMD > $OUT/tl_lz$v.log 2>&1
done
rm -f /dev/shm/lz100.bam /dev/shm/lz.tsv
find $OUT -name "*.csv" -size +12M -delete
ls -la $OUT/tl_lz2/*/
