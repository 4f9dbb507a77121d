#!/bin/bash
# HISTORICAL (round 3): COVERM_NO_CRC_PROBE no longer exists; COVERM_INFLATE_ABLATE is still read by the lane-per-block kernel only.
# k_inflate ablations (results are wrong by construction; only the kernel time matters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ab in 0 1 2 3; do
  OUT=$R/gpurun_out/prof_ablate$ab
  mkdir -p $OUT
  COVERM_INFLATE_ABLATE=$ab COVERM_NO_CRC_PROBE=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/tools/ingest_probe.py 20000000 16 > $OUT/log.txt 2>&1
  find $OUT -name "*kernel_trace.csv" -delete
done
