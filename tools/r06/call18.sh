#!/bin/bash
# round 6, call 18: is the device the bound of the end-to-end run?  Kernel trace of coverm-amd over a 100 M-read BAM (device ingest)
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_ingest.sh r06ing 100000000
python $R/tools/r06/ingest_timeline.py "$R/gpurun_out/prof_r06ing/trace/**/*kernel_trace.csv" $R/gpurun_out/prof_r06ing/timeline.json | head -60
cat $R/gpurun_out/prof_r06ing/plain_run.log | tail -8
