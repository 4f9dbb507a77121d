#!/bin/bash
# round 6, call 30: the device's timeline under the faster feed (100 M reads, kernel trace)
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_ingest.sh r06ing3 100000000
python $R/tools/r06/ingest_timeline.py "$R/gpurun_out/prof_r06ing3/trace/**/*kernel_trace.csv" $R/gpurun_out/prof_r06ing3/timeline.json | head -12
