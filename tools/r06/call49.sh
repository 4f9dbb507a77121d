#!/bin/bash
# round 6, call 49: six 50 M-read files on one GPU, one session against two sessions on the device
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call49
timeout 1200 python $R/tools/r06/two_sessions_per_device.py 6 50000000 $R/gpurun_out/r06_call49/two_sessions.json 2>&1 | tee $R/gpurun_out/r06_call49/two_sessions.log | tail -24
