#!/bin/bash
# round 6, call 27: GPU_MAX_HW_QUEUES against the process's start and end (each queue's 173 MB save area)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call27
timeout 1200 python $R/tools/r06/queues_probe.py 100000000 $R/gpurun_out/r06_call27/queues_probe.json 2>&1 | tee $R/gpurun_out/r06_call27/queues_probe.log | tail -60
