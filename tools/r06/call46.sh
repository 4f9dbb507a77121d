#!/bin/bash
# round 6, call 46: the timing lines of the binary over a 2 M-contig sample
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call46
timeout 900 python $R/tools/r06/two_million_timing.py 2>&1 | tee $R/gpurun_out/r06_call46/timing.log | tail -40
