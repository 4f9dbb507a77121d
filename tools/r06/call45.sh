#!/bin/bash
# round 6, call 45: the contig sweep again (the binary's table at 200 k / 2 M contigs with the threaded printer)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call45
timeout 1700 python $R/tools/r06/contig_sweep.py --out $R/gpurun_out/r06_contig_sweep.json > $R/gpurun_out/r06_call45/sweep.log 2> $R/gpurun_out/r06_call45/sweep.err; grep "wall_s" $R/gpurun_out/r06_call45/sweep.log | cut -c1-700
