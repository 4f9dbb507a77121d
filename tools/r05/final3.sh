#!/bin/bash
# round 5, with what was left of the GPU budget: kernel stats and HBM traffic counters of `python bench.py --no-cpu-baseline` on the build with
# k_prep8s (tag r05f) — three rocprofv3 passes
R=$GRAFT_REPO_ROOT; cd $R
PROF_TRAFFIC_ONLY=1 timeout 100 tools/prof_bench.sh r05f > $R/gpurun_out/prof_bench_r05f.log 2>&1
tail -3 $R/gpurun_out/prof_bench_r05f.log; ls $R/gpurun_out/profbench_r05f/
grep -h "k_prep\|k_pileup" $R/gpurun_out/profbench_r05f/trace/*/*kernel_stats.csv | head -5
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/profbench_r05f/*.log | head
