#!/bin/bash
# round 5, the last seconds of the GPU budget: the SQ counters of the same command (two more passes into profbench_r05f)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profbench_r05f; mkdir -p $OUT
export COVERM_BENCH_CACHE=/dev/shm
CMD="python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2"
timeout 45 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 25 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*counter_collection.csv" -size +16M -exec sh -c 'head -4000 "$1" > "$1.head"; rm "$1"' _ {} \;
ls $OUT/pmc_sq/*/ $OUT/pmc_sq2/*/ 2>&1 | tail -6
