#!/bin/bash
# rocprofv3 over bench.py (same command the driver runs, minus the CPU / end-to-end legs): kernel trace + stats, then separate PMC passes.
# usage: tools/prof_bench.sh <tag> [bench args...]      -> gpurun_out/profbench_<tag>/ ; then tools/summarize_prof.py <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/profbench_$TAG
mkdir -p $OUT
export COVERM_BENCH_CACHE=/dev/shm
CMD="python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 $@"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
[ "${PROF_TRAFFIC_ONLY:-0}" = "1" ] && { tail -2 $OUT/trace.log; rm -f /dev/shm/reads_*.npz; exit 0; }      # (kernel stats + HBM traffic only: three passes)
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 150 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
timeout 150 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH --output-format csv -d $OUT/pmc_sq3 -- $CMD > $OUT/pmc_sq3.log 2>&1
# keep only summaries small enough to merge back
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +16M -exec sh -c 'head -4000 "$1" > "$1.head"; rm "$1"' _ {} \;
tail -2 $OUT/trace.log
rm -f /dev/shm/reads_*.npz
