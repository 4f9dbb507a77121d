#!/bin/bash
# rocprofv3 over bench.py (same command the driver runs): kernel trace + stats, then separate PMC passes.
# usage: tools/prof_bench.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/profbench_$TAG
mkdir -p $OUT
CMD="python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 $@"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
# keep only summaries small enough to merge back
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +16M -exec sh -c 'head -2000 "$1" > "$1.head"; rm "$1"' _ {} \;
tail -2 $OUT/trace.log
ls -laR $OUT | head -40
