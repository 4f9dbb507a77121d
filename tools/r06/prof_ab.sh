#!/bin/bash
# rocprofv3 over tools/r06/kernel_ab.py with ONE variant: kernel trace + stats, then separate PMC passes (never combined with a trace domain).
# usage: tools/r06/prof_ab.sh <tag> <kernel_ab args...>   -> gpurun_out/r06prof_<tag>/ ; summary: python tools/r06/summ.py <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/r06prof_$TAG
mkdir -p $OUT; rm -rf $OUT/*
CMD="python $R/tools/r06/kernel_ab.py --shapes 0 --rounds 1 --steps 10 $@"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH"; do
  n=$(echo $grp | cut -d' ' -f1)
  [ -n "$PROF_SKIP_PMC" ] && break
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$n -- $CMD > $OUT/pmc_$n.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +16M -exec sh -c 'head -6000 "$1" > "$1.head"; rm "$1"' _ {} \;
find $OUT -name "*agent_info.csv" -delete
tail -3 $OUT/trace.log
