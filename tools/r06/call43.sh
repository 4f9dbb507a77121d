#!/bin/bash
# round 6, call 43: a 2 M-read file takes 0.09 s or 0.15 s in `samples`, at random: every timing line of 16 runs
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call43
python $R/tools/make_bam.py /dev/shm/s.bam 2000000 8 > /dev/null 2>&1
for i in $(seq 1 16); do
  echo "== run $i"
  COVERM_CLI_TIMING=1 $R/coverm_amd/coverm-amd contig -b /dev/shm/s.bam -m mean -t 16 -o /dev/shm/s.tsv 2>&1 | grep -v "mapping \|Rss\|VmRSS\|In sample"
  sleep 0.7
done > $R/gpurun_out/r06_call43/runs.log 2>&1
rm -f /dev/shm/s.bam /dev/shm/s.tsv
grep -c "== run" $R/gpurun_out/r06_call43/runs.log
