#!/bin/bash
# round 6, call 42: which stamp of cov_create is late in the runs that start slowly (about one in eight: sessions 0.2 s instead of 0.07 s)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call42
python $R/tools/make_bam.py /dev/shm/s.bam 2000000 8 > /dev/null 2>&1
for i in $(seq 1 40); do
  COVERM_CLI_TIMING=1 $R/coverm_amd/coverm-amd contig -b /dev/shm/s.bam -m mean -t 16 -o /dev/shm/s.tsv 2>&1 | grep "cov_create\|main:" | sed 's/\[covermhip\] cov_create: //; s/\[coverm-amd\] main: //' | tr '\n' '|'
  echo
  sleep 0.7
done | tee $R/gpurun_out/r06_call42/create_stamps.log | awk -F'|' '{print $1" | "$4" | "$5" | "$6}' | cut -c1-250
rm -f /dev/shm/s.bam /dev/shm/s.tsv
