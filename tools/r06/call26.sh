#!/bin/bash
# round 6, call 26: what is resident when coverm-amd ends (its largest mappings), 50 M reads
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call26
python $R/tools/make_bam.py /dev/shm/m.bam 50000000 16 > /dev/null 2>&1
for i in 1 2 3; do
COVERM_CLI_TIMING=1 $R/coverm_amd/coverm-amd contig -b /dev/shm/m.bam -m mean -t 16 -o /dev/shm/m.tsv 2>&1 | grep "mapping\|Rss\|VmHWM\|wall clock" 
sleep 2
done | tee $R/gpurun_out/r06_call26/mappings.log
rm -f /dev/shm/m.bam /dev/shm/m.tsv
