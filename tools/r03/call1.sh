#!/bin/bash
# round 3, GPU call 1: GPU suite after the ADVICE fixes, file -> HBM probes, e2e baseline at 50 M reads
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03_call1; mkdir -p $OUT
cd $R
nproc > $OUT/box.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/box.txt; free -g >> $OUT/box.txt; df -h /dev/shm >> $OUT/box.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
python tools/make_bam.py /dev/shm/p.bam 50000000 16 > $OUT/make.log 2>&1
timeout 300 tools/ubench/io_probe /dev/shm/p.bam 16 > $OUT/io_probe.log 2>&1
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/p.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/p.tsv"
for i in 1 2 3; do sleep 3; /usr/bin/time -f "wall %e s" env COVERM_CLI_TIMING=1 $CMD 2>> $OUT/e2e_50M.log; done
rm -f /dev/shm/p.bam /dev/shm/p.tsv
cat $OUT/io_probe.log; grep -h "wall\|device ingest:\|main:" $OUT/e2e_50M.log
