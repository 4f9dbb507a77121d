#!/bin/bash
# round 6, call 22: the feed — threads moving a tmpfs file into page-locked memory: pread against populated mapping + (non-temporal) copy
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call22
head -c 9000000000 /dev/urandom > /dev/shm/copy_probe.bin
for t in 14 10; do $R/tools/ubench/copy_probe /dev/shm/copy_probe.bin $t 8; done 2>&1 | tee $R/gpurun_out/r06_call22/copy_probe.log
rm -f /dev/shm/copy_probe.bin
