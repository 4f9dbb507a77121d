#!/bin/bash
# round 6, call 12: (b) can two RCCL ranks share one device?  (a) `coverm-amd --devices 0,0,0,0,0,0,0,0` at config-4 and
# config-5 size on the one-GPU box with the box's real CPU quota: eight feeders' read / registration stamps, threads per feeder, total wall —
# the host-side contention part of DESIGN section 7's model as a measurement (ONE device behind the eight feeders: not a scaling figure)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call12; mkdir -p $OUT; rm -rf $OUT/*; cd $R
grep -E "MemTotal|MemAvailable" /proc/meminfo | tee $OUT/meminfo.txt; df -h /dev/shm | tee -a $OUT/meminfo.txt; nproc | tee -a $OUT/meminfo.txt
timeout 400 python tools/r06/rccl_same_device_probe.py > $OUT/rccl_same_device_probe.json 2> $OUT/rccl_probe.err; tail -25 $OUT/rccl_same_device_probe.json; tail -3 $OUT/rccl_probe.err
AVAIL=$(awk '/MemAvailable/{print int($2/1048576)}' /proc/meminfo)
DEVS=0,0,0,0,0,0,0,0; [ "$AVAIL" -lt 220 ] && DEVS=0,0,0,0
echo "MemAvailable ${AVAIL} GB -> devices $DEVS" | tee -a $OUT/meminfo.txt
( COVERM_BENCH_MULTI_DEVICE_CHECK=$DEVS timeout 2400 python bench.py --no-cpu-baseline --time-limit 2400 > $OUT/bench_multi_feeders.json 2> $OUT/bench_multi_feeders.err ); echo "rc $?" >> $OUT/bench_multi_feeders.err
tail -c 3000 $OUT/bench_multi_feeders.json; tail -5 $OUT/bench_multi_feeders.err
( timeout 600 python -m pytest tests/test_gpu_abi_parity.py -x -q -m gpu -k "assemblies or two_table or prep_kernel" --timeout 300 2>&1 | tail -4 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
