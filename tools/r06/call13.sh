#!/bin/bash
# round 6, call 13: k_prep_generic's atomics batched over the runs of a step (tests, times at 200 000 / 2 000 000 contigs); eight feeders: mapped file or staging slots
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call13; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_estimates.py tests/test_gpu_bounded_store.py -x -q -m gpu --timeout 300 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
for spec in "5000 1000000000 2000" "200000 1000000000 1000" "2000000 2000000000 1000"; do
  set -- $spec
  timeout 400 python tools/r06/kernel_ab.py --shapes 0 --rounds 2 --steps 5 --contigs $1 --bp $2 --min-len $3 --variant default= --variant k_prep7s=COVERM_PREP_KERNEL:7 > $OUT/contigs_$1.log 2> $OUT/contigs_$1.err; echo "exit $?" >> $OUT/contigs_$1.log
  tail -5 $OUT/contigs_$1.log; tail -3 $OUT/contigs_$1.err
done
timeout 1500 python tools/r06/eight_feeders_io.py > $OUT/eight_feeders_io.json 2> $OUT/eight_feeders_io.err; python -c "import json; d=json.load(open('$OUT/eight_feeders_io.json')); print(d['median_wall_s'])"; tail -3 $OUT/eight_feeders_io.err
