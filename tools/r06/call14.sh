#!/bin/bash
# round 6, call 14: k_prep_generic with the next step's loads in flight; staging slots as the default for any number of feeders; tests + times
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_call14; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 1200 python -m pytest tests/test_gpu_abi_parity.py tests/test_gpu_estimates.py tests/test_gpu_bounded_store.py tests/test_gpu_configs.py tests/test_gpu_ingest.py -x -q -m gpu --timeout 300 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1; cat $OUT/pytest.log
for spec in "5000 1000000000 2000" "200000 1000000000 1000" "2000000 2000000000 1000"; do
  set -- $spec
  timeout 400 python tools/r06/kernel_ab.py --shapes 0 --rounds 2 --steps 5 --contigs $1 --bp $2 --min-len $3 --variant default= > $OUT/contigs_$1.log 2> $OUT/contigs_$1.err; echo "exit $?" >> $OUT/contigs_$1.log
  tail -4 $OUT/contigs_$1.log; tail -3 $OUT/contigs_$1.err
done
