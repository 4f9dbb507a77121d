#!/bin/bash
# round 6, call 44: ingest windows sized by the file — tests, then the 2 M-read runs again (samples 0.09 / 0.15 s before)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call44
timeout 1500 python -m pytest $R/tests/test_gpu_ingest.py $R/tests/test_gpu_bounded_store.py $R/tests/test_cli_binary.py $R/tests/test_gpu_configs.py $R/tests/test_gpu_pair_filter.py $R/tests/test_genes.py -q -m gpu -x 2>&1 | tail -9 | head -4 | tee $R/gpurun_out/r06_call44/pytest.log
python $R/tools/make_bam.py /dev/shm/s.bam 2000000 8 > /dev/null 2>&1
for i in $(seq 1 24); do
  COVERM_CLI_TIMING=1 $R/coverm_amd/coverm-amd contig -b /dev/shm/s.bam -m mean -t 16 -o /dev/shm/s.tsv 2>&1 | grep "windows of\|main:" | sed 's/.*device allocations \([0-9.]*\)s.*/alloc \1/; s/.*device sessions \([0-9.]*\)s, samples \([0-9.]*\)s.*/sessions \1 samples \2/' | tr '\n' ' '
  echo; sleep 0.7
done | tee $R/gpurun_out/r06_call44/small_runs.log
rm -f /dev/shm/s.bam /dev/shm/s.tsv
