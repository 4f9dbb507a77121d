#!/bin/bash
# round 5: k_bam_extract writing its columns four records at a time (fields buffered in LDS): tests, kernel time and WRITE_SIZE
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_call14; mkdir -p $OUT; rm -rf $OUT/*; cd $R
( timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pair_filter.py tests/test_cli_binary.py tests/test_gpu_bounded_store.py tests/test_genes.py -x -q -m gpu --timeout 300 2>&1 | tail -6 ) > $OUT/pytest.log 2>&1
cat $OUT/pytest.log
PROF_PMC=1 timeout 900 tools/prof_ingest.sh r05ext2 20000000 > $OUT/prof_ingest.log 2>&1
tail -3 $OUT/prof_ingest.log
