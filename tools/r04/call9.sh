#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call9; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 200 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --timeout 90 -k "spans_leave or hands_irregular" 2>&1 | tail -60 ) > $OUT/pytest.log 2>&1
cat $OUT/pytest.log
