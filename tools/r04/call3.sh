#!/bin/bash
# Third GPU call of round 4: k_inflate_lds (one workgroup per BGZF block, the block assembled in LDS) — the ingest tests with it, its launch
# times against k_inflate_wave + k_lz_resolve with the phases ablated, end to end at 200 M reads.
#   gpurun --timeout 600 -- tools/r04/call3.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call3; mkdir -p $OUT
cd $R
( timeout 300 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --timeout 120 -k "workgroup" 2>&1 | tail -8 ) > $OUT/pytest_lds.log 2>&1
timeout 150 tools/r03/wave_variants.sh r04_call3 20000000 "v3:X=0 v4:COVERM_INFLATE_V=4 v4_tables:COVERM_INFLATE_V=4,COVERM_INFLATE_ABLATE=1 v4_plan:COVERM_INFLATE_V=4,COVERM_INFLATE_ABLATE=2 v4_pass3:COVERM_INFLATE_V=4,COVERM_INFLATE_ABLATE=3" > /dev/null 2>&1
SWEEP_ONLY_LDS=1 timeout 200 python tools/r04/window_sweep.py 200000000 16 > $OUT/window_sweep.log 2>&1
cat $OUT/pytest_lds.log $OUT/variants.log $OUT/window_sweep.log
