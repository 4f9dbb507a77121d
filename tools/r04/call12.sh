#!/bin/bash
# two one-line experiments: k_inflate_wave held to 96 registers (five waves per SIMD), k_lz_resolve's source loads at workgroup scope (L1 hits allowed)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call12; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout 150 tools/r03/wave_variants.sh r04_call12 20000000 "base:X=0 waves5:COVERM_INFLATE_WAVES=5 lz_wg:COVERM_LZ_SCOPE=wg both:COVERM_INFLATE_WAVES=5,COVERM_LZ_SCOPE=wg base2:X=0" > /dev/null 2>&1
( COVERM_INFLATE_WAVES=5 COVERM_LZ_SCOPE=wg timeout 200 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --timeout 60 -k "default" 2>&1 | tail -4 ) > $OUT/pytest.log 2>&1
cat $OUT/variants.log $OUT/pytest.log
