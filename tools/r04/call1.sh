#!/bin/bash
# First GPU call of round 4 (≈ 60 s of box time): price the store shapes, time the four Sink kernels against k_inflate, and run the ingest
# tests under COVERM_INFLATE_V=3.   gpurun --timeout 420 -- tools/r04/call1.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call1; mkdir -p $OUT
cd $R
timeout 60 tools/ubench/store_probe > $OUT/store_probe.log 2>&1
timeout 150 tools/r03/wave_variants.sh r04_call1 20000000 "v1:X=0 v1_lz2:COVERM_LZ_UNROLL=2 v1_lz4:COVERM_LZ_UNROLL=4 v1_lz8:COVERM_LZ_UNROLL=8 v3_st1:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=1 v3_st2:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=2 v3_st3:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=3 v3_st4:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=4 v3_st5:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=5 v3_st6:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=6 v3_st7:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=7 v3_st2_cur2:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=2,COVERM_INFLATE_WAVE_CURSOR=2 v3_st5_cur2:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=5,COVERM_INFLATE_WAVE_CURSOR=2 v3_st5_pass2:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=5,COVERM_INFLATE_ABLATE=3 v3_st5_cur2_pass2:COVERM_INFLATE_V=3,COVERM_INFLATE_WAVE_STORES=5,COVERM_INFLATE_WAVE_CURSOR=2,COVERM_INFLATE_ABLATE=3" > /dev/null 2>&1
cd $R
( COVERM_INFLATE_V=3 timeout 240 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --timeout 120 2>&1 | tail -5 ) > $OUT/pytest_v3.log 2>&1
# replicated histogram bins in k_pileup_fast: parity (the coverage parity tests) and time
for r in 2 4; do
  ( COVERM_PILEUP_HREP=$r timeout 200 python -m pytest tests/test_gpu_abi_parity.py -m gpu -x -q --timeout 120 2>&1 | tail -3 ) > $OUT/pytest_hrep$r.log 2>&1
  ( COVERM_PILEUP_HREP=$r timeout 120 python bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>&1 | tail -2 ) > $OUT/bench_hrep$r.log 2>&1
done
cat $OUT/store_probe.log $OUT/variants.log $OUT/pytest_v3.log $OUT/pytest_hrep*.log $OUT/bench_hrep*.log
