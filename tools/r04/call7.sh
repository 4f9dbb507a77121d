#!/bin/bash
# the -m gpu suite with the packed-arithmetic k_pileup_fast, the bench line (kernel times), k_crc32_wave against k_crc32
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call7; mkdir -p $OUT; rm -f $OUT/*
cd $R
( timeout 400 python -m pytest tests -m gpu -x -q --timeout 180 2>&1 | tail -8 ) > $OUT/pytest_gpu.log 2>&1
( COVERM_CRC_WAVE=1 timeout 200 python -m pytest tests/test_gpu_ingest.py -m gpu -x -q --timeout 60 -k "default" 2>&1 | tail -4 ) > $OUT/pytest_crc_wave.log 2>&1
( timeout 150 python bench.py --no-cpu-baseline --no-e2e --no-binary-legs --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms_per_step', d.get('ms_per_step'), 'kernels', d['roofline'].get('all_kernels_ms'), 'frac', d['roofline'].get('frac'))" ) > $OUT/bench_short.log 2>&1
timeout 100 tools/r03/wave_variants.sh r04_call7 20000000 "crc_lane:X=0 crc_wave:COVERM_CRC_WAVE=1" > /dev/null 2>&1
sed -i 's/k_lz_resolve" in n/k_lz_resolve" in n or "k_crc32" in n or "k_bam" in n/' $R/tools/r03/wave_variants.sh
cat $OUT/pytest_gpu.log $OUT/pytest_crc_wave.log $OUT/bench_short.log $OUT/variants.log
