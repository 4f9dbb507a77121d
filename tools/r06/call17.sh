#!/bin/bash
# round 6, call 17: the whole -m gpu suite after the knobs moved into COVERM_KNOBS
mkdir -p gpurun_out/r06_call17
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -15 > gpurun_out/r06_call17/pytest_gpu.log
cat gpurun_out/r06_call17/pytest_gpu.log
