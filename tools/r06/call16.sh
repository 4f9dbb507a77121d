#!/bin/bash
# round 6, call 16: the step-shape parity cases (tests/test_gpu_step_shapes.py) alone
mkdir -p gpurun_out/r06_call16
timeout 1200 python -m pytest tests/test_gpu_step_shapes.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r06_call16/pytest.log
cat gpurun_out/r06_call16/pytest.log
