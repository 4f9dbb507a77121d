#!/bin/bash
# round 6, call 36: the whole -m gpu suite and smoke() on HEAD (after the launcher's pid check and the launcher tests)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call36; cd $R
( timeout 1500 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -6 ) > gpurun_out/r06_call36/pytest_gpu.log 2>&1; cat gpurun_out/r06_call36/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -1
ps -eo stat,comm | grep -c "^Z.*coverm" || true
