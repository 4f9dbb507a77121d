#!/bin/bash
# round 6, call 39: how long until the first piece's upload call (the ingest's start-up), 200 M reads, default against pread, two rounds
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_call39
timeout 1200 python $R/tools/r06/feed_ab.py 200000000 2 $R/gpurun_out/r06_call39/feed_ab.json 2>&1 | tee $R/gpurun_out/r06_call39/feed_ab.log | grep "^{'mode'" | sed "s/'bytes_from.*//; s/'vm_hwm_mb.*'sessions_s'/'sessions_s'/" | cut -c1-360
