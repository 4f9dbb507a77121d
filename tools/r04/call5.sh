#!/bin/bash
# Who shares the device with whom: the timeline of the ingest kernels (k_inflate_wave fills every wave slot of a CU: do k_lz_resolve /
# k_crc32 / the parse kernels of the window before run beside it at all?), end-to-end time with k_inflate_wave held to fewer resident
# waves by unused LDS, and with more hardware queues than the runtime's default four (the session has five streams).
#      gpurun --timeout 420 -- tools/r04/call5.sh
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04_call5; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout 150 python tools/make_bam.py /dev/shm/e2e.bam 100000000 16 > $OUT/make.log 2>&1
ls -la /dev/shm/e2e.bam >> $OUT/make.log
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/e2e.bam -m mean trimmed_mean covered_fraction variance count --min-read-percent-identity 95 --min-read-aligned-length 50 --proper-pairs-only -t 16 -o /dev/shm/e2e.tsv"
run() {  # name, env...
  name=$1; shift
  for rep in 1 2 3; do
    rm -f /dev/shm/e2e.tsv
    s=$(date +%s%N); env "$@" COVERM_CLI_TIMING=1 timeout 30 $CMD 2> /tmp/err.log; e=$(date +%s%N)
    echo "$name rep $rep: wall $(( (e - s) / 1000000 )) ms | $(grep -h 'device ingest: buffers' /tmp/err.log | sed 's/.*file read/file read/') | $(md5sum /dev/shm/e2e.tsv | cut -c1-8)" >> $OUT/sweep.log
  done
}
run "pad 0" X=0
run "pad 3" COVERM_INFLATE_WAVE_PAD_KB=3
run "pad 6" COVERM_INFLATE_WAVE_PAD_KB=6
run "queues 8" GPU_MAX_HW_QUEUES=8
run "queues 8 pad 6" GPU_MAX_HW_QUEUES=8 COVERM_INFLATE_WAVE_PAD_KB=6
cd /tmp && export TMPDIR=/tmp
tl() {  # name, env...
  name=$1; shift
  rm -rf /tmp/tl_$name
  env "$@" COVERM_NO_FAST_EXIT=1 timeout 60 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$name -- $CMD > /tmp/tl_$name.log 2>&1
  f=$(find /tmp/tl_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $OUT/timeline_$name.txt <<PY
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(); t0 = rows[0][0]
print("# start ms, end ms, duration ms, queue, kernel (kernels longer than 0.3 ms)")
for s, e, q, k in rows:
    if e - s > 300000: print("%9.2f %9.2f %8.2f  q%s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, k[:40]))
PY
}
tl default X=0
tl queues8 GPU_MAX_HW_QUEUES=8
cat $OUT/make.log $OUT/sweep.log; sed -n 1,70p $OUT/timeline_default.txt; echo ----; sed -n 1,70p $OUT/timeline_queues8.txt
rm -f /dev/shm/e2e.bam /dev/shm/e2e.tsv
