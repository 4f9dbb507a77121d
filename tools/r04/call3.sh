#!/bin/bash
# Counters of k_inflate_wave (third GPU call of round 4, ≈ 40 s): instruction mix, waits, and what its stores become on the way to L2 / HBM.
#   gpurun --timeout 200 -- tools/r04/call3.sh [stores] [cursor]
R=$GRAFT_REPO_ROOT; ST=${1:-2}; CUR=${2:-1}; OUT=$R/gpurun_out/r04_call3; mkdir -p $OUT
cd $R
timeout 60 python tools/make_bam.py /dev/shm/ikt.bam 20000000 16 > $OUT/make.log 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="$R/coverm_amd/coverm-amd contig -b /dev/shm/ikt.bam -m mean trimmed_mean covered_fraction variance -t 16 -o /dev/shm/ikt.tsv"
export COVERM_INFLATE_V=3 COVERM_INFLATE_WAVE_STORES=$ST COVERM_INFLATE_WAVE_CURSOR=$CUR COVERM_NO_FAST_EXIT=1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE WRITE_SIZE" \
           "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_REQ_sum TCC_WRITE_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 30 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- $CMD > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  echo "== $set" >> $OUT/pmc.log
  if [ -n "$f" ]; then python - "$f" >> $OUT/pmc.log <<PY
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_inflate" in k or "k_lz_resolve" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print("   %-40s %s" % (k[:40], "  ".join("%s %.4g/launch" % (c, v / max(1, n[(k, c)])) for c, v in acc[k].items())))
PY
  else tail -3 /tmp/pmc_$i.log >> $OUT/pmc.log; fi
done
cat $OUT/pmc.log
rm -f /dev/shm/ikt.bam /dev/shm/ikt.tsv
