#!/bin/bash
# round 6, call 20: counters of the ingest kernels (one full round of 81 920 blocks + one of ~12 k: a 20 M-read file), for profiles/r06_ingest_*
R=$GRAFT_REPO_ROOT
PROF_PMC=1 bash $R/tools/prof_ingest.sh r06ingpmc 20000000
ls $R/gpurun_out/prof_r06ingpmc
