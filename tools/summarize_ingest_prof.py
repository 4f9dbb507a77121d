"""gpurun_out/prof_<tag>/ (written by PROF_PMC=1 tools/prof_ingest.sh <tag> 20000000) -> profiles/<out>_ingest_kernel_stats.csv and
profiles/<out>_ingest_pmc_summary.json.

    python tools/summarize_ingest_prof.py r05ing r05

A 20 M-read level-1 BAM is one full round of 81 920 BGZF blocks and one of ~12 k: per kernel and counter the LARGER of its two
dispatches is the full round (`derived_full_round`); HBM bytes follow MI355X_MICROARCH.md (FETCH_SIZE / WRITE_SIZE count KiB;
gfx950 correction: reads = 2 x FETCH_SIZE)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
assert ks, "no kernel_stats.csv under " + src
shutil.copy(ks[0], os.path.join(dst, out + "_ingest_kernel_stats.csv"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv*"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
counters = {k: {c: {"max_dispatch": max(v), "sum": sum(v), "dispatches": len(v)} for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
derived = {}
for k, cs in counters.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f, w = cs["FETCH_SIZE"]["max_dispatch"], cs["WRITE_SIZE"]["max_dispatch"]
        derived[k] = {"full_round_blocks": 81920, "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w, "hbm_read_bytes_gfx950_corrected": int(2 * f * 1024),
                      "hbm_write_bytes": int(w * 1024)}
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CU_CYCLES"):
            if c in cs:
                derived[k][c] = cs[c]["max_dispatch"]
json.dump({"_source": tag, "_units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; derived_full_round = the larger dispatch of each kernel (81 920 blocks)",
           "_algorithmic_per_full_round": {"covi::k_inflate_wave": 81920 * (21100 + 62900), "covi::k_lz_resolve": 81920 * (5900 * 2 + 2 * 50600)},
           "derived_full_round": derived, "counters": counters}, open(os.path.join(dst, out + "_ingest_pmc_summary.json"), "w"), indent=1)
print("wrote", out + "_ingest_kernel_stats.csv", out + "_ingest_pmc_summary.json", "for", len(derived), "kernels")
