"""gpurun_out/profbench_<tag>/ (written by tools/prof_bench.sh) -> profiles/<tag>_bench_kernel_stats.csv,
profiles/<tag>_bench_pmc_summary.json and profiles/pmc_traffic.json (HBM bytes per launch per kernel)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "profbench_" + tag)
if not os.path.isdir(src):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)      # tools/prof_cmd.sh
dst = os.path.join(ROOT, "profiles")
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
assert ks, "no kernel_stats.csv under " + src
shutil.copy(ks[0], os.path.join(dst, tag + "_bench_kernel_stats.csv"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv*"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
summ = {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in sorted(cs.items())}
        for k, cs in sorted(acc.items())}
json.dump(summ, open(os.path.join(dst, tag + "_bench_pmc_summary.json"), "w"), indent=1)
traffic = {"_source": tag, "_formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB -> bytes, per launch (MI355X_MICROARCH.md, gfx950 correction)"}
short = {"k_prep_lean<": "k_prep", "k_prep_generic<": "k_prep", "k_prep<": "k_prep", "k_prep6<": "k_prep", "k_prep5p<": "k_prep", "k_prep8s<": "k_prep", "k_prep7s<": "k_prep", "k_post_prep": "k_post_prep", "k_estimate": "k_estimate", "k_ranges<": "k_ranges", "k_pileup_fast": "k_pileup", "k_pileup_stream": "k_pileup_slow_tiles", "k_cx_expand": "k_cx_expand",
         "k_tile_scan": "k_tile_scan"}
for k, cs in summ.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        b = (2 * cs["FETCH_SIZE"]["mean_per_dispatch"] + cs["WRITE_SIZE"]["mean_per_dispatch"]) * 1024
        for pat, key in short.items():
            if pat in k:
                traffic[key] = traffic.get(key, 0) + int(b)
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
# pipe utilisation of the two big kernels (bench.py reports these as the on-chip roofline of k_pileup_fast)
pipes = {"_source": tag, "_units": "SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles; SQ_BUSY_CU_CYCLES counts cycles per CU (x4 SIMDs = SIMD-cycles); "
         "the VALU issue fraction is NOT taken from SQ_ACTIVE_INST_VALU (it ticks once per instruction, VERDICT round 2): tools/valu_mix.py derives it "
         "from SQ_INSTS_VALU, the kernel's instruction classes and the measured per-class issue costs; the two valu_frac_if_* entries are its bounds"}
kstats = {}
for row in csv.DictReader(open(ks[0])):
    kstats[row["Name"].split("(")[0].replace("void ", "").strip()] = float(row["AverageNs"]) / 1e6
for k, cs in summ.items():
    key = "k_pileup" if "k_pileup_fast" in k else "k_prep" if ("k_prep_lean<" in k or "k_prep<" in k or "k_prep6<" in k or "k_prep5p<" in k or "k_prep8s<" in k or "k_prep7s<" in k) else None
    if not key or "SQ_BUSY_CU_CYCLES" not in cs:
        continue
    g = lambda c: cs.get(c, {}).get("mean_per_dispatch")
    simd_cycles = 4.0 * g("SQ_BUSY_CU_CYCLES")
    p_ = {"kernel": k, "simd_cycles": simd_cycles}
    for kn, ms in kstats.items():
        if kn.split("<")[0] in k and (("<" not in kn) or kn == k):
            p_["kernel_ms"] = ms
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_INSTS_VALU"):
        p_["lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_INSTS_VALU"))      # active lanes per VALU instruction / 64
    if g("SQ_INSTS_VALU"):
        p_["valu_insts"] = g("SQ_INSTS_VALU")
        p_["valu_frac_if_all_full_rate"] = 2.0 * g("SQ_INSTS_VALU") / simd_cycles
        p_["valu_frac_if_all_half_rate"] = 4.0 * g("SQ_INSTS_VALU") / simd_cycles
    if g("SQ_LDS_IDX_ACTIVE"):
        p_["lds_busy_frac"] = g("SQ_LDS_IDX_ACTIVE") / g("SQ_BUSY_CU_CYCLES")
        p_["lds_bank_conflict_frac_of_lds_cycles"] = (g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_ACTIVE_INST_SCA"):
        p_["scalar_busy_frac"] = 4.0 * g("SQ_ACTIVE_INST_SCA") / simd_cycles
    if g("SQ_WAVE_CYCLES"):
        for c, n in (("SQ_WAIT_ANY", "wave_time_waiting_on_waitcnt"), ("SQ_WAIT_INST_ANY", "wave_time_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "wave_time_issuing")):
            if g(c) is not None:
                p_[n] = g(c) / g("SQ_WAVE_CYCLES")
        p_["waves_per_simd_avg"] = 4.0 * g("SQ_WAVE_CYCLES") / simd_cycles
    for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_LDS_ATOMIC"):
        if g(c) is not None:
            p_[c.lower()] = g(c)
    pipes[key] = p_
json.dump(pipes, open(os.path.join(dst, "pmc_pipes.json"), "w"), indent=1)
print(open(os.path.join(dst, tag + "_bench_kernel_stats.csv")).read()[:3000])
print(json.dumps(traffic, indent=1))
