"""gpurun_out/r06prof_<tag>/ (tools/r06/prof_ab.sh) -> profiles/r06_<tag>_kernel_stats.csv + profiles/r06_<tag>_pmc.json (per kernel, mean per
dispatch; traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "r06prof_" + tag)
dst = os.path.join(ROOT, "profiles")
ks = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
kst = {}
if ks:
    shutil.copy(ks[0], os.path.join(dst, "r06_%s_kernel_stats.csv" % tag))
    for row in csv.DictReader(open(ks[0])):
        kst[row["Name"].split("(")[0].replace("void ", "").strip()] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv*"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    d = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_traffic_bytes"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if k in kst:
        d["avg_us"], d["calls"] = kst[k]
    if d.get("SQ_BUSY_CU_CYCLES") and d.get("SQ_INSTS_VALU"):
        d["valu_per_simd_cycle_x4"] = 4.0 * d["SQ_INSTS_VALU"] / (4.0 * d["SQ_BUSY_CU_CYCLES"])      # 1.0 = every SIMD issues a 4-cycle VALU instruction back to back
    if d.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in d:
                d[c + "_frac_of_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    out[k] = d
json.dump(out, open(os.path.join(dst, "r06_%s_pmc.json" % tag), "w"), indent=1)
big = sorted(kst.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:12]
for k, (us, n) in big:
    d = out.get(k, {})
    print("%-60s %9.1f us x %4d  VALU %s SALU %s VMEM %s LDS %s  traffic %s" % (k[:60], us, n, *["%.1fM" % (d[c] / 1e6) if c in d else "-" for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS")],
          "%.3f GB" % (d["hbm_traffic_bytes"] / 1e9) if "hbm_traffic_bytes" in d else "-"))
