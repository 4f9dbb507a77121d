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
short = {"k_prep<": "k_prep", "k_ranges<": "k_ranges", "k_pileup_fast": "k_pileup", "k_pileup_stream": "k_pileup_slow_tiles", "k_cx_expand": "k_cx_expand",
         "k_tile_scan": "k_tile_scan"}
for k, cs in summ.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        b = (2 * cs["FETCH_SIZE"]["mean_per_dispatch"] + cs["WRITE_SIZE"]["mean_per_dispatch"]) * 1024
        for pat, key in short.items():
            if pat in k:
                traffic[key] = traffic.get(key, 0) + int(b)
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, tag + "_bench_kernel_stats.csv")).read()[:3000])
print(json.dumps(traffic, indent=1))
