"""Runs tools/ubench/valu_rate (built with hipcc from valu_rate.hip if missing) and writes its table as JSON for tools/valu_mix.py:
    python tools/valu_rate_json.py profiles/r03_valu_rate.json [waves_per_simd=4]"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tools", "ubench", "valu_rate")
if not os.path.exists(exe):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "ubench", "valu_rate.hip"), "-o", exe])
out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
wps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
names = {"v_fma_f32": "v_fma_f32", "v_add_u32": "v_add_u32", "v_mad_u32_u24": "v_mad_u32_u24", "v_add_u32_sdwa sext": "sdwa", "min3/max3/add3": "min3/max3/add3",
         "v_add_u32_dpp row_shr": "dpp", "v_cmp + v_cndmask": "cmp+cndmask", "cmp/saveexec/2 valu/or (x5)": "predicated block", "ds_add_u32 random": "ds_add_u32",
         "s_bcnt1 + s_add": "salu"}
table, allw, ghz = {}, {}, None
for l in out.splitlines():
    m = re.match(r"device .* clock ([\d.]+) GHz", l)
    if m: ghz = float(m.group(1))
    m = re.match(r"(.+?)\s+waves/SIMD (\d+) : ([\d.]+) ms, ([\d.]+) ns per wave-instruction per SIMD = ([\d.]+) cycles", l)
    if m:
        k = names.get(m.group(1).strip(), m.group(1).strip())
        allw.setdefault(k, {})[m.group(2)] = float(m.group(5))
        if int(m.group(2)) == wps: table[k] = float(m.group(5))
json.dump({"source": "tools/ubench/valu_rate.hip on this GPU", "clock_ghz": ghz, "waves_per_simd": wps, "cycles_per_wave_instruction": table,
           "cmp_cndmask_is_pair": False, "all_occupancies": allw, "raw": out.splitlines()}, open(sys.argv[1], "w"), indent=1)
print(json.dumps(table))
