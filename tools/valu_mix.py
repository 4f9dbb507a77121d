"""VALU instruction mix of a kernel and the issue-cycle fraction that follows from it.

    python tools/valu_mix.py --kernel k_pileup_fast7ILb1 [--asm /tmp/cov.s] [--rates profiles/r03_valu_rate.json]
                             [--pmc profiles/pmc_pipes.json --pmc-key k_pileup] [--weights loop|flat|<json>] [--json out.json]

What it does, in the order a reader who does not trust any issue-rate assumption would redo it:
  1. the kernel's ISA: `hipcc --offload-arch=gfx950 -O3 -S` of coverm_amd/csrc/covermhip.hip (or --asm), one function;
  2. every VALU instruction gets a CLASS by its encoding (SDWA, DPP, 24-bit multiply-add, three-operand VOP3, compare / select,
     64-bit, packed, plain two-operand, ...);
  3. a class costs what tools/ubench/valu_rate.hip MEASURES on this GPU for a chain of such instructions (cycles per wave64
     instruction per SIMD, committed as profiles/r03_valu_rate.json); classes the micro-benchmark does not cover take the cost of the
     nearest covered form and are listed as "assumed";
  4. basic blocks are weighted by how often they run: `flat` = once each (the static mix), `loop` = 8^depth from the compiler's own
     loop annotations, or a JSON of per-label trip counts from an instrumented run;
  5. with the DYNAMIC instruction count of a launch from the counters (SQ_INSTS_VALU) and the launch's SIMD-cycles
     (duration x clock x 4 SIMDs x CUs), the VALU issue fraction is  INSTS_VALU x mean cycles per instruction / SIMD-cycles.
The mean lies between the all-full-rate and the all-half-rate bounds; SQ_ACTIVE_INST_VALU is not used (it ticks once per instruction).
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# class -> key of the measured rate it is costed with (tools/ubench/valu_rate.hip kinds), True when that very form was measured
CLASS_RATE = {
    "plain2": ("v_add_u32", True),            # two-operand VOP1 / VOP2 / simple VOP3: mov, add, sub, and, or, xor, shifts, min, max
    "fma": ("v_fma_f32", True),
    "mad24": ("v_mad_u32_u24", True),         # 24-bit multiply(-add)
    "sdwa": ("sdwa", True),
    "dpp": ("dpp", True),
    "vop3_3op": ("min3/max3/add3", True),     # three register operands: add3, min3, max3, lshl_add, lshl_or, and_or, bfe, alignbit, perm, bfi
    "cmp_sel": ("cmp+cndmask", True),         # v_cmp* / v_cndmask*: costed per instruction at half of the measured pair
    "mul32": ("v_mad_u32_u24", False),        # v_mul_lo / v_mul_hi / v_mad_u64: quarter rate on paper, costed as half rate (lower bound)
    "b64": ("min3/max3/add3", False),         # 64-bit shifts / adds (v_lshlrev_b64, v_lshl_add_u64, v_add_co + addc pairs are plain2)
    "packed": ("v_add_u32", False),           # v_pk_* / v_dot2*: one pass
    "lane": ("dpp", False),                   # v_readlane / v_readfirstlane / v_writelane / v_permlane
    "other": ("min3/max3/add3", False),
}


def classify(mn, ops):
    if mn.endswith("_sdwa") or "src0_sel" in ops or "dst_sel" in ops:
        return "sdwa"
    if mn.endswith("_dpp") or "row_" in ops or "quad_perm" in ops or "wave_sh" in ops or "row_bcast" in ops:
        return "dpp"
    if mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_bpermute")):
        return "lane"
    if mn.startswith(("v_cmp", "v_cndmask")):
        return "cmp_sel"
    if mn.startswith(("v_mad_u32_u24", "v_mad_i32_i24", "v_mul_u32_u24", "v_mul_i32_i24", "v_mul_hi_u32_u24")):
        return "mad24"
    if mn.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")):
        return "mul32"
    if mn.startswith(("v_pk_", "v_dot")):
        return "packed"
    if mn.startswith(("v_fma", "v_mac", "v_mad_f")):
        return "fma"
    if re.search(r"_(b|u|i)64", mn) or mn.startswith("v_lshl_add_u64"):
        return "b64"
    if mn.startswith(("v_add3", "v_min3", "v_max3", "v_med3", "v_lshl_add", "v_lshl_or", "v_add_lshl", "v_and_or", "v_or3", "v_xad", "v_bfe", "v_bfi", "v_alignbit",
                      "v_alignbyte", "v_perm", "v_mad", "v_sad", "v_cubeid")):
        return "vop3_3op"
    if mn.startswith(("v_mov", "v_add", "v_sub", "v_and", "v_or", "v_xor", "v_not", "v_lshl", "v_lshr", "v_ashr", "v_min", "v_max", "v_bfrev", "v_ffb", "v_cvt", "v_bcnt",
                      "v_mbcnt", "v_accvgpr", "v_nop", "v_swap", "v_sat", "v_ldexp", "v_rcp", "v_exp", "v_log", "v_trunc", "v_floor", "v_mul_f", "v_sqrt")):
        return "plain2"
    return "other"


def kernel_asm(asm_path, pattern):
    txt = open(asm_path).read().splitlines()
    start = None
    for i, l in enumerate(txt):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and re.search(pattern, l) and not l.startswith(".L"):
            start = i
            break
    if start is None:
        raise SystemExit("no function matching %r in %s" % (pattern, asm_path))
    body = []
    for l in txt[start + 1:]:
        body.append(l)
        if l.strip().startswith("s_endpgm"):
            break
    return txt[start].split(":")[0], body


def blocks(body):
    """[(label, loop_depth, [(mnemonic, operands)])]"""
    out, cur, depth, label = [], [], 0, "entry"
    for l in body:
        m = re.match(r"^(\.LBB[\w]+):", l)
        if m:
            if cur:
                out.append((label, depth, cur))
            label, cur = m.group(1), []
            d = re.search(r"Depth=(\d+)", l)
            # a label line may carry the loop header note; notes on following comment lines are picked up below
            depth_note = int(d.group(1)) if d else None
            if depth_note is not None:
                depth = depth_note
            elif "in Loop" not in l:
                depth = 0 if "Loop" not in l else depth
            continue
        s = l.strip()
        if s.startswith(";") or not s:
            d = re.search(r"Depth=(\d+)", s)
            if d and not cur:
                depth = int(d.group(1))
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        cur.append((parts[0], parts[1].split(";")[0] if len(parts) > 1 else ""))
    if cur:
        out.append((label, depth, cur))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True, help="regex on the mangled name, e.g. k_pileup_fast7ILb1")
    ap.add_argument("--asm", default=None)
    ap.add_argument("--rates", default=os.path.join(ROOT, "profiles", "r03_valu_rate.json"))
    ap.add_argument("--weights", default="loop")
    ap.add_argument("--pmc", default=None, help="JSON with per-kernel counters (profiles/pmc_pipes.json)")
    ap.add_argument("--pmc-key", default=None)
    ap.add_argument("--kernel-ms", type=float, default=None)
    ap.add_argument("--clock-ghz", type=float, default=None)
    ap.add_argument("--cus", type=int, default=256)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    asm = a.asm
    if asm is None:
        asm = "/tmp/covermhip_valu_mix.s"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", asm,
                               "-x", "hip", os.path.join(ROOT, "coverm_amd", "csrc", "covermhip.hip"), "-Wno-unused-function", "-Wno-pass-failed"],
                              stderr=subprocess.DEVNULL)
    name, body = kernel_asm(asm, a.kernel)
    bl = blocks(body)
    rates = json.load(open(a.rates)) if os.path.exists(a.rates) else None
    weights = None
    if a.weights not in ("loop", "flat"):
        weights = json.load(open(a.weights))
    tot = {}
    n_valu = n_salu = n_lds = n_vmem = 0.0
    for label, depth, ins in bl:
        w = 1.0 if a.weights == "flat" else (8.0 ** depth if weights is None else float(weights.get(label, 0.0)))
        for mn, ops in ins:
            if mn.startswith("v_"):
                c = classify(mn, ops)
                tot[c] = tot.get(c, 0.0) + w
                n_valu += w
            elif mn.startswith("s_"):
                n_salu += w
            elif mn.startswith("ds_"):
                n_lds += w
            elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
                n_vmem += w
    res = {"kernel": name, "weights": a.weights, "basic_blocks": len(bl), "static_valu": sum(1 for _, _, i in bl for m, _ in i if m.startswith("v_")),
           "weighted": {"valu": n_valu, "salu": n_salu, "lds": n_lds, "vmem": n_vmem}, "class_share": {k: v / n_valu for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}}
    if rates:
        cyc = rates["cycles_per_wave_instruction"]
        full, half = cyc["v_add_u32"], max(cyc.values())
        mean, assumed = 0.0, 0.0
        for k, v in tot.items():
            key, measured = CLASS_RATE[k]
            c = cyc[key] / 2.0 if k == "cmp_sel" and key == "cmp+cndmask" and rates.get("cmp_cndmask_is_pair", True) else cyc[key]
            mean += v / n_valu * c
            if not measured:
                assumed += v / n_valu
        res.update(mean_cycles_per_valu_instruction=mean, full_rate_cycles=full, half_rate_cycles=half, share_costed_by_assumption=assumed,
                   rates_source=os.path.relpath(a.rates, ROOT))
        if a.pmc and a.pmc_key:
            p = json.load(open(a.pmc))[a.pmc_key]
            insts = float(p["valu_insts"])
            ghz = a.clock_ghz or rates.get("clock_ghz", 2.4)
            ms = a.kernel_ms or p.get("kernel_ms")
            if p.get("simd_cycles"):      # 4 x SQ_BUSY_CU_CYCLES of the same launch: no clock assumed
                simd_cycles = float(p["simd_cycles"])
                res.update(valu_insts_per_launch=insts, kernel_ms=ms, simd_cycles=simd_cycles, simd_cycles_source="4 x SQ_BUSY_CU_CYCLES",
                           valu_issue_frac=insts * mean / simd_cycles, valu_issue_frac_if_all_full_rate=insts * full / simd_cycles,
                           valu_issue_frac_if_all_half_rate=insts * half / simd_cycles, lane_utilisation=p.get("lane_utilisation"))
            elif ms:
                simd_cycles = ms * 1e-3 * ghz * 1e9 * 4 * a.cus
                res.update(valu_insts_per_launch=insts, kernel_ms=ms, clock_ghz=ghz, simd_cycles=simd_cycles,
                           valu_issue_frac=insts * mean / simd_cycles, valu_issue_frac_if_all_full_rate=insts * full / simd_cycles,
                           valu_issue_frac_if_all_half_rate=insts * half / simd_cycles,
                           lane_utilisation=p.get("lane_utilisation"))
    print(json.dumps(res, indent=1))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
