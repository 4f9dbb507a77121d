"""The numbers the documents quote for the current round come from the committed profiles — mechanically.

DESIGN.md and README.md each carry a block between `<!-- numbers:begin -->` and `<!-- numbers:end -->`: a table of the round's figures, every row
naming the file under profiles/ and the key it was read from.  `python tools/check_docs.py --write` regenerates the blocks from the files;
`python tools/check_docs.py` (and tests/test_docs_follow_profiles.py, on the CPU) fails when a block differs from what the files say, when a cited
file is missing, or when the prose outside the blocks cites a profiles/ path that does not exist.  (files were
overwritten after the prose that quoted them.)
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- numbers:begin -->", "<!-- numbers:end -->"
DOCS = ["DESIGN.md", "README.md"]
TAG = "r06g"      # the round's final profile set (tools/r06/final.sh <tag>)


def jget(path, keys):
    d = json.load(open(os.path.join(ROOT, path)))
    for k in keys:
        d = d[k]
    return d


def kstat(path, name, col="AverageNs"):
    for row in csv.DictReader(open(os.path.join(ROOT, path))):
        if name in row["Name"]:
            return float(row[col])
    raise KeyError(name + " not in " + path)


def sig(x, n=3):
    return "%.*g" % (n, float(x))


def rows():
    b = "profiles/%s_bench_line.json" % TAG
    ks = "profiles/%s_bench_kernel_stats.csv" % TAG
    pm = "profiles/%s_bench_pmc_summary.json" % TAG
    sw = "profiles/r06_contig_sweep.json"
    two = "profiles/r06_two_samples_in_flight.json"
    R = []
    add = lambda what, val, src: R.append((what, val, src))
    add("device-resident step, config 2 (`ms_per_step`)", sig(jget(b, ["ms_per_step"]), 4) + " ms", b + ": ms_per_step")
    add("aligned reads/s, device-resident (`value`)", sig(jget(b, ["value"]) / 1e9, 4) + " G/s", b + ": value")
    add("k_prep group (k_prep_lean + k_prep_generic + k_post_prep), HIP events of the bench", sig(jget(b, ["roofline", "all_kernels_ms", "k_prep"]), 3) + " ms", b + ": roofline.all_kernels_ms.k_prep")
    add("k_pileup group, HIP events of the bench", sig(jget(b, ["roofline", "all_kernels_ms", "k_pileup"]), 3) + " ms", b + ": roofline.all_kernels_ms.k_pileup")
    add("k_prep group: algorithmic bytes / time against 8 TB/s", sig(jget(b, ["roofline", "kernels", "k_prep", "hbm_frac_of_8TBps"]), 3), b + ": roofline.kernels.k_prep.hbm_frac_of_8TBps")
    add("k_pileup: algorithmic bytes / time against 8 TB/s", sig(jget(b, ["roofline", "kernels", "k_pileup", "hbm_frac_of_8TBps"]), 3), b + ": roofline.kernels.k_pileup.hbm_frac_of_8TBps")
    add("`roofline.frac` of the line (its dominant kernel)", sig(jget(b, ["roofline", "frac"]), 3) + " (" + str(jget(b, ["roofline", "kernel"])) + ")", b + ": roofline.frac")
    add("k_prep_lean, rocprofv3 kernel trace, average", sig(kstat(ks, "k_prep_lean") / 1e3, 4) + " us", ks + ": k_prep_lean AverageNs")
    add("k_prep_generic, rocprofv3 kernel trace, average", sig(kstat(ks, "k_prep_generic") / 1e3, 3) + " us", ks + ": k_prep_generic AverageNs")
    add("k_pileup_fast, rocprofv3 kernel trace, average", sig(kstat(ks, "k_pileup_fast") / 1e3, 4) + " us", ks + ": k_pileup_fast AverageNs")
    for kern, short in (("covk::k_prep_lean<false, false, false>", "k_prep_lean"), ("covk::k_pileup_fast<true>", "k_pileup_fast")):
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
            try:
                v = jget(pm, [kern, c, "mean_per_dispatch"])
            except KeyError:
                continue
            if v:
                add("%s %s per launch" % (short, c), sig(v / 1e6, 4) + " M", pm + ": " + short + "." + c)
    add("k_prep HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE)", sig(jget("profiles/pmc_traffic.json", ["k_prep"]) / 1e9, 4) + " GB", "profiles/pmc_traffic.json: k_prep")
    add("k_pileup HBM traffic per launch", sig(jget("profiles/pmc_traffic.json", ["k_pileup"]) / 1e9, 4) + " GB", "profiles/pmc_traffic.json: k_pileup")
    for key, what in (("e2e_l1_s", "end to end, 200 M reads, BGZF level 1: wall (the command returns when the table is written)"), ("e2e_l1_x_overlapped", "... x the CPU path, decode and scan overlapped"),
                      ("e2e_l1_one_process_s", "the same as ONE process (`COVERM_NO_FAST_EXIT=1`: the caller also waits for the runtime's teardown)"),
                      ("e2e_l1_x_overlapped_one_process", "... x the CPU path"),
                      ("e2e_l6_s", "end to end, BGZF level 6: wall"), ("e2e_l6_x_overlapped", "... x the CPU path, overlapped"),
                      ("cpu_decode_s", "CPU decode (16 threads, libdeflate), best of three"), ("cpu_scan_s", "CPU scan (oracle, one thread), best"),
                      ("cfg2_binary_s", "config 2 through the binary, median"), ("cfg3_binary_s", "config 3 through the binary, median")):
        try:
            v = jget(b, [key])
        except KeyError:
            continue          # (a bench line from before the field existed)
        add(what, ("%s" % v) + (" s" if key.endswith("_s") else ""), b + ": " + key)
    add("parity of the bench's own check (`parity_equal`, `tables_equal`)", "%s, %s" % (jget(b, ["parity_equal"]), jget(b, ["tables_equal"])), b + ": parity_equal, tables_equal")
    for c in jget(sw, ["configs"]):
        n = c["contigs"]
        add("%d contigs x 50 M reads: kernels of a step" % n, sig(c["kernels_sum_ms"], 4) + " ms (k_prep %s, k_pileup %s, k_estimate %s)" % (
            sig(c["kernel_ms"]["k_prep"], 3), sig(c["kernel_ms"]["k_pileup"], 3), sig(c["kernel_ms"]["k_estimate"], 3)), sw + ": configs[contigs=%d].kernel_ms" % n)
        br = c.get("binary_runs") or []
        if br:
            walls = sorted(r["wall_s"] for r in br)
            add("%d contigs: `coverm-amd contig` over the BAM file, median wall of three" % n, sig(walls[len(walls) // 2], 3) + " s (peak RSS %s MB)" % sig(float(br[0]["vm_hwm_kb"]) / 1024, 4),
                sw + ": configs[contigs=%d].binary_runs" % n)
    add("two samples in flight on one GPU: aggregate over one after the other", sig(jget(two, ["concurrent_over_sequential"]), 3) + " x", two + ": concurrent_over_sequential")
    return R


def block():
    out = [BEGIN, "", "| quantity | value | read from |", "|---|---|---|"]
    for what, val, src in rows():
        out.append("| %s | %s | `%s` |" % (what, val, src))
    out += ["", END]
    return "\n".join(out)


def cited_paths(text):
    return set(re.findall(r"profiles/[A-Za-z0-9_./-]*[A-Za-z0-9]", text))


def main():
    write = "--write" in sys.argv
    want = block()
    bad = []
    for doc in DOCS:
        p = os.path.join(ROOT, doc)
        s = open(p).read()
        if BEGIN not in s or END not in s:
            bad.append("%s: no numbers block" % doc)
            continue
        a, z = s.index(BEGIN), s.index(END) + len(END)
        if s[a:z] != want:
            if write:
                open(p, "w").write(s[:a] + want + s[z:])
                print("rewrote the numbers block of", doc)
            else:
                bad.append("%s: the numbers block differs from what the profiles say (python tools/check_docs.py --write)" % doc)
    for doc in DOCS + ["profiles/README.md"]:
        s = open(os.path.join(ROOT, doc)).read()
        for q in sorted(cited_paths(s)):
            if "*" in q or q.endswith(("_", "/")):
                continue
            if not os.path.exists(os.path.join(ROOT, q)) and not any(f.startswith(os.path.basename(q)) for f in os.listdir(os.path.join(ROOT, "profiles"))):
                bad.append("%s cites %s, which does not exist" % (doc, q))
    for b_ in bad:
        print("check_docs:", b_)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
