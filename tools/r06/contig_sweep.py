"""Round 6: the many-contig regime.  For 5 000 / 200 000 / 2 000 000 contigs at equal reads:

  (a) device-resident step (cov_finish over the records in HBM + the floats' fetch): per kernel group (cov_kernel_ms) and the host's wall time,
      with the integer statistics + floats of every contig compared between k_prep_lean / k_prep_generic's default choice and k_prep7s;
  (b) the same sample written as a BAM file (tmpfs) through `coverm-amd contig` with COVERM_CLI_TIMING: wall, the binary's own stamps (sessions,
      ingest, finish + fetch, scan drivers + table), peak RSS, size of the table.

    python tools/r06/contig_sweep.py [--reads N] [--out profiles/r06_contig_sweep.json] [--no-binary]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from coverm_amd import bam as cbam  # noqa: E402
from coverm_amd import host, synth  # noqa: E402
from coverm_amd.engine import FilterConfig, Session  # noqa: E402
from coverm_amd.host import CoverageEstimator as E  # noqa: E402

FIELDS = ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")
CONFIGS = [(5_000, 1_000_000_000, 2000), (200_000, 1_000_000_000, 1000), (2_000_000, 2_000_000_000, 1000)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_contig_sweep.json"))
    ap.add_argument("--no-binary", action="store_true")
    ap.add_argument("--tmp", default="/dev/shm")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    est = [E.new_estimator_mean(0.0, 75, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, 75),
           E.new_estimator_covered_fraction(0.0), E.new_estimator_variance(0.0, 75)]
    res = {"reads": a.reads, "methods": "mean trimmed_mean covered_fraction variance", "configs": []}
    for n_contigs, bp, min_len in CONFIGS:
        t0 = time.time()
        ref = synth.make_reference(n_contigs, bp, seed=1, min_len=min_len)
        batch = synth.make_reads(ref, a.reads, seed=2)
        dt = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in FIELDS}
        torch.cuda.synchronize()
        row = {"contigs": n_contigs, "bp": int(ref.lengths.sum()), "records": int(batch.n_records), "generated_s": round(time.time() - t0, 1)}
        print("== %d contigs, %d bp, %d records (%.1f s)" % (n_contigs, row["bp"], batch.n_records, row["generated_s"]), flush=True)
        outs = {}
        for vname, env in (("k_prep7s", {"COVERM_PREP_KERNEL": "7"}), ("default", {})):
            os.environ.pop("COVERM_PREP_KERNEL", None)
            os.environ.update(env)
            want_hist, _ = host.wants(est)
            s = Session(0, FilterConfig(), 75, want_hist, False)
            s.set_targets(ref.lengths)
            s.set_estimators(est)
            s.push_device(dt, batch.n_records)
            stats, summ = s.finish()
            outs[vname] = (stats.tobytes(), s.estimates().tobytes(), int(summ.n_considered))
            if vname == "default":
                for _ in range(2):
                    s.finish(); s.estimates()
                torch.cuda.synchronize()
                acc = {}
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    s.finish(); s.estimates()
                    for k, v in s.kernel_ms().items():
                        acc[k] = acc.get(k, 0.0) + v[0]
                row["step_ms_python_harness"] = round((time.perf_counter() - t1) / a.steps * 1e3, 3)
                row["kernel_ms"] = {k: round(v / a.steps, 4) for k, v in acc.items() if v > 0}
                row["kernels_sum_ms"] = round(sum(row["kernel_ms"].values()), 4)
            s.close()
        os.environ.pop("COVERM_PREP_KERNEL", None)
        row["statistics_and_floats_equal_k_prep7s"] = outs["k_prep7s"] == outs["default"]
        row["considered"] = outs["default"][2]
        print(json.dumps(row), flush=True)
        del dt
        torch.cuda.empty_cache()
        if not a.no_binary:
            path = os.path.join(a.tmp, "r06_sweep_%d.bam" % n_contigs)
            t0 = time.time()
            cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=1, threads=min(32, os.cpu_count() or 8))
            row["bam_bytes"] = os.path.getsize(path)
            row["bam_written_s"] = round(time.time() - t0, 1)
            exe = os.path.join(ROOT, "coverm_amd", "coverm-amd")
            runs = []
            for k in range(3):
                time.sleep(4.0)      # (a process that held 30 GB of device memory is still being reaped when the next one starts: bench.py's run_binary waits too)
                t0 = time.perf_counter()
                p = subprocess.run([exe, "contig", "-b", path, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "-o", os.path.join(a.tmp, "r06_sweep.tsv"), "-t", "16"],
                                   env=dict(os.environ, COVERM_CLI_TIMING="1"), capture_output=True, text=True)
                wall = time.perf_counter() - t0
                err = p.stderr
                g = lambda pat: (re.search(pat, err).group(1) if re.search(pat, err) else None)
                runs.append({"wall_s": round(wall, 3), "rc": p.returncode,
                             "sessions_s": g(r"arguments \+ device sessions ([0-9.]+)s"), "samples_s": g(r"samples ([0-9.]+)s"), "scan_drivers_and_table_s": g(r"scan drivers \+ table ([0-9.]+)s"),
                             "ingest_s": g(r"ingest \(decode\+push\) ([0-9.]+)s"), "finish_fetch_s": g(r"finish\+fetch ([0-9.]+)s"),
                             "vm_hwm_kb": g(r"VmHWM:\s+(\d+) kB")})
                if p.returncode != 0:
                    print(err[-2000:], flush=True)
            row["binary_runs"] = runs
            row["table_bytes"] = os.path.getsize(os.path.join(a.tmp, "r06_sweep.tsv")) if os.path.exists(os.path.join(a.tmp, "r06_sweep.tsv")) else None
            os.remove(path)
            print(json.dumps(runs), flush=True)
        res["configs"].append(row)
        del batch, ref
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print("written", a.out)


if __name__ == "__main__":
    main()
