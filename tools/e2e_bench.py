"""End-to-end timing: synthetic BAM on disk -> `coverm-amd contig` (C++ reader + GPU engine).

Reports (i) BAM decode time (C++ reader, t threads), (ii) H2D + device pipeline, (iii) whole binary wall time,

Used for DESIGN.md's end-to-end table; not the bench.  (The CPU port is timed only by bench.py's cpu_baseline leg.)
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from coverm_amd import bam as cbam  # noqa: E402
from coverm_amd import synth  # noqa: E402
from coverm_amd.engine import FilterConfig, Session  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--contigs", type=int, default=1000)
ap.add_argument("--bp", type=int, default=200_000_000)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--cpu", type=int, default=0, help="unused (the CPU port is timed by bench.py's cpu_baseline leg only)")
ap.add_argument("--genes", type=int, default=0, help="also time --gff with one gene per this many bases")
ap.add_argument("--samples", type=int, default=0, help="also time N lean BAMs -> one dense table (config 4 shape)")
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "coverm_amd", "coverm-amd")

ref = synth.make_reference(a.contigs, a.bp, seed=1)
batch = synth.make_reads(ref, a.reads, seed=2)
tmp = tempfile.mkdtemp(prefix="covbam")
for with_seq in (True, False):
    path = os.path.join(tmp, "synth_%s.bam" % ("seq" if with_seq else "lean"))
    t = time.time()
    cbam.write_bam(path, ref.names, ref.lengths, batch, with_seq=with_seq, threads=a.threads)
    tw = time.time() - t
    size = os.path.getsize(path)
    t = time.time()
    af = cbam.read_alignment_file(path, threads=a.threads, want_names=False)
    td = time.time() - t
    t = time.time()
    with Session(0, FilterConfig(), 75, want_hist=True) as s:
        s.set_targets(af.ref_lens)
        s.push(af.records)
        st, summ = s.finish()
        h = s.hist()
    tg = time.time() - t
    t = time.time()
    r = subprocess.run([BIN, "contig", "-b", path, "-m", "mean", "trimmed_mean", "covered_fraction", "variance", "-t",
                        str(a.threads), "-o", os.path.join(tmp, "out.tsv")], capture_output=True, text=True)
    tb = time.time() - t
    assert r.returncode == 0, r.stderr
    print("%s BAM: %.2f GB written in %.1fs | decode(%d thr) %.2fs = %.1f M rec/s | session create+push+finish %.3fs | "
          "coverm-amd binary wall %.2fs = %.2f M reads/s end-to-end" % (
              "full-SEQ" if with_seq else "lean", size / 1e9, tw, a.threads, td, a.reads / td / 1e6, tg, tb,
              int(summ.n_considered) / tb / 1e6), flush=True)
if a.genes:
    gff = os.path.join(tmp, "genes.gff")
    rng = np.random.default_rng(7)
    with open(gff, "w") as fh:
        ng = 0
        for name, L in zip(ref.names, ref.lengths):
            p0 = 1
            while p0 + 300 < L:
                ln = int(rng.integers(300, 2 * a.genes - 300))
                e0 = min(int(L), p0 + ln)
                fh.write("%s\tsyn\tCDS\t%d\t%d\t.\t+\t0\tID=g%d\n" % (name, p0, e0, ng))
                ng += 1
                p0 = e0 + int(rng.integers(1, 200))
    path = os.path.join(tmp, "synth_lean.bam")
    t = time.time()
    r = subprocess.run([BIN, "contig", "-b", path, "--gff", gff, "-m", "mean", "covered_fraction", "count", "-t", str(a.threads),
                        "-o", os.path.join(tmp, "genes.tsv")], capture_output=True, text=True)
    tb = time.time() - t
    assert r.returncode == 0, r.stderr
    print("per-gene: %d genes over %d contigs, lean BAM: coverm-amd --gff wall %.2fs" % (ng, len(ref.names), tb), flush=True)
if a.samples:
    paths = []
    for k in range(a.samples):
        bk = synth.make_reads(ref, a.reads, seed=10 + k)
        pk = os.path.join(tmp, "s%d.bam" % k)
        cbam.write_bam(pk, ref.names, ref.lengths, bk, with_seq=True, threads=a.threads)
        paths.append(pk)
    t = time.time()
    r = subprocess.run([BIN, "contig", "-b"] + paths + ["-m", "mean", "variance", "rpkm", "-t", str(a.threads), "-o",
                        os.path.join(tmp, "out.tsv")], capture_output=True, text=True)
    tb = time.time() - t
    assert r.returncode == 0, r.stderr
    sys.stderr.write(r.stderr)
    print("%d full-SEQ BAMs x %d reads -> dense table: coverm-amd wall %.2fs = %.2f M reads/s end-to-end" % (
        a.samples, a.reads, tb, a.samples * a.reads / tb / 1e6), flush=True)
