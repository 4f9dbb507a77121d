"""The round's last 40 GPU-seconds: the tests added after the last full -m gpu run, called directly (no pytest, no torch import)."""
import os, sys, time
t0 = time.time()
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
out = open(os.path.join(R, "gpurun_out", "r04_call21.log"), "w")
def say(*a):
    print("%6.1fs" % (time.time() - t0), *a, flush=True); print("%6.1fs" % (time.time() - t0), *a, file=out, flush=True)
import tests.test_gpu_abi_parity as T
say("imported")
T._piles_between_the_bin_counts(); say("piles between the bin counts, default kernel (k_pileup_fast7): equal")
os.environ["COVERM_FAST_WAVES"] = "6"
T._piles_between_the_bin_counts(); say("piles between the bin counts, COVERM_FAST_WAVES=6 (k_pileup_fast): equal")
T.test_fast_kernel_deep_and_interleaved_tiles(); say("deep and interleaved tiles, six waves: equal")
del os.environ["COVERM_FAST_WAVES"]
T.test_fast_kernel_deep_and_interleaved_tiles(); say("deep and interleaved tiles, default: equal")
import tempfile, pathlib
import numpy as np
import tests.test_gpu_ingest as G
from coverm_amd import synth, bam as cbam
os.environ["COVERM_EXT_PARTS"] = "1"
with tempfile.TemporaryDirectory() as d:
    ref = synth.make_reference(40, 6_000_000, seed=18, min_len=5000, max_len=800_000)
    b = synth.make_reads(ref, 120_000, seed=19)
    p = str(pathlib.Path(d) / "s.bam")
    cbam.write_bam(p, ref.names, ref.lengths, b, with_seq=2, threads=4)
    w = G._check(p)
    np.testing.assert_array_equal(w.records.pos, b.pos)
say("extraction with a lane per segment: equal")
