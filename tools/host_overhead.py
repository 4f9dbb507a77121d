"""Breaks one bench step into: cov_finish wall, sum of kernel times, hist fetch, C++ host finalisation, Python glue."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coverm_amd import host, synth
from coverm_amd.engine import FilterConfig, Session
from coverm_amd.host import CoverageEstimator as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ref = synth.make_reference(n // 10000, n * 20, seed=1)
batch = synth.make_reads(ref, n, seed=2)
dev = torch.device("cuda", 0)
dt = {k: torch.from_numpy(getattr(batch, k)).to(dev) for k in ("tid", "pos", "flag", "mapq", "nm", "nm_kind", "l_seq", "cigar_off", "cigar")}
torch.cuda.synchronize()
est = [E.new_estimator_mean(0.0, 75, False), E.new_estimator_trimmed_mean(0.05, 0.95, 0.0, 75), E.new_estimator_covered_fraction(0.0), E.new_estimator_variance(0.0, 75)]
s = Session(0, FilterConfig(), 75, True, False)
s.set_targets(ref.lengths); s.push_device(dt, batch.n_records)
for it in range(6):
    t0 = time.perf_counter(); st, summ = s.finish(); t1 = time.perf_counter(); h = s.hist(); t2 = time.perf_counter()
    taker = host.CoverageTaker.new_cached_single_float_coverage_taker(4)
    sample = host.SampleResult("s", st, h, int(summ.num_detected_primary_alignments))
    rm = host.contig_coverage(ref.names, ref.lengths, [sample], taker, est, True); t3 = time.perf_counter()
    km = s.kernel_ms()
    print("finish %.3f ms (kernels %.3f) | hist %.3f | host finalise %.3f | total %.3f" % ((t1 - t0) * 1e3, sum(v[0] for v in km.values()), (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
