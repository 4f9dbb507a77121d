"""Multi-GPU sharding of the coverage path: one process per GPU, `torch.distributed` (RCCL on ROCm, gloo on CPU).

Two ways the path shards, neither needs a data-path collective:

* **by sample** — one BAM per rank (`contig.rs:22` processes BAMs independently); rank 0 receives every rank's
  per-contig statistics with ONE gather and runs the scan drivers / printers over all samples;
* **by tid range** — one sorted BAM split into contiguous reference-id ranges (a contiguous span of records);
  per-contig statistics of different ranks are disjoint rows, merged on rank 0 after the same single gather,
  and `num_detected_primary_alignments` is the sum of the shards' counts.

The gathered payload is the raw `cov_contig_stats` array (128 B per contig) plus the compact histograms, moved as
byte tensors so that integers stay exact.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import native
from .engine import RecordBatch
from .host import SampleResult


def tid_range_shards(target_len, world: int, weights=None) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) tid ranges, balanced by `weights` (default: target length)."""
    w = np.asarray(target_len if weights is None else weights, dtype=np.float64)
    n = len(w)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1] if cum[-1] > 0 else 1.0
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r / world, side="left")))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.minimum(cuts, n))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def shard_records(batch: RecordBatch, lo_tid: int, hi_tid: int, include_unplaced: bool = False) -> RecordBatch:
    """Records of a coordinate-sorted batch whose tid lies in [lo_tid, hi_tid).  Records without a reference
    (tid = -1, sorted last) go to the shard with include_unplaced so that primaries are counted once."""
    tid = batch.tid
    neg = np.nonzero(tid < 0)[0]
    placed_end = int(neg[0]) if len(neg) else len(tid)
    # tid is non-decreasing over [0, placed_end)
    lo = int(np.searchsorted(tid[:placed_end], lo_tid, side="left"))
    hi = int(np.searchsorted(tid[:placed_end], hi_tid, side="left"))
    if include_unplaced and placed_end < len(tid):
        idx = np.concatenate([np.arange(lo, hi), np.arange(placed_end, len(tid))])
        n = (batch.cigar_off[1:].astype(np.int64) - batch.cigar_off[:-1].astype(np.int64))[idx]
        off = np.zeros(len(idx) + 1, dtype=np.uint32)
        np.cumsum(n, out=off[1:])
        cig = np.concatenate([batch.cigar[batch.cigar_off[i]:batch.cigar_off[i + 1]] for i in idx]) if len(idx) else \
            np.zeros(0, np.uint32)
        return RecordBatch(tid[idx], batch.pos[idx], batch.flag[idx], batch.mapq[idx], batch.nm[idx],
                           batch.nm_kind[idx], batch.l_seq[idx], off, cig.astype(np.uint32))
    return batch.slice(lo, hi)


def _gather_bytes(payload: np.ndarray, dist, device, dst=0) -> Optional[List[np.ndarray]]:
    """Variable-length byte gather: sizes first (all_gather of one int64), then one padded gather."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    raw = np.ascontiguousarray(payload).view(np.uint8).reshape(-1)
    size = torch.tensor([raw.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if raw.size:
        buf[:raw.size] = torch.from_numpy(raw.copy()).to(device)
    out = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return [o[:sizes[r]].cpu().numpy() for r, o in enumerate(out)]


def gather_samples(local: SampleResult, dist, device="cpu", dst=0) -> Optional[List[SampleResult]]:
    """By-sample sharding: every rank contributes one whole SampleResult; rank `dst` gets them in rank order."""
    stats = _gather_bytes(local.stats, dist, device, dst)
    hist = _gather_bytes(local.hist if local.hist is not None else np.zeros(0, np.uint64), dist, device, dst)
    meta = np.frombuffer(local.stoit_name.encode(), dtype=np.uint8)
    names = _gather_bytes(meta, dist, device, dst)
    prim = _gather_bytes(np.asarray([local.num_detected_primary_alignments], dtype=np.uint64), dist, device, dst)
    if stats is None:
        return None
    res = []
    for r in range(len(stats)):
        st = stats[r].view(native.CONTIG_STATS_DTYPE).copy()
        h = hist[r].view(np.uint64).copy() if local.hist is not None else None
        res.append(SampleResult(bytes(names[r]).decode(), st, h, int(prim[r].view(np.uint64)[0])))
    return res


def gather_tid_shards(local: SampleResult, tid_range: Tuple[int, int], dist, device="cpu", dst=0
                      ) -> Optional[SampleResult]:
    """By-tid-range sharding of ONE sample: rows [lo, hi) of each rank's statistics are authoritative; histogram
    slices are re-based into one concatenated array; primaries are summed."""
    parts = gather_samples(local, dist, device, dst)
    rng = _gather_bytes(np.asarray(tid_range, dtype=np.int64), dist, device, dst)
    if parts is None:
        return None
    merged = np.zeros_like(parts[0].stats)
    hists, base, prim = [], 0, 0
    for r, p in enumerate(parts):
        lo, hi = (int(x) for x in rng[r].view(np.int64))
        rows = p.stats[lo:hi].copy()
        if p.hist is not None:
            rows["hist_off"] += base
            hists.append(p.hist)
            base += len(p.hist)
        merged[lo:hi] = rows
        prim += p.num_detected_primary_alignments
    hist = np.concatenate(hists) if hists else None
    return SampleResult(parts[0].stoit_name, merged, hist, prim)
