"""Multi-GPU sharding of the coverage path: one process per GPU, `torch.distributed` (RCCL on ROCm, gloo on CPU).

Two ways the path shards, neither needs a data-path collective:

* **by sample** — one BAM per rank (`contig.rs:22` processes BAMs independently); rank 0 receives every rank's
  per-contig statistics with ONE gather and runs the scan drivers / printers over all samples;
* **by tid range** — one sorted BAM split into contiguous reference-id ranges (a contiguous span of records);
  per-contig statistics of different ranks are disjoint rows, merged on rank 0 after the same single gather,
  and `num_detected_primary_alignments` is the sum of the shards' counts.

The gathered payload is ONE packed byte buffer per rank (a 304-byte header + the raw `cov_contig_stats` array, 128 B per
contig): for tid shards its size is the same on every rank, so a single `gather` moves it with no size exchange; samples
with different reference sets agree on the largest target count first and pad.  Compact histograms, when a
method needs them, follow point to point (their sizes are in the gathered headers).  Bytes, so that integers stay exact.
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


_NAME_BYTES = 256
# stoit name | name length (u32) + has-histogram flag (u32) | num_detected_primary_alignments | tid range lo, hi | histogram bins | contigs
_HEADER_BYTES = _NAME_BYTES + 8 + 8 + 16 + 8 + 8


def _pack(local: SampleResult, tid_range: Tuple[int, int], pad_to: Optional[int] = None) -> np.ndarray:
    """The fixed-size part of one rank's result: header + the raw cov_contig_stats rows (128 B per contig), padded with zero
    rows up to `pad_to` contigs.  For tid shards of ONE file every rank has the same number of targets, so ONE gather moves
    the buffers without any size exchange; by-sample sharding agrees on `pad_to` first (gather_samples)."""
    name = local.stoit_name.encode()
    if len(name) > _NAME_BYTES:                         # cut on a character boundary, never inside a UTF-8 sequence
        name = name[:_NAME_BYTES].decode(errors="ignore").encode()
    n = len(local.stats)
    head = np.zeros(_HEADER_BYTES, np.uint8)
    head[:len(name)] = np.frombuffer(name, np.uint8)
    o = _NAME_BYTES
    head[o:o + 8] = np.asarray([len(name), 0 if local.hist is None else 1], np.uint32).view(np.uint8)
    head[o + 8:o + 16] = np.asarray([local.num_detected_primary_alignments], np.uint64).view(np.uint8)
    head[o + 16:o + 32] = np.asarray(tid_range, np.int64).view(np.uint8)
    head[o + 32:o + 40] = np.asarray([0 if local.hist is None else len(local.hist)], np.uint64).view(np.uint8)
    head[o + 40:o + 48] = np.asarray([n], np.uint64).view(np.uint8)
    rows = np.ascontiguousarray(local.stats).view(np.uint8).reshape(-1)
    if pad_to is not None and pad_to > n:
        rows = np.concatenate([rows, np.zeros((pad_to - n) * native.CONTIG_STATS_DTYPE.itemsize, np.uint8)])
    return np.concatenate([head, rows])


def _unpack(raw: np.ndarray):
    o = _NAME_BYTES
    name_len, has_hist = (int(x) for x in raw[o:o + 8].view(np.uint32))
    name = bytes(raw[:min(name_len, _NAME_BYTES)]).decode(errors="replace")
    prim = int(raw[o + 8:o + 16].view(np.uint64)[0])
    lo, hi = (int(x) for x in raw[o + 16:o + 32].view(np.int64))
    n_hist = int(raw[o + 32:o + 40].view(np.uint64)[0])
    n = int(raw[o + 40:o + 48].view(np.uint64)[0])
    stats = raw[_HEADER_BYTES:_HEADER_BYTES + n * native.CONTIG_STATS_DTYPE.itemsize].view(native.CONTIG_STATS_DTYPE).copy()
    return name, prim, (lo, hi), n_hist, stats, bool(has_hist)


def gather_packed(local: SampleResult, tid_range: Tuple[int, int], dist, device="cpu", dst=0, pad_to: Optional[int] = None):
    """ONE gather of every rank's packed (header + per-contig statistics) buffer to rank `dst`; the buffers must have one size
    (same number of targets, or padded to `pad_to`).  Histograms (only present for trimmed_mean / coverage_histogram) have a
    data-dependent size: rank `dst` learns every size — and whether a rank has one at all — from the gathered headers and the
    other ranks send theirs point to point — no collective, no size exchange.
    Returns, on `dst`, a list of (SampleResult, (lo, hi)) in rank order; None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    packed = torch.from_numpy(_pack(local, tid_range, pad_to)).to(device)
    out = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, out, dst=dst)
    if rank != dst:
        if local.hist is not None and len(local.hist):
            dist.send(torch.from_numpy(np.ascontiguousarray(local.hist).view(np.uint8).copy()).to(device), dst=dst)
        return None
    res = []
    for r, o in enumerate(out):
        name, prim, rng, n_hist, stats, has_hist = _unpack(o.cpu().numpy())
        hist = None
        if has_hist:
            if r == dst:
                hist = np.ascontiguousarray(local.hist, np.uint64).copy()
            elif n_hist:
                t = torch.empty(n_hist * 8, dtype=torch.uint8, device=device)
                dist.recv(t, src=r)
                hist = t.cpu().numpy().view(np.uint64).copy()
            else:
                hist = np.zeros(0, np.uint64)
        res.append((SampleResult(name, stats, hist, prim), rng))
    return res


def gather_samples(local: SampleResult, dist, device="cpu", dst=0) -> Optional[List[SampleResult]]:
    """By-sample sharding: every rank contributes one whole SampleResult; rank `dst` gets them in rank order.  Different BAMs may
    bring different reference sets (contig.rs:29-32 reads each file's own header), so the ranks first agree on the largest
    number of targets (one 8-byte all_gather) and pad their rows to it; the single gather then moves equal-sized buffers."""
    world = dist.get_world_size()
    mine = torch.tensor([len(local.stats)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    pad_to = max(int(t.item()) for t in sizes)
    got = gather_packed(local, (0, len(local.stats)), dist, device, dst, pad_to=pad_to)
    return None if got is None else [g[0] for g in got]


def gather_tid_shards(local: SampleResult, tid_range: Tuple[int, int], dist, device="cpu", dst=0
                      ) -> Optional[SampleResult]:
    """By-tid-range sharding of ONE sample: rows [lo, hi) of each rank's statistics are authoritative; histogram
    slices are re-based into one concatenated array; primaries are summed."""
    got = gather_packed(local, tid_range, dist, device, dst)
    if got is None:
        return None
    if any(len(p.stats) != len(got[0][0].stats) for p, _ in got):
        raise ValueError("gather_tid_shards: the ranks hold different numbers of targets (tid shards must come from one file)")
    merged = np.zeros_like(got[0][0].stats)
    hists, base, prim = [], 0, 0
    for p, (lo, hi) in got:
        rows = p.stats[lo:hi].copy()
        if p.hist is not None:
            rows["hist_off"] += base
            hists.append(p.hist)
            base += len(p.hist)
        merged[lo:hi] = rows
        prim += p.num_detected_primary_alignments
    hist = np.concatenate(hists) if hists else None
    return SampleResult(got[0][0].stoit_name, merged, hist, prim)


def wait_for_root(dist, rank: int, key: str = "coverm_root_done", minutes: float = 45.0, root: int = 0) -> bool:
    """Ranks other than `root` wait ON THE HOST until `root` calls this: a key in the process group's store, not a collective.  A barrier
    of the nccl (= RCCL) backend is a kernel that spins on every waiting rank's GPU — while rank 0 of `bench.py --gpus N` runs the product's
    own multi-device legs on those very GPUs.  True when the store was used; on any failure of that path the caller's next barrier still
    orders the ranks (the store is an optimisation of WHERE they wait, not of correctness)."""
    import datetime
    import sys
    # every call of a process uses its own key (the calls of all ranks pair up in order): a key that was set once would let the second wait
    # on it return at once
    n = _WAIT_CALLS[key] = _WAIT_CALLS.get(key, 0) + 1
    k = "%s#%d" % (key, n)
    try:
        store = dist.distributed_c10d._get_default_store()
        if rank == root:
            store.set(k, "1")
        else:
            store.wait([k], datetime.timedelta(minutes=minutes))
        return True
    except Exception as ex:      # (a real timeout included: say so, the caller's barrier still orders the ranks)
        print("[coverm_amd.distributed] wait_for_root(%s): %s: %s" % (k, type(ex).__name__, str(ex)[:200]), file=sys.stderr, flush=True)
        return False


_WAIT_CALLS = {}
