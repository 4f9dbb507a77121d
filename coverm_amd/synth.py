"""Deterministic synthetic sorted-BAM record generator (numpy), the workload of BASELINE.json's configs.

Profile: contig lengths log-uniform in [2 kb, 2 Mb] rescaled to a target total, 150 bp
paired reads (flags 99/147/83/163, proper-pair bit on 95 %), contig chosen ~ length x lognormal(sigma=1)
abundance, CIGAR mix 90 % `150M`, 6 % `kM iI (150-k-i)M`, 3 % `kM dD (150-k)M`, 1 % soft-clipped, plus
0.1 % each of `=`/`X` and `N` runs (ops absent from the reference's fixtures), NM ~ Poisson(1.5) + indel
length (type C), MAPQ in {0, 1..60}, 1 % secondary, 1 % supplementary, 2 % unmapped-with-tid (no CIGAR,
no NM).  Records are coordinate sorted (tid, pos).  Everything derives from the two seeds.
"""
from dataclasses import dataclass
from typing import List

import numpy as np

from .engine import RecordBatch


@dataclass
class SynthReference:
    names: List[str]
    lengths: np.ndarray          # int64
    genome_of_contig: np.ndarray  # int32: genome index of every contig (g{k}~c{j} naming)
    genomes: List[str]


def make_reference(n_contigs=5000, total_bp=1_000_000_000, seed=1, contigs_per_genome=10,
                   min_len=2000, max_len=2_000_000) -> SynthReference:
    rng = np.random.default_rng(seed)
    raw = np.exp(rng.uniform(np.log(min_len), np.log(max_len), n_contigs))
    lens = np.maximum(min_len, np.round(raw * (total_bp / raw.sum()))).astype(np.int64)
    n_genomes = (n_contigs + contigs_per_genome - 1) // contigs_per_genome
    g_of = (np.arange(n_contigs) // contigs_per_genome).astype(np.int32)
    names = ["g%d~c%d" % (i // contigs_per_genome, i % contigs_per_genome) for i in range(n_contigs)]
    return SynthReference(names, lens, g_of, ["g%d" % g for g in range(n_genomes)])


def make_reads(ref: SynthReference, n_reads: int, seed=2, read_len=150) -> RecordBatch:
    """n_reads records over `ref`, coordinate sorted."""
    rng = np.random.default_rng(seed)
    L = ref.lengths
    n = int(n_reads)
    w = L * rng.lognormal(0.0, 1.0, len(L))
    cdf = np.cumsum(w) / w.sum()
    tid = np.searchsorted(cdf, rng.random(n), side="right").astype(np.int32)
    np.minimum(tid, len(L) - 1, out=tid)
    room = np.maximum(L[tid] - (read_len + 250), 1)
    pos = (rng.random(n) * room).astype(np.int32)

    # CIGAR classes
    u = rng.random(n)
    cls = np.zeros(n, dtype=np.int8)            # 0: 150M
    cls[u >= 0.90] = 1                          # kM iI rM
    cls[u >= 0.96] = 2                          # kM dD rM
    cls[u >= 0.988] = 3                         # sS rM
    cls[u >= 0.998] = 4                         # k= 1X r=
    cls[u >= 0.999] = 5                         # kM nN rM
    k = rng.integers(20, 100, n).astype(np.uint32)
    ilen = rng.integers(1, 6, n).astype(np.uint32)
    dlen = rng.integers(1, 6, n).astype(np.uint32)
    slen = rng.integers(1, 40, n).astype(np.uint32)
    nlen = rng.integers(50, 200, n).astype(np.uint32)
    RL = np.uint32(read_len)

    # flags
    fr = rng.random(n)
    first = rng.random(n) < 0.5
    rev = rng.random(n) < 0.5
    flag = np.where(first, 0x40, 0x80).astype(np.uint16) | np.uint16(0x1)
    flag |= np.where(rev, 0x10, 0x20).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.95, 0x2, 0).astype(np.uint16)
    flag |= np.where(fr < 0.01, 0x100, 0).astype(np.uint16)
    flag |= np.where((fr >= 0.01) & (fr < 0.02), 0x800, 0).astype(np.uint16)
    unmapped = (fr >= 0.02) & (fr < 0.04)
    flag = np.where(unmapped, (flag & ~np.uint16(0x2)) | np.uint16(0x4), flag).astype(np.uint16)
    cls[unmapped] = -1

    mapq = rng.integers(0, 61, n).astype(np.uint8)
    mapq[rng.random(n) < 0.05] = 0
    indel = np.where(cls == 1, ilen, np.where(cls == 2, dlen, 0)).astype(np.uint32)
    nm = (rng.poisson(1.5, n).astype(np.uint32) + indel + (cls == 4).astype(np.uint32)).astype(np.uint32)
    nm_kind = np.where(unmapped, 0, 1).astype(np.uint8)
    nm[unmapped] = 0
    l_seq = np.full(n, read_len, dtype=np.uint32)

    # coordinate sort
    order = np.argsort((tid.astype(np.int64) << 32) | pos.astype(np.int64), kind="stable")
    tid, pos, cls, k, ilen, dlen, slen, nlen = (a[order] for a in (tid, pos, cls, k, ilen, dlen, slen, nlen))
    flag, mapq, nm, nm_kind = (a[order] for a in (flag, mapq, nm, nm_kind))

    n_ops = np.select([cls == -1, cls == 0, cls == 3], [0, 1, 2], default=3).astype(np.uint32)
    cigar_off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(n_ops, out=cigar_off[1:])
    cigar = np.zeros(int(cigar_off[-1]), dtype=np.uint32)
    o = cigar_off[:-1].astype(np.int64)

    def put(mask, slot, length, op):
        idx = o[mask] + slot
        cigar[idx] = (length[mask].astype(np.uint32) << 4) | np.uint32(op)

    full = np.full(n, read_len, dtype=np.uint32)
    put(cls == 0, 0, full, 0)
    m = cls == 1
    put(m, 0, k, 0); put(m, 1, ilen, 1); put(m, 2, RL - k - ilen, 0)
    m = cls == 2
    put(m, 0, k, 0); put(m, 1, dlen, 2); put(m, 2, RL - k, 0)
    m = cls == 3
    put(m, 0, slen, 4); put(m, 1, RL - slen, 0)
    m = cls == 4
    put(m, 0, k, 7); put(m, 1, np.ones(n, np.uint32), 8); put(m, 2, RL - k - 1, 7)
    m = cls == 5
    put(m, 0, k, 0); put(m, 1, nlen, 3); put(m, 2, RL - k, 0)
    return RecordBatch(tid, pos, flag, mapq, nm, nm_kind, l_seq, cigar_off, cigar)


def aligned_bases(batch: RecordBatch) -> int:
    """Total M/=/X bases of all records (for the Gbp/s figure)."""
    op = batch.cigar & 15
    ln = (batch.cigar >> 4).astype(np.int64)
    return int(ln[(op == 0) | (op == 7) | (op == 8)].sum())


def make_long_reads(ref: SynthReference, n_reads: int, seed=2, mean_len=10_000, indel_rate=0.08) -> RecordBatch:
    """Long-read profile (ONT/PacBio-like mappings): read lengths ~ lognormal around `mean_len`, one 1-3 base I or D
    every ~1/indel_rate aligned bases (so thousands of CIGAR operations per read), leading/trailing soft clips on a
    fifth of the reads.  Coordinate sorted; vectorised so tens of millions of CIGAR words take seconds."""
    rng = np.random.default_rng(seed)
    L = ref.lengths
    n = int(n_reads)
    rl = np.clip(rng.lognormal(np.log(mean_len) - 0.125, 0.5, n), 500, 60_000)
    j = rng.poisson(rl * indel_rate).astype(np.int64)              # indel events per read
    seg_read = np.repeat(np.arange(n), j + 1)                       # read of every M segment
    n_seg = len(seg_read)
    mlen = 1 + rng.geometric(indel_rate, n_seg).astype(np.int64)
    np.minimum(mlen, 400, out=mlen)
    seg_start = np.zeros(n + 1, dtype=np.int64); np.cumsum(j + 1, out=seg_start[1:])
    k_in = np.arange(n_seg) - seg_start[seg_read]                   # index of the segment within its read
    has_x = k_in < j[seg_read]                                      # every segment but the last is followed by I/D
    xop = np.where(rng.random(n_seg) < 0.5, 1, 2).astype(np.uint32)  # I / D
    xlen = rng.integers(1, 4, n_seg).astype(np.int64)
    clip5 = rng.random(n) < 0.2
    clip3 = rng.random(n) < 0.2
    nops = 2 * j + 1 + clip5 + clip3
    coff = np.zeros(n + 1, dtype=np.int64); np.cumsum(nops, out=coff[1:])
    ref_span = np.bincount(seg_read, weights=mlen + np.where(has_x & (xop == 2), xlen, 0), minlength=n).astype(np.int64)
    w = np.maximum(L - ref_span.max() - 1, 0) * rng.lognormal(0.0, 1.0, len(L))
    assert w.sum() > 0, "reference contigs are shorter than the longest read"
    cdf = np.cumsum(w) / w.sum()
    tid = np.minimum(np.searchsorted(cdf, rng.random(n), side="right"), len(L) - 1).astype(np.int32)
    pos = (rng.random(n) * (L[tid] - ref_span - 1)).astype(np.int32)
    cig = np.zeros(int(coff[-1]), dtype=np.uint32)
    base = coff[:-1] + clip5
    cig[coff[:-1][clip5]] = (rng.integers(5, 200, int(clip5.sum())).astype(np.uint32) << 4) | 4
    cig[(coff[1:] - 1)[clip3]] = (rng.integers(5, 200, int(clip3.sum())).astype(np.uint32) << 4) | 4
    mi = base[seg_read] + 2 * k_in
    cig[mi] = (mlen.astype(np.uint32) << 4)
    cig[mi[has_x] + 1] = (xlen[has_x].astype(np.uint32) << 4) | xop[has_x]
    aligned = np.bincount(seg_read, weights=mlen + np.where(has_x, xlen, 0), minlength=n)
    nm = rng.poisson(aligned * 0.03).astype(np.uint32)
    order = np.lexsort((pos, tid))
    new_off = np.zeros(n + 1, dtype=np.int64); np.cumsum(nops[order], out=new_off[1:])
    src = np.repeat(coff[:-1][order] - new_off[:-1], nops[order]) + np.arange(int(new_off[-1]))
    flag = np.where(rng.random(n) < 0.03, 0x800, 0).astype(np.uint16) | np.where(rng.random(n) < 0.5, 16, 0).astype(np.uint16) | 2
    lseq = (aligned + 0).astype(np.uint32)
    return RecordBatch.from_arrays(tid[order], pos[order], flag[order], np.full(n, 60), nm[order], np.ones(n), lseq[order],
                                   new_off.astype(np.uint32), cig[src])
