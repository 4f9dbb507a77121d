"""GPU parity cases built for the decision points of round 6's kernels, each against the CPU oracle through the C ABI (bit-exact: statistics,
histograms, identity sums, depth arrays).

k_prep_lean (csrc/prep_lean.hip.h) decides per STEP of 64 consecutive records whether its loop takes the step (one contig, the contig of
the record in front, at most three walked CIGAR operations per record, no operation of 2^24 bases) or lists it for k_prep_generic; a
wave owns 16 steps, a workgroup 4 waves.  k_pileup_fast (csrc/pileup_kernels.hip.h) walks interior tiles of 1024 bases with the stripped
loop and a contig's last tile with the clipped one; tiles of more than 8191 candidate runs go to k_pileup_stream; depths of 512 and more
leave the LDS histogram.  The cases put ONE deviation at a chosen lane / step / tile offset so that each branch is taken next to records
that take the other one.
"""
import numpy as np
import pytest

from coverm_amd.engine import RecordBatch
from tests.test_gpu_abi_parity import compare, to_bamdata

pytestmark = pytest.mark.gpu

FILTER = dict(min_percent_identity_single=0.95, min_aligned_length_single=50, min_aligned_percent_single=0.5)


def _batch(tid, pos, cigars, flag=None, nm=None, nm_kind=None, mapq=None, l_seq=None):
    n = len(tid)
    lens = np.fromiter((len(c) for c in cigars), dtype=np.int64, count=n)
    coff = np.zeros(n + 1, dtype=np.uint32)
    coff[1:] = np.cumsum(lens)
    cig = np.concatenate([np.asarray(c, dtype=np.uint32) for c in cigars]) if n else np.zeros(0, np.uint32)
    flag = np.zeros(n, np.uint16) if flag is None else np.asarray(flag, np.uint16)
    nm = np.ones(n, np.uint32) if nm is None else np.asarray(nm, np.uint32)
    nm_kind = np.ones(n, np.uint8) if nm_kind is None else np.asarray(nm_kind, np.uint8)
    mapq = np.full(n, 30, np.uint8) if mapq is None else np.asarray(mapq, np.uint8)
    l_seq = np.full(n, 100, np.int64) if l_seq is None else np.asarray(l_seq)
    return RecordBatch.from_arrays(np.asarray(tid, np.int32), np.asarray(pos, np.int32), flag, mapq, nm, nm_kind, l_seq, coff, cig)


def op(n, code):
    return (int(n) << 4) | "MIDNSHP=X".index(code)


M100 = [op(100, "M")]
KINDS = {
    # name -> (cigar of the deviating record, flag, does the contig change at it, does its step leave k_prep_lean's loop)
    "border": (M100, 0, True, True),
    "five_operations": ([op(30, "M"), op(1, "I"), op(30, "M"), op(2, "D"), op(39, "M")], 0, False, True),
    "three_operations_soft_clips": ([op(5, "S"), op(90, "M"), op(5, "S")], 0, False, False),
    "insertion_and_deletion_in_three": ([op(50, "M"), op(3, "D"), op(50, "M")], 0, False, False),
    "unmapped": (M100, 0x4, False, False),
    "supplementary": (M100, 0x800, False, False),
    # (secondary alignments are excluded by the flag filter of these cases: a record that is not walked does not count its operations)
    "secondary_with_long_cigar": ([op(10, "M"), op(1, "I")] * 9 + [op(10, "M")], 0x100, False, False),
}
# record index of the deviation = one of these + lane: the first step of a workgroup's second wave, the last step of its third wave, the
# next workgroup's first step
STEP_STARTS = [4096 + 1024, 4096 + 2048 + 15 * 64, 2 * 4096]


@pytest.mark.parametrize("filtered", [False, True], ids=["scan", "reader-filter"])
@pytest.mark.parametrize("lane", [0, 1, 31, 32, 62, 63])
@pytest.mark.parametrize("kind", sorted(KINDS))
def test_one_record_turns_a_step(kind, lane, filtered):
    """12 800 reads of 100M at a stride of 37 bases over one contig (a second contig behind it); ONE record differs.  Steps around it stay in
    k_prep_lean's loop, the step that holds it (and, for a border, the step whose predecessor it changes) goes to k_prep_generic."""
    cigar, flag_k, border, listed = KINDS[kind]
    n = 12_800
    for start in STEP_STARTS:
        k = start + lane
        tid = np.zeros(n, np.int32)
        pos = (np.arange(n) * 37).astype(np.int32)
        if border:
            tid[k:] = 1
            pos[k:] -= pos[k]
        cigars = [M100] * n
        cigars[k] = cigar
        flag = np.zeros(n, np.uint16)
        flag[k] = flag_k
        nm = (np.arange(n) % 4).astype(np.uint32)
        ref_lens = np.asarray([n * 37 + 200, n * 37 + 200, 5_000], dtype=np.int64)
        b = to_bamdata(_batch(tid, pos, cigars, flag=flag, nm=nm), ref_lens)
        paths = {}
        compare(b, ff=(True, True, False), excl=75, fp=FILTER if filtered else None, check_depth=[0, 1], paths_out=paths)
        # the sample's first step (no record in front of it) and its last one (no record behind it) are always listed
        assert paths["listed_steps"] == 2 + (1 if listed else 0) and not paths["generic_only"], paths


@pytest.mark.parametrize("lane", [0, 63])
def test_an_operation_of_two_to_the_24_bases_in_a_three_operation_cigar(lane):
    """20M 17000000N 20M: three operations, so the step passes the operation count, and the closed form then meets a length that does not fit
    its 24-bit fields (CigSum::big): the step is listed and k_prep_generic walks it in 64 bits."""
    n = 4096 + 2048
    k = 4096 + 64 * 3 + lane
    tid = np.zeros(n, np.int32)
    pos = (np.arange(n) * 11).astype(np.int32)
    cigars = [M100] * n
    cigars[k] = [op(20, "M"), op(17_000_000, "N"), op(20, "M")]
    ref_lens = np.asarray([17_200_000], dtype=np.int64)
    paths = {}
    compare(to_bamdata(_batch(tid, pos, cigars), ref_lens), ff=(True, True, False), excl=75, check_depth=[0], paths_out=paths)
    assert paths["listed_steps"] == 3, paths         # the first step, the last one and the one with the long skip


@pytest.mark.parametrize("read_len", [100, 3000])
@pytest.mark.parametrize("stride", [0, 1, 15, 16, 17, 1023, 1024, 1025, 2048, 5000])
def test_tile_index_at_position_strides(stride, read_len):
    """The tile index is built by telescoping sums (a record that opens a tile adds -i to it and +i to the tile before) and the tiles a read
    enters hear of it through an atomicMin: strides below, at and above the tile size, reads shorter and longer than a tile."""
    n = 2560 if stride == 0 else 6000
    pos = (np.arange(n, dtype=np.int64) * stride).astype(np.int32)
    tid = np.zeros(n, np.int32)
    ref_lens = np.asarray([int(pos[-1]) + read_len + 300, 2_000], dtype=np.int64)
    cigars = [[op(read_len, "M")]] * n
    b = to_bamdata(_batch(tid, pos, cigars, l_seq=np.full(n, read_len)), ref_lens)
    compare(b, ff=(True, True, False), excl=75, check_depth=[0], chunks=2)
    compare(b, ff=(True, True, False), excl=0)


@pytest.mark.parametrize("excl", [0, 75])
@pytest.mark.parametrize("d", [-17, -16, -15, -1, 0, 1, 15, 16, 17, 511])
def test_contig_ends_around_a_tile_border(d, excl):
    """A contig of 3 x 1024 + d bases covered up to its last base, another contig behind it: the last tile is walked by the clipped loop
    (whole 16-base lane slices, then the slice that holds the end), and with d <= 0 the contig has no partial tile at all."""
    rng = np.random.default_rng(100 + d)
    L = 3 * 1024 + d
    n0, n1 = 9000, 3000
    pos0 = np.sort(rng.integers(0, L - 100 + 1, n0))
    pos0[-40:] = L - 100                      # reads that end exactly at the contig's last base
    pos1 = np.sort(rng.integers(0, 900, n1))
    tid = np.concatenate([np.zeros(n0, np.int32), np.ones(n1, np.int32)])
    pos = np.concatenate([pos0, pos1]).astype(np.int32)
    b = to_bamdata(_batch(tid, pos, [M100] * (n0 + n1), nm=rng.integers(0, 3, n0 + n1)), np.asarray([L, 1000], dtype=np.int64))
    compare(b, ff=(True, True, False), excl=excl, check_depth=[0, 1])


@pytest.mark.parametrize("records", [1279, 1280, 1281])
def test_the_threshold_below_which_k_prep_generic_walks_every_step(records):
    """cov_finish launches k_prep_generic alone when a sample has fewer than 128 records per contig (covermhip.hip, gen_all): ten contigs and
    a record count just below, at and above 1280."""
    rng = np.random.default_rng(records)
    ref_lens = np.full(10, 30_000, dtype=np.int64)
    tid = np.sort(rng.integers(0, 10, records)).astype(np.int32)
    pos = np.zeros(records, np.int32)
    for t in range(10):
        m = tid == t
        pos[m] = np.sort(rng.integers(0, 29_800, int(m.sum())))
    cigars = [M100 if i % 9 else [op(40, "M"), op(1, "I"), op(59, "M")] for i in range(records)]
    b = to_bamdata(_batch(tid, pos, cigars), ref_lens)
    paths = {}
    compare(b, ff=(True, True, False), excl=75, check_depth=[0, 9], paths_out=paths)
    assert paths["generic_only"] == (records < 1280), paths
    compare(b, ff=(False, True, True), excl=0, fp=FILTER)


@pytest.mark.parametrize("runs", [8190, 8191, 8192, 8193])
def test_a_tile_at_the_fast_kernels_candidate_limit(runs):
    """k_pileup_fast takes a tile of at most 8191 candidate runs (FAST_MAX_CAND: the biased 16-bit fields hold 4 x the count); one tile of
    an interior position with exactly that many reads starting in it, one more and one fewer, between ordinary tiles."""
    rng = np.random.default_rng(runs)
    ref_lens = np.asarray([8 * 1024 + 100], dtype=np.int64)
    pos_hot = np.sort(rng.integers(3 * 1024, 4 * 1024 - 100, runs))       # start AND end inside tile 3
    pos_other = np.sort(np.concatenate([rng.integers(0, 3 * 1024 - 100, 700), rng.integers(4 * 1024, 8 * 1024, 900)]))
    pos = np.sort(np.concatenate([pos_hot, pos_other])).astype(np.int32)
    n = len(pos)
    b = to_bamdata(_batch(np.zeros(n, np.int32), pos, [M100] * n), ref_lens)
    paths = {}
    compare(b, ff=(True, True, False), excl=75, check_depth=[0], paths_out=paths)
    assert paths["slow_tiles"] == (1 if runs > 8191 else 0), paths


@pytest.mark.parametrize("where", ["interior", "last-tile"])
@pytest.mark.parametrize("depth", [511, 512, 513])
def test_a_pile_at_the_lds_histograms_last_bin(depth, where):
    """`depth` identical reads on one spot: the LDS histogram of k_pileup_fast has 512 bins (depths 0..511), deeper positions are counted in
    the arena — in an interior tile (stripped loop gives way to the general one) and in a contig's last tile (clipped loop)."""
    L = 5 * 1024 + 300
    at = 2 * 1024 + 200 if where == "interior" else 5 * 1024 + 100
    rng = np.random.default_rng(depth)
    bg = np.sort(rng.integers(0, L - 100, 2000))
    pos = np.sort(np.concatenate([bg, np.full(depth, at)])).astype(np.int32)
    n = len(pos)
    b = to_bamdata(_batch(np.zeros(n, np.int32), pos, [M100] * n), np.asarray([L, 700], dtype=np.int64))
    compare(b, ff=(True, True, False), excl=0, check_depth=[0])
