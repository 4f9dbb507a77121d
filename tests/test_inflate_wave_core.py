"""csrc/inflate_wave_core.h (one wave per BGZF block: three passes over a shared Huffman table) run on the CPU.

The header is the whole algorithm of k_inflate_wave; tests/c/inflate_wave_host.cpp instantiates it with a loop over the 64 lanes.  Every
block is inflated, its match tokens are executed the way k_lz_resolve does, and the bytes must be zlib's.  The reference's decoder for
this step is htslib's bgzf inflate (bam_generator.rs:125-129 hands the file to it); zlib is the same RFC 1951.
"""
import ctypes as C
import glob
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module", params=["lanes 0..63", "lanes 63..0", "lanes 0..63, 8-byte sink", "lanes 63..0, 8-byte sink", "lanes 63..0, 3 extra literals"])
def host(request, tmp_path_factory):
    # the lanes of a COVW_PARFOR region run concurrently on the device; here they run one after the other, in both orders
    so = str(tmp_path_factory.mktemp("covw") / "covw_host.so")
    subprocess.check_call(["g++", "-O2", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "c", "inflate_wave_host.cpp")]
                          + (["-DCOVW_REVERSE"] if "..0" in request.param else []) + (["-DCOVW_SINK_OLD"] if "8-byte" in request.param else []) + (["-DCOVW_EXTRA_LITS=3"] if "extra" in request.param else []))
    L = C.CDLL(so)
    L.covw_host_inflate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.covw_host_inflate.restype = C.c_int
    L.covw_host_resolve.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.covw_host_resolve.restype = C.c_int
    L.covw_host_wave_bytes.restype = C.c_uint32
    L.covw_host_last_deflate_blocks.restype = C.c_uint32
    L.covw_host_last_chunks.restype = C.c_uint32
    return L


def inflate(L, payload: bytes, isize: int, misalign=0, fill=0xA5):
    """-> (status, bytes or None, tokens, pass-2 rounds of the last Huffman block)"""
    out = np.zeros(isize + 16, dtype=np.uint8)
    tok = np.zeros(21888, dtype=np.uint16)
    nt, rounds = C.c_uint32(0), C.c_uint32(0)
    src = np.frombuffer(payload, dtype=np.uint8) if payload else np.zeros(1, dtype=np.uint8)
    st = L.covw_host_inflate(src.ctypes.data, len(payload), misalign, fill, out.ctypes.data, isize, tok.ctypes.data, C.byref(nt), C.byref(rounds))
    if st != 0:
        return st, None, 0, rounds.value
    body = out[8:8 + isize]
    assert L.covw_host_resolve(body.ctypes.data, isize, tok.ctypes.data, nt.value) == 0
    return 0, body.tobytes(), nt.value, rounds.value


def raw_deflate(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=None) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if flush_every is None:
        return co.compress(data) + co.flush()
    parts = []
    for i in range(0, len(data), flush_every):
        parts.append(co.compress(data[i:i + flush_every]))
        parts.append(co.flush(zlib.Z_FULL_FLUSH))        # ends the Huffman block, adds an empty stored block
    parts.append(co.flush())
    return b"".join(parts)


def bgzf_blocks(path):
    """(payload, isize, crc) of every BGZF block of a file (SAM spec 4.1)"""
    buf = open(path, "rb").read()
    p = 0
    while p < len(buf):
        assert buf[p:p + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", buf, p + 10)[0]
        q, bsize = p + 12, None
        while q < p + 12 + xlen:
            si1, si2, slen = buf[q], buf[q + 1], struct.unpack_from("<H", buf, q + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", buf, q + 4)[0] + 1
            q += 4 + slen
        payload = buf[p + 12 + xlen:p + bsize - 8]
        crc, isize = struct.unpack_from("<II", buf, p + bsize - 8)
        yield payload, isize, crc
        p += bsize


def bam_like(rng, n):
    """bytes with the statistics of BAM records: small little-endian integers, 4-bit packed bases, repetitive qualities, short tags"""
    parts = []
    size = 0
    while size < n:
        l_seq = int(rng.integers(50, 151))
        core = struct.pack("<iiBBHHHIiii", 300 + l_seq, int(rng.integers(0, 50)), 20, int(rng.integers(0, 61)), 4681, 2, int(rng.integers(0, 4)) * 16, l_seq,
                           -1, -1, 0)
        name = b"read_%07d\0" % int(rng.integers(0, 10 ** 7))
        cigar = struct.pack("<II", (l_seq - 10) << 4, 10 << 4 | 4)
        seq = rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8).tobytes() if rng.random() < 0.5 else bytes([0x11, 0x24, 0x48, 0x82]) * ((l_seq + 7) // 8)
        qual = bytes(rng.choice(np.array([37, 37, 37, 25, 11, 2], dtype=np.uint8), l_seq))
        tags = b"NMC" + bytes([int(rng.integers(0, 5))]) + b"ASC" + bytes([int(rng.integers(0, 200))])
        rec = core + name + cigar + seq + qual + tags
        parts.append(rec)
        size += len(rec)
    return b"".join(parts)[:n]


def test_state_fits_twenty_five_waves_per_cu(host):
    assert host.covw_host_wave_bytes() <= 6528      # 160 KiB of LDS per CU / 6.4 KiB: 25 waves (the registers allow 16)


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("misalign", [0, 1, 3])
def test_bam_like_blocks_match_zlib(host, level, misalign):
    rng = np.random.default_rng(100 * level + misalign)
    for size in (0xff00, 40000, 3000, 64, 1):
        data = bam_like(rng, size)
        comp = raw_deflate(data, level)
        st, got, nt, rounds = inflate(host, comp, len(data), misalign)
        assert st == 0 and got == data, (level, size, st)
        assert rounds <= 3, (size, rounds)


def test_fixed_huffman_stored_and_multi_block_payloads(host):
    rng = np.random.default_rng(7)
    data = bam_like(rng, 30000)
    for comp in (raw_deflate(data, 6, zlib.Z_FIXED),                      # fixed codes
                 raw_deflate(data, 0),                                     # stored
                 raw_deflate(data, 6, flush_every=7000),                   # five Huffman blocks with empty stored blocks between them
                 raw_deflate(data, 1, zlib.Z_HUFFMAN_ONLY),                # no matches at all
                 raw_deflate(bytes(30000), 9),                             # one literal and maximal matches at distance 1
                 raw_deflate(rng.integers(0, 256, 30000, dtype=np.uint8).tobytes(), 6)):      # incompressible: zlib stores or codes 8-9 bit literals
        want = zlib.decompress(comp, -15)
        for fill in (0x00, 0xff, 0xA5):
            st, got, nt, rounds = inflate(host, comp, len(want), 2, fill)
            assert st == 0 and got == want


def test_blocks_behind_an_early_end_are_taken_in_chunks(host):
    # libdeflate splits a BGZF block into several DEFLATE blocks; what follows a short first block is decoded in chunks of about its length,
    # continuing with the same tables until the end-of-block symbol turns up
    rng = np.random.default_rng(23)
    data = bam_like(rng, 60000)
    for cuts in ([3000], [3000, 6000, 40000], [500, 1000, 1500], [30000], [59990]):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        parts, at = [], 0
        for c in cuts + [len(data)]:
            parts.append(co.compress(data[at:c]))
            parts.append(co.flush(zlib.Z_FULL_FLUSH) if c < len(data) else co.flush())
            at = c
        comp = b"".join(parts)
        st, got, nt, rounds = inflate(host, comp, len(data), 1)
        assert st == 0 and got == data, cuts
        n_blocks, n_chunks = host.covw_host_last_deflate_blocks(), host.covw_host_last_chunks()
        assert n_blocks >= len(cuts) + 1
        if cuts[0] <= 3000:
            assert n_chunks > n_blocks - len(cuts), (cuts, n_blocks, n_chunks)      # some block took more than one chunk


def test_long_codes_behind_the_primary_tables(host):
    # a skewed alphabet gives 12-15 bit literal codes (primary table: 11 bits) and long distance codes
    rng = np.random.default_rng(11)
    p = 0.5 ** np.arange(1, 257, dtype=np.float64) + 1e-5
    data = rng.choice(256, 60000, p=p / p.sum()).astype(np.uint8).tobytes()
    for level in (1, 6, 9):
        comp = raw_deflate(data, level)
        st, got, nt, rounds = inflate(host, comp, len(data))
        assert st == 0 and got == data


def test_empty_block_and_eof_marker(host):
    eof = bytes.fromhex("0300")              # the payload of the BGZF EOF block: one fixed-code block holding end-of-block only
    st, got, nt, rounds = inflate(host, eof, 0)
    assert st == 0 and got == b""


def test_damaged_streams_are_errors_not_crashes(host):
    rng = np.random.default_rng(3)
    data = bam_like(rng, 20000)
    comp = raw_deflate(data, 6)
    # wrong size in the trailer
    assert inflate(host, comp, len(data) - 1)[0] != 0
    assert inflate(host, comp, len(data) + 1)[0] != 0
    # truncated payload
    assert inflate(host, comp[:len(comp) // 2], len(data))[0] != 0
    assert inflate(host, comp[:5], len(data))[0] != 0
    assert inflate(host, b"", len(data))[0] != 0
    # reserved block type
    assert inflate(host, b"\x07" + comp[1:], len(data))[0] != 0
    # flipped bits anywhere: an error or different bytes (which the CRC kernel catches), never a write outside the block (-1 = canary)
    n_err = 0
    for k in range(200):
        bad = bytearray(comp)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        st, got, nt, rounds = inflate(host, bytes(bad), len(data))
        assert st >= 0
        n_err += st != 0 or got != data
    assert n_err >= 190
    # random bytes as a payload
    for k in range(50):
        st = inflate(host, rng.integers(0, 256, 500, dtype=np.uint8).tobytes(), 4000)[0]
        assert st >= 0


def test_every_block_of_the_reference_bams(host):
    files = sorted(glob.glob(os.path.join(HERE, "golden", "raw", "*.bam")))
    assert len(files) >= 19
    n = 0
    for f in files:
        for payload, isize, crc in bgzf_blocks(f):
            st, got, nt, rounds = inflate(host, payload, isize, n & 3)
            assert st == 0, (f, n, st)
            assert zlib.crc32(got) == crc, (f, n)
            n += 1
    assert n >= 40


def test_blocks_written_by_the_product_writer(host, tmp_path):
    # the kind of file bench.py measures on: covh_bam_write at level 1 and 6, with sequences and qualities
    from coverm_amd import bam as cbam, synth
    ref = synth.make_reference(9, 300_000, seed=2, min_len=1500, max_len=80_000)
    b = synth.make_reads(ref, 30_000, seed=3)
    rounds_seen = []
    for level in (1, 6):
        path = str(tmp_path / ("w%d.bam" % level))
        cbam.write_bam(path, ref.names, ref.lengths, b, with_seq=2, level=level, threads=2)
        n = 0
        for payload, isize, crc in bgzf_blocks(path):
            st, got, nt, rounds = inflate(host, payload, isize, n % 4)
            assert st == 0 and zlib.crc32(got) == crc
            rounds_seen.append(rounds)
            n += 1
        assert n >= 10
    assert np.mean(rounds_seen) < 1.5, np.bincount(rounds_seen)     # pass 2 runs once on nearly every block


def test_sanitizer_run_over_valid_and_damaged_streams(tmp_path):
    """tests/c/inflate_wave_fuzz.cpp: exact-size buffers under AddressSanitizer + UBSan — the 64 readable bytes behind the payload, the isize
    output bytes and the TOK_CAP token positions are all the core may touch, whatever the stream holds."""
    exe = str(tmp_path / "covw_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe, os.path.join(HERE, "c", "inflate_wave_fuzz.cpp"), "-lz"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime / zlib headers here: " + r.stderr[-200:])
    for seed in ("1", "7"):
        r = subprocess.run([exe, "120", seed], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "120 valid streams exact" in r.stdout

