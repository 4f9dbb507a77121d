"""csrc/crc_wave_core.h (the CRC-32 of a BGZF block by one wave: 64 columns of 8-byte words, tables for a distance of 512 bytes, a scan
over the lanes) run on the CPU: tests/c/crc_wave_host.cpp loops over the 64 lanes; every result must be zlib's crc32 — what htslib checks
for every block the reference reads (bgzf.c inflate_block behind bam_generator.rs:125-129)."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("crcw") / "crcw_host.so")
    subprocess.check_call(["g++", "-O2", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "c", "crc_wave_host.cpp")])
    L = C.CDLL(so)
    L.crcw_host_crc.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.crcw_host_crc.restype = C.c_uint32
    L.crcw_host_tables.restype = C.POINTER(C.c_uint32)
    L.crcw_host_table_words.restype = C.c_uint32
    return L


def crc(L, buf, off, n, reverse=0):
    return L.crcw_host_crc(buf.ctypes.data + off, n, reverse)


def test_tables(host):
    """HI8[3] is the classic byte table; LO_D[j][b] is (b << 8 j) advanced over D zero bytes — checked through zlib: the CRC register after
    a message is linear, so crc(m || zeros(D)) ^ crc(zeros(len(m) + D)) only depends on m's own register."""
    n = host.crcw_host_table_words()
    assert n == 9 * 1024
    T = np.ctypeslib.as_array(host.crcw_host_tables(), shape=(n,)).copy()
    classic = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (0xEDB88320 ^ (c >> 1)) if c & 1 else c >> 1
        classic[i] = c
    np.testing.assert_array_equal(T[1024 + 768:1024 + 1024], classic)

    def advance(v, zeros):       # the register v over `zeros` zero bytes, bytewise
        for _ in range(zeros):
            v = int(classic[v & 0xFF]) ^ (v >> 8)
        return v
    for base, dist in [(0, 8), (2048, 16), (3072, 32), (6144, 256), (7168, 512)]:
        for j in range(4):
            for b in (1, 0x80, 0xA7, 0xFF):
                assert int(T[base + 256 * j + b]) == advance(b << (8 * j), dist), (dist, j, b)
    for j in range(4):
        for b in (1, 0x5C, 0xFF):
            assert int(T[1024 + 256 * j + b]) == advance(b << (8 * j), 4)
            assert int(T[8192 + 256 * j + b]) == advance(b << (8 * j), 508)


SIZES = sorted(set(list(range(0, 41)) + [63, 64, 65, 127, 128, 129, 255, 256, 257] + list(range(500, 531)) + list(range(1015, 1035))
                   + [4095, 4096, 4097, 32768, 65279, 65280, 65281, 65535, 65536]))


@pytest.mark.parametrize("fill", ["random", "zeros", "ones"])
def test_every_size_class_and_alignment(host, fill):
    """Block lengths around every boundary of the scheme (the bytewise limit of 16, a word, a row of 512 bytes, two rows, htslib's 0xff00
    and the format's 65536) at all eight alignments, the lanes in both orders."""
    rng = np.random.default_rng(5)
    buf = {"random": rng.integers(0, 256, 70_000, dtype=np.uint8), "zeros": np.zeros(70_000, np.uint8), "ones": np.full(70_000, 0xFF, np.uint8)}[fill]
    for n in SIZES:
        for a in range(8):
            off = 64 + a
            want = zlib.crc32(buf[off:off + n].tobytes())
            assert crc(host, buf, off, n) == want, (n, a)
            assert crc(host, buf, off, n, reverse=1) == want, (n, a, "reverse")


def test_random_blocks_inside_a_stream(host):
    """Blocks as the ingest sees them: consecutive pieces of random lengths of one inflated stream, each starting where the last ended."""
    rng = np.random.default_rng(9)
    buf = rng.integers(0, 256, 3_000_000, dtype=np.uint8)
    buf[100_000:400_000] = 0          # low-entropy stretches
    buf[900_000:1_000_000] = 0x21
    off = 8
    while off + 65536 + 8 < len(buf):
        n = int(rng.choice([int(rng.integers(0, 700)), int(rng.integers(0, 65537)), 65280]))
        assert crc(host, buf, off, n, reverse=int(rng.integers(0, 2))) == zlib.crc32(buf[off:off + n].tobytes()), (off, n)
        off += n


def test_a_flipped_bit_is_seen(host):
    rng = np.random.default_rng(11)
    buf = rng.integers(0, 256, 70_000, dtype=np.uint8)
    for n in (16, 17, 511, 512, 513, 65280):
        good = crc(host, buf, 67, n)
        for pos in (0, 1, n // 2, n - 2, n - 1):
            buf[67 + pos] ^= 0x10
            assert crc(host, buf, 67, n) != good
            buf[67 + pos] ^= 0x10
        assert crc(host, buf, 67, n) == good
