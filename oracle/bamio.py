"""TEST INFRASTRUCTURE ONLY — pure-Python BGZF/BAM/SAM reader + minimal BAM writer.

Nothing in the product path (coverm_amd/, bench.py's timed GPU leg) may import this module;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do (as the checker).

The reference reads BAM through rust-htslib 0.46.0 -> hts-sys 2.2.0 -> htslib C
(Cargo.lock:1643-1645, 872-874), which is not vendored under /root/reference.  This file
restates the published BGZF + BAM record layout (SAM spec v1, section 4) so the oracle
can consume the reference's fixture BAMs without htslib.  Call sites anchoring what is
consumed: bam_generator.rs:113-119 (read), contig.rs:124-168 (tid/pos/cigar/flags),
lib.rs:138-158 (NM aux, accepted types C/S/I), filter.rs:164-176 (qname/mtid, pair mode).

The writer half produces BGZF-compressed BAM from the same SoA record layout; it exists
so tests can build synthetic BAMs (ops `=`/`X`/`N`, ragged/empty inputs) that no reference
fixture contains, following the record field choices of tests/test_cmdline.rs:4212-4312.
"""
import struct
import zlib
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# nm_kind codes shared with include/covermhip.h (COV_NM_*)
NM_ABSENT = 0      # no NM aux tag          -> reference panics in nm() (lib.rs:149-156)
NM_UNSIGNED = 1    # type C / S / I         -> accepted (lib.rs:141-143)
NM_BADTYPE = 2     # type c / s / i / other -> reference panics (lib.rs:144-147)

CIGAR_OPS = "MIDNSHP=X"


@dataclass
class BamData:
    """SoA view of a whole BAM/SAM file, in file order."""
    ref_names: List[str]
    ref_lens: np.ndarray          # int64[n_ref]
    tid: np.ndarray               # int32[R]
    pos: np.ndarray               # int32[R]
    flag: np.ndarray              # uint16[R]
    mapq: np.ndarray              # uint8[R]
    l_seq: np.ndarray             # int32[R]   (uint32 across the ABI)
    nm: np.ndarray                # uint32[R]
    nm_kind: np.ndarray           # uint8[R]
    cigar_off: np.ndarray         # uint32[R+1]  CSR into `cigar`
    cigar: np.ndarray             # uint32[C]    len<<4 | op
    mtid: np.ndarray              # int32[R]
    mpos: np.ndarray              # int32[R]
    tlen: np.ndarray              # int32[R]
    qname: List[bytes] = field(default_factory=list)
    header_text: str = ""

    @property
    def n_records(self) -> int:
        return int(self.tid.shape[0])

    def select(self, idx) -> "BamData":
        """Records idx (array of indices, any order) as a new BamData (CIGAR re-packed)."""
        idx = np.asarray(idx, dtype=np.int64)
        n = (self.cigar_off[1:] - self.cigar_off[:-1]).astype(np.int64)[idx]
        off = np.zeros(len(idx) + 1, dtype=np.uint32)
        np.cumsum(n, out=off[1:])
        cig = np.zeros(int(off[-1]), dtype=np.uint32)
        for j, i in enumerate(idx):
            cig[off[j]:off[j + 1]] = self.cigar[self.cigar_off[i]:self.cigar_off[i + 1]]
        return BamData(self.ref_names, self.ref_lens, self.tid[idx], self.pos[idx], self.flag[idx],
                       self.mapq[idx], self.l_seq[idx], self.nm[idx], self.nm_kind[idx], off, cig,
                       self.mtid[idx], self.mpos[idx], self.tlen[idx],
                       [self.qname[i] for i in idx] if self.qname else [], self.header_text)


# ----------------------------------------------------------------------------- BGZF
def bgzf_decompress(raw: bytes) -> bytes:
    """Concatenated gzip members, each with the BC extra subfield; payload is raw DEFLATE."""
    out = []
    p = 0
    n = len(raw)
    while p < n:
        if raw[p:p + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("not a BGZF block at offset %d" % p)
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        q = p + 12
        bsize = None
        while q < p + 12 + xlen:
            si1, si2, slen = struct.unpack_from("<BBH", raw, q)
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", raw, q + 4)[0] + 1
            q += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC subfield")
        cdata = raw[p + 12 + xlen: p + bsize - 8]
        data = zlib.decompress(cdata, -15)
        crc, isize = struct.unpack_from("<II", raw, p + bsize - 8)
        if len(data) != isize or (zlib.crc32(data) & 0xFFFFFFFF) != crc:
            raise ValueError("BGZF block CRC/ISIZE mismatch")
        out.append(data)
        p += bsize
    return b"".join(out)


_AUX_FIXED = {ord("A"): 1, ord("c"): 1, ord("C"): 1, ord("s"): 2, ord("S"): 2,
              ord("i"): 4, ord("I"): 4, ord("f"): 4}


def _scan_aux_nm(buf: bytes, p: int, end: int):
    """Linear aux scan for NM (what htslib's bam_aux_get does).  Returns (nm, nm_kind)."""
    while p + 3 <= end:
        tag = buf[p:p + 2]
        typ = buf[p + 2]
        p += 3
        if typ in _AUX_FIXED:
            sz = _AUX_FIXED[typ]
            if tag == b"NM":
                if typ == ord("C"):
                    return buf[p], NM_UNSIGNED
                if typ == ord("S"):
                    return struct.unpack_from("<H", buf, p)[0], NM_UNSIGNED
                if typ == ord("I"):
                    return struct.unpack_from("<I", buf, p)[0], NM_UNSIGNED
                return 0, NM_BADTYPE
            p += sz
        elif typ in (ord("Z"), ord("H")):
            z = buf.index(b"\x00", p)
            if tag == b"NM":
                return 0, NM_BADTYPE
            p = z + 1
        elif typ == ord("B"):
            sub = buf[p]
            cnt = struct.unpack_from("<I", buf, p + 1)[0]
            if tag == b"NM":
                return 0, NM_BADTYPE
            p += 5 + cnt * _AUX_FIXED[sub]
        else:
            raise ValueError("unknown aux type %r" % chr(typ))
    return 0, NM_ABSENT


def read_bam(path: str) -> BamData:
    with open(path, "rb") as fh:
        raw = fh.read()
    buf = bgzf_decompress(raw)
    if buf[:4] != b"BAM\x01":
        raise ValueError("bad BAM magic")
    l_text = struct.unpack_from("<i", buf, 4)[0]
    text = buf[8:8 + l_text].decode("utf-8", "replace")
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", buf, p)[0]
    p += 4
    names, lens = [], np.zeros(n_ref, dtype=np.int64)
    for i in range(n_ref):
        l_name = struct.unpack_from("<i", buf, p)[0]
        names.append(buf[p + 4:p + 4 + l_name - 1].decode())
        lens[i] = struct.unpack_from("<i", buf, p + 4 + l_name)[0]
        p += 8 + l_name
    tid, pos, flag, mapq, lseq, nm, nmk, ncig, mtid, mpos, tlen, qn = ([] for _ in range(12))
    cig_chunks = []
    n = len(buf)
    while p < n:
        block_size = struct.unpack_from("<i", buf, p)[0]
        (refid, rpos, l_read_name, mq, _bin, n_cigar, flg, l_seq, nref, npos,
         tl) = struct.unpack_from("<iiBBHHHiiii", buf, p + 4)
        q = p + 36
        qn.append(buf[q:q + l_read_name - 1])
        q += l_read_name
        cig = np.frombuffer(buf, dtype="<u4", count=n_cigar, offset=q).astype(np.uint32)
        q += 4 * n_cigar
        q += (l_seq + 1) // 2 + l_seq
        v, k = _scan_aux_nm(buf, q, p + 4 + block_size)
        tid.append(refid); pos.append(rpos); flag.append(flg); mapq.append(mq)
        lseq.append(l_seq); nm.append(v); nmk.append(k); ncig.append(n_cigar)
        mtid.append(nref); mpos.append(npos); tlen.append(tl)
        cig_chunks.append(cig)
        p += 4 + block_size
    return _pack(names, lens, tid, pos, flag, mapq, lseq, nm, nmk, ncig, cig_chunks, mtid, mpos, tlen, qn, text)


def _pack(names, lens, tid, pos, flag, mapq, lseq, nm, nmk, ncig, cig_chunks, mtid, mpos, tlen, qn, text):
    off = np.zeros(len(tid) + 1, dtype=np.uint32)
    if len(tid):
        np.cumsum(np.asarray(ncig, dtype=np.int64), out=off[1:])
    cig = np.concatenate(cig_chunks).astype(np.uint32) if cig_chunks else np.zeros(0, np.uint32)
    return BamData(list(names), np.asarray(lens, dtype=np.int64),
                   np.asarray(tid, np.int32), np.asarray(pos, np.int32), np.asarray(flag, np.uint16),
                   np.asarray(mapq, np.uint8), np.asarray(lseq, np.int32), np.asarray(nm, np.uint32),
                   np.asarray(nmk, np.uint8), off, cig, np.asarray(mtid, np.int32),
                   np.asarray(mpos, np.int32), np.asarray(tlen, np.int32), list(qn), text)


# ----------------------------------------------------------------------------- SAM text
def read_sam(path: str) -> BamData:
    """Minimal SAM text reader (only for fixtures such as tests/data/mapq_test.sam, which the
    reference opens through htslib's format auto-detection, filter.rs:758)."""
    names, lens = [], []
    rows = []
    text = []
    with open(path, "r") as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line:
                continue
            if line.startswith("@"):
                text.append(line)
                if line.startswith("@SQ"):
                    d = dict(f.split(":", 1) for f in line.split("\t")[1:])
                    names.append(d["SN"]); lens.append(int(d["LN"]))
                continue
            rows.append(line.split("\t"))
    idx = {n: i for i, n in enumerate(names)}
    tid, pos, flag, mapq, lseq, nm, nmk, ncig, mtid, mpos, tlen, qn = ([] for _ in range(12))
    cig_chunks = []
    for f in rows:
        qn.append(f[0].encode())
        flag.append(int(f[1]))
        t = idx[f[2]] if f[2] != "*" else -1
        tid.append(t)
        pos.append(int(f[3]) - 1)
        mapq.append(int(f[4]))
        ops = []
        if f[5] != "*":
            num = ""
            for ch in f[5]:
                if ch.isdigit():
                    num += ch
                else:
                    ops.append((int(num) << 4) | CIGAR_OPS.index(ch))
                    num = ""
        cig_chunks.append(np.asarray(ops, dtype=np.uint32))
        ncig.append(len(ops))
        mt = t if f[6] == "=" else (idx[f[6]] if f[6] != "*" else -1)
        mtid.append(mt)
        mpos.append(int(f[7]) - 1)
        tlen.append(int(f[8]))
        lseq.append(0 if f[9] == "*" else len(f[9]))
        v, k = 0, NM_ABSENT
        for aux in f[11:]:
            tag, typ, val = aux.split(":", 2)
            if tag == "NM":
                # htslib stores SAM 'i' aux in the smallest fitting type; non-negative -> unsigned
                if typ == "i" and int(val) >= 0:
                    v, k = int(val), NM_UNSIGNED
                else:
                    v, k = 0, NM_BADTYPE
        nm.append(v); nmk.append(k)
    return _pack(names, lens, tid, pos, flag, mapq, lseq, nm, nmk, ncig, cig_chunks, mtid, mpos, tlen, qn,
                 "\n".join(text))


def read_alignment_file(path: str) -> BamData:
    with open(path, "rb") as fh:
        magic = fh.read(2)
    return read_bam(path) if magic == b"\x1f\x8b" else read_sam(path)


# ----------------------------------------------------------------------------- writer
def _bgzf_block(data: bytes, level: int) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = co.compress(data) + co.flush()
    bsize = len(cdata) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def write_bam(path: str, d: BamData, level: int = 1, with_seq: bool = True, block: int = 0xFF00,
              nm_type: str = "C") -> None:
    """Write `d` as BGZF BAM.  SEQ is all 'A' / QUAL 0xff when with_seq (l_seq bytes), else
    SEQ='*' with l_seq forced to 0.  NM is written with BAM type `nm_type` for NM_UNSIGNED
    records ('C','S','I'), as type 'c' for NM_BADTYPE records, and omitted for NM_ABSENT."""
    parts = [b"BAM\x01"]
    text = d.header_text or "@HD\tVN:1.6\tSO:coordinate\n" + "".join(
        "@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in zip(d.ref_names, d.ref_lens))
    tb = text.encode()
    parts.append(struct.pack("<i", len(tb))); parts.append(tb)
    parts.append(struct.pack("<i", len(d.ref_names)))
    for n, l in zip(d.ref_names, d.ref_lens):
        nb = n.encode() + b"\x00"
        parts.append(struct.pack("<i", len(nb))); parts.append(nb); parts.append(struct.pack("<i", int(l)))
    fmt = {"C": "<B", "S": "<H", "I": "<I"}[nm_type]
    for i in range(d.n_records):
        cig = d.cigar[d.cigar_off[i]:d.cigar_off[i + 1]]
        qn = (d.qname[i] if d.qname else b"r%d" % i) + b"\x00"
        l_seq = int(d.l_seq[i]) if with_seq else 0
        span = int(sum(int(c) >> 4 for c in cig if (int(c) & 15) in (0, 2, 3, 7, 8)))
        p = int(d.pos[i])
        body = struct.pack("<iiBBHHHiiii", int(d.tid[i]), p, len(qn), int(d.mapq[i]),
                           _reg2bin(max(p, 0), max(p, 0) + max(span, 1)), len(cig), int(d.flag[i]), l_seq,
                           int(d.mtid[i]), int(d.mpos[i]), int(d.tlen[i]))
        body += qn + cig.astype("<u4").tobytes()
        body += b"\x11" * ((l_seq + 1) // 2) + b"\xff" * l_seq
        k = int(d.nm_kind[i])
        if k == NM_UNSIGNED:
            body += b"NM" + nm_type.encode() + struct.pack(fmt, int(d.nm[i]))
        elif k == NM_BADTYPE:
            body += b"NMc" + struct.pack("<b", 1)
        parts.append(struct.pack("<i", len(body))); parts.append(body)
    raw = b"".join(parts)
    with open(path, "wb") as fh:
        for s in range(0, len(raw), block):
            fh.write(_bgzf_block(raw[s:s + block], level))
        fh.write(BGZF_EOF)
