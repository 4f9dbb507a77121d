"""Feasibility probe for a WAVE-per-block inflate: how quickly does a DEFLATE decoder that starts at an arbitrary bit
offset inside a BGZF block's stream fall into step with the true symbol sequence?

    python tools/proto/deflate_sync.py <file.bam> [n_blocks=12] [lanes=64]

For every (dynamic / fixed Huffman) DEFLATE block inside the first n BGZF blocks that are full-size: the true unit boundaries (a unit =
literal, or length + extra bits + distance + extra bits, or end of block) from a plain serial decode; then `lanes` evenly spaced start
offsets, each decoded with the block's own tables as if a unit began there, until the trajectory lands on a true boundary (then it stays on
it for good) or leaves the block.  Prints the distribution of the synchronisation distance in bits and in units, and what two / three
passes of "start where the left neighbour ended" would resolve.  Test infrastructure: pure Python, no product code involved."""
import sys
import zlib

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
CLO = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self, data):
        self.v = int.from_bytes(data, "little")
        self.n = len(data) * 8

    def get(self, pos, k):
        return (self.v >> pos) & ((1 << k) - 1)


def table(lens):
    """canonical Huffman: dict (length, code MSB-first) -> symbol"""
    cnt = [0] * 16
    for l in lens:
        cnt[l] += 1
    cnt[0] = 0
    code, nxt = 0, [0] * 16
    for l in range(1, 16):
        code = (code + cnt[l - 1]) << 1
        nxt[l] = code
    t = {}
    for s, l in enumerate(lens):
        if l:
            t[(l, nxt[l])] = s
            nxt[l] += 1
    return t


def sym(b, pos, t):
    code = 0
    for l in range(1, 16):
        code = (code << 1) | b.get(pos + l - 1, 1)
        s = t.get((l, code))
        if s is not None:
            return s, l
    return None, 0


def unit(b, pos, tl, td, end):
    """one unit at `pos`: (next pos, kind) with kind 'lit' / 'match' / 'eob' / None (invalid here)"""
    if pos >= end:
        return None, None
    s, l = sym(b, pos, tl)
    if s is None:
        return None, None
    pos += l
    if s < 256:
        return pos, "lit"
    if s == 256:
        return pos, "eob"
    if s > 285:
        return None, None
    pos += LEXT[s - 257]
    d, l = sym(b, pos, td)
    if d is None or d > 29:
        return None, None
    return pos + l + DEXT[d], "match"


def deflate_blocks(b):
    """yields (first unit bit, tables, [true unit boundaries incl. the position behind EOB]) for every Huffman block of the stream"""
    pos, last = 0, 0
    while not last:
        last = b.get(pos, 1); typ = b.get(pos + 1, 2); pos += 3
        if typ == 0:
            pos = (pos + 7) & ~7
            ln = b.get(pos, 16); pos += 32 + 8 * ln
            continue
        if typ == 1:
            ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
            dl = [5] * 30
        else:
            hlit = b.get(pos, 5) + 257; hdist = b.get(pos + 5, 5) + 1; hclen = b.get(pos + 10, 4) + 4; pos += 14
            cl = [0] * 19
            for i in range(hclen):
                cl[CLO[i]] = b.get(pos, 3); pos += 3
            tc = table(cl)
            lens = []
            while len(lens) < hlit + hdist:
                s, l = sym(b, pos, tc); pos += l
                if s < 16: lens.append(s)
                elif s == 16: r = 3 + b.get(pos, 2); pos += 2; lens += [lens[-1]] * r
                elif s == 17: r = 3 + b.get(pos, 3); pos += 3; lens += [0] * r
                else: r = 11 + b.get(pos, 7); pos += 7; lens += [0] * r
            ll, dl = lens[:hlit], lens[hlit:]
        tl, td = table(ll), table(dl)
        first, bounds = pos, [pos]
        while True:
            pos, kind = unit(b, pos, tl, td, b.n)
            bounds.append(pos)
            if kind == "eob":
                break
        yield first, tl, td, bounds, ll


def main():
    path = sys.argv[1]
    n_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    raw = open(path, "rb").read(n_blocks * 70000 + 200000)
    q, done = 0, 0
    dist_bits, dist_units, unsynced, n_starts = [], [], 0, 0
    per_block = []
    while done < n_blocks and q + 18 < len(raw):
        bs = int.from_bytes(raw[q + 16:q + 18], "little") + 1
        payload = raw[q + 18:q + bs - 8]
        isize = int.from_bytes(raw[q + bs - 4:q + bs], "little")
        q += bs
        if isize < 60000:
            continue
        assert len(zlib.decompress(payload, -15)) == isize
        b = Bits(payload + b"\0" * 8)
        nblk = 0
        for first, tl, td, bounds, ll in deflate_blocks(b):
            nblk += 1
            end = bounds[-1]
            span = end - first
            if span < lanes * 64:
                continue
            bset = set(bounds)
            index = {p: k for k, p in enumerate(bounds)}
            S = span // lanes
            worst = 0
            for i in range(1, lanes):
                p = first + i * S
                n_starts += 1
                units = 0
                start = p
                while p is not None and p not in bset and p < end:
                    p, _ = unit(b, p, tl, td, end)
                    units += 1
                if p is None or p not in bset:
                    unsynced += 1
                    worst = max(worst, S * 4)
                else:
                    dist_bits.append(p - start); dist_units.append(units); worst = max(worst, p - start)
            per_block.append((nblk, span, len(bounds) - 1, S, worst, sum(1 for x in ll if x), max(ll)))
        done += 1
    dist_bits.sort(); dist_units.sort()
    pct = lambda a, f: a[min(len(a) - 1, int(f * len(a)))] if a else None
    print("%d BGZF blocks, %d Huffman blocks probed, %d starts (%d lanes per block): %d never fell into step inside the block" % (done, len(per_block), n_starts, lanes, unsynced))
    print("synchronisation distance, bits : median %s  p90 %s  p99 %s  max %s" % (pct(dist_bits, .5), pct(dist_bits, .9), pct(dist_bits, .99), dist_bits[-1] if dist_bits else None))
    print("synchronisation distance, units: median %s  p90 %s  p99 %s  max %s" % (pct(dist_units, .5), pct(dist_units, .9), pct(dist_units, .99), dist_units[-1] if dist_units else None))
    for nblk, span, units, S, worst, nsym, maxlen in per_block[:16]:
        print("   block #%d of its BGZF block: %6d bits, %5d units, %4d bits per lane, worst start in step after %5d bits (%.2f of a lane's share); %d codes, longest %d bits" % (nblk, span, units, S, worst, worst / S, nsym, maxlen))
    bad = sum(1 for x in per_block if x[4] > x[3])
    print("Huffman blocks in which some lane needs more than its own share to fall into step (a third pass): %d of %d" % (bad, len(per_block)))


if __name__ == "__main__":
    main()
