// BGZF / BAM / SAM reader of the host layer (include/coverm_host.h, covh_bam_*).
//
// The reference reads alignments through rust-htslib -> htslib (not vendored): bam::Reader::from_path
// bam_generator.rs:366, Reader::read :114, set_threads(n-1) = BGZF inflate pool :125-129.  This file
// restates the published container formats (SAM spec section 4: BGZF blocks, BAM header, alignment record
// layout, aux encoding) and yields exactly the fields the scan consumes, as the SoA batch of covermhip.h:
// tid, pos, flag, mapq, l_seq, NM (+ how it was typed, lib.rs:138-158), raw CIGAR; plus read names and mate
// tid for pair-mode filtering (filter.rs:164-176).
//
// Decode strategy: BGZF blocks are independent gzip members, so the file is split at block boundaries and
// inflated by a pool of threads straight into one contiguous buffer (offsets from each block's ISIZE);
// record boundaries are then found in one cheap serial hop over block_size fields and the per-record field
// extraction (including the linear aux scan for NM) runs in parallel again.
#include <dlfcn.h>
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/coverm_host.h"

namespace {

// Allocator for the record arrays: resize() leaves elements uninitialised (the parallel extraction loop is the
// first touch, so page faults are spread over the threads instead of a serial zero-fill), and large arrays come
// from anonymous mappings advised to use transparent huge pages.
std::atomic<bool> g_pinned_records{false};   // covh_bam_set_pinned

template <class T>
struct RecAlloc {
    using value_type = T;
    RecAlloc() = default;
    template <class U> RecAlloc(const RecAlloc<U> &) {}
    static size_t mapped_len(size_t bytes) { return (bytes + (2u << 20) - 1) & ~((size_t)(2u << 20) - 1); }
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= (1u << 20) && g_pinned_records.load(std::memory_order_relaxed))
            if (void *h = cov_host_alloc(bytes)) return (T *)h;
        if (bytes >= (8u << 20)) {
            void *m = mmap(nullptr, mapped_len(bytes), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m == MAP_FAILED) throw std::bad_alloc();
            (void)madvise(m, mapped_len(bytes), MADV_HUGEPAGE);
            return (T *)m;
        }
        void *q = malloc(bytes ? bytes : 1);
        if (!q) throw std::bad_alloc();
        return (T *)q;
    }
    void deallocate(T *q, size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= (1u << 20) && cov_host_free(q)) return;
        if (bytes >= (8u << 20)) munmap(q, mapped_len(bytes)); else free(q);
    }
    template <class U> void construct(U *q) { ::new ((void *)q) U; }   // default-init: no zeroing
    template <class U, class... A> void construct(U *q, A &&...a) { ::new ((void *)q) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const RecAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const RecAlloc<U> &) const { return false; }
};
template <class T> using RecVec = std::vector<T, RecAlloc<T>>;

struct Bam {
    std::string path, err;
    std::vector<std::string> names;
    std::vector<uint64_t> lens;
    std::string header_text;
    // SoA
    RecVec<int32_t> tid, pos, mtid;
    RecVec<uint16_t> flag;
    RecVec<uint8_t> mapq, nm_kind;
    RecVec<uint32_t> nm, l_seq, cigar_off, cigar, qname_off;
    std::string qnames;
    int threads = 1;
    bool want_names = false;
};

inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

template <typename F>
void parallel_for(size_t n, int threads, F fn) {
    threads = std::max(1, std::min<int>(threads, (int)std::max<size_t>(1, n)));
    if (threads == 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<size_t> next{0};
    const size_t grain = std::max<size_t>(1, n / (threads * 16));
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&]() {
            for (;;) {
                size_t b = next.fetch_add(grain);
                if (b >= n) break;
                size_t e = std::min(n, b + grain);
                for (size_t i = b; i < e; i++) fn(i);
            }
        });
    for (auto &th : pool) th.join();
}

// ---- uninitialised byte buffer (a std::vector would memset gigabytes before inflate overwrites them)
// Large buffers come from anonymous mappings advised to use transparent huge pages: dozens of inflate threads
// first-touching gigabytes of 4 KiB pages otherwise spend most of their time in page faults.  Unmapping gigabytes
// is itself ~0.15 s, so released mappings are either parked in a one-slot cache for the next file (opt-in,
// covh_bam_set_buffer_cache) or unmapped on a detached thread.
struct MapCache {
    static constexpr int SLOTS = 4;
    std::mutex m;
    bool enabled = false;
    void *p[SLOTS] = {nullptr, nullptr, nullptr, nullptr}; size_t len[SLOTS] = {0, 0, 0, 0};
    static void unmap_async(void *q, size_t l) {
        if (!q) return;
        try { std::thread([q, l] { munmap(q, l); }).detach(); } catch (...) { munmap(q, l); }
    }
    // best fit, but never hand a block more than 4x the request (the compressed-file buffer must not take the
    // slot sized for the inflated stream)
    void *take(size_t need, size_t &got) {
        std::lock_guard<std::mutex> lk(m);
        int best = -1;
        for (int i = 0; i < SLOTS; i++)
            if (p[i] && len[i] >= need && len[i] <= 4 * need && (best < 0 || len[i] < len[best])) best = i;
        if (best < 0) return nullptr;
        void *q = p[best]; got = len[best]; p[best] = nullptr; len[best] = 0;
        return q;
    }
    void give(void *q, size_t l) {
        void *old = q; size_t oldl = l;
        {
            std::lock_guard<std::mutex> lk(m);
            if (enabled) {
                int slot = -1;
                for (int i = 0; i < SLOTS && slot < 0; i++) if (!p[i]) slot = i;
                if (slot < 0) { slot = 0; for (int i = 1; i < SLOTS; i++) if (len[i] < len[slot]) slot = i; }
                if (!p[slot] || len[slot] < l) { old = p[slot]; oldl = len[slot]; p[slot] = q; len[slot] = l; }
            }
        }
        unmap_async(old, oldl);
    }
    void set(bool on) {
        void *old[SLOTS]; size_t oldl[SLOTS];
        {
            std::lock_guard<std::mutex> lk(m);
            enabled = on;
            for (int i = 0; i < SLOTS; i++) { old[i] = nullptr; oldl[i] = 0; if (!on) { old[i] = p[i]; oldl[i] = len[i]; p[i] = nullptr; len[i] = 0; } }
        }
        for (int i = 0; i < SLOTS; i++) if (old[i]) munmap(old[i], oldl[i]);
    }
};
MapCache g_map_cache;

struct Buf {
    uint8_t *p = nullptr; size_t n = 0; size_t mapped = 0;
    Buf() = default;
    Buf(const Buf &) = delete;
    ~Buf() { release(); }
    void release() { if (mapped) g_map_cache.give(p, mapped); else free(p); p = nullptr; n = 0; mapped = 0; }
    bool alloc(size_t k) {
        release();
        if (k >= (64u << 20)) {
            const size_t len = (k + (2u << 20) - 1) & ~((size_t)(2u << 20) - 1);
            size_t got = 0;
            if (void *c = g_map_cache.take(len, got)) { p = (uint8_t *)c; n = k; mapped = got; return true; }
            void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, len, MADV_HUGEPAGE);
                p = (uint8_t *)m; n = k; mapped = len;
                return true;
            }
        }
        p = (uint8_t *)malloc(k ? k : 1); n = k;
        return p != nullptr;
    }
    const uint8_t &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    const uint8_t *data() const { return p; }
};

// ---- libdeflate, if the runtime has it (no header in this image: prototypes declared by hand); else zlib
struct LibDeflate {
    void *(*alloc)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
    LibDeflate() {
        if (getenv("COVERM_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        crc32 = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc && decompress && release && crc32;
    }
};
const LibDeflate &libdeflate() { static LibDeflate L; return L; }
struct TlsDecompressor {
    void *d = nullptr;
    ~TlsDecompressor() { if (d) libdeflate().release(d); }
};

// ---- BGZF: block table, parallel inflate
bool bgzf_inflate_all(const Buf &raw, Buf &out, int threads, std::string &err) {
    struct Blk { size_t in_off, in_len; size_t out_off; uint32_t isize, crc; };
    std::vector<Blk> blocks;
    size_t p = 0, total = 0;
    while (p < raw.size()) {
        if (p + 18 > raw.size() || raw[p] != 0x1f || raw[p + 1] != 0x8b || raw[p + 2] != 8 || !(raw[p + 3] & 4)) {
            err = "not a BGZF block"; return false;
        }
        const uint16_t xlen = rd16(&raw[p + 10]);
        size_t q = p + 12, bsize = 0;
        while (q + 4 <= p + 12 + xlen) {
            const uint16_t slen = rd16(&raw[q + 2]);
            if (raw[q] == 66 && raw[q + 1] == 67 && slen == 2) bsize = (size_t)rd16(&raw[q + 4]) + 1;
            q += 4 + slen;
        }
        if (bsize == 0 || p + bsize > raw.size()) { err = "truncated BGZF block"; return false; }
        Blk b;
        b.in_off = p + 12 + xlen; b.in_len = bsize - 12 - xlen - 8;
        b.crc = rd32(&raw[p + bsize - 8]); b.isize = rd32(&raw[p + bsize - 4]);
        b.out_off = total;
        total += b.isize;
        blocks.push_back(b);
        p += bsize;
    }
    if (!out.alloc(total)) { err = "out of memory"; return false; }
    std::atomic<bool> ok{true};
    const LibDeflate &L = libdeflate();
    parallel_for(blocks.size(), threads, [&](size_t i) {
        const Blk &b = blocks[i];
        if (b.isize == 0) return;
        uint8_t *dst = out.p + b.out_off;
        if (L.ok) {
            thread_local TlsDecompressor tls;
            if (!tls.d) tls.d = L.alloc();
            size_t got = 0;
            if (!tls.d || L.decompress(tls.d, raw.p + b.in_off, b.in_len, dst, b.isize, &got) != 0 || got != b.isize) { ok = false; return; }
            if (L.crc32(0, dst, b.isize) != b.crc) ok = false;
            return;
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { ok = false; return; }
        zs.next_in = const_cast<Bytef *>(raw.p + b.in_off); zs.avail_in = (uInt)b.in_len;
        zs.next_out = dst; zs.avail_out = b.isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != b.isize) { ok = false; return; }
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.isize) != b.crc) ok = false;
    });
    if (!ok) { err = "BGZF inflate / CRC failure"; return false; }
    return true;
}

bool read_file(const std::string &path, Buf &out, std::string &err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "Unable to find BAM file " + path; return false; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (!out.alloc(n > 0 ? (size_t)n : 0)) { fclose(f); err = "out of memory"; return false; }
    size_t got = n > 0 ? fread(out.p, 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != out.size()) { err = "short read on " + path; return false; }
    return true;
}

// linear aux scan for NM (what htslib's bam_aux_get does); returns nm_kind
uint8_t scan_nm(const uint8_t *p, const uint8_t *end, uint32_t &nm) {
    while (p + 3 <= end) {
        const bool is_nm = p[0] == 'N' && p[1] == 'M';
        const uint8_t t = p[2];
        p += 3;
        size_t sz = 0;
        switch (t) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': {
            const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
            if (is_nm) return COV_NM_BADTYPE;
            if (!z) return COV_NM_ABSENT;
            p = z + 1;
            continue;
        }
        case 'B': {
            if (p + 5 > end) return COV_NM_ABSENT;
            const uint8_t st = p[0];
            const uint32_t cnt = rd32(p + 1);
            const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
            if (is_nm) return COV_NM_BADTYPE;
            p += 5 + (size_t)cnt * es;
            continue;
        }
        default: return COV_NM_ABSENT;  // malformed aux area
        }
        if (is_nm) {
            if (p + sz > end) return COV_NM_ABSENT;
            if (t == 'C') { nm = p[0]; return COV_NM_UNSIGNED; }
            if (t == 'S') { nm = rd16(p); return COV_NM_UNSIGNED; }
            if (t == 'I') { nm = rd32(p); return COV_NM_UNSIGNED; }
            return COV_NM_BADTYPE;   // c / s / i / A / f : the reference panics (lib.rs:144-147)
        }
        p += sz;
    }
    return COV_NM_ABSENT;
}

bool parse_bam(Bam &b, const Buf &u) {
    if (u.size() < 12 || memcmp(u.data(), "BAM\1", 4) != 0) { b.err = "bad BAM magic"; return false; }
    size_t p = 4;
    const uint32_t l_text = rd32(&u[p]); p += 4;
    if (p + l_text + 4 > u.size()) { b.err = "truncated BAM header"; return false; }
    b.header_text.assign((const char *)&u[p], l_text); p += l_text;
    const uint32_t n_ref = rd32(&u[p]); p += 4;
    b.names.reserve(n_ref); b.lens.reserve(n_ref);
    for (uint32_t i = 0; i < n_ref; i++) {
        if (p + 4 > u.size()) { b.err = "truncated BAM header"; return false; }
        const uint32_t l_name = rd32(&u[p]); p += 4;
        if (p + l_name + 4 > u.size()) { b.err = "truncated BAM header"; return false; }
        b.names.emplace_back((const char *)&u[p], l_name ? l_name - 1 : 0); p += l_name;
        b.lens.push_back(rd32(&u[p])); p += 4;
    }
    // ---- record boundaries.  Each record only says how long it is, so finding them is a pointer chase; done
    // serially it is a chain of cache misses (~60 ns per record) and bounds the whole decode.  Parallel scheme:
    // cut the buffer into segments, let every thread FIND a record start inside its segment (an offset from which
    // a chain of 8 records is plausible), hop its own segment, and accept the result only if each segment's
    // chain lands exactly on the start the next segment found — then it equals the serial hop by induction.
    // Any mismatch falls back to the plain serial hop, so the outcome never depends on the heuristic.
    const size_t N = u.size();
    const int32_t n_ref_i = (int32_t)n_ref;
    auto plausible = [&](size_t o) -> size_t {   // returns record end, or 0 if `o` cannot start a record
        if (o + 36 > N) return 0;
        const uint32_t bs = rd32(&u[o]);
        if (bs < 32 || o + 4 + (size_t)bs > N) return 0;
        const int32_t rid = (int32_t)rd32(&u[o + 4]), pos = (int32_t)rd32(&u[o + 8]), nrid = (int32_t)rd32(&u[o + 24]);
        const uint32_t lname = u[o + 12], ncig = rd16(&u[o + 16]), lseq = rd32(&u[o + 20]);
        if (rid < -1 || rid >= n_ref_i || nrid < -1 || nrid >= n_ref_i || pos < -1 || lname == 0) return 0;
        if (rid >= 0 && (uint64_t)pos > b.lens[rid]) return 0;
        const uint64_t fixed = 32ull + lname + 4ull * ncig + (lseq + 1ull) / 2 + lseq;
        if (fixed > bs) return 0;
        if (u[o + 36 + lname - 1] != 0) return 0;
        return o + 4 + bs;
    };
    struct Seg { size_t start = 0, stop = 0; bool found = false; std::vector<size_t> rec; size_t landed = 0; };
    RecVec<size_t> rec;
    bool parallel_ok = false;
    const size_t body = N - p;
    int nseg = b.threads > 1 ? std::min<int>(b.threads * 4, (int)(body / (1 << 20))) : 0;
    if (nseg >= 2) {
        std::vector<Seg> seg(nseg);
        for (int k = 0; k < nseg; k++) seg[k].stop = p + body * (size_t)(k + 1) / nseg;
        seg[0].start = p; seg[0].found = true;
        parallel_for((size_t)nseg, b.threads, [&](size_t k) {
            Seg &S = seg[k];
            if (k > 0) {
                const size_t from = seg[k - 1].stop;
                for (size_t o = from; o < S.stop && !S.found; o++) {
                    size_t q = o; int chain = 0;
                    while (chain < 8) { const size_t e = plausible(q); if (!e) break; q = e; chain++; if (q == N) break; }
                    if (chain == 8 || (chain > 0 && q == N)) { S.start = o; S.found = true; }
                }
            }
        });
        // a segment without a start (one huge record spans it) is absorbed by its predecessor
        std::vector<int> live;
        for (int k = 0; k < nseg; k++) if (seg[k].found) live.push_back(k);
        parallel_for(live.size(), b.threads, [&](size_t j) {
            Seg &S = seg[live[j]];
            const size_t limit = j + 1 < live.size() ? seg[live[j + 1]].start : N;
            size_t q = S.start;
            S.rec.reserve((limit - q) / 200 + 16);
            while (q < limit && q + 4 <= N) {
                const uint32_t bs = rd32(&u[q]);
                if (bs < 32 || q + 4 + (size_t)bs > N) break;
                S.rec.push_back(q);
                q += 4 + (size_t)bs;
            }
            S.landed = q;
        });
        parallel_ok = true;
        for (size_t j = 0; j < live.size() && parallel_ok; j++) {
            const size_t want = j + 1 < live.size() ? seg[live[j + 1]].start : N;
            if (seg[live[j]].landed != want) parallel_ok = false;
        }
        if (parallel_ok) {
            std::vector<size_t> base(live.size() + 1, 0);
            for (size_t j = 0; j < live.size(); j++) base[j + 1] = base[j] + seg[live[j]].rec.size();
            rec.resize(base.back());
            parallel_for(live.size(), b.threads, [&](size_t j) {
                const auto &v = seg[live[j]].rec;
                if (!v.empty()) memcpy(&rec[base[j]], v.data(), v.size() * sizeof(size_t));
            });
        }
    }
    if (!parallel_ok) {   // serial hop (small inputs, one thread, or a failed speculation)
        rec.clear();
        rec.reserve(body / 200 + 16);
        size_t q = p;
        while (q + 4 <= N) {
            const uint32_t bs = rd32(&u[q]);
            if (bs < 32 || q + 4 + (size_t)bs > N) { b.err = "truncated BAM record"; return false; }
            rec.push_back(q);
            q += 4 + (size_t)bs;
        }
        if (q != N) { b.err = "truncated BAM record"; return false; }
    }
    // CIGAR / name offsets: per-record counts in parallel, then a prefix sum
    {
        const size_t Rn = rec.size();
        b.cigar_off.resize(Rn + 1); b.cigar_off[0] = 0;
        if (b.want_names) { b.qname_off.resize(Rn + 1); b.qname_off[0] = 0; }
        // blocked inclusive scan: per-block counts, serial scan of the block totals, per-block fix-up
        const size_t nblk = std::max<size_t>(1, std::min<size_t>((size_t)b.threads * 4, Rn / 65536 + 1));
        std::vector<uint64_t> cs(nblk + 1, 0), qs(nblk + 1, 0);
        parallel_for(nblk, b.threads, [&](size_t k) {
            const size_t lo = Rn * k / nblk, hi = Rn * (k + 1) / nblk;
            uint64_t c = 0, q = 0;
            for (size_t i = lo; i < hi; i++) {
                c += rd16(&u[rec[i] + 16]); b.cigar_off[i + 1] = (uint32_t)c;
                if (b.want_names) { q += u[rec[i] + 12] ? u[rec[i] + 12] - 1u : 0u; b.qname_off[i + 1] = (uint32_t)q; }
            }
            cs[k + 1] = c; qs[k + 1] = q;
        });
        for (size_t k = 0; k < nblk; k++) { cs[k + 1] += cs[k]; qs[k + 1] += qs[k]; }
        if (cs[nblk] > 0xffffffffull || qs[nblk] > 0xffffffffull) { b.err = "BAM too large for 32-bit CIGAR/name offsets"; return false; }
        parallel_for(nblk, b.threads, [&](size_t k) {
            if (!k) return;
            const size_t lo = Rn * k / nblk, hi = Rn * (k + 1) / nblk;
            const uint32_t c = (uint32_t)cs[k], q = (uint32_t)qs[k];
            for (size_t i = lo; i < hi; i++) { b.cigar_off[i + 1] += c; if (b.want_names) b.qname_off[i + 1] += q; }
        });
    }
    const size_t R = rec.size();
    b.tid.resize(R); b.pos.resize(R); b.mtid.resize(R); b.flag.resize(R); b.mapq.resize(R); b.nm_kind.resize(R);
    b.nm.resize(R); b.l_seq.resize(R);
    b.cigar.resize(b.cigar_off[R]);
    if (b.want_names) b.qnames.resize(b.qname_off[R]);
    std::atomic<bool> ok{true};
    parallel_for(R, b.threads, [&](size_t i) {
        const uint8_t *r = &u[rec[i]];
        const uint32_t bs = rd32(r);
        const uint8_t *end = r + 4 + bs;
        b.tid[i] = (int32_t)rd32(r + 4); b.pos[i] = (int32_t)rd32(r + 8);
        const uint32_t l_read_name = r[12];
        b.mapq[i] = r[13];
        const uint32_t n_cig = rd16(r + 16);
        b.flag[i] = rd16(r + 18);
        const uint32_t l_seq = rd32(r + 20);
        b.l_seq[i] = l_seq;
        b.mtid[i] = (int32_t)rd32(r + 24);
        const uint8_t *q = r + 36;
        if (q + l_read_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq > end) { ok = false; return; }
        if (b.want_names && l_read_name) memcpy(&b.qnames[b.qname_off[i]], q, l_read_name - 1);
        q += l_read_name;
        if (n_cig) memcpy(&b.cigar[b.cigar_off[i]], q, 4ull * n_cig);
        q += 4ull * n_cig + (l_seq + 1) / 2 + l_seq;
        uint32_t nm = 0;
        b.nm_kind[i] = scan_nm(q, end, nm);
        b.nm[i] = nm;
    });
    if (!ok) { b.err = "corrupt BAM record"; return false; }
    return true;
}

// ---- SAM text (htslib auto-detects the format; needed for e.g. tests/data/mapq_test.sam, filter.rs:758)
bool parse_sam(Bam &b, const Buf &raw) {
    const char *s = (const char *)raw.data(), *e = s + raw.size();
    std::vector<std::pair<std::string, size_t>> dummy;
    auto find_ref = [&](const std::string &n) -> int32_t {
        for (size_t i = 0; i < b.names.size(); i++) if (b.names[i] == n) return (int32_t)i;
        return -1;
    };
    b.cigar_off.assign(1, 0u);
    if (b.want_names) b.qname_off.assign(1, 0u);
    while (s < e) {
        const char *nl = (const char *)memchr(s, '\n', (size_t)(e - s));
        const char *le = nl ? nl : e;
        std::string line(s, le);
        s = nl ? nl + 1 : e;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '@') {
            b.header_text += line; b.header_text += '\n';
            if (line.compare(0, 3, "@SQ") == 0) {
                std::string sn; uint64_t ln = 0;
                size_t p = 3;
                while (p < line.size()) {
                    size_t t = line.find('\t', p + 1);
                    std::string f = line.substr(p + 1, t == std::string::npos ? std::string::npos : t - p - 1);
                    if (f.compare(0, 3, "SN:") == 0) sn = f.substr(3);
                    else if (f.compare(0, 3, "LN:") == 0) ln = strtoull(f.c_str() + 3, nullptr, 10);
                    if (t == std::string::npos) break;
                    p = t;
                }
                b.names.push_back(sn); b.lens.push_back(ln);
            }
            continue;
        }
        std::vector<std::string> f;
        size_t p = 0;
        for (;;) {
            size_t t = line.find('\t', p);
            f.push_back(line.substr(p, t == std::string::npos ? std::string::npos : t - p));
            if (t == std::string::npos) break;
            p = t + 1;
        }
        if (f.size() < 11) { b.err = "malformed SAM line"; return false; }
        const int32_t t = f[2] == "*" ? -1 : find_ref(f[2]);
        b.tid.push_back(t);
        b.flag.push_back((uint16_t)strtoul(f[1].c_str(), nullptr, 10));
        b.pos.push_back((int32_t)strtol(f[3].c_str(), nullptr, 10) - 1);
        b.mapq.push_back((uint8_t)strtoul(f[4].c_str(), nullptr, 10));
        if (f[5] != "*") {
            uint32_t num = 0;
            for (char ch : f[5]) {
                if (ch >= '0' && ch <= '9') num = num * 10 + (uint32_t)(ch - '0');
                else {
                    const char *ops = "MIDNSHP=X";
                    const char *o = strchr(ops, ch);
                    b.cigar.push_back((num << 4) | (uint32_t)(o ? o - ops : 15));
                    num = 0;
                }
            }
        }
        b.cigar_off.push_back((uint32_t)b.cigar.size());
        b.mtid.push_back(f[6] == "=" ? t : (f[6] == "*" ? -1 : find_ref(f[6])));
        b.l_seq.push_back(f[9] == "*" ? 0u : (uint32_t)f[9].size());
        uint32_t nm = 0; uint8_t k = COV_NM_ABSENT;
        for (size_t a = 11; a < f.size(); a++) {
            if (f[a].compare(0, 3, "NM:") == 0 && f[a].size() > 5) {
                // htslib stores a SAM `i` value in the smallest fitting type; non-negative => unsigned
                if (f[a][3] == 'i' && f[a][5] != '-') { nm = (uint32_t)strtoul(f[a].c_str() + 5, nullptr, 10); k = COV_NM_UNSIGNED; }
                else k = COV_NM_BADTYPE;
            }
        }
        b.nm.push_back(nm); b.nm_kind.push_back(k);
        if (b.want_names) { b.qnames += f[0]; b.qname_off.push_back((uint32_t)b.qnames.size()); }
    }
    return true;
}

}  // namespace

extern "C" {

struct covh_bam { Bam b; };

covh_bam *covh_bam_open(const char *path, int threads, int want_names, char *err, size_t errcap) {
    const bool timing = getenv("COVERM_BAM_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(), t1 = t0, t2 = t0, t3 = t0;
    covh_bam *h = new covh_bam();
    h->b.path = path; h->b.threads = std::max(1, threads); h->b.want_names = want_names != 0;
    Buf raw;
    bool ok = read_file(path, raw, h->b.err);
    if (ok && raw.size() == 0) { ok = false; h->b.err = std::string(path) + ": empty file (no BAM/SAM header)"; }   // htslib: sam_hdr_read fails
    t1 = now();
    if (ok) {
        if (raw.size() >= 2 && raw[0] == 0x1f && raw[1] == 0x8b) {
            Buf u;
            ok = bgzf_inflate_all(raw, u, h->b.threads, h->b.err);
            raw.alloc(0);
            t2 = now();
            if (ok) ok = parse_bam(h->b, u);
            t3 = now();
            const size_t umb = u.size() >> 20;
            u.alloc(0);
            if (timing) fprintf(stderr, "[covh_bam] read %.3fs inflate %.3fs (%zu MB) parse %.3fs release %.3fs\n", t1 - t0, t2 - t1, umb, t3 - t2, now() - t3);
        } else ok = parse_sam(h->b, raw);
    }
    if (!ok) {
        if (err && errcap) { strncpy(err, h->b.err.c_str(), errcap - 1); err[errcap - 1] = 0; }
        delete h;
        return nullptr;
    }
    return h;
}
void covh_bam_close(covh_bam *h) { delete h; }
void covh_bam_set_buffer_cache(int on) { g_map_cache.set(on != 0); }
void covh_bam_set_pinned(int on) { g_pinned_records.store(on != 0); }
uint32_t covh_bam_n_targets(const covh_bam *h) { return (uint32_t)h->b.names.size(); }
const char *covh_bam_target_name(const covh_bam *h, uint32_t i) { return h->b.names[i].c_str(); }
uint64_t covh_bam_target_len(const covh_bam *h, uint32_t i) { return h->b.lens[i]; }
uint64_t covh_bam_n_records(const covh_bam *h) { return h->b.tid.size(); }
void covh_bam_batch(const covh_bam *h, cov_batch *out) {
    const Bam &b = h->b;
    out->tid = b.tid.data(); out->pos = b.pos.data(); out->flag = b.flag.data(); out->mapq = b.mapq.data();
    out->nm = b.nm.data(); out->nm_kind = b.nm_kind.data(); out->l_seq = b.l_seq.data();
    out->cigar_off = b.cigar_off.data(); out->cigar = b.cigar.data(); out->n_records = b.tid.size();
}
const int32_t *covh_bam_mtid(const covh_bam *h) { return h->b.mtid.data(); }
const uint32_t *covh_bam_qname_off(const covh_bam *h) { return h->b.qname_off.empty() ? nullptr : h->b.qname_off.data(); }
const char *covh_bam_qnames(const covh_bam *h) { return h->b.qnames.data(); }
uint64_t covh_bam_n_cigar(const covh_bam *h) { return h->b.cigar.size(); }

// ---- writer: SoA batch -> BGZF BAM (synthetic benchmark inputs; SEQ is all 'A', QUAL 0xff, or '*' when !with_seq).
// Records are serialised in parallel slices, cut into <= 0xff00-byte blocks and deflated by the thread pool.
int covh_bam_write(const char *path, uint32_t n_targets, const char *const *names, const uint64_t *lens,
                   const cov_batch *b, int with_seq, int level, int threads) {
    std::vector<uint8_t> head;
    auto put32 = [](std::vector<uint8_t> &v, uint32_t x) { uint8_t t[4]; memcpy(t, &x, 4); v.insert(v.end(), t, t + 4); };
    head.insert(head.end(), {'B', 'A', 'M', 1});
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (uint32_t i = 0; i < n_targets; i++) text += std::string("@SQ\tSN:") + names[i] + "\tLN:" + std::to_string(lens[i]) + "\n";
    put32(head, (uint32_t)text.size()); head.insert(head.end(), text.begin(), text.end());
    put32(head, n_targets);
    for (uint32_t i = 0; i < n_targets; i++) {
        const size_t l = strlen(names[i]) + 1;
        put32(head, (uint32_t)l); head.insert(head.end(), names[i], names[i] + l); put32(head, (uint32_t)lens[i]);
    }
    const uint64_t R = b->n_records;
    // per-record sizes -> offsets
    std::vector<uint64_t> off(R + 1, 0);
    for (uint64_t i = 0; i < R; i++) {
        const uint32_t nc = b->cigar_off[i + 1] - b->cigar_off[i];
        const uint32_t ls = with_seq ? b->l_seq[i] : 0;
        char qn[32]; const int lq = snprintf(qn, sizeof qn, "r%llu", (unsigned long long)i) + 1;
        const uint32_t aux = b->nm_kind[i] == COV_NM_UNSIGNED ? (b->nm[i] < 256 ? 4 : b->nm[i] < 65536 ? 5 : 7) : b->nm_kind[i] == COV_NM_BADTYPE ? 4 : 0;
        off[i + 1] = off[i] + 36 + lq + 4ull * nc + (ls + 1) / 2 + ls + aux;
    }
    std::vector<uint8_t> raw(head.size() + off[R]);
    memcpy(raw.data(), head.data(), head.size());
    uint8_t *base = raw.data() + head.size();
    parallel_for(R, threads, [&](size_t i) {
        uint8_t *p = base + off[i];
        const uint32_t nc = b->cigar_off[i + 1] - b->cigar_off[i];
        const uint32_t ls = with_seq ? b->l_seq[i] : 0;
        char qn[32]; const int lq = snprintf(qn, sizeof qn, "r%llu", (unsigned long long)i) + 1;
        const uint32_t bs = (uint32_t)(off[i + 1] - off[i] - 4);
        auto w32 = [&](uint32_t x) { memcpy(p, &x, 4); p += 4; };
        w32(bs); w32((uint32_t)b->tid[i]); w32((uint32_t)b->pos[i]);
        *p++ = (uint8_t)lq; *p++ = b->mapq[i];
        const uint16_t bin = 4680, ncg = (uint16_t)nc, fl = b->flag[i];
        memcpy(p, &bin, 2); p += 2; memcpy(p, &ncg, 2); p += 2; memcpy(p, &fl, 2); p += 2;
        w32(ls); w32((uint32_t)-1); w32((uint32_t)-1); w32(0);
        memcpy(p, qn, lq); p += lq;
        if (nc) { memcpy(p, b->cigar + b->cigar_off[i], 4ull * nc); p += 4ull * nc; }
        memset(p, 0x11, (ls + 1) / 2); p += (ls + 1) / 2;
        memset(p, 0xff, ls); p += ls;
        if (b->nm_kind[i] == COV_NM_UNSIGNED) {
            *p++ = 'N'; *p++ = 'M';
            if (b->nm[i] < 256) { *p++ = 'C'; *p++ = (uint8_t)b->nm[i]; }
            else if (b->nm[i] < 65536) { *p++ = 'S'; const uint16_t v = (uint16_t)b->nm[i]; memcpy(p, &v, 2); p += 2; }
            else { *p++ = 'I'; memcpy(p, &b->nm[i], 4); p += 4; }
        } else if (b->nm_kind[i] == COV_NM_BADTYPE) { *p++ = 'N'; *p++ = 'M'; *p++ = 'c'; *p++ = 1; }
    });
    const size_t BLK = 0xff00;
    const size_t nblk = (raw.size() + BLK - 1) / BLK;
    std::vector<std::vector<uint8_t>> comp(nblk);
    std::atomic<bool> ok{true};
    parallel_for(nblk, threads, [&](size_t k) {
        const size_t s0 = k * BLK, n = std::min(BLK, raw.size() - s0);
        std::vector<uint8_t> &o = comp[k];
        o.resize(n + n / 8 + 128);
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok = false; return; }
        zs.next_in = &raw[s0]; zs.avail_in = (uInt)n; zs.next_out = o.data() + 18; zs.avail_out = (uInt)(o.size() - 26);
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { ok = false; deflateEnd(&zs); return; }
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(o.data(), hdr, 16);
        const uint16_t bsz = (uint16_t)(clen + 25);
        memcpy(o.data() + 16, &bsz, 2);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), &raw[s0], (uInt)n), isz = (uint32_t)n;
        memcpy(o.data() + 18 + clen, &crc, 4); memcpy(o.data() + 22 + clen, &isz, 4);
        o.resize(clen + 26);
    });
    if (!ok) return 1;
    FILE *f = fopen(path, "wb");
    if (!f) return 2;
    for (auto &o : comp) fwrite(o.data(), 1, o.size(), f);
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, f);
    fclose(f);
    return 0;
}

}  // extern "C"
