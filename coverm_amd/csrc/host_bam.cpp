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

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <unordered_map>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <emmintrin.h>      // SSE2 (x86-64 baseline): non-temporal stores for the staging slots

#include "../../include/coverm_host.h"
#include "reader_filter.h"
#include "roctx_ranges.h"
#include "knobs.h"

namespace {

// Allocator for the record arrays: resize() leaves elements uninitialised (the parallel extraction loop is the
// first touch, so page faults are spread over the threads instead of a serial zero-fill), and large arrays come
// from anonymous mappings advised to use transparent huge pages.
std::atomic<bool> g_pinned_records{false};   // covh_bam_set_pinned
std::atomic<bool> g_release_staging{false};  // covh_bam_set_release_staging
std::atomic<int> g_feeders{1};               // covh_bam_set_concurrent_feeders: device ingests the caller runs at once

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

// Persistent worker pool for the streamed reader: windows are small (tens of MB), so spawning threads per parallel
// loop (parallel_for above: ~50 us per thread) would cost more than the loop.  run() hands out index ranges from an
// atomic counter; the calling thread works too.
class Pool {
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void(size_t)> fn_;
    size_t n_ = 0, grain_ = 1;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    int busy_ = 0;
    bool stop_ = false;
    void drain() {
        for (;;) {
            const size_t b = next_.fetch_add(grain_);
            if (b >= n_) break;
            const size_t e = std::min(n_, b + grain_);
            for (size_t i = b; i < e; i++) fn_(i);
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            drain();
            { std::lock_guard<std::mutex> lk(m_); if (--busy_ == 0) done_.notify_all(); }
        }
    }
public:
    explicit Pool(int threads) { for (int t = 1; t < std::max(1, threads); t++) th_.emplace_back([this] { worker(); }); }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size() + 1; }
    template <typename F>
    void run(size_t n, F f, size_t min_grain = 1) {
        if (n == 0) return;
        if (th_.empty() || n <= min_grain) { for (size_t i = 0; i < n; i++) f(i); return; }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = f; n_ = n; grain_ = std::max(min_grain, n / ((th_.size() + 1) * 8)); next_ = 0;
            busy_ = (int)th_.size(); gen_++;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return busy_ == 0; });
    }
};

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
    void *(*alloc_c)(int) = nullptr;
    size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    void (*release_c)(void *) = nullptr;
    bool ok = false, ok_c = false;
    LibDeflate() {
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        crc32 = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc && decompress && release && crc32;
        alloc_c = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        release_c = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        ok_c = ok && alloc_c && compress && release_c;
    }
};
const LibDeflate &libdeflate() { static LibDeflate L; return L; }
struct TlsDecompressor {
    void *d = nullptr;
    ~TlsDecompressor() { if (d) libdeflate().release(d); }
};
// one compressor per thread and level, given back when the thread ends (the writers' loops run on short-lived threads)
struct TlsCompressor {
    void *c = nullptr; int level = -1;
    void *get(int lv) {
        const LibDeflate &L = libdeflate();
        if (!L.ok_c) return nullptr;
        if (!c || level != lv) { if (c) L.release_c(c); c = L.alloc_c(std::max(1, std::min(12, lv))); level = lv; }
        return c;
    }
    ~TlsCompressor() { if (c) libdeflate().release_c(c); }
};

// ---- BGZF: block table, parallel inflate
struct Blk { size_t in_off, in_len; size_t out_off; uint32_t isize, crc; };

// Parses the BGZF blocks that lie completely inside raw[p, n): `p` advances past them, `total` grows by their ISIZE.
// Returns false on a malformed block (`err` set).  A block cut off by `n` is an error unless partial_tail_ok (streaming:
// the tail is re-presented with the next window).  Every header read is bounds-checked against `n`.
bool bgzf_block_table(const uint8_t *raw, size_t n, size_t &p, std::vector<Blk> &blocks, size_t &total, bool partial_tail_ok,
                      std::string &err) {
    while (p < n) {
        if (p + 18 > n) { if (partial_tail_ok) return true; err = "truncated BGZF block"; return false; }
        if (raw[p] != 0x1f || raw[p + 1] != 0x8b || raw[p + 2] != 8 || !(raw[p + 3] & 4)) { err = "not a BGZF block"; return false; }
        const size_t xlen = rd16(&raw[p + 10]);
        if (p + 12 + xlen > n) { if (partial_tail_ok) return true; err = "truncated BGZF block"; return false; }
        size_t q = p + 12, bsize = 0;
        while (q + 4 <= p + 12 + xlen) {
            const size_t slen = rd16(&raw[q + 2]);
            if (raw[q] == 66 && raw[q + 1] == 67 && slen == 2 && q + 6 <= p + 12 + xlen) bsize = (size_t)rd16(&raw[q + 4]) + 1;
            q += 4 + slen;
        }
        if (bsize == 0 || bsize < 12 + xlen + 8) { err = "malformed BGZF block header"; return false; }
        if (p + bsize > n) { if (partial_tail_ok) return true; err = "truncated BGZF block"; return false; }
        Blk b;
        b.in_off = p + 12 + xlen; b.in_len = bsize - 12 - xlen - 8;
        b.crc = rd32(&raw[p + bsize - 8]); b.isize = rd32(&raw[p + bsize - 4]);
        if (b.isize > 65536) { err = "BGZF block inflates to more than 64 KiB"; return false; }
        b.out_off = total;
        total += b.isize;
        blocks.push_back(b);
        p += bsize;
    }
    return true;
}

bool inflate_block(const uint8_t *raw, const Blk &b, uint8_t *dst) {
    if (b.isize == 0) return true;
    const LibDeflate &L = libdeflate();
    if (L.ok) {
        thread_local TlsDecompressor tls;
        if (!tls.d) tls.d = L.alloc();
        size_t got = 0;
        if (!tls.d || L.decompress(tls.d, raw + b.in_off, b.in_len, dst, b.isize, &got) != 0 || got != b.isize) return false;
        return L.crc32(0, dst, b.isize) == b.crc;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(raw + b.in_off); zs.avail_in = (uInt)b.in_len;
    zs.next_out = dst; zs.avail_out = b.isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END || zs.total_out != b.isize) return false;
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.isize) == b.crc;
}

bool bgzf_inflate_all(const Buf &raw, Buf &out, int threads, std::string &err) {
    std::vector<Blk> blocks;
    size_t p = 0, total = 0;
    if (!bgzf_block_table(raw.p, raw.size(), p, blocks, total, false, err)) return false;
    if (!out.alloc(total)) { err = "out of memory"; return false; }
    std::atomic<bool> ok{true};
    parallel_for(blocks.size(), threads, [&](size_t i) { if (!inflate_block(raw.p, blocks[i], out.p + blocks[i].out_off)) ok = false; });
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

// Linear aux scan (what htslib's bam_aux_get does): NM value + how it was typed (returns nm_kind), and the CG:B,I/i
// array if present (the real CIGAR of a record with more than 65535 operations, SAM spec 4.2.2).
uint8_t scan_aux(const uint8_t *p, const uint8_t *end, uint32_t &nm, const uint8_t **cg, uint32_t *cg_cnt) {
    uint8_t kind = COV_NM_ABSENT;
    bool have_nm = false;
    while (p + 3 <= end) {
        const bool is_nm = !have_nm && p[0] == 'N' && p[1] == 'M';
        const bool is_cg = cg != nullptr && p[0] == 'C' && p[1] == 'G';
        const uint8_t t = p[2];
        p += 3;
        size_t sz = 0;
        switch (t) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': {
            const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
            if (is_nm) { kind = COV_NM_BADTYPE; have_nm = true; }
            if (!z) return kind;
            p = z + 1;
            continue;
        }
        case 'B': {
            if (p + 5 > end) return kind;
            const uint8_t st = p[0];
            const uint32_t cnt = rd32(p + 1);
            const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
            if (is_nm) { kind = COV_NM_BADTYPE; have_nm = true; }
            if ((size_t)(end - p - 5) / es < cnt) return kind;
            if (is_cg && (st == 'I' || st == 'i') && *cg == nullptr) { *cg = p + 5; *cg_cnt = cnt; }
            p += 5 + (size_t)cnt * es;
            continue;
        }
        default: return kind;  // malformed aux area
        }
        if (p + sz > end) return kind;
        if (is_nm) {
            have_nm = true;
            if (t == 'C') { nm = p[0]; kind = COV_NM_UNSIGNED; }
            else if (t == 'S') { nm = rd16(p); kind = COV_NM_UNSIGNED; }
            else if (t == 'I') { nm = rd32(p); kind = COV_NM_UNSIGNED; }
            else kind = COV_NM_BADTYPE;   // c / s / i / A / f : the reference panics (lib.rs:144-147)
            if (cg == nullptr) return kind;
        }
        p += sz;
    }
    return kind;
}

// A record whose CIGAR has more than 65535 operations stores `<l_seq>S<ref_len>N` in the CIGAR field and the real one in
// CG:B,I; htslib swaps it back in while reading (bam_tag2cigar, sam.c), so record.cigar() at contig.rs:168 sees the real
// CIGAR.  Same conditions here.  Returns the CG payload and count, or nullptr.
const uint8_t *real_cigar_from_cg(const uint8_t *r, const uint8_t *end, uint32_t &cnt) {
    const uint32_t n_cig = rd16(r + 16), l_read_name = r[12], l_seq = rd32(r + 20);
    if (n_cig == 0 || (int32_t)rd32(r + 4) < 0 || (int32_t)rd32(r + 8) < 0) return nullptr;
    const uint8_t *cg0 = r + 36 + l_read_name;
    if (cg0 + 4ull * n_cig > end) return nullptr;
    const uint32_t c0 = rd32(cg0);
    if ((c0 & 15u) != 4u || (c0 >> 4) != l_seq) return nullptr;
    const uint8_t *aux = cg0 + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq;
    if (aux > end) return nullptr;
    const uint8_t *cg = nullptr; uint32_t nm = 0; cnt = 0;
    (void)scan_aux(aux, end, nm, &cg, &cnt);
    if (cg == nullptr || cnt < n_cig || cnt >= (1u << 29)) return nullptr;
    return cg;
}
inline bool maybe_long_cigar(const uint8_t *r) {   // cheap pre-test: the placeholder always has exactly two operations
    return rd16(r + 16) == 2;
}

// BAM header (magic, text, reference dictionary) at u[0, N).  Returns 1 and sets `p` to the first record when complete,
// 0 when more bytes are needed, -1 on error.
int parse_bam_header(const uint8_t *u, size_t N, size_t &p, std::vector<std::string> &names, std::vector<uint64_t> &lens,
                     std::string &text, std::string &err) {
    if (N < 12) return 0;
    if (memcmp(u, "BAM\1", 4) != 0) { err = "bad BAM magic"; return -1; }
    size_t q = 4;
    const uint32_t l_text = rd32(&u[q]); q += 4;
    if (q + (size_t)l_text + 4 > N) return 0;
    const size_t text_at = q;
    q += l_text;
    const uint32_t n_ref = rd32(&u[q]); q += 4;
    if ((size_t)n_ref > (N - q) / 9 + 1) return 0;   // every entry takes at least 9 bytes: not all here yet
    // first walk: is the whole dictionary here?  (The caller asks again after every MiB it has inflated; a header of 2 M references is
    // 40 MB, and building — and dropping — a million names per question made reading it 0.5 s.)
    {
        size_t w = q;
        for (uint32_t i = 0; i < n_ref; i++) {
            if (w + 4 > N) return 0;
            const uint32_t l_name = rd32(&u[w]); w += 4;
            if (w + (size_t)l_name + 4 > N) return 0;
            w += (size_t)l_name + 4;
        }
    }
    std::vector<std::string> nm; std::vector<uint64_t> ln;
    nm.reserve(n_ref); ln.reserve(n_ref);
    for (uint32_t i = 0; i < n_ref; i++) {
        const uint32_t l_name = rd32(&u[q]); q += 4;
        nm.emplace_back((const char *)&u[q], l_name ? l_name - 1 : 0); q += l_name;
        ln.push_back(rd32(&u[q])); q += 4;
    }
    text.assign((const char *)&u[text_at], l_text);
    names.swap(nm); lens.swap(ln);
    p = q;
    return 1;
}

// Can a record start at offset o?  Returns the record end, or 0.
inline size_t plausible_record(const uint8_t *u, size_t N, size_t o, int32_t n_ref, const uint64_t *lens) {
    if (o + 36 > N) return 0;
    const uint32_t bs = rd32(&u[o]);
    if (bs < 32 || o + 4 + (size_t)bs > N) return 0;
    const int32_t rid = (int32_t)rd32(&u[o + 4]), pos = (int32_t)rd32(&u[o + 8]), nrid = (int32_t)rd32(&u[o + 24]);
    const uint32_t lname = u[o + 12], ncig = rd16(&u[o + 16]), lseq = rd32(&u[o + 20]);
    if (rid < -1 || rid >= n_ref || nrid < -1 || nrid >= n_ref || pos < -1 || lname == 0) return 0;
    if (rid >= 0 && (uint64_t)pos > lens[rid]) return 0;
    const uint64_t fixed = 32ull + lname + 4ull * ncig + (lseq + 1ull) / 2 + lseq;
    if (fixed > bs) return 0;
    if (u[o + 36 + lname - 1] != 0) return 0;
    return o + 4 + bs;
}
// First offset in [from, to) from which a chain of 8 plausible records starts (or a shorter chain that ends exactly at N
// when `n_is_end`); SIZE_MAX if none.
size_t find_record_start(const uint8_t *u, size_t N, size_t from, size_t to, int32_t n_ref, const uint64_t *lens, bool n_is_end) {
    for (size_t o = from; o < to; o++) {
        size_t q = o; int chain = 0;
        while (chain < 8) { const size_t e = plausible_record(u, N, q, n_ref, lens); if (!e) break; q = e; chain++; if (q == N) break; }
        if (chain == 8 || (n_is_end && chain > 0 && q == N)) return o;
    }
    return (size_t)-1;
}

// Record boundaries in u[p, N), the first record starting at p.  Each record only says how long it is, so finding them is a
// pointer chase; done serially it is a chain of cache misses (~60 ns per record) and bounds the whole decode.  Parallel
// scheme: cut the buffer into segments, let every thread FIND a record start inside its segment (an offset from which a
// chain of 8 records is plausible), hop its own segment, and accept the result only if each segment's chain lands exactly
// on the start the next segment found — then it equals the serial hop by induction.  Any mismatch falls back to the plain
// serial hop, so the outcome never depends on the heuristic.  `end` receives the offset just past the last complete record
// (== N for a whole file; a streamed window may end inside a record).
template <typename PF>
bool find_records(const uint8_t *u, size_t p, size_t N, int32_t n_ref, const uint64_t *lens, int threads, PF &&pfor,
                  RecVec<size_t> &rec, size_t &end) {
    struct Seg { size_t start = 0, stop = 0; bool found = false; std::vector<size_t> rec; size_t landed = 0; };
    bool parallel_ok = false;
    const size_t body = N - p;
    int nseg = threads > 1 ? std::min<int>(threads * 4, (int)(body / (1 << 20))) : 0;
    rec.clear();
    if (nseg >= 2) {
        std::vector<Seg> seg(nseg);
        for (int k = 0; k < nseg; k++) seg[k].stop = p + body * (size_t)(k + 1) / nseg;
        seg[0].start = p; seg[0].found = true;
        pfor((size_t)nseg, [&](size_t k) {
            Seg &S = seg[k];
            if (k > 0) {
                const size_t o = find_record_start(u, N, seg[k - 1].stop, S.stop, n_ref, lens, true);
                if (o != (size_t)-1) { S.start = o; S.found = true; }
            }
        });
        // a segment without a start (one huge record spans it) is absorbed by its predecessor
        std::vector<int> live;
        for (int k = 0; k < nseg; k++) if (seg[k].found) live.push_back(k);
        pfor(live.size(), [&](size_t j) {
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
        for (size_t j = 0; j + 1 < live.size() && parallel_ok; j++)
            if (seg[live[j]].landed != seg[live[j + 1]].start) parallel_ok = false;
        if (parallel_ok) {
            std::vector<size_t> base(live.size() + 1, 0);
            for (size_t j = 0; j < live.size(); j++) base[j + 1] = base[j] + seg[live[j]].rec.size();
            rec.resize(base.back());
            pfor(live.size(), [&](size_t j) {
                const auto &v = seg[live[j]].rec;
                if (!v.empty()) memcpy(&rec[base[j]], v.data(), v.size() * sizeof(size_t));
            });
            end = seg[live.back()].landed;
        }
    }
    if (!parallel_ok) {   // serial hop (small inputs, one thread, or a failed speculation)
        rec.clear();
        rec.reserve(body / 200 + 16);
        size_t q = p;
        while (q + 4 <= N) {
            const uint32_t bs = rd32(&u[q]);
            if (bs < 32) return false;
            if (q + 4 + (size_t)bs > N) break;
            rec.push_back(q);
            q += 4 + (size_t)bs;
        }
        end = q;
    }
    return true;
}

// Destination of the per-record field extraction: the SoA batch of covermhip.h plus mate tid / read names.
struct SoaOut {
    int32_t *tid = nullptr, *pos = nullptr, *mtid = nullptr;
    uint16_t *flag = nullptr;
    uint8_t *mapq = nullptr, *nm_kind = nullptr;
    uint32_t *nm = nullptr, *l_seq = nullptr, *cigar_off = nullptr, *cigar = nullptr, *qname_off = nullptr;
    char *qnames = nullptr;
};

// Pass 1 of the extraction: cigar_off[0..R] (and qname_off) as prefix sums of the per-record counts.  `ensure` is called
// once with the totals so the caller can size cigar / qnames.
template <typename PF, typename EN>
bool record_offsets(const uint8_t *u, size_t N, const RecVec<size_t> &rec, int threads, PF &&pfor, uint32_t *cigar_off,
                    uint32_t *qname_off, EN &&ensure, std::string &err) {
    const size_t Rn = rec.size();
    cigar_off[0] = 0;
    if (qname_off) qname_off[0] = 0;
    const size_t nblk = std::max<size_t>(1, std::min<size_t>((size_t)threads * 4, Rn / 65536 + 1));
    std::vector<uint64_t> cs(nblk + 1, 0), qs(nblk + 1, 0);
    pfor(nblk, [&](size_t k) {
        const size_t lo = Rn * k / nblk, hi = Rn * (k + 1) / nblk;
        uint64_t c = 0, q = 0;
        for (size_t i = lo; i < hi; i++) {
            const uint8_t *r = &u[rec[i]];
            uint32_t nc = rd16(r + 16);
            if (maybe_long_cigar(r)) { uint32_t cnt = 0; if (real_cigar_from_cg(r, r + 4 + rd32(r), cnt)) nc = cnt; }
            c += nc; cigar_off[i + 1] = (uint32_t)c;
            if (qname_off) { q += r[12] ? r[12] - 1u : 0u; qname_off[i + 1] = (uint32_t)q; }
        }
        cs[k + 1] = c; qs[k + 1] = q;
    });
    for (size_t k = 0; k < nblk; k++) { cs[k + 1] += cs[k]; qs[k + 1] += qs[k]; }
    if (cs[nblk] > 0xffffffffull || qs[nblk] > 0xffffffffull) { err = "BAM too large for 32-bit CIGAR/name offsets"; return false; }
    pfor(nblk, [&](size_t k) {
        if (!k) return;
        const size_t lo = Rn * k / nblk, hi = Rn * (k + 1) / nblk;
        const uint32_t c = (uint32_t)cs[k], q = (uint32_t)qs[k];
        for (size_t i = lo; i < hi; i++) { cigar_off[i + 1] += c; if (qname_off) qname_off[i + 1] += q; }
    });
    (void)N;
    return ensure(cs[nblk], qs[nblk]);
}

// Pass 2: every field of record i (cigar_off / qname_off already final).
inline bool extract_record(const uint8_t *u, size_t at, size_t i, const SoaOut &o) {
    const uint8_t *r = &u[at];
    const uint32_t bs = rd32(r);
    const uint8_t *end = r + 4 + bs;
    o.tid[i] = (int32_t)rd32(r + 4); o.pos[i] = (int32_t)rd32(r + 8);
    const uint32_t l_read_name = r[12];
    o.mapq[i] = r[13];
    const uint32_t n_cig = rd16(r + 16);
    o.flag[i] = rd16(r + 18);
    const uint32_t l_seq = rd32(r + 20);
    o.l_seq[i] = l_seq;
    if (o.mtid) o.mtid[i] = (int32_t)rd32(r + 24);
    const uint8_t *q = r + 36;
    if (q + l_read_name + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq > end) return false;
    if (o.qnames && l_read_name) memcpy(&o.qnames[o.qname_off[i]], q, l_read_name - 1);
    q += l_read_name;
    const uint32_t want = o.cigar_off[i + 1] - o.cigar_off[i];
    const uint8_t *aux = q + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq;
    uint32_t nm = 0;
    if (want != n_cig) {       // the real CIGAR lives in CG:B,I (record_offsets already verified the conditions)
        const uint8_t *cg = nullptr; uint32_t cnt = 0;
        o.nm_kind[i] = scan_aux(aux, end, nm, &cg, &cnt);
        if (cg == nullptr || cnt != want) return false;
        memcpy(&o.cigar[o.cigar_off[i]], cg, 4ull * want);
    } else {
        if (n_cig) memcpy(&o.cigar[o.cigar_off[i]], q, 4ull * n_cig);
        o.nm_kind[i] = scan_aux(aux, end, nm, nullptr, nullptr);
    }
    o.nm[i] = nm;
    return true;
}

bool parse_bam(Bam &b, const Buf &u) {
    size_t p = 0;
    const int hrc = parse_bam_header(u.data(), u.size(), p, b.names, b.lens, b.header_text, b.err);
    if (hrc < 0) return false;
    if (hrc == 0) { b.err = u.size() < 12 ? "bad BAM magic" : "truncated BAM header"; return false; }
    const size_t N = u.size();
    auto pfor = [&](size_t n, auto fn) { parallel_for(n, b.threads, fn); };
    RecVec<size_t> rec;
    size_t end = p;
    if (!find_records(u.data(), p, N, (int32_t)b.names.size(), b.lens.data(), b.threads, pfor, rec, end) || end != N) {
        b.err = "truncated BAM record"; return false;
    }
    const size_t R = rec.size();
    b.cigar_off.resize(R + 1);
    if (b.want_names) b.qname_off.resize(R + 1);
    auto ensure = [&](uint64_t ncig, uint64_t nq) { b.cigar.resize(ncig); if (b.want_names) b.qnames.resize(nq); return true; };
    if (!record_offsets(u.data(), N, rec, b.threads, pfor, b.cigar_off.data(), b.want_names ? b.qname_off.data() : nullptr, ensure, b.err))
        return false;
    b.tid.resize(R); b.pos.resize(R); b.mtid.resize(R); b.flag.resize(R); b.mapq.resize(R); b.nm_kind.resize(R);
    b.nm.resize(R); b.l_seq.resize(R);
    SoaOut o;
    o.tid = b.tid.data(); o.pos = b.pos.data(); o.mtid = b.mtid.data(); o.flag = b.flag.data(); o.mapq = b.mapq.data();
    o.nm_kind = b.nm_kind.data(); o.nm = b.nm.data(); o.l_seq = b.l_seq.data(); o.cigar_off = b.cigar_off.data();
    o.cigar = b.cigar.data(); o.qname_off = b.want_names ? b.qname_off.data() : nullptr;
    o.qnames = b.want_names ? &b.qnames[0] : nullptr;
    std::atomic<bool> ok{true};
    parallel_for(R, b.threads, [&](size_t i) { if (!extract_record(u.data(), rec[i], i, o)) ok = false; });
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

// ---------------------------------------------------------------------------------------------- streamed reader
// covh_bam_stream: the same decode as covh_bam_open, but window by window with bounded memory, so that inflate, record
// parsing, the H2D copy of finished batches and (for several files) the GPU pipeline overlap inside ONE file:
//
//   coordinator I : pread ~32 MiB of compressed bytes -> block table -> inflate pool -> window queue
//   coordinator P : window (+ the bytes of the record cut off at the end of the previous window) -> record boundaries
//                   (speculative parallel hop, verified) -> field extraction into a page-locked SoA batch -> batch queue
//   consumer      : covh_bam_stream_next() hands out one batch at a time (cov_push_batch it, ask for the next)
//
// Three windows and three batches circulate; nothing else scales with the file.  A span (tid range) can be selected for
// multi-GPU runs: the rank's first BGZF block is located by probing the file (no index needed: find a block header,
// inflate, find a plausible record chain, read its tid), so every rank inflates only its own part of the file.
struct StreamBatch {
    void *mem[9] = {};
    size_t cap_rec = 0, cap_cig = 0;
    uint64_t n = 0, ncig = 0;
    static void *grab(size_t bytes) { void *p = cov_host_alloc(bytes); return p ? p : malloc(bytes); }
    static void drop(void *p) { if (p && !cov_host_free(p)) free(p); }
    bool ensure_rec(size_t R) {
        if (R <= cap_rec) return true;
        const size_t c = std::max(R, cap_rec + cap_rec / 2);
        static const size_t esz[8] = {4, 4, 2, 1, 4, 1, 4, 4};   // tid pos flag mapq nm nm_kind l_seq cigar_off
        for (int k = 0; k < 8; k++) { drop(mem[k]); mem[k] = grab((c + 1) * esz[k]); if (!mem[k]) return false; }
        cap_rec = c;
        return true;
    }
    bool ensure_cig(size_t C) {
        if (C <= cap_cig) return true;
        const size_t c = std::max(C, cap_cig + cap_cig / 2);
        drop(mem[8]); mem[8] = grab((c + 1) * 4);
        if (!mem[8]) return false;
        cap_cig = c;
        return true;
    }
    void fill(cov_batch *o) const {
        o->tid = (const int32_t *)mem[0]; o->pos = (const int32_t *)mem[1]; o->flag = (const uint16_t *)mem[2];
        o->mapq = (const uint8_t *)mem[3]; o->nm = (const uint32_t *)mem[4]; o->nm_kind = (const uint8_t *)mem[5];
        o->l_seq = (const uint32_t *)mem[6]; o->cigar_off = (const uint32_t *)mem[7]; o->cigar = (const uint32_t *)mem[8];
        o->n_records = n;
    }
    ~StreamBatch() { for (auto &m : mem) drop(m); }
};

struct StreamWindow {
    uint8_t *buf = nullptr; size_t cap = 0;
    size_t head = 0;   // bytes reserved in front of the inflated data for the previous window's cut-off record
    size_t len = 0;    // inflated bytes at buf + head
    bool last = false;
    bool reserve(size_t head_want, size_t len_want, bool keep) {   // keep: preserve the `len` bytes at buf + head
        if (head_want <= head && head + len_want <= cap) return true;
        const size_t nh = std::max(head, head_want), nc = nh + std::max(len_want, len) + (1u << 16);
        uint8_t *nb = (uint8_t *)malloc(nc);
        if (!nb) return false;
        if (keep && buf) memcpy(nb + nh, buf + head, len);
        free(buf);
        buf = nb; cap = nc; head = nh;
        return true;
    }
    ~StreamWindow() { free(buf); }
};

constexpr int64_t KEY_INF = 0x7fffffff;
inline int64_t tid_key(int32_t tid) { return tid < 0 ? KEY_INF : tid; }

struct Stream {
    std::string path, err;
    int fd = -1;
    uint64_t file_size = 0;
    std::vector<std::string> names; std::vector<uint64_t> lens; std::string header_text;
    int threads = 1;
    size_t window_comp = 32u << 20;
    // span
    uint64_t start_off = 0; bool mid_start = false;
    int64_t key_lo = 0, key_hi = KEY_INF + 1; bool empty_span = false;
    // machinery
    std::unique_ptr<Pool> pool_i, pool_p;
    std::thread th_i, th_p;
    std::mutex m; std::condition_variable cv;
    std::deque<StreamWindow *> win_free, win_full;
    std::deque<StreamBatch *> bat_free, bat_full;
    std::vector<std::unique_ptr<StreamWindow>> wins;
    std::vector<std::unique_ptr<StreamBatch>> bats;
    bool stop = false, i_done = false, p_done = false, failed = false;
    bool span_unsorted = false;   // the failure is "keys decrease inside a span" (covh_bam_stream_next returns -2)
    StreamBatch *held = nullptr;
    uint64_t n_records = 0;
    std::atomic<uint64_t> peak_bytes{0};
    double t_read = 0, t_inflate = 0, t_parse = 0, t_wait_i = 0, t_wait_p = 0;

    void fail(const std::string &e) { std::lock_guard<std::mutex> lk(m); if (!failed) { failed = true; err = e; } stop = true; cv.notify_all(); }
    void account() {
        uint64_t b = 0;
        for (auto &w : wins) b += w->cap;
        for (auto &x : bats) b += x->cap_rec * 24 + x->cap_cig * 4;
        b += window_comp + (1u << 20);
        uint64_t cur = peak_bytes.load();
        while (b > cur && !peak_bytes.compare_exchange_weak(cur, b)) {}
    }

    void run_inflate() {
        std::vector<uint8_t> cbuf(window_comp + (256u << 10));
        size_t left = 0; uint64_t fpos = start_off;
        std::vector<Blk> blocks;
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        for (;;) {
            StreamWindow *w = nullptr;
            {
                const double t0 = now();
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !win_free.empty(); });
                if (stop) break;
                w = win_free.front(); win_free.pop_front();
                t_wait_i += now() - t0;
            }
            const double t0 = now();
            size_t got = 0;
            while (got < window_comp) {
                const ssize_t r = pread(fd, cbuf.data() + left + got, window_comp - got, (off_t)(fpos + got));
                if (r < 0) { fail("read error on " + path); return; }
                if (r == 0) break;
                got += (size_t)r;
            }
            fpos += got;
            const bool eof = got < window_comp;
            const size_t n = left + got;
            blocks.clear();
            size_t p = 0, total = 0;
            std::string e;
            if (!bgzf_block_table(cbuf.data(), n, p, blocks, total, !eof, e)) { fail(e); return; }
            if (!eof && blocks.empty() && n > (128u << 10)) { fail("BGZF block larger than 64 KiB"); return; }
            const double t1 = now();
            w->len = 0;
            if (!w->reserve(1u << 20, total, false)) { fail("out of memory"); return; }
            std::atomic<bool> ok{true};
            uint8_t *dst = w->buf + w->head;
            const uint8_t *src = cbuf.data();
            pool_i->run(blocks.size(), [&](size_t i) { if (!inflate_block(src, blocks[i], dst + blocks[i].out_off)) ok = false; });
            if (!ok) { fail("BGZF inflate / CRC failure"); return; }
            w->len = total; w->last = eof;
            left = n - p;
            if (left) memmove(cbuf.data(), cbuf.data() + p, left);
            if (cbuf.size() < left + window_comp) cbuf.resize(left + window_comp);
            t_read += t1 - t0; t_inflate += now() - t1;
            account();
            { std::lock_guard<std::mutex> lk(m); win_full.push_back(w); }
            cv.notify_all();
            if (eof) break;
        }
        { std::lock_guard<std::mutex> lk(m); i_done = true; }
        cv.notify_all();
    }

    void run_parse() {
        std::vector<uint8_t> carry;
        bool header_done = mid_start, need_start = mid_start, finished = false;
        int64_t span_last_key = -1;
        RecVec<size_t> rec;
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        auto pfor = [&](size_t n, auto fn) { pool_p->run(n, fn); };
        const int32_t n_ref = (int32_t)names.size();
        while (!finished) {
            StreamWindow *w = nullptr;
            {
                const double t0 = now();
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || !win_full.empty() || i_done; });
                if (stop) break;
                if (win_full.empty()) break;   // inflate side ended without a `last` window: only after an error
                w = win_full.front(); win_full.pop_front();
                t_wait_p += now() - t0;
            }
            const double t0 = now();
            const size_t c = carry.size();
            if (c > w->head && !w->reserve(c + (c >> 2), w->len, true)) { fail("out of memory"); return; }
            uint8_t *base = w->buf + w->head - c;
            if (c) memcpy(base, carry.data(), c);
            const size_t N = c + w->len;
            const bool last = w->last;
            size_t p = 0, end = 0;
            bool have = true;
            if (!header_done) {
                std::vector<std::string> nm; std::vector<uint64_t> ln; std::string tx, e;
                const int rc = parse_bam_header(base, N, p, nm, ln, tx, e);
                if (rc < 0) { fail(e); return; }
                if (rc == 0) { if (last) { fail("truncated BAM header"); return; } have = false; end = 0; }
                else header_done = true;   // names/lens were read at open time
            }
            if (have && need_start) {   // span starting in the middle of the file: locate the first record of this window
                const size_t o = find_record_start(base, N, 0, N, n_ref, lens.data(), last);
                if (o == (size_t)-1) { have = false; end = last ? N : 0; if (!last && N > (64u << 20)) { fail("no record boundary found at the span start"); return; } }
                else { p = o; need_start = false; }
            }
            size_t i0 = 0, i1 = 0;
            if (have) {
                if (!find_records(base, p, N, n_ref, lens.data(), pool_p->size(), pfor, rec, end)) { fail("truncated BAM record"); return; }
                if (last && end != N) { fail("truncated BAM record"); return; }
                i1 = rec.size();
                if (key_lo > 0 || key_hi <= KEY_INF || mid_start) {
                    // A span trusts the file's order when it drops its neighbours' records, so it checks that order over everything
                    // it parses (the spans' byte ranges overlap: together they cover the file).  contig.rs:129-132 stops at the first
                    // record whose tid is lower than its predecessor's; the whole-file path reports that from cov_finish.
                    int64_t prev = span_last_key;
                    for (size_t i = 0; i < i1; i++) {
                        const int64_t k = tid_key((int32_t)rd32(base + rec[i] + 4));
                        if (k < prev) {
                            { std::lock_guard<std::mutex> lk(m); if (!failed) span_unsorted = true; }
                            fail("BAM file appears to be unsorted. Input BAM files must be sorted by reference (i.e. by samtools sort)"); return;
                        }
                        prev = k;
                    }
                    span_last_key = prev;
                }
                // span: keep records with key_lo <= key(tid) < key_hi (file order; ranges are contiguous in a sorted file)
                if (key_lo > 0) while (i0 < i1 && tid_key((int32_t)rd32(base + rec[i0] + 4)) < key_lo) i0++;
                if (key_hi <= KEY_INF) {
                    size_t j = i0;
                    while (j < i1 && tid_key((int32_t)rd32(base + rec[j] + 4)) < key_hi) j++;
                    if (j < i1) { i1 = j; finished = true; }
                }
            }
            StreamBatch *b = nullptr;
            if (i1 > i0) {
                {
                    const double tw = now();
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return stop || !bat_free.empty(); });
                    if (stop) break;
                    b = bat_free.front(); bat_free.pop_front();
                    t_wait_p += now() - tw;
                }
                const size_t R = i1 - i0;
                if (!b->ensure_rec(R)) { fail("out of memory"); return; }
                RecVec<size_t> sub;   // record_offsets / extraction work on the kept sub-range
                const RecVec<size_t> *rv = &rec;
                if (i0 != 0 || i1 != rec.size()) { sub.assign(rec.begin() + i0, rec.begin() + i1); rv = &sub; }
                std::string e;
                auto ensure = [&](uint64_t ncig, uint64_t) { b->ncig = ncig; return b->ensure_cig((size_t)ncig); };
                if (!record_offsets(base, N, *rv, pool_p->size(), pfor, (uint32_t *)b->mem[7], nullptr, ensure, e)) { fail(e.empty() ? "out of memory" : e); return; }
                SoaOut o;
                o.tid = (int32_t *)b->mem[0]; o.pos = (int32_t *)b->mem[1]; o.flag = (uint16_t *)b->mem[2]; o.mapq = (uint8_t *)b->mem[3];
                o.nm = (uint32_t *)b->mem[4]; o.nm_kind = (uint8_t *)b->mem[5]; o.l_seq = (uint32_t *)b->mem[6];
                o.cigar_off = (uint32_t *)b->mem[7]; o.cigar = (uint32_t *)b->mem[8];
                std::atomic<bool> ok{true};
                const RecVec<size_t> &rr = *rv;
                pool_p->run(R, [&](size_t i) { if (!extract_record(base, rr[i], i, o)) ok = false; }, 256);
                if (!ok) { fail("corrupt BAM record"); return; }
                b->n = R;
            }
            carry.assign(base + end, base + N);
            if (last) finished = true;
            t_parse += now() - t0;
            account();
            {
                std::lock_guard<std::mutex> lk(m);
                win_free.push_back(w);
                if (b) bat_full.push_back(b);
                if (finished) { p_done = true; stop_inflate_locked(); }
            }
            cv.notify_all();
        }
        { std::lock_guard<std::mutex> lk(m); p_done = true; }
        cv.notify_all();
    }
    void stop_inflate_locked() { if (!i_done) stop = true; }

    ~Stream() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        if (th_i.joinable()) th_i.join();
        if (th_p.joinable()) th_p.join();
        if (fd >= 0) close(fd);
    }
};

// Inflates blocks from `off` until `want(buffer)` says it has enough; used for the header pre-read and the span probes.
template <typename W>
bool inflate_from(int fd, uint64_t file_size, uint64_t off, std::vector<uint8_t> &out, W &&want, std::string &err) {
    std::vector<uint8_t> c;
    uint64_t fpos = off;
    size_t left = 0;
    out.clear();
    for (;;) {
        const size_t chunk = 1u << 20;
        c.resize(left + chunk);
        const ssize_t r = fpos < file_size ? pread(fd, c.data() + left, chunk, (off_t)fpos) : 0;
        if (r < 0) { err = "read error"; return false; }
        fpos += (uint64_t)r;
        const size_t n = left + (size_t)r;
        const bool eof = r == 0 || fpos >= file_size;
        std::vector<Blk> blocks; size_t p = 0, total = 0;
        if (!bgzf_block_table(c.data(), n, p, blocks, total, !eof, err)) return false;
        const size_t base = out.size();
        out.resize(base + total);
        if (blocks.size() >= 32) {      // a long header (an assembly's reference dictionary is tens of MB): the blocks of a chunk on a few threads
            const size_t T = std::min<size_t>({(size_t)8, (size_t)std::max(1u, std::thread::hardware_concurrency()), blocks.size() / 8});
            std::atomic<size_t> next{0};
            std::atomic<bool> ok{true};
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; t++)
                th.emplace_back([&] {
                    for (size_t i; ok.load(std::memory_order_relaxed) && (i = next.fetch_add(1)) < blocks.size();)
                        if (!inflate_block(c.data(), blocks[i], out.data() + base + blocks[i].out_off)) ok = false;
                });
            for (auto &x : th) x.join();
            if (!ok) { err = "BGZF inflate / CRC failure"; return false; }
        } else
        for (auto &b : blocks) if (!inflate_block(c.data(), b, out.data() + base + b.out_off)) { err = "BGZF inflate / CRC failure"; return false; }
        if (want(out)) return true;
        if (eof) return true;   // caller decides whether what it got is enough
        left = n - p;
        memmove(c.data(), c.data() + p, left);
    }
}

// First BGZF block at or after file offset `off` (header magic + BC subfield + a second block header right behind it),
// SIZE_MAX if none before the end of the file.
uint64_t find_block_start(int fd, uint64_t file_size, uint64_t off) {
    std::vector<uint8_t> c(192u << 10);
    while (off < file_size) {
        const ssize_t r = pread(fd, c.data(), c.size(), (off_t)off);
        if (r < 28) return (uint64_t)-1;
        const size_t n = (size_t)r;
        for (size_t q = 0; q + 18 <= n; q++) {
            if (c[q] != 0x1f || c[q + 1] != 0x8b || c[q + 2] != 8 || c[q + 3] != 4) continue;
            if (rd16(&c[q + 10]) != 6 || c[q + 12] != 66 || c[q + 13] != 67 || rd16(&c[q + 14]) != 2) continue;
            const size_t bsize = (size_t)rd16(&c[q + 16]) + 1;
            if (bsize < 26) continue;
            if (off + q + bsize == file_size) return off + q;
            if (q + bsize + 4 <= n && c[q + bsize] == 0x1f && c[q + bsize + 1] == 0x8b && c[q + bsize + 2] == 8 && c[q + bsize + 3] == 4) return off + q;
        }
        off += n - 28;
    }
    return (uint64_t)-1;
}

// Key (tid, or KEY_INF for unmapped-without-reference) of the first record that can be located at or after file offset
// `off`; *block_off receives the BGZF block the search started in.  KEY_INF + 1 when the file ends first; PROBE_FAILED when
// nothing could be decided (a header signature inside compressed data that does not inflate, even after moving on a few times;
// no record boundary within 96 MB) — callers must not read that as "beyond every key".
constexpr int64_t PROBE_FAILED = -2;
int64_t probe_key(int fd, uint64_t file_size, uint64_t off, int32_t n_ref, const uint64_t *lens, uint64_t *block_off) {
    for (int attempt = 0; attempt < 8; attempt++) {
        const uint64_t b = find_block_start(fd, file_size, off);
        *block_off = b;
        if (b == (uint64_t)-1) return KEY_INF + 1;
        std::vector<uint8_t> u; std::string err;
        size_t found = (size_t)-1;
        bool gave_up = false;
        auto want = [&](std::vector<uint8_t> &buf) {
            found = find_record_start(buf.data(), buf.size(), 0, buf.size(), n_ref, lens, false);
            gave_up = found == (size_t)-1 && buf.size() > (96u << 20);
            return found != (size_t)-1 || gave_up;
        };
        if (!inflate_from(fd, file_size, b, u, want, err)) { off = b + 1; continue; }     // a false signature: the next candidate
        if (found == (size_t)-1 && !gave_up) found = find_record_start(u.data(), u.size(), 0, u.size(), n_ref, lens, true);   // read to the end of the file
        if (found == (size_t)-1) return gave_up ? PROBE_FAILED : KEY_INF + 1;
        return tid_key((int32_t)rd32(&u[found + 4]));
    }
    return PROBE_FAILED;
}

// Span boundaries of a file cut into `span_count` tid spans: B[k] = one past the key found at k / count of the file, boff[k] the
// BGZF block that probe started in.  Every rank computes the same.  A probe that cannot be decided makes the whole file ONE
// span (span 0 takes everything, the others are empty): slower, never wrong.
bool span_boundaries(int fd, uint64_t file_size, uint32_t span_count, int32_t n_ref, const uint64_t *lens, std::vector<int64_t> &B,
                     std::vector<uint64_t> &boff) {
    B.assign(span_count + 1, 0); boff.assign(span_count + 1, 0);
    bool ok = true;
    for (uint32_t k = 1; k < span_count && ok; k++) {
        uint64_t bo = 0;
        const int64_t key = probe_key(fd, file_size, file_size / span_count * k, n_ref, lens, &bo);
        if (key == PROBE_FAILED) { ok = false; break; }
        B[k] = std::max(B[k - 1], key >= KEY_INF ? KEY_INF : key + 1);
        boff[k] = bo;
    }
    B[span_count] = KEY_INF + 1;
    if (!ok) for (uint32_t k = 1; k < span_count; k++) { B[k] = KEY_INF + 1; boff[k] = (uint64_t)-1; }
    return ok;
}

}  // namespace

extern "C" {

struct covh_bam_stream { Stream s; };

covh_bam_stream *covh_bam_stream_open(const char *path, int threads, uint32_t span_index, uint32_t span_count, char *err, size_t errcap) {
    auto bail = [&](covh_bam_stream *h, const std::string &e) -> covh_bam_stream * {
        if (err && errcap) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; }
        delete h;
        return nullptr;
    };
    covh_bam_stream *h = new covh_bam_stream();
    Stream &s = h->s;
    s.path = path; s.threads = std::max(1, threads);
    s.fd = open(path, O_RDONLY);
    if (s.fd < 0) return bail(h, std::string("Unable to find BAM file ") + path);
    {
        const off_t e = lseek(s.fd, 0, SEEK_END);
        s.file_size = e > 0 ? (uint64_t)e : 0;
    }
    if (s.file_size == 0) return bail(h, std::string(path) + ": empty file (no BAM/SAM header)");
    uint8_t magic[2] = {0, 0};
    if (pread(s.fd, magic, 2, 0) != 2 || magic[0] != 0x1f || magic[1] != 0x8b) return bail(h, "not a BGZF file (the streamed reader takes BAM only)");
    { long long v; if (covknob::get("stream_window_kb", v) && v >= 64) s.window_comp = (size_t)v << 10; }
    // header: inflate from the start until the reference dictionary is complete
    {
        std::vector<uint8_t> u; std::string e;
        size_t p = 0; int rc = 0;
        auto want = [&](std::vector<uint8_t> &buf) {
            std::string e2;
            rc = parse_bam_header(buf.data(), buf.size(), p, s.names, s.lens, s.header_text, e2);
            if (rc < 0) e = e2;
            return rc != 0;
        };
        if (!inflate_from(s.fd, s.file_size, 0, u, want, e)) return bail(h, e);
        if (rc < 0) return bail(h, e);
        if (rc == 0) return bail(h, u.size() < 12 ? "bad BAM magic" : "truncated BAM header");
    }
    if (span_count > 1) {
        if (span_index >= span_count) return bail(h, "span index out of range");
        const int32_t n_ref = (int32_t)s.names.size();
        // boundaries B[k], k = 1 .. count-1: one past the tid found at k/count of the file; every rank computes the same
        std::vector<int64_t> B; std::vector<uint64_t> boff;
        span_boundaries(s.fd, s.file_size, span_count, n_ref, s.lens.data(), B, boff);
        s.key_lo = B[span_index]; s.key_hi = B[span_index + 1];
        if (span_index > 0) {
            if (boff[span_index] == (uint64_t)-1 || s.key_lo >= s.key_hi) s.empty_span = true;
            else { s.start_off = boff[span_index]; s.mid_start = true; }
        } else if (s.key_hi <= 0) s.empty_span = true;
    }
    if (s.empty_span) { s.p_done = true; s.i_done = true; return h; }
    const int ti = s.threads, tp = std::max(1, s.threads / 2);
    s.pool_i.reset(new Pool(ti)); s.pool_p.reset(new Pool(tp));
    for (int k = 0; k < 3; k++) {
        s.wins.emplace_back(new StreamWindow()); s.win_free.push_back(s.wins.back().get());
        s.bats.emplace_back(new StreamBatch()); s.bat_free.push_back(s.bats.back().get());
    }
    s.th_i = std::thread([h] { h->s.run_inflate(); });
    s.th_p = std::thread([h] { h->s.run_parse(); });
    return h;
}

uint32_t covh_bam_stream_n_targets(const covh_bam_stream *h) { return (uint32_t)h->s.names.size(); }
const char *covh_bam_stream_target_name(const covh_bam_stream *h, uint32_t i) { return h->s.names[i].c_str(); }
uint64_t covh_bam_stream_target_len(const covh_bam_stream *h, uint32_t i) { return h->s.lens[i]; }
const char *covh_bam_stream_error(const covh_bam_stream *h) { return h->s.err.c_str(); }
const char *covh_inflate_backend(void) { return libdeflate().ok ? "libdeflate" : "zlib"; }
uint64_t covh_bam_stream_peak_bytes(const covh_bam_stream *h) { return h->s.peak_bytes.load(); }
uint64_t covh_bam_stream_n_records(const covh_bam_stream *h) { return h->s.n_records; }

int covh_bam_stream_next(covh_bam_stream *h, cov_batch *out) {
    Stream &s = h->s;
    std::unique_lock<std::mutex> lk(s.m);
    if (s.held) { s.bat_free.push_back(s.held); s.held = nullptr; s.cv.notify_all(); }
    s.cv.wait(lk, [&] { return !s.bat_full.empty() || s.p_done || s.failed; });
    if (s.failed) return s.span_unsorted ? -2 : -1;
    if (s.bat_full.empty()) return 0;
    s.held = s.bat_full.front(); s.bat_full.pop_front();
    s.held->fill(out);
    s.n_records += s.held->n;
    return 1;
}

void covh_bam_stream_timing(const covh_bam_stream *h, double *out5) {
    const Stream &s = h->s;
    out5[0] = s.t_read; out5[1] = s.t_inflate; out5[2] = s.t_parse; out5[3] = s.t_wait_i; out5[4] = s.t_wait_p;
}

void covh_bam_stream_close(covh_bam_stream *h) { delete h; }

}  // extern "C"

namespace {
}  // namespace

// ---------------------------------------------------------------------------------------------- device ingest driver
// Host side of cov_ingest_*: reads the file into two alternating page-locked buffers (parallel pread), hops the BGZF block
// headers of every piece (18 bytes per block; a header or a block may straddle two pieces) and feeds bytes + completed
// blocks to the session; the GPU inflates, checks CRCs, finds the records and fills its own record store.
namespace {
struct HeaderOnly {
    std::vector<std::string> names; std::vector<uint64_t> lens; std::string text;
    uint64_t first_record = 0, file_size = 0;
};
}  // namespace

extern "C" {

struct covh_bam_header { HeaderOnly h; };

covh_bam_header *covh_bam_read_header(const char *path, char *err, size_t errcap) {
    auto bail = [&](covh_bam_header *h, int fd, const std::string &e) -> covh_bam_header * {
        if (err && errcap) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; }
        if (fd >= 0) close(fd);
        delete h;
        return nullptr;
    };
    covh_bam_header *h = new covh_bam_header();
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return bail(h, -1, std::string("Unable to find BAM file ") + path);
    const off_t e = lseek(fd, 0, SEEK_END);
    h->h.file_size = e > 0 ? (uint64_t)e : 0;
    if (h->h.file_size == 0) return bail(h, fd, std::string(path) + ": empty file (no BAM/SAM header)");
    uint8_t magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) != 2 || magic[0] != 0x1f || magic[1] != 0x8b) return bail(h, fd, "not a BGZF file");
    std::vector<uint8_t> u; std::string msg;
    size_t p = 0; int rc = 0;
    auto want = [&](std::vector<uint8_t> &buf) {
        std::string e2;
        rc = parse_bam_header(buf.data(), buf.size(), p, h->h.names, h->h.lens, h->h.text, e2);
        if (rc < 0) msg = e2;
        return rc != 0;
    };
    if (!inflate_from(fd, h->h.file_size, 0, u, want, msg)) return bail(h, fd, msg);
    if (rc < 0) return bail(h, fd, msg);
    if (rc == 0) return bail(h, fd, u.size() < 12 ? "bad BAM magic" : "truncated BAM header");
    h->h.first_record = p;
    close(fd);
    return h;
}
void covh_bam_header_free(covh_bam_header *h) { delete h; }
uint32_t covh_bam_header_n_targets(const covh_bam_header *h) { return (uint32_t)h->h.names.size(); }
const char *covh_bam_header_target_name(const covh_bam_header *h, uint32_t i) { return h->h.names[i].c_str(); }
uint64_t covh_bam_header_target_len(const covh_bam_header *h, uint32_t i) { return h->h.lens[i]; }
uint64_t covh_bam_header_first_record(const covh_bam_header *h) { return h->h.first_record; }

// Bytes into a page-locked staging slot with non-temporal stores: the destination's lines are not read for ownership and do not stay in
// the caches (the next reader of them is the DMA engine).  dst 16-byte aligned (a slot is page-aligned, chunks are multiples of 512 KiB);
// src anywhere (a tid span starts at a block, not at a page).
static void stream_copy(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t o = 0;
    if (((uintptr_t)dst & 15u) == 0) {
        for (; o + 64 <= n; o += 64) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(src + o)), b = _mm_loadu_si128((const __m128i *)(src + o + 16));
            const __m128i c = _mm_loadu_si128((const __m128i *)(src + o + 32)), d = _mm_loadu_si128((const __m128i *)(src + o + 48));
            _mm_stream_si128((__m128i *)(dst + o), a); _mm_stream_si128((__m128i *)(dst + o + 16), b);
            _mm_stream_si128((__m128i *)(dst + o + 32), c); _mm_stream_si128((__m128i *)(dst + o + 48), d);
        }
    }
    if (o < n) memcpy(dst + o, src + o, n - o);
    _mm_sfence();      // before the piece is announced: the upload that reads the slot is enqueued by another thread
}

// 0 = records are in the session's store; 1 = the file needs the CPU reader (reason in err; nothing appended); -1 = error;
// -2 = the record keys decrease inside the span (the reference's "appears to be unsorted" text in err).
// timing (optional, 8 doubles): seconds reading the file, waiting for staging slots, in cov_ingest_end, total, until the first piece's upload call
// (cov_ingest_begin's streams, events and buffers beside the first reads), in the header walk, in the feed calls, [7] see below.
int covh_bam_gpu_ingest(const char *path, int threads, cov_session *s, const covh_bam_header *hd, int check_crc, uint64_t *n_records,
                        double *timing, char *err, size_t errcap) {
    return covh_bam_gpu_ingest_span(path, threads, s, hd, check_crc, 0, 1, n_records, timing, err, errcap);
}

// One tid span of the file through the device ingest (same span definition as covh_bam_stream_open: boundaries one past the
// tid found at k / count of the file).  The bytes fed run from the BGZF block at the span's probe point to a margin behind the
// first block that begins with a record of a later span (found by bisection over file offsets: the keys are sorted); the
// device drops the records of the neighbouring spans at both ends.
int covh_bam_gpu_ingest_span(const char *path, int threads, cov_session *s, const covh_bam_header *hd, int check_crc, uint32_t span_index,
                             uint32_t span_count, uint64_t *n_records, double *timing, char *err, size_t errcap) {
    auto fail = [&](int rc, const std::string &e) { if (err && errcap) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; } return rc; };
    covr::Range rr("device ingest of one file span (covh_bam_gpu_ingest_span)");
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    if (n_records) *n_records = 0;
    if (span_count == 0 || span_index >= span_count) return fail(-1, "span index out of range");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(-1, std::string("Unable to find BAM file ") + path);
    struct FdClose { int fd; ~FdClose() { close(fd); } } fdc{fd};
    const uint64_t file_size = hd->h.file_size;
    uint64_t f_lo = 0, size = file_size;          // bytes [f_lo, size) are fed
    int64_t key_lo = 0, key_hi = KEY_INF + 1;
    uint64_t first_record = hd->h.first_record;
    if (span_count > 1) {
        const int32_t n_ref = (int32_t)hd->h.names.size();
        std::vector<int64_t> B; std::vector<uint64_t> boff;
        span_boundaries(fd, file_size, span_count, n_ref, hd->h.lens.data(), B, boff);
        key_lo = B[span_index]; key_hi = B[span_index + 1];
        if (span_index > 0) {
            if (boff[span_index] == (uint64_t)-1 || key_lo >= key_hi) return 0;          // empty span: nothing to read
            f_lo = boff[span_index]; first_record = 0;
        } else if (key_hi <= 0) return 0;
        if (span_index + 1 < span_count && boff[span_index + 1] != (uint64_t)-1) {
            // smallest offset from which the first record found has a key >= key_hi
            uint64_t lo = boff[span_index + 1], hi = file_size;
            bool decided = true;
            while (hi - lo > 65536 && decided) {
                const uint64_t mid = lo + (hi - lo) / 2;
                uint64_t bo = 0;
                const int64_t key = probe_key(fd, file_size, mid, n_ref, hd->h.lens.data(), &bo);
                if (key == PROBE_FAILED) decided = false;        // not "beyond the span": feed up to the end of the file instead
                else if (key >= key_hi) hi = mid; else lo = mid;
            }
            // every record of this span starts before that point and may run on for up to the carry size of the device ingest
            const uint64_t want = hi + ((uint64_t)17 << 20);
            if (decided && want < file_size) { const uint64_t e = find_block_start(fd, file_size, want); if (e != (uint64_t)-1) size = e; }
        }
    }
    const bool mid_start = f_lo != 0, open_end = size != file_size;
    // Where the DMA engine takes the compressed bytes from (COVERM_INGEST_IO = pread | mmap):
    //   pread (default) staging slots in page-locked memory filled by threaded preads;
    //   mmap  the file is mapped and its pages are registered with the device piece by piece, ahead of the uploads, so that the
    //         bytes go from the page cache to HBM with no copy by the CPU.  Alone on an idle device this reaches the link's rate
    //         (tools/ubench/io_probe: 57 GB/s against 42-54 GB/s through staging slots), but inside the running pipeline
    //         hipHostRegister drops to ~20 GB/s, the copies take longer to enqueue and unregistering 20 GB at the end costs
    //         another 0.4 s (profiles/r03_io_modes.log: 200 M reads 1.78 s against 0.98 s): kept as an option, not the default.
    //   More than two devices fed at once (covh_bam_set_concurrent_feeders; coverm-amd --devices): until round 6 the mapping, registered up
    //   front, was the default there, on a model (a staged byte crosses the host's memory three times, a mapped one once).  Measured with eight
    //   feeders on the one-GPU box (profiles/r06_eight_feeders_io.json): the eight registrations do NOT run beside one another — 0.2 to 1.2 s
    //   per 2.6 GB span, 20.6 GB in ~1.2 s all told = 17 GB/s for the process —, which alone is longer than the whole single-device run
    //   (0.76-0.96 s), while eight staged readers under the same 16-CPU quota copy at the one reader's rate (~49 GB/s in aggregate).  The
    //   staging slots are therefore the default for any number of feeders; COVERM_INGEST_IO=mmap-upfront | mmap remain as options.
    //   How the staging slots are FILLED (round 6, tools/ubench/copy_probe, profiles/r06_copy_probe.log): the feed was bound by the host's
    //   memory system — 14 threads pread at 80-88 GB/s alone, but beside them the DMA out of the slots falls from 58 to 37-52 GB/s (pread's
    //   copy_to_user reads the source, reads the destination lines for ownership and writes them; the DMA reads them once more), and the
    //   pipeline settles where both run at ~50 GB/s.  Copying from a MAPPING of the file with non-temporal stores (no read for ownership,
    //   nothing of the destination left in the caches) runs at 99-102 GB/s and leaves the DMA its 58-60 GB/s; zapping every chunk's pages
    //   behind the copy (74-83 GB/s; a 20 GB mapping's page tables would otherwise cost ~0.4 s when the process ends) still does.  That is
    //   the default; COVERM_INGEST_IO=pread keeps the preads (and is what a refused mapping falls back to).
    const char *io = getenv("COVERM_INGEST_IO");
    bool use_map = io && (!strcmp(io, "mmap") || !strcmp(io, "mmap-upfront"));
    const bool map_upfront = use_map && !strcmp(io, "mmap-upfront");
    bool copy_map = !use_map && !(io && !strcmp(io, "pread"));
    uint8_t *map = nullptr;
    const uint64_t PG = 4096, map_len = (file_size + PG - 1) / PG * PG;
    if (use_map || copy_map) {
        void *m = mmap(nullptr, (size_t)file_size, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) use_map = copy_map = false; else map = (uint8_t *)m;
    }
    constexpr int NS = COV_INGEST_SLOTS;
    uint8_t *buf[NS];
    for (int k = 0; k < NS; k++) buf[k] = nullptr;       // page-locked on first use by the reader thread: slots 1.. are pinned while piece 0 is already on its way
    // Every way out of this function after cov_ingest_begin first gives up whatever is still queued on the device (no-op once
    // cov_ingest_end has run), and only then parks the staging buffers / unregisters the mapping: an upload may still be reading
    // them, and an extraction still writing the store the CPU reader is about to push into.
    struct BufFree {
        cov_session *s; uint8_t **b; uint8_t *&map; uint64_t file_size; std::vector<std::pair<uint8_t *, uint64_t>> regs; std::mutex m;
        ~BufFree() {
            (void)cov_ingest_abort(s);
            for (int k = 0; k < NS; k++) if (b[k]) cov_host_free(b[k]);
            for (auto &r : regs) (void)cov_host_unregister(s, r.first);
            if (map) munmap(map, (size_t)file_size);
        }
    } bf{s, buf, map, file_size, {}, {}};
    // cov_ingest_begin (four streams, a dozen events, two page-locked tables: ~25 ms) runs BESIDE the reader's first pieces (their
    // page-locked slots cost ~5 ms each to obtain): the coordinator waits for it in front of the first feed.
    std::future<int> begun = std::async(std::launch::async, [&]() -> int {
        if (cov_ingest_begin(s, file_size, first_record, check_crc) != COV_OK) return 1;
        if (span_count > 1 && cov_ingest_span(s, key_lo, key_hi, mid_start ? 1 : 0, open_end ? 1 : 0, f_lo, size) != COV_OK) return 1;
        return 0;
    });
    struct BegunWait { std::future<int> &f; ~BegunWait() { if (f.valid()) (void)f.get(); } } begun_wait{begun};     // (destroyed before bf: abort sees a finished begin)
    if (use_map && begun.get() != 0) return fail(-1, cov_last_error(s));      // registering the mapping needs the session's device set up
    // staging slots: page-locked memory costs ~0.17 s per GiB to obtain and ~0.13 s per GiB to give back when the process ends
    // (tools/ubench/exit_probe), so the slots are as small as the reader's rate allows: 4 x 32 MiB in 2 MiB chunks read 5 GB in
    // 0.098 s inside the pipeline, 4 x 64 MiB in 4 MiB chunks in 0.114 s (profiles/r03_reader_sweep_50M.log)
    size_t piece = use_map ? (size_t)256 << 20 : (size_t)32 << 20;
    long long piece_kb = 0;
    const bool piece_env = covknob::get("ingest_piece_kb", piece_kb);      // tests: many small pieces
    if (piece_env && piece_kb >= 64) piece = (size_t)piece_kb << 10;
    // registered so far: [reg_lo0, reg_hi) of the mapping, in whole pages; a piece registers what of its pages is not registered yet
    uint64_t reg_hi = f_lo / PG * PG;
    auto register_piece = [&](uint64_t off, uint64_t n) -> bool {
        const uint64_t hi = std::min<uint64_t>(map_len, (off + n + PG - 1) / PG * PG);
        if (hi <= reg_hi) return true;
        if (cov_host_register(s, map + reg_hi, (size_t)(hi - reg_hi)) != COV_OK) return false;
        { std::lock_guard<std::mutex> lk(bf.m); bf.regs.emplace_back(map + reg_hi, hi - reg_hi); }
        reg_hi = hi;
        return true;
    };
    if (use_map && !register_piece(f_lo, map_upfront ? size - f_lo : std::min<uint64_t>(piece, size - f_lo))) {     // refused: staging slots instead
        use_map = false; copy_map = true;      // (the mapping itself is there: its bytes are copied into the slots)
        if (!piece_env) piece = (size_t)32 << 20;
    }
    const double t_begin = now() - t_start;
    std::vector<cov_bgzf_block> blocks;
    uint64_t next_blk = f_lo, out_off = 0, pending_bsize = 0;     // absolute file offset of the next block header; running inflated size; BSIZE of a block whose header is read but whose end is not here yet
    uint8_t tail[64]; uint64_t tail_end = 0; size_t tail_len = 0;   // last bytes of the previous piece (a header may straddle)
    double t_read = 0, t_wait = 0, t_walk = 0, t_feed = 0, t_first_feed = 0;
    const uint64_t n_pieces = (size - f_lo + piece - 1) / piece;
    // Hopping the block headers is a chain of dependent cache misses (~0.3 us per block, 0.28 s for the 944 k blocks of a 200 M-read
    // file) if one thread does it after the fact.  The pool thread that has just read a 4 MiB chunk hops the blocks that lie
    // entirely inside it while the bytes are still in its cache (first header found by its 16-byte signature); the coordinator
    // takes over a chunk's list when the chain arrives exactly at the list's first header, and hops by itself otherwise (the blocks
    // that straddle chunks, or a chunk whose first signature was a coincidence inside compressed data).
    // chunks well below a piece / threads: a piece ends in a barrier, and with one chunk per thread the slowest thread sets its
    // pace (64 MiB pieces: 4 MiB chunks 0.114 s per 5 GB, 1 MiB chunks 0.091 s; the coordinator hops the blocks that straddle chunks)
    size_t chunk = 512u << 10;
    struct PreBlock { uint64_t hdr; uint32_t bsize, crc, isize; };
    struct PreChunk { uint64_t first = ~0ull, next = 0; std::vector<PreBlock> blocks; };
    const size_t chunks_per_piece = (piece + chunk - 1) / chunk;
    std::vector<PreChunk> pre((size_t)NS * chunks_per_piece);
    auto prewalk = [](const uint8_t *p, size_t n, uint64_t abs, PreChunk &out) {
        out.first = ~0ull; out.next = 0; out.blocks.clear();
        auto is_hdr = [&](size_t q) {
            return q + 18 <= n && p[q] == 0x1f && p[q + 1] == 0x8b && p[q + 2] == 8 && p[q + 3] == 4 && p[q + 10] == 6 && p[q + 11] == 0 && p[q + 12] == 66 &&
                   p[q + 13] == 67 && p[q + 14] == 2 && p[q + 15] == 0;
        };
        size_t q = 0;
        for (;;) {     // first header signature in the chunk
            const void *f = q < n ? memchr(p + q, 0x1f, n - q) : nullptr;
            if (!f) return;
            q = (size_t)((const uint8_t *)f - p);
            if (is_hdr(q)) break;
            q++;
        }
        out.first = abs + q;
        while (is_hdr(q)) {
            const size_t bsize = (size_t)(p[q + 16] | (p[q + 17] << 8)) + 1;
            if (bsize < 26 || q + bsize > n) break;                  // ends in a later chunk: the coordinator's business
            PreBlock b; b.hdr = abs + q; b.bsize = (uint32_t)bsize;
            memcpy(&b.crc, p + q + bsize - 8, 4); memcpy(&b.isize, p + q + bsize - 4, 4);
            if (b.isize > 65536u) break;
            out.blocks.push_back(b);
            q += bsize;
        }
        out.next = abs + q;
    };
    // The reader thread fills staging slot k % NS with piece k (threaded preads) as soon as the upload of piece k - NS has left
    // the slot; this thread walks the block headers of the pieces in order and feeds them: file reading, the serial header walk
    // and the device never wait for one another in turn.
    std::mutex mu; std::condition_variable cv;
    uint64_t ready = 0, fed = 0;        // pieces read so far / pieces handed to cov_ingest_feed so far
    bool reader_failed = false, reader_soft = false, stop = false;      // soft: the file can still go to the CPU reader
    std::string reader_err;
    // The mapping's pages leave the page table again behind the copy (a 20 GB mapping left to the end of the process costs ~0.4 s there, and
    // counts as resident meanwhile).  HOW matters: a madvise per 512 KiB chunk interrupts every thread of the process 41 000 times per 20 GB
    // (a TLB shootdown each) and tripled the time the coordinator spends in the feed calls; the piece in front, in 4 MiB parts handed to the
    // same pool, is an eighth of the calls (COVERM_KNOBS ingest_zap: 1 = per chunk, 2 = the piece in front (default), 0 = all at the end).
    long long zap_mode = 2;
    (void)covknob::get("ingest_zap", zap_mode);
    const uint64_t ZPART = 4ull << 20;
    auto zap_range = [&](uint64_t a, uint64_t b) {
        const uint64_t z0 = (a + PG - 1) / PG * PG, z1 = b / PG * PG;
        if (z1 > z0) (void)madvise(map + z0, (size_t)(z1 - z0), MADV_DONTNEED);
    };
    std::thread reader([&]() {
        Pool pool(std::max(1, threads > 6 ? threads - 2 : threads));     // this thread's caller and the coordinator need CPUs too (12 threads read no slower than 16 under a 16-CPU quota)
        for (uint64_t k = 0; k < n_pieces; k++) {
            const int slot = (int)(k % NS);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || k < NS || fed + NS > k; });      // piece k - NS has been fed: its upload is on the copy stream
                if (stop) return;
            }
            double t0 = now();
            const uint64_t off = f_lo + k * piece, n = std::min<uint64_t>(piece, size - off);
            const size_t nch = (size_t)((n + chunk - 1) / chunk);
            std::atomic<bool> ok{true};
            if (use_map) {
                // the piece's pages become device-readable (a few ms per 256 MiB), the pool hops its block headers in the mapping
                if (!register_piece(off, n)) { std::lock_guard<std::mutex> lk(mu); reader_failed = true; reader_soft = true; reader_err = cov_last_error(s); cv.notify_all(); return; }
                t_wait += now() - t0;
                t0 = now();
                const uint8_t *src = map + off;
                pool.run(nch, [&](size_t c) {
                    const size_t o0 = c * chunk, e = (size_t)std::min<uint64_t>(n, (uint64_t)o0 + chunk);
                    prewalk(src + o0, e - o0, off + o0, pre[(size_t)slot * chunks_per_piece + c]);
                });
                t_read += now() - t0;
            } else {
            if (k >= NS && cov_ingest_slot_wait(s, slot) != COV_OK) {
                std::lock_guard<std::mutex> lk(mu); reader_failed = true; reader_err = cov_last_error(s); cv.notify_all(); return;
            }
            t_wait += now() - t0;
            if (!buf[slot]) {
                buf[slot] = (uint8_t *)cov_host_alloc(std::min<uint64_t>(piece, size - f_lo));
                if (!buf[slot]) { std::lock_guard<std::mutex> lk(mu); reader_failed = true; reader_err = "no page-locked staging memory (is a HIP device usable?)"; cv.notify_all(); return; }
            }
            t0 = now();
            uint8_t *dst = buf[slot];
            const size_t nz = (copy_map && zap_mode == 2 && k > 0) ? (size_t)((piece + ZPART - 1) / ZPART) : 0;      // the piece in front, in parts
            pool.run(nch + nz, [&](size_t c) {
                if (c >= nch) { const uint64_t a = off - piece + (uint64_t)(c - nch) * ZPART; zap_range(a, std::min<uint64_t>(a + ZPART, off)); return; }
                const size_t o0 = c * chunk;
                size_t o = o0; const size_t e = (size_t)std::min<uint64_t>(n, (uint64_t)o + chunk);
                if (copy_map) {
                    // from the mapping, with stores that go past the caches; the block headers are hopped in the SOURCE, whose lines the
                    // copy has just loaded
                    const uint8_t *src = map + off + o0;
                    stream_copy(dst + o0, src, e - o0);
                    prewalk(src, e - o0, off + o0, pre[(size_t)slot * chunks_per_piece + c]);
                    if (zap_mode == 1) zap_range(off + o0, off + e);
                    return;
                }
                while (o < e) {
                    const ssize_t r = pread(fd, dst + o, e - o, (off_t)(off + o));
                    if (r <= 0) { ok = false; return; }
                    o += (size_t)r;
                }
                prewalk(dst + o0, e - o0, off + o0, pre[(size_t)slot * chunks_per_piece + c]);
            });
            t_read += now() - t0;
            }
            std::lock_guard<std::mutex> lk(mu);
            if (!ok) { reader_failed = true; reader_err = std::string("read error on ") + path; cv.notify_all(); return; }
            ready = k + 1;
            cv.notify_all();
        }
        // what of the mapping is still in the page table goes now, beside the device's last window (the pool has nothing else to do)
        if (copy_map && zap_mode != 1 && n_pieces) {
            const uint64_t a0 = zap_mode == 2 ? f_lo + (n_pieces - 1) * piece : f_lo;
            const size_t parts = (size_t)((size - a0 + ZPART - 1) / ZPART);
            pool.run(parts, [&](size_t c) { const uint64_t a = a0 + (uint64_t)c * ZPART; zap_range(a, std::min<uint64_t>(a + ZPART, size)); });
        }
    });
    struct Join { std::thread &t; std::mutex &mu; std::condition_variable &cv; bool &stop;
                  ~Join() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); } } joiner{reader, mu, cv, stop};
    for (uint64_t k = 0; k < n_pieces; k++) {
        const int slot = (int)(k % NS);
        const uint64_t off = f_lo + k * piece, n = std::min<uint64_t>(piece, size - off);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return reader_failed || ready > k; });
            if (reader_failed) return fail(reader_soft ? 1 : -1, reader_err);
        }
        const uint8_t *dst = use_map ? map + off : buf[slot];
        double t0 = now();
        // ---- block headers completed by this piece
        auto byte_at = [&](uint64_t a) -> uint8_t {   // absolute file offset, within this piece or the saved tail of the previous one
            if (a >= off) return dst[a - off];
            return tail[tail_len - (size_t)(tail_end - a)];
        };
        blocks.clear();
        const uint64_t have = off + n;
        for (;;) {
            if (pending_bsize == 0 && next_blk >= off && next_blk < have) {     // a chunk's own hop starts exactly here: take its blocks
                const PreChunk &P = pre[(size_t)slot * chunks_per_piece + (size_t)((next_blk - off) / chunk)];
                if (P.first == next_blk && !P.blocks.empty()) {
                    for (const PreBlock &pb : P.blocks) {
                        cov_bgzf_block b;
                        b.in_off = pb.hdr + 18; b.in_len = pb.bsize - 26; b.crc = pb.crc; b.isize = pb.isize; b.out_off = out_off; b.pad = 0;
                        out_off += pb.isize;
                        blocks.push_back(b);
                    }
                    next_blk = P.next;
                    continue;
                }
            }
            if (pending_bsize == 0) {          // header of the next block (may straddle into the saved tail of the previous piece)
                if (next_blk + 18 > have) break;
                uint8_t hb[18];
                for (int q = 0; q < 18; q++) hb[q] = byte_at(next_blk + (uint64_t)q);
                if (hb[0] != 0x1f || hb[1] != 0x8b || hb[2] != 8 || !(hb[3] & 4)) return fail(1, "not a BGZF block (device ingest hands the file to the CPU reader)");
                const uint32_t xlen = hb[10] | (hb[11] << 8);
                if (xlen != 6 || hb[12] != 66 || hb[13] != 67 || hb[14] != 2 || hb[15] != 0)    // extra subfields besides BC: rare, let the CPU reader take it
                    return fail(1, "BGZF block with extra subfields (device ingest hands the file to the CPU reader)");
                pending_bsize = (uint64_t)(hb[16] | (hb[17] << 8)) + 1;
                if (pending_bsize < 26) return fail(1, "malformed BGZF block header");
            }
            const uint64_t bsize = pending_bsize;
            if (next_blk + bsize > have) break;             // completed by a later piece (its header is not read again)
            cov_bgzf_block b;
            b.in_off = next_blk + 18; b.in_len = (uint32_t)(bsize - 26);
            uint8_t tr[8];
            for (int q = 0; q < 8; q++) tr[q] = byte_at(next_blk + bsize - 8 + (uint64_t)q);
            memcpy(&b.crc, tr, 4); memcpy(&b.isize, tr + 4, 4);
            if (b.isize > 65536u) return fail(1, "BGZF block inflates to more than 64 KiB");
            b.out_off = out_off; b.pad = 0;
            out_off += b.isize;
            blocks.push_back(b);
            next_blk += bsize;
            pending_bsize = 0;
        }
        tail_len = (size_t)std::min<uint64_t>(sizeof tail, n);
        memcpy(tail, dst + n - tail_len, tail_len);
        tail_end = have;
        t_walk += now() - t0;
        t0 = now();
        if (begun.valid() && begun.get() != 0) return fail(-1, cov_last_error(s));
        if (k == 0) t_first_feed = now() - t_start;
        if (cov_ingest_feed(s, slot, dst, off, n, blocks.data(), (uint32_t)blocks.size()) != COV_OK) return fail(-1, cov_last_error(s));
        t_feed += now() - t0;
        { std::lock_guard<std::mutex> lk(mu); fed = k + 1; }
        cv.notify_all();
    }
    reader.join();
    if (begun.valid() && begun.get() != 0) return fail(-1, cov_last_error(s));      // (a span without pieces)
    if (next_blk != size) return fail(1, "truncated BGZF block at the end of the file");
    double t0 = now();
    // (Giving the staging slots back to the system BESIDE the device's last rounds was measured — exit 0.03 s shorter, tail as much
    // longer, hipHostFree waits for the device: profiles/r03_tail_variants.log — and it had a second thread call into the session that
    // cov_ingest_end is working on.  With covh_bam_set_release_staging the slots now go back right after the ingest has ended.)
    uint64_t nrec = 0;
    const cov_status rc = cov_ingest_end(s, &nrec);
    if (g_release_staging.load() && !use_map) {
        for (int k = 0; k < NS; k++)
            if (buf[k]) { cov_host_free(buf[k]); buf[k] = nullptr; }
        cov_host_trim();
    }
    const double t_end = now() - t0;
    if (timing) { timing[0] = t_read; timing[1] = t_wait; timing[2] = t_end; timing[3] = now() - t_start; timing[4] = t_first_feed > 0 ? t_first_feed : t_begin; timing[5] = t_walk; timing[6] = t_feed; timing[7] = use_map ? (map_upfront ? 2 : 1) : copy_map ? 3 : 0; }      /* [7]: where the DMA read the bytes: 0 staging slots filled by pread, 1 mapped file, 2 mapped and registered up front, 3 staging slots filled from the mapping */
    if (rc == COV_ERR_INGEST_FALLBACK) return fail(1, cov_last_error(s));
    if (rc == COV_ERR_UNSORTED) return fail(-2, cov_last_error(s));      // keys decrease inside the span: the caller may send the file through one device whole
    if (rc != COV_OK) return fail(-1, cov_last_error(s));
    if (n_records) *n_records = nrec;
    return 0;
}

}  // extern "C"

extern "C" {

struct covh_bam { Bam b; };

covh_bam *covh_bam_open(const char *path, int threads, int want_names, char *err, size_t errcap) {
    const bool timing = covh_timing_on();
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
void covh_bam_set_release_staging(int on) { g_release_staging.store(on != 0); }
void covh_bam_set_concurrent_feeders(int n) { g_feeders.store(n < 1 ? 1 : n); }
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

// ---- writer: SoA batch -> BGZF BAM (synthetic benchmark / test inputs).
// with_seq: 0 = SEQ '*'; 1 = SEQ all 'A', QUAL 0xff, names r<i> (compresses to ~14 B per read: inflate cost far below any
// real BAM's); 2 = realistic entropy: uniformly random bases, Phred-like binned qualities (37/25/11/2 with probabilities
// .88/.08/.03/.01, independent per base, i.e. a little MORE entropy than real instrument output), Illumina-style read
// names — about 60-70 compressed bytes per 150 bp read, like a real short-read BAM.
// Records are serialised and deflated chunk by chunk (bounded memory), every stage threaded.
namespace {
inline uint64_t mix64(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
inline int synth_name(char *qn, uint64_t i, int mode) {
    if (mode < 2) return snprintf(qn, 64, "r%llu", (unsigned long long)i) + 1;
    // instrument : run : flowcell : lane : tile : x : y — distinct for distinct i: (flowcell, lane, tile, x) encode i bijectively
    // (bits 24.., 22-23, 14-21, and the low 14 bits through an odd multiplier), the rest is hash noise
    const uint64_t h = mix64(i * 2 + 1);
    return snprintf(qn, 64, "A00%03u:%u:HXX%05u:%u:%u:%u:%u", (unsigned)(h % 7), 100 + (unsigned)((h >> 8) % 5), (unsigned)(i >> 24) + 17000,
                    1 + (unsigned)((i >> 22) & 3), 1101 + (unsigned)((i >> 14) & 255), (unsigned)(((i & 16383) * 7919u) & 16383u) + 1000,
                    (unsigned)((h >> 40) % 36000) + 1000) + 1;
}
}  // namespace

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
    FILE *f = fopen(path, "wb");
    if (!f) return 2;
    Pool pool(std::max(1, threads));
    const uint64_t R = b->n_records;
    // with_seq = 3: as 2, and records 2k / 2k + 1 that lie on the same reference are MATES: one read name, next_refID = their
    // reference, next_pos = the other's position (the pair-mode filter's input; every other record keeps a name of its own)
    const bool paired = with_seq == 3;
    auto name_id = [&](uint64_t i) -> uint64_t { if (!paired) return i; const uint64_t m = i ^ 1ull; return (m < R && b->tid[m] == b->tid[i]) ? (i & ~1ull) : i; };
    const uint64_t CH = 1u << 21;                      // records per chunk
    const size_t BLK = 0xff00;
    std::vector<uint8_t> raw(head);                    // bytes not yet written as blocks (header first)
    std::vector<uint64_t> off;
    std::vector<std::vector<uint8_t>> comp;
    const LibDeflate &LD = libdeflate();
    std::atomic<bool> ok{true};
    uint8_t qtab[256];
    for (int v = 0; v < 256; v++) qtab[v] = v < 225 ? 37 : v < 246 ? 25 : v < 254 ? 11 : 2;
    static const uint8_t btab[4] = {1, 2, 4, 8};
    auto flush_blocks = [&](bool final) {
        const size_t nblk = final ? (raw.size() + BLK - 1) / BLK : raw.size() / BLK;
        if (comp.size() < nblk) comp.resize(nblk);
        pool.run(nblk, [&](size_t k) {
            const size_t s0 = k * BLK, n = std::min(BLK, raw.size() - s0);
            std::vector<uint8_t> &o = comp[k];
            o.resize(n + n / 8 + 256);
            size_t clen = 0;
            if (LD.ok_c) {
                thread_local TlsCompressor tc;
                void *cmp = tc.get(level);
                clen = cmp ? LD.compress(cmp, &raw[s0], n, o.data() + 18, o.size() - 26) : 0;
            }
            if (clen == 0) {
                z_stream zs; memset(&zs, 0, sizeof zs);
                if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok = false; return; }
                zs.next_in = &raw[s0]; zs.avail_in = (uInt)n; zs.next_out = o.data() + 18; zs.avail_out = (uInt)(o.size() - 26);
                if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { ok = false; deflateEnd(&zs); return; }
                clen = zs.total_out;
                deflateEnd(&zs);
            }
            static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            memcpy(o.data(), hdr, 16);
            const uint16_t bsz = (uint16_t)(clen + 25);
            memcpy(o.data() + 16, &bsz, 2);
            const uint32_t crc = LD.ok ? LD.crc32(0, &raw[s0], n) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), &raw[s0], (uInt)n), isz = (uint32_t)n;
            memcpy(o.data() + 18 + clen, &crc, 4); memcpy(o.data() + 22 + clen, &isz, 4);
            o.resize(clen + 26);
        });
        for (size_t k = 0; k < nblk; k++) fwrite(comp[k].data(), 1, comp[k].size(), f);
        const size_t used = std::min(raw.size(), nblk * BLK);
        raw.erase(raw.begin(), raw.begin() + used);
    };
    for (uint64_t c0 = 0; c0 < R || c0 == 0; c0 += CH) {
        const uint64_t c1 = std::min(R, c0 + CH), n = c1 - c0;
        off.assign(n + 1, 0);
        // per-record sizes -> offsets (blocked prefix sum)
        const size_t nb = std::max<size_t>(1, std::min<size_t>((size_t)pool.size() * 4, n / 16384 + 1));
        std::vector<uint64_t> tot(nb + 1, 0);
        pool.run(nb, [&](size_t k) {
            const uint64_t lo = n * k / nb, hi = n * (k + 1) / nb;
            uint64_t acc = 0;
            for (uint64_t j = lo; j < hi; j++) {
                const uint64_t i = c0 + j;
                const uint32_t nc = b->cigar_off[i + 1] - b->cigar_off[i];
                const uint32_t ls = with_seq ? b->l_seq[i] : 0;
                char qn[64]; const int lq = synth_name(qn, name_id(i), with_seq);
                const uint32_t aux = b->nm_kind[i] == COV_NM_UNSIGNED ? (b->nm[i] < 256 ? 4 : b->nm[i] < 65536 ? 5 : 7) : b->nm_kind[i] == COV_NM_BADTYPE ? 4 : 0;
                acc += 36 + lq + 4ull * nc + (ls + 1) / 2 + ls + aux;
                off[j + 1] = acc;
            }
            tot[k + 1] = acc;
        });
        for (size_t k = 0; k < nb; k++) tot[k + 1] += tot[k];
        pool.run(nb, [&](size_t k) {
            if (!k) return;
            const uint64_t lo = n * k / nb, hi = n * (k + 1) / nb;
            for (uint64_t j = lo; j < hi; j++) off[j + 1] += tot[k];
        });
        const size_t base0 = raw.size();
        raw.resize(base0 + off[n]);
        uint8_t *base = raw.data() + base0;
        pool.run(n, [&](size_t j) {
            const uint64_t i = c0 + j;
            uint8_t *p = base + off[j];
            const uint32_t nc = b->cigar_off[i + 1] - b->cigar_off[i];
            const uint32_t ls = with_seq ? b->l_seq[i] : 0;
            char qn[64]; const int lq = synth_name(qn, name_id(i), with_seq);
            const uint32_t bs = (uint32_t)(off[j + 1] - off[j] - 4);
            auto w32 = [&](uint32_t x) { memcpy(p, &x, 4); p += 4; };
            w32(bs); w32((uint32_t)b->tid[i]); w32((uint32_t)b->pos[i]);
            *p++ = (uint8_t)lq; *p++ = b->mapq[i];
            const uint16_t bin = 4680, ncg = (uint16_t)nc, fl = b->flag[i];
            memcpy(p, &bin, 2); p += 2; memcpy(p, &ncg, 2); p += 2; memcpy(p, &fl, 2); p += 2;
            if (paired) { const uint64_t m = i ^ 1ull; const bool has = m < R && b->tid[m] == b->tid[i]; w32(ls); w32((uint32_t)b->tid[i]); w32(has ? (uint32_t)b->pos[m] : (uint32_t)b->pos[i]); w32(0); }
            else { w32(ls); w32((uint32_t)-1); w32((uint32_t)-1); w32(0); }
            memcpy(p, qn, lq); p += lq;
            if (nc) { memcpy(p, b->cigar + b->cigar_off[i], 4ull * nc); p += 4ull * nc; }
            if (with_seq >= 2) {
                uint64_t st = mix64(i ^ 0x5eedull);
                const uint32_t nb2 = (ls + 1) / 2;
                for (uint32_t k = 0; k < nb2; k += 16) {         // 16 bytes = 32 bases per 64-bit draw
                    uint64_t r = st = mix64(st);
                    for (uint32_t q = k; q < std::min(nb2, k + 16); q++, r >>= 4) p[q] = (uint8_t)((btab[r & 3] << 4) | btab[(r >> 2) & 3]);
                }
                if (ls & 1) p[nb2 - 1] &= 0xf0;
                p += nb2;
                for (uint32_t k = 0; k < ls; k += 8) {
                    uint64_t r = st = mix64(st);
                    for (uint32_t q = k; q < std::min(ls, k + 8); q++, r >>= 8) p[q] = qtab[r & 255];
                }
                p += ls;
            } else {
                memset(p, 0x11, (ls + 1) / 2); p += (ls + 1) / 2;
                memset(p, 0xff, ls); p += ls;
            }
            if (b->nm_kind[i] == COV_NM_UNSIGNED) {
                *p++ = 'N'; *p++ = 'M';
                if (b->nm[i] < 256) { *p++ = 'C'; *p++ = (uint8_t)b->nm[i]; }
                else if (b->nm[i] < 65536) { *p++ = 'S'; const uint16_t v = (uint16_t)b->nm[i]; memcpy(p, &v, 2); p += 2; }
                else { *p++ = 'I'; memcpy(p, &b->nm[i], 4); p += 4; }
            } else if (b->nm_kind[i] == COV_NM_BADTYPE) { *p++ = 'N'; *p++ = 'M'; *p++ = 'c'; *p++ = 1; }
        }, 64);
        flush_blocks(c1 >= R);
        if (!ok) { fclose(f); return 1; }
        if (R == 0) break;
    }
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, f);
    const bool werr = ferror(f) != 0;
    if (fclose(f) != 0 || werr) return 3;
    return 0;
}

}  // extern "C"

namespace {

// A BGZF stream written as it grows: blocks of 0xff00 bytes (what htslib writes); whenever 1024 of them are full they are handed to ONE
// background task that compresses them on `threads` threads and writes them, while the caller goes on filling the next batch (at most
// one batch is in flight: memory stays at two batches); the last (short) block and the EOF marker at close().  libdeflate when the
// runtime has it (2-3 x zlib at the same level), as in the synthetic writer above.
class BgzfOut {
    FILE *f_; const int level_, threads_;
    std::vector<uint8_t> pend_, busy_;
    std::vector<std::vector<uint8_t>> comp_;
    std::thread task_;
    std::atomic<bool> ok_{true};
    static constexpr size_t BLK = 0xff00, FLUSH_BLOCKS = 1024;
    // busy_ (whole blocks, or everything when `all`) -> compressed -> the file
    void write_busy() {
        const size_t n = busy_.size(), nblk = (n + BLK - 1) / BLK;
        if (comp_.size() < nblk) comp_.resize(nblk);
        const LibDeflate &LD = libdeflate();
        const uint8_t *data = busy_.data(); const int level = level_;
        std::atomic<bool> &ok = ok_;
        parallel_for(nblk, threads_, [&](size_t k) {
            const size_t s0 = k * BLK, len = std::min(BLK, n - s0);
            std::vector<uint8_t> &o = comp_[k];
            o.resize(len + len / 8 + 256);
            size_t clen = 0;
            if (LD.ok_c) {
                thread_local TlsCompressor tc;
                void *cmp = tc.get(level);
                clen = cmp ? LD.compress(cmp, data + s0, len, o.data() + 18, o.size() - 26) : 0;
            }
            if (clen == 0) {
                z_stream zs; memset(&zs, 0, sizeof zs);
                if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok = false; return; }
                zs.next_in = const_cast<uint8_t *>(data + s0); zs.avail_in = (uInt)len; zs.next_out = o.data() + 18; zs.avail_out = (uInt)(o.size() - 26);
                if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { ok = false; deflateEnd(&zs); return; }
                clen = zs.total_out;
                deflateEnd(&zs);
            }
            static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            memcpy(o.data(), hdr, 16);
            const uint16_t bsz = (uint16_t)(clen + 25);
            memcpy(o.data() + 16, &bsz, 2);
            const uint32_t crc = LD.ok ? LD.crc32(0, data + s0, len) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), data + s0, (uInt)len), isz = (uint32_t)len;
            memcpy(o.data() + 18 + clen, &crc, 4); memcpy(o.data() + 22 + clen, &isz, 4);
            o.resize(clen + 26);
        });
        if (!ok_) return;
        for (size_t k = 0; k < nblk; k++) if (fwrite(comp_[k].data(), 1, comp_[k].size(), f_) != comp_[k].size()) { ok_ = false; return; }
    }
    bool wait() { if (task_.joinable()) task_.join(); return ok_; }
    bool flush(bool all) {
        if (!wait()) return false;                       // the batch before this one is on the file
        const size_t take = all ? pend_.size() : pend_.size() / BLK * BLK;
        if (!take) return true;
        busy_.assign(pend_.begin(), pend_.begin() + (ptrdiff_t)take);
        pend_.erase(pend_.begin(), pend_.begin() + (ptrdiff_t)take);
        task_ = std::thread([this] { write_busy(); });
        return true;
    }
public:
    BgzfOut(FILE *f, int level, int threads) : f_(f), level_(level), threads_(std::max(1, threads)) {}
    BgzfOut(const BgzfOut &) = delete;
    ~BgzfOut() { if (task_.joinable()) task_.join(); }
    bool put(const uint8_t *p, size_t n) {
        pend_.insert(pend_.end(), p, p + n);
        return pend_.size() < FLUSH_BLOCKS * BLK || flush(false);
    }
    bool close() {
        if (!flush(true) || !wait()) return false;
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        return fwrite(eof, 1, 28, f_) == 28;
    }
};

}  // namespace

extern "C" {

// `coverm filter` (bin/coverm.rs:408-472): bam::Reader -> ReferenceSortedBamFilter -> bam::Writer under the reader's header, a record at a
// time.  Here the file goes through in WINDOWS of 64 MiB of BGZF blocks: inflated on `threads` threads behind the cut-off record of the
// window before, the records' filter fields summarised on `threads` threads, the selection by the reference's own state machine
// (csrc/reader_filter.h) in file order, the selected records appended byte for byte (names, bases, qualities, tags) to the output
// stream behind the input's header bytes.  Memory: one window, the output blocks in flight, and — in the pair branch — a copy of every
// first mate that is still waiting for its second (what the reference's first_set holds).
int covh_bam_filter_file(const char *in_path, const char *out_path, const covh_pair_filter *f, int filter_pairs, int include_supplementary,
                         int include_secondary, int filter_out, int level, int threads, uint64_t *n_in, uint64_t *n_out, char *err, size_t errcap) {
    auto fail = [&](const std::string &e) { if (err && errcap) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; } return -1; };
    if (!in_path || !out_path || !f) return fail("covh_bam_filter_file: invalid argument");
    if (n_in) *n_in = 0;
    if (n_out) *n_out = 0;
    const int T = std::max(1, threads);
    const bool timing = covh_timing_on();
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now(), t_part[5] = {0, 0, 0, 0, 0};
    auto stamp = [&](int k) { const double t = now(); t_part[k] += t - t_prev; t_prev = t; };
    struct File { FILE *f = nullptr; ~File() { if (f) fclose(f); } } fi, fo;
    fi.f = fopen(in_path, "rb");
    if (!fi.f) return fail(std::string("Unable to find BAM file ") + in_path);
    size_t CHUNK = (size_t)64 << 20;
    { long long v; if (covknob::get("filter_window_kb", v) && v >= 1) CHUNK = (size_t)v << 10; }      // tests: many small windows
    std::vector<uint8_t> cbuf, u;       // compressed bytes not yet inflated; inflated bytes not yet consumed (a cut-off record in front)
    size_t carry = 0;
    bool eof = false, first_read = true, header_done = false;
    std::vector<std::string> names; std::vector<uint64_t> lens; std::string header_text, e2;
    std::unique_ptr<BgzfOut> W;
    covf::ReaderFilter M(*f, filter_pairs != 0, include_supplementary != 0, include_secondary != 0, filter_out != 0);
    std::unordered_map<uint64_t, std::vector<uint8_t>> parked;      // first mates waiting for their second, by the id the machine knows them by
    uint64_t next_id = 0, n_rec = 0, n_sel = 0;
    std::vector<Blk> blocks;
    std::vector<size_t> rec;
    std::vector<covf::RecSum> sums;
    while (!eof || !cbuf.empty()) {
        if (!eof) {
            const size_t old = cbuf.size();
            cbuf.resize(old + CHUNK);
            const size_t got = fread(cbuf.data() + old, 1, CHUNK, fi.f);
            cbuf.resize(old + got);
            if (got < CHUNK) { if (ferror(fi.f)) return fail(std::string("short read on ") + in_path); eof = true; }
        }
        if (first_read) {
            first_read = false;
            if (cbuf.size() < 2 || cbuf[0] != 0x1f || cbuf[1] != 0x8b) return fail(std::string(in_path) + ": `filter` reads BAM files");
        }
        stamp(0);
        blocks.clear();
        size_t p = 0, total = 0;
        if (!bgzf_block_table(cbuf.data(), cbuf.size(), p, blocks, total, !eof, e2)) return fail(e2);
        u.resize(carry + total);
        {
            std::atomic<bool> ok{true};
            uint8_t *dst = u.data() + carry; const uint8_t *src = cbuf.data();
            parallel_for(blocks.size(), T, [&](size_t i) { if (!inflate_block(src, blocks[i], dst + blocks[i].out_off)) ok = false; });
            if (!ok) return fail("BGZF inflate / CRC failure");
        }
        cbuf.erase(cbuf.begin(), cbuf.begin() + (ptrdiff_t)p);
        stamp(1);
        const size_t N = u.size();
        size_t q = 0;
        if (!header_done) {
            const int hrc = parse_bam_header(u.data(), N, q, names, lens, header_text, e2);
            if (hrc < 0) return fail(e2);
            if (hrc == 0) {
                if (eof && cbuf.empty()) return fail(N < 12 ? "bad BAM magic" : "truncated BAM header");
                carry = N;
                continue;
            }
            header_done = true;
            fo.f = fopen(out_path, "wb");
            if (!fo.f) return fail(std::string("Failed to write BAM file ") + out_path);
            W.reset(new BgzfOut(fo.f, level, T));
            if (!W->put(u.data(), q)) return fail(std::string("Failed to write BAM file ") + out_path);       // the header bytes as they are
        }
        // where the records of this window begin; the last one may be cut off by the window's end
        rec.clear();
        while (q + 4 <= N) {
            const uint32_t bs = rd32(u.data() + q);
            if (bs < 32) return fail("corrupt BAM record");
            if (q + 4 + (size_t)bs > N) break;
            rec.push_back(q);
            q += 4 + (size_t)bs;
        }
        const size_t R = rec.size();
        sums.resize(R);
        {
            std::atomic<bool> ok{true};
            const uint8_t *ub = u.data();
            parallel_for((R + 4095) / 4096, T, [&](size_t blk) {
                for (size_t i = blk * 4096, hi = std::min(R, i + 4096); i < hi; i++) {
                    const uint8_t *r = ub + rec[i], *end = r + 4 + rd32(r);
                    const uint32_t l_read_name = r[12], n_cig = rd16(r + 16), l_seq = rd32(r + 20);
                    const uint8_t *cg0 = r + 36 + l_read_name;
                    if (cg0 + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq > end) { ok = false; return; }
                    covf::RecSum &s = sums[i];
                    s.tid = (int32_t)rd32(r + 4); s.mtid = (int32_t)rd32(r + 24); s.flag = rd16(r + 18); s.mapq = r[13]; s.l_seq = l_seq;
                    s.nm = 0;
                    s.nm_kind = scan_aux(cg0 + 4ull * n_cig + (l_seq + 1ull) / 2 + l_seq, end, s.nm, nullptr, nullptr);
                    const uint8_t *cig = cg0; uint32_t nc = n_cig, cnt = 0;
                    if (maybe_long_cigar(r)) if (const uint8_t *cg = real_cigar_from_cg(r, end, cnt)) { cig = cg; nc = cnt; }      // what record.cigar() returns
                    uint32_t a = 0, d = 0;
                    for (uint32_t c = 0; c < nc; c++) {
                        const uint32_t w = rd32(cig + 4ull * c), op = w & 15u, len = w >> 4;
                        if (op == 0 || op == 1 || op == 7 || op == 8) a += len;
                        else if (op == 2) d += len;
                    }
                    s.aligned_no_del = a; s.aligned_with_del = a + d;
                }
            });
            if (!ok) return fail("corrupt BAM record");
        }
        stamp(2);
        for (size_t i = 0; i < R; i++) {
            const uint8_t *r = u.data() + rec[i];
            const size_t len = 4 + (size_t)rd32(r);
            const uint32_t l_read_name = r[12];
            uint64_t partner = 0; bool forget = false;
            const covf::ReaderFilter::Act act = M.push(sums[i], (const char *)r + 36, l_read_name ? l_read_name - 1u : 0u, next_id, &partner, &forget);
            if (M.err == COV_ERR_NM_MISSING) return fail("Mapping record encountered that does not have an 'NM' auxiliary tag in the SAM/BAM format");
            if (M.err) return fail("Unexpected data type of NM aux tag");
            if (forget) parked.clear();
            bool okw = true;
            switch (act) {
            case covf::ReaderFilter::EMIT: okw = W->put(r, len); n_sel++; break;
            case covf::ReaderFilter::PARK: parked.emplace(next_id++, std::vector<uint8_t>(r, r + len)); break;
            case covf::ReaderFilter::EMIT_PAIR: {
                auto it = parked.find(partner);
                if (it == parked.end()) return fail("covh_bam_filter_file: a parked record is missing");
                okw = W->put(it->second.data(), it->second.size()) && W->put(r, len);
                parked.erase(it); n_sel += 2;
                break;
            }
            case covf::ReaderFilter::DROP_PAIR: parked.erase(partner); break;
            case covf::ReaderFilter::DROP: break;
            }
            if (!okw) return fail(std::string("Failed to write BAM file ") + out_path);
        }
        n_rec += R;
        stamp(3);
        carry = N - q;
        if (carry && q) memmove(u.data(), u.data() + q, carry);
    }
    if (!header_done) return fail("truncated BAM header");
    if (carry) return fail("truncated BAM record");
    const bool okc = W->close();
    const bool werr = ferror(fo.f) != 0;
    const int crc = fclose(fo.f); fo.f = nullptr;
    if (crc != 0 || werr || !okc) return fail(std::string("Failed to write BAM file ") + out_path);
    stamp(4);
    if (timing) fprintf(stderr, "[coverm-amd] filter %s: file read %.3fs, inflate %.3fs, records %.3fs, reader filter + deflate + write %.3fs, close %.3fs\n", in_path,
                        t_part[0], t_part[1], t_part[2], t_part[3], t_part[4]);
    if (n_in) *n_in = n_rec;
    if (n_out) *n_out = n_sel;
    return 0;
}

}  // extern "C"
