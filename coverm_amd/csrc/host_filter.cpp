// host_filter.cpp — the reader-stage PAIR filter on the host (C++, threaded).
//
// ReferenceSortedBamFilter::read, pair branch (src/filter.rs:117-228) with filter_out = true, as the coverage
// commands use it: among primary proper-pair records (`!secondary && !supplementary && proper_pair`, :133-136) a
// record whose query name has not been seen in the current reference is parked if its mate maps to the same
// reference (:164-171); when the second record of that name arrives the pair is judged (single-read predicate on
// both when single thresholds are set, :176-190, then read_pair_passes_filter :281-336) and, if it passes, the
// FIRST record is returned followed by the SECOND (:191-208).  The parked set is cleared whenever the reference
// changes (:139-149).  Output here = the indices the reference would return, in that order.
//
// The single-read branch runs on the device (k_prep); this branch needs read names, which never cross the C ABI, so
// it stays above it, like the reference's own filter sits above the scan.  References are independent, so they
// are processed by a thread pool, each with an open-addressing table keyed by a 64-bit name hash (names compared in
// full on a hit: no reliance on hash uniqueness).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/coverm_host.h"
#include "reader_filter.h"

namespace {

thread_local std::string t_err;

inline uint64_t hash_name(const char *p, size_t n) {   // FNV-1a, 64 bit
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (uint8_t)p[i]; h *= 1099511628211ull; }
    return h ^ (h >> 29);
}

struct Judge {
    const cov_batch &b;
    const covh_pair_filter &f;
    int err = 0;   // COV_ERR_NM_MISSING / COV_ERR_NM_BADTYPE seen while judging (reference: nm() panics, lib.rs:138-158)
    uint32_t aligned_of(uint64_t i, bool with_del) const {
        uint32_t a = 0;
        for (uint32_t c = b.cigar_off[i]; c < b.cigar_off[i + 1]; c++) {
            const uint32_t op = b.cigar[c] & 15u, len = b.cigar[c] >> 4;
            if (op == 0 || op == 1 || op == 7 || op == 8 || (with_del && op == 2)) a += len;
        }
        return a;
    }
    bool nm(uint64_t i, uint64_t &v) {
        if (b.nm_kind[i] == COV_NM_UNSIGNED) { v = b.nm[i]; return true; }
        if (!err) err = b.nm_kind[i] == COV_NM_ABSENT ? COV_ERR_NM_MISSING : COV_ERR_NM_BADTYPE;
        return false;
    }
    bool single_ok(uint64_t i) {   // filter.rs:243-279
        if (f.min_mapq != 255 && (b.mapq[i] < f.min_mapq || b.mapq[i] == 255)) return false;
        uint64_t e;
        if (!nm(i, e)) return false;
        const uint32_t al = aligned_of(i, true);
        return al >= f.min_aligned_length_single && (float)al / (float)b.l_seq[i] >= f.min_aligned_percent_single &&
               1.0f - (float)e / (float)al >= f.min_percent_identity_single;
    }
    bool pair_ok(uint64_t i2, uint64_t i1) {   // filter.rs:281-336
        if (f.min_mapq != 255 && (b.mapq[i1] < f.min_mapq || b.mapq[i2] < f.min_mapq || b.mapq[i1] == 255 || b.mapq[i2] == 255))
            return false;
        uint64_t e2, e1;
        if (!nm(i2, e2) || !nm(i1, e1)) return false;
        const uint64_t e = e2 + e1;
        const uint32_t al = aligned_of(i2, false) + aligned_of(i1, false);
        return al >= f.min_aligned_length_pair &&
               (float)al / (float)((uint64_t)b.l_seq[i1] + b.l_seq[i2]) >= f.min_aligned_percent_pair &&
               1.0f - (float)e / (float)al >= f.min_percent_identity_pair;
    }
};

// One maximal run [lo, hi) of eligible records that share a reference.  `out` receives (first, second) pairs.
void pair_segment(const cov_batch &b, const int32_t *mtid, const uint32_t *qoff, const char *qn, const covh_pair_filter &f,
                  uint64_t lo, uint64_t hi, std::vector<uint64_t> &slots, std::vector<uint64_t> &out, int &err) {
    Judge J{b, f};
    uint64_t n_el = 0;
    for (uint64_t i = lo; i < hi; i++) n_el += !((b.flag[i] & 0x900) || !(b.flag[i] & 0x2));
    uint64_t cap = 16;
    while (cap < 2 * n_el + 2) cap <<= 1;
    const uint64_t EMPTY = ~0ull, TOMB = ~0ull - 1;
    slots.assign(cap, EMPTY);
    const int32_t cur = b.tid[lo];
    for (uint64_t i = lo; i < hi; i++) {
        const uint16_t flag = b.flag[i];
        if ((flag & 0x900) || !(flag & 0x2)) continue;
        const char *q = qn + qoff[i];
        const size_t ql = qoff[i + 1] - qoff[i];
        uint64_t h = hash_name(q, ql) & (cap - 1);
        uint64_t tomb = EMPTY;
        bool found = false;
        for (;; h = (h + 1) & (cap - 1)) {
            const uint64_t v = slots[h];
            if (v == EMPTY) break;
            if (v == TOMB) { if (tomb == EMPTY) tomb = h; continue; }
            const size_t vl = qoff[v + 1] - qoff[v];
            if (vl == ql && memcmp(qn + qoff[v], q, ql) == 0) { found = true; break; }
        }
        if (!found) {
            if (mtid[i] == cur) slots[tomb != EMPTY ? tomb : h] = i;     // :164-171
        } else {
            const uint64_t i1 = slots[h];
            slots[h] = TOMB;
            if ((!f.filter_single || (J.single_ok(i1) && J.single_ok(i))) && J.pair_ok(i, i1)) { out.push_back(i1); out.push_back(i); }
        }
    }
    if (J.err && !err) err = J.err;
}

template <typename F>
void run_pool(size_t n, int threads, F fn) {
    threads = std::max(1, std::min<int>(threads, (int)std::max<size_t>(1, n)));
    if (threads == 1) { for (size_t i = 0; i < n; i++) fn(i, 0); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t]() { for (size_t i; (i = next.fetch_add(1)) < n;) fn(i, t); });
    for (auto &th : pool) th.join();
}

}  // namespace

extern "C" {

void covh_free(void *p) { free(p); }

int covh_pair_mode_order(const cov_batch *b, const int32_t *mtid, const uint32_t *qname_off, const char *qnames,
                         const covh_pair_filter *f, int threads, uint64_t **order_out, uint64_t *n_out) {
    if (!b || !f || !order_out || !n_out || (b->n_records && (!mtid || !qname_off || !qnames))) return COV_ERR_INVALID_ARG;
    *order_out = nullptr; *n_out = 0;
    const uint64_t R = b->n_records;
    // segments of the ELIGIBLE subsequence that share a reference (the parked set is cleared when it changes)
    std::vector<std::pair<uint64_t, uint64_t>> seg;
    {
        bool open = false; int32_t cur = 0; uint64_t lo = 0, last = 0;
        for (uint64_t i = 0; i < R; i++) {
            const uint16_t flag = b->flag[i];
            if ((flag & 0x900) || !(flag & 0x2)) continue;
            if (!open || b->tid[i] != cur) {
                if (open) seg.emplace_back(lo, last + 1);
                open = true; cur = b->tid[i]; lo = i;
            }
            last = i;
        }
        if (open) seg.emplace_back(lo, last + 1);
    }
    // inside [lo, hi) every eligible record has the segment's reference; ineligible ones are skipped by pair_segment
    std::vector<std::vector<uint64_t>> outs(seg.size());
    threads = std::max(1, threads);
    std::vector<std::vector<uint64_t>> slots((size_t)threads);
    std::vector<int> errs((size_t)threads, 0);
    // large references first, so that the pool does not end on one of them
    std::vector<size_t> by_size(seg.size());
    for (size_t k = 0; k < seg.size(); k++) by_size[k] = k;
    std::sort(by_size.begin(), by_size.end(), [&](size_t x, size_t y) { return seg[x].second - seg[x].first > seg[y].second - seg[y].first; });
    run_pool(seg.size(), threads, [&](size_t j, int t) {
        const size_t k = by_size[j];
        pair_segment(*b, mtid, qname_off, qnames, *f, seg[k].first, seg[k].second, slots[(size_t)t], outs[k], errs[(size_t)t]);
    });
    for (int e : errs) if (e) return e;
    uint64_t tot = 0;
    for (auto &o : outs) tot += o.size();
    uint64_t *ord = (uint64_t *)malloc(std::max<uint64_t>(1, tot) * sizeof(uint64_t));
    if (!ord) return COV_ERR_INVALID_ARG;
    uint64_t w = 0;
    for (auto &o : outs) { if (!o.empty()) memcpy(ord + w, o.data(), o.size() * sizeof(uint64_t)); w += o.size(); }
    *order_out = ord; *n_out = tot;
    return COV_OK;
}

// ReferenceSortedBamFilter::read as a whole (src/filter.rs:84-228): both branches, filter_out = true (what the coverage commands and
// `coverm filter` use) or false (`coverm filter --inverse`).  Serial, in file order, names compared in full: this is the `filter`
// subcommand's selection (bin/coverm.rs:408-472), whose cost is the BAM rewrite around it; the coverage path has its own fast
// implementations (the device's k_prep / cov_pair_filter_apply, covh_pair_mode_order above).
//   single branch (filter_single && !filter_pairs, :88-116): an unmapped record is returned when !filter_out; a record that passes
//     the flag test (:100-102) is returned iff single_read_passes_filter == filter_out; every other record is dropped;
//   pair branch (:117-228): an unmapped record is returned when !filter_out; secondary / supplementary records are dropped; a record
//     that is not a proper pair is dropped (filter_out) or returned (!filter_out); proper pairs are matched by name within one
//     reference and the pair is returned, first mate then second, iff (its judgement == filter_out).
int covh_reader_filter_order(const cov_batch *b, const int32_t *mtid, const uint32_t *qname_off, const char *qnames, const covh_pair_filter *f,
                             int filter_pairs, int include_supplementary, int include_secondary, int filter_out, uint64_t **order_out, uint64_t *n_out) {
    if (!b || !f || !order_out || !n_out || (b->n_records && (!mtid || !qname_off || !qnames))) return COV_ERR_INVALID_ARG;
    *order_out = nullptr; *n_out = 0;
    const uint64_t R = b->n_records;
    covf::ReaderFilter M(*f, filter_pairs != 0, include_supplementary != 0, include_secondary != 0, filter_out != 0);
    std::vector<uint64_t> ord;
    for (uint64_t i = 0; i < R; i++) {
        covf::RecSum r;
        r.tid = b->tid[i]; r.mtid = mtid[i]; r.flag = b->flag[i]; r.mapq = b->mapq[i]; r.nm_kind = b->nm_kind[i]; r.nm = b->nm[i]; r.l_seq = b->l_seq[i];
        covf::aligned_lengths(b->cigar + b->cigar_off[i], b->cigar_off[i + 1] - b->cigar_off[i], r.aligned_with_del, r.aligned_no_del);
        uint64_t partner = 0; bool forget = false;
        const covf::ReaderFilter::Act act = M.push(r, qnames + qname_off[i], qname_off[i + 1] - qname_off[i], i, &partner, &forget);
        if (M.err) return M.err;
        if (act == covf::ReaderFilter::EMIT) ord.push_back(i);
        else if (act == covf::ReaderFilter::EMIT_PAIR) { ord.push_back(partner); ord.push_back(i); }
    }
    uint64_t *o = (uint64_t *)malloc(std::max<size_t>(1, ord.size()) * sizeof(uint64_t));
    if (!o) return COV_ERR_INVALID_ARG;
    if (!ord.empty()) memcpy(o, ord.data(), ord.size() * sizeof(uint64_t));
    *order_out = o; *n_out = ord.size();
    return COV_OK;
}

// Gathers the records `order[0..n)` of `src` into a new batch (page-locked when a device is usable): what the scan
// receives after the reader stage.  Release with covh_batch_free.
int covh_batch_select(const cov_batch *src, const uint64_t *order, uint64_t n, int threads, cov_batch *out) {
    if (!src || !out || (n && !order)) return COV_ERR_INVALID_ARG;
    auto alloc = [](size_t bytes) -> void * { void *p = cov_host_alloc(bytes ? bytes : 1); return p ? p : malloc(bytes ? bytes : 1); };
    std::vector<uint64_t> coff(n + 1, 0);
    for (uint64_t j = 0; j < n; j++) coff[j + 1] = coff[j] + (src->cigar_off[order[j] + 1] - src->cigar_off[order[j]]);
    if (coff[n] >= 0xfffffff0ull) return COV_ERR_INVALID_ARG;
    int32_t *tid = (int32_t *)alloc(n * 4), *pos = (int32_t *)alloc(n * 4);
    uint16_t *flag = (uint16_t *)alloc(n * 2);
    uint8_t *mapq = (uint8_t *)alloc(n), *nmk = (uint8_t *)alloc(n);
    uint32_t *nm = (uint32_t *)alloc(n * 4), *lseq = (uint32_t *)alloc(n * 4), *co = (uint32_t *)alloc((n + 1) * 4),
             *cig = (uint32_t *)alloc((size_t)coff[n] * 4 + 4);
    const size_t blocks = (size_t)((n + 65535) / 65536);
    run_pool(blocks, threads, [&](size_t bk, int) {
        const uint64_t lo = (uint64_t)bk * 65536, hi = std::min<uint64_t>(n, lo + 65536);
        for (uint64_t j = lo; j < hi; j++) {
            const uint64_t i = order[j];
            tid[j] = src->tid[i]; pos[j] = src->pos[i]; flag[j] = src->flag[i]; mapq[j] = src->mapq[i]; nmk[j] = src->nm_kind[i];
            nm[j] = src->nm[i]; lseq[j] = src->l_seq[i]; co[j] = (uint32_t)coff[j];
            const uint32_t c0 = src->cigar_off[i], k = src->cigar_off[i + 1] - c0;
            if (k) memcpy(cig + coff[j], src->cigar + c0, (size_t)k * 4);
        }
    });
    co[n] = (uint32_t)coff[n];
    out->tid = tid; out->pos = pos; out->flag = flag; out->mapq = mapq; out->nm = nm; out->nm_kind = nmk; out->l_seq = lseq;
    out->cigar_off = co; out->cigar = cig; out->n_records = n;
    return COV_OK;
}

void covh_batch_free(cov_batch *b) {
    if (!b) return;
    const void *ps[] = {b->tid, b->pos, b->flag, b->mapq, b->nm, b->nm_kind, b->l_seq, b->cigar_off, b->cigar};
    for (const void *p : ps)
        if (p && !cov_host_free(const_cast<void *>(p))) free(const_cast<void *>(p));
    memset(b, 0, sizeof *b);
}

}  // extern "C"
