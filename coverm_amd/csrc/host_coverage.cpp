// Host side above the covermhip ABI (include/coverm_host.h): estimator finalisation from integer
// statistics, the three scan drivers, CoverageTaker implementations and CoveragePrinter.
// Pure C++17, no HIP: this is where the reference's Rust `calculate_coverage` / takers / printers sit.
#include "../../include/coverm_host.h"
#include "knobs.h"

#include <algorithm>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstring>
#include <limits>
#include <optional>
#include <string>
#include <thread>
#include <string_view>
#include <vector>

typedef uint64_t u64;
typedef uint32_t u32;

namespace {

thread_local std::string g_err;

// ------------------------------------------------------------------ Rust float Display
template <typename F>
size_t fmt_float(F v, char *buf, size_t cap) {
    std::string s;
    if (std::isnan(v)) s = "NaN";
    else if (std::isinf(v)) s = v > 0 ? "inf" : "-inf";
    else if (v == 0) s = std::signbit(v) ? "-0" : "0";
    else {
        char tmp[512];
        auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::fixed);  // shortest round-trip
        s.assign(tmp, r.ptr);
    }
    if (cap) {
        size_t n = std::min(cap - 1, s.size());
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}
std::string f32s(float v) { char b[512]; fmt_float(v, b, sizeof b); return b; }
std::string f64s(double v) { char b[512]; fmt_float(v, b, sizeof b); return b; }
// the same digits appended in place (the printers' rows: no temporary strings)
template <typename F>
void put_float(std::string &o, F v) {
    if (std::isnan(v)) { o += "NaN"; return; }
    if (std::isinf(v)) { o += v > 0 ? "inf" : "-inf"; return; }
    if (v == 0) { o += std::signbit(v) ? "-0" : "0"; return; }
    char tmp[512];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::fixed);
    o.append(tmp, (size_t)(r.ptr - tmp));
}
// Rows of a table formatted on several threads when there are many (an assembly's table is millions of rows and several times as many
// numbers: 0.56 s of a 1.2 s run at 2 M contigs on one thread): `row(out, k)` appends row k to `out`; the pieces are joined in order.
template <typename Row>
void format_rows(std::string &o, size_t n_rows, size_t bytes_per_row, Row row) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t T = std::min<size_t>({(size_t)8, (size_t)hw, n_rows / 32768 + 1});
    if (T <= 1) { for (size_t k = 0; k < n_rows; k++) row(o, k); return; }
    const size_t n_parts = T * 4, per = (n_rows + n_parts - 1) / n_parts;
    std::vector<std::string> parts(n_parts);
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
        th.emplace_back([&] {
            for (;;) {
                const size_t p = next.fetch_add(1);
                if (p >= n_parts) break;
                const size_t lo = p * per, hi = std::min(n_rows, lo + per);
                if (hi > lo) parts[p].reserve((hi - lo) * bytes_per_row);
                for (size_t k = lo; k < hi; k++) row(parts[p], k);
            }
        });
    for (auto &x : th) x.join();
    size_t total = 0;
    for (auto &x : parts) total += x.size();
    o.reserve(o.size() + total);
    for (auto &x : parts) { o += x; std::string().swap(x); }
}

// Rust `x as usize` for f32: saturating, NaN -> 0
u64 f32_to_usize(float x) {
    if (!(x == x) || x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return std::numeric_limits<u64>::max();
    return (u64)x;
}

// ------------------------------------------------------------------ entry accumulator
// What the estimators of one entry (contig or genome) would hold after add_contig over its contigs
// (estimators.rs:366-528), expressed in the integer statistics the device returns.
struct EntryAcc {
    u64 win_len = 0, win_sum_d = 0, win_sum_d2 = 0, win_covered = 0;
    u64 full_len = 0, full_covered = 0, n_reads = 0, mismatches = 0;
    u32 win_min_d = 0xffffffffu;
    double sum_identity = 0.0;
    std::vector<u64> hist;  // merged counts; hist.size() == counts.len()
    // an entry made of ONE contig reads that contig's bins where they lie (no copy); a second contig materialises `hist`
    const u64 *hview = nullptr; size_t hview_n = 0;
    size_t hsize() const { return hview ? hview_n : hist.size(); }
    u64 hat(size_t i) const { return hview ? hview[i] : hist[i]; }
    void own_hist() { if (hview) { hist.assign(hview, hview + hview_n); hview = nullptr; hview_n = 0; } }
    void reset() {   // keeps hist's capacity: one reset per contig
        win_len = win_sum_d = win_sum_d2 = win_covered = full_len = full_covered = n_reads = mismatches = 0;
        win_min_d = 0xffffffffu; sum_identity = 0.0; hist.clear(); hview = nullptr; hview_n = 0; hist_len_only = 0;
    }
    void add_contig(const cov_contig_stats &s, u64 L, u64 excl, u64 n, double identity, const u64 *h) {
        n_reads += n;
        mismatches += s.sum_nm - s.sum_indel;   // total_edit_distance - total_indels (u64 wrapping, contig.rs:59)
        sum_identity += identity;
        full_len += L;
        full_covered += s.full_covered;
        if (2 * excl < L) {                      // estimators.rs:386-392, 436-445
            win_len += L - 2 * excl;
            win_sum_d += s.win_sum_d; win_sum_d2 += s.win_sum_d2; win_covered += s.win_covered;
            win_min_d = std::min(win_min_d, s.win_min_d);
            if (h && s.hist_len) {
                if (!hview && hist.empty()) { hview = h + s.hist_off; hview_n = s.hist_len; }
                else {
                    own_hist();
                    if (hist.size() < s.hist_len) hist.resize(s.hist_len, 0);
                    for (u32 d = 0; d < s.hist_len; d++) hist[d] += h[s.hist_off + d];
                }
            } else if (!h) {
                // no histogram requested: remember only how long counts would be
                if (hist_len_only < (u64)s.win_max_d + 1) hist_len_only = (u64)s.win_max_d + 1;
            }
        }
    }
    // same, from an explicit entry (unit tests)
    u64 hist_len_only = 0;
};

u64 unobserved_bases(const u64 *u, size_t n, u64 excl) {  // estimators.rs:226-242
    u64 s = 0, e = 2 * excl;
    for (size_t i = 0; i < n; i++) s += u[i] < e ? u[i] : u[i] - e;
    return s;
}
u64 sum_u64(const u64 *u, size_t n) { u64 s = 0; for (size_t i = 0; i < n; i++) s += u[i]; return s; }

// estimators.rs:530-839, every branch in the reference's operation order
float calculate(const covh_estimator &e, const EntryAcc &a, const u64 *unobs, size_t n_unobs) {
    const float minfrac = e.min_fraction_covered_bases;
    switch (e.kind) {
    case COVH_MEAN: {
        const u64 T = a.win_len + unobserved_bases(unobs, n_unobs, e.contig_end_exclusion);
        if (T == 0 || ((float)a.win_covered / (float)T) < minfrac) return 0.0f;
        const float num = e.exclude_mismatches ? (float)(a.win_sum_d - a.mismatches) : (float)a.win_sum_d;
        return num / (float)T;
    }
    case COVH_TRIMMED_MEAN: {
        const u64 U = unobserved_bases(unobs, n_unobs, e.contig_end_exclusion);
        const u64 T = a.win_len + U;
        if (T == 0) return 0.0f;
        if (((float)a.win_covered / (float)T) < minfrac) return 0.0f;
        const u64 min_index = f32_to_usize(std::floor(e.trim_min * (float)T));
        const u64 max_index = f32_to_usize(std::ceil(e.trim_max * (float)T));
        if (a.win_covered == 0) return 0.0f;
        u64 acc = 0, total = 0;
        bool started = false;
        for (size_t i = 0, nh = a.hsize(); i < nh; i++) {
            const u64 n = a.hat(i) + (i == 0 ? U : 0);  // counts[0] += unobserved (:596)
            acc += n;
            if (acc >= min_index) {
                if (started) {
                    if (acc > max_index) {
                        const u64 excess = acc - n;
                        total += (max_index >= excess ? max_index - excess + 1 : 0) * (u64)i;
                        break;
                    } else total += n * (u64)i;
                } else if (acc > max_index) { total = (max_index - min_index + 1) * (u64)i; started = true; }  // no break (:626-629)
                else if (acc < min_index) {}
                else { total = (acc - min_index + 1) * (u64)i; started = true; }
            }
        }
        return (float)total / (float)(max_index - min_index);
    }
    case COVH_PILEUP_COUNTS: {
        if (a.win_len == 0) return 0.0f;
        const u64 T = a.win_len + unobserved_bases(unobs, n_unobs, e.contig_end_exclusion);
        if (((float)a.win_covered / (float)T) < minfrac) return 0.0f;
        return (float)(T - a.win_covered + 1);
    }
    case COVH_COVERED_FRACTION: {
        const u64 T = a.full_len + sum_u64(unobs, n_unobs);
        if (T == 0 || ((float)a.full_covered / (float)T) < minfrac) return 0.0f;
        return (float)a.full_covered / (float)T;
    }
    case COVH_COVERED_BASES: {
        const u64 T = a.full_len + sum_u64(unobs, n_unobs);
        if (T == 0 || ((float)a.full_covered / (float)T) < minfrac) return 0.0f;
        return (float)a.full_covered;
    }
    case COVH_RPKM: {
        const u64 T = a.full_len + sum_u64(unobs, n_unobs);
        if (T == 0 || ((float)a.full_covered / (float)T) < minfrac) return 0.0f;
        return (float)(a.n_reads * 1000000000ull) / (float)T;
    }
    case COVH_TPM: {
        const u64 T = a.full_len + sum_u64(unobs, n_unobs);
        if (T == 0 || ((float)a.full_covered / (float)T) < minfrac) return 0.0f;
        return (float)std::exp(std::log((double)a.n_reads) - std::log((double)T));
    }
    case COVH_VARIANCE: {
        const u64 U = unobserved_bases(unobs, n_unobs, e.contig_end_exclusion);
        const u64 T = a.win_len + U;
        if (T == 0) return 0.0f;
        // counts.is_empty()  <=>  no contig contributed a window position  <=>  win_len == 0
        if (((float)a.win_covered / (float)T) < minfrac || T < 3 || a.win_len == 0) return 0.0f;
        // shifted sums: k = lowest occupied depth (0 as soon as unobserved bases land in counts[0]); the
        // reference's ex = sum (x-k) n and ex2 = sum (x-k)^2 n over the histogram are these integers
        const u64 k = U > 0 ? 0 : a.win_min_d;
        const u64 N = a.win_len + U;
        const u64 ex = a.win_sum_d - k * N;
        const u64 ex2 = a.win_sum_d2 - 2 * k * a.win_sum_d + k * k * N;
        return ((float)ex2 - (float)(ex * ex) / (float)T) / (float)(T - 1);
    }
    case COVH_LENGTH: return (float)(a.full_len + sum_u64(unobs, n_unobs));
    case COVH_READ_COUNT: return (float)a.n_reads;
    case COVH_READS_PER_BASE: return (float)a.n_reads / (float)(a.full_len + sum_u64(unobs, n_unobs));
    case COVH_ANIR: return a.n_reads == 0 ? 0.0f : (float)(a.sum_identity / (double)a.n_reads);
    }
    return 0.0f;
}

}  // namespace

// ------------------------------------------------------------------ takers (coverage_takers.rs)
struct covh_taker {
    int kind;
    size_t num_coverages;
    std::string text;
    std::string cur_stoit, cur_entry_name;
    // cached
    std::vector<std::string> stoit_names;
    // entry names live in one arena (a std::string per entry costs an allocation per contig per sample)
    struct NameRef { int64_t off = -1; uint32_t len = 0; explicit operator bool() const { return off >= 0; } };
    struct NameTable {
        std::string arena;
        std::vector<NameRef> refs;
        size_t size() const { return refs.size(); }
        void resize(size_t n) { refs.resize(n); }
        struct Entry {
            const NameTable *t; size_t i;
            explicit operator bool() const { return (bool)t->refs[i]; }
            std::string operator*() const { return std::string(t->arena.data() + t->refs[i].off, t->refs[i].len); }
            std::string_view view() const { return std::string_view(t->arena.data() + t->refs[i].off, t->refs[i].len); }
        };
        Entry operator[](size_t i) const { return Entry{this, i}; }
        void set(size_t i, std::string_view n) { refs[i].off = (int64_t)arena.size(); refs[i].len = (uint32_t)n.size(); arena.append(n); }
        struct Iter { const NameTable *t; size_t i; Entry operator*() const { return Entry{t, i}; } Iter &operator++() { ++i; return *this; } bool operator!=(const Iter &o) const { return i != o.i; } };
        Iter begin() const { return Iter{this, 0}; }
        Iter end() const { return Iter{this, refs.size()}; }
    } entry_names;
    std::vector<std::vector<std::pair<size_t, float>>> coverages;
    size_t cur_stoit_i = 0, cur_entry_i = 0;
    bool mismatch = false;

    void start_stoit(const std::string &n) {
        if (kind == COVH_TAKER_CACHED) { stoit_names.push_back(n); coverages.emplace_back(); cur_stoit_i = stoit_names.size() - 1; }
        else cur_stoit = n;
    }
    // capacity hint from the scan drivers: entries per sample, coverages per entry, bytes of all entry names
    void reserve(size_t entries, size_t per_entry, size_t name_bytes) {
        if (kind != COVH_TAKER_CACHED) { text.reserve(text.size() + entries * (per_entry * 12 + 24) + name_bytes); return; }
        if (!coverages.empty()) coverages.back().reserve(entries * per_entry);
        if (entry_names.refs.size() < entries) entry_names.refs.resize(entries);        // unset refs: ids arrive in any order, and not all of them
        if (entry_names.arena.capacity() < name_bytes) entry_names.arena.reserve(name_bytes);
    }
    // Cached taker, every target of a header an entry (zero rows printed): the name table in one pass.  True when the table now holds exactly
    // these names under ids 0 .. n - 1 (it was empty, or already held this very header); false: the caller goes entry by entry.
    bool names_bulk(const char *blob, const uint32_t *off, size_t n) {
        if (kind != COVH_TAKER_CACHED) return false;
        const size_t bytes = off[n] - off[0];
        if (entry_names.refs.empty() || (entry_names.arena.empty() && entry_names.refs.size() <= n)) {
            entry_names.arena.assign(blob + off[0], bytes);
            entry_names.refs.resize(n);
            for (size_t i = 0; i < n; i++) { entry_names.refs[i].off = (int64_t)(off[i] - off[0]); entry_names.refs[i].len = off[i + 1] - off[i]; }
            return true;
        }
        if (entry_names.refs.size() != n || entry_names.arena.size() != bytes || memcmp(entry_names.arena.data(), blob + off[0], bytes) != 0) return false;
        for (size_t i = 0; i < n; i++)
            if (entry_names.refs[i].off != (int64_t)(off[i] - off[0]) || entry_names.refs[i].len != off[i + 1] - off[i]) return false;
        return true;
    }
    void start_entry(size_t id, std::string_view name) {
        if (kind == COVH_TAKER_STREAM) { text += cur_stoit; text += '\t'; text.append(name); }
        else if (kind == COVH_TAKER_PILEUP) cur_entry_name.assign(name);
        else {
            if (id >= entry_names.size()) entry_names.resize(id + 1);
            if (!entry_names[id]) entry_names.set(id, name);
            else if (entry_names[id].view() != name) mismatch = true;  // coverage_takers.rs:140-148 (process::exit(1))
            cur_entry_i = id;
        }
    }
    void add_single_coverage(float c) {
        if (kind == COVH_TAKER_STREAM) { text += '\t'; text += (c == 0.0f) ? std::string("0") : f32s(c); }
        else if (kind == COVH_TAKER_CACHED) coverages[cur_stoit_i].emplace_back(cur_entry_i, c);
    }
    void add_coverage_entry(u64 num_reads, u64 num_bases) {
        if (kind == COVH_TAKER_PILEUP) {
            text += cur_stoit; text += '\t'; text += cur_entry_name; text += '\t';
            text += std::to_string(num_reads); text += '\t'; text += std::to_string(num_bases); text += '\n';
        }
    }
    void finish_entry() { if (kind == COVH_TAKER_STREAM) text += '\n'; }
};

namespace {

struct EntryAndCoverages { size_t entry_index, stoit_index; std::vector<float> coverages; };

// CoverageTakerTypeIterator, coverage_takers.rs:265-377
std::vector<EntryAndCoverages> iterate_cached(const covh_taker &t) {
    std::vector<EntryAndCoverages> out;
    const size_t ns = t.stoit_names.size(), nc = t.num_coverages;
    if (ns == 0) return out;
    std::vector<size_t> nxt(ns, 0);
    size_t cur = 0;
    std::optional<size_t> last;
    while (cur <= ns) {
        std::optional<size_t> lowest;
        for (size_t si = 0; si < ns; si++) {
            if (nxt[si] < t.coverages[si].size()) {
                const size_t ei = t.coverages[si][nxt[si]].first;
                if (!last || ei > *last) { if (!lowest || ei < *lowest) lowest = ei; }
            }
        }
        if (lowest) {
            const size_t chosen = nxt[cur];
            const auto &lst = t.coverages[cur];
            EntryAndCoverages e{*lowest, cur, {}};
            if (chosen >= lst.size() || lst[chosen].first != *lowest) e.coverages.assign(nc, 0.0f);
            else for (size_t k = 0; k < nc; k++) e.coverages.push_back(lst[chosen + k].second);
            for (size_t si = 0; si < ns; si++)
                if (t.coverages[si].size() > nxt[si] && t.coverages[si][nxt[si]].first == *lowest) nxt[si] += nc;
            last = *lowest;
            out.push_back(std::move(e));
        } else {
            cur++;
            if (cur >= ns) break;
            std::fill(nxt.begin(), nxt.end(), 0);
            last.reset();
        }
    }
    return out;
}

void print_coverage(const covh_estimator &e, const EntryAcc &a, float coverage, covh_taker &t) {  // estimators.rs:936-969
    if (e.kind != COVH_PILEUP_COUNTS) { t.add_single_coverage(coverage); return; }
    for (size_t i = 0, nh = a.hsize(); i < nh; i++) {
        u64 cov;
        if (i == 0) { const u64 c = f32_to_usize(std::floor(coverage)); cov = c == 0 ? 0 : c - 1; }
        else cov = a.hat(i);
        t.add_coverage_entry(i, cov);
    }
}
void print_zero_coverage(const covh_estimator &e, covh_taker &t, u64 entry_length) {  // estimators.rs:971-991
    if (e.kind == COVH_PILEUP_COUNTS) return;
    t.add_single_coverage(e.kind == COVH_LENGTH ? (float)entry_length : 0.0f);
}

std::string_view target_name(const covh_header *h, u32 tid) {
    return std::string_view(h->names + h->name_off[tid], h->name_off[tid + 1] - h->name_off[tid]);
}

bool check_excl(const covh_estimator *est, size_t n) {
    bool have = false; u64 x = 0;
    for (size_t i = 0; i < n; i++) {
        const int k = est[i].kind;
        if (k == COVH_MEAN || k == COVH_TRIMMED_MEAN || k == COVH_PILEUP_COUNTS || k == COVH_VARIANCE) {
            if (have && x != est[i].contig_end_exclusion) return false;
            have = true; x = est[i].contig_end_exclusion;
        }
    }
    return true;
}
u64 session_excl(const covh_estimator *est, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const int k = est[i].kind;
        if (k == COVH_MEAN || k == COVH_TRIMMED_MEAN || k == COVH_PILEUP_COUNTS || k == COVH_VARIANCE)
            return est[i].contig_end_exclusion;
    }
    return 0;
}

}  // namespace

extern "C" {

const char *covh_last_error(void) { return g_err.c_str(); }
size_t covh_format_f32(float v, char *buf, size_t cap) { return fmt_float(v, buf, cap); }
size_t covh_format_f64(double v, char *buf, size_t cap) { return fmt_float(v, buf, cap); }

covh_taker *covh_taker_new(int kind, size_t num_coverages) {
    covh_taker *t = new covh_taker();
    t->kind = kind; t->num_coverages = num_coverages;
    return t;
}
void covh_taker_free(covh_taker *t) { delete t; }
const char *covh_taker_text(const covh_taker *t, size_t *len) { if (len) *len = t->text.size(); return t->text.c_str(); }
void covh_taker_clear_text(covh_taker *t) { t->text.clear(); }
size_t covh_taker_cached_coverages(const covh_taker *t, size_t stoit, float *out, size_t cap) {
    if (t->kind != COVH_TAKER_CACHED || stoit >= t->coverages.size()) return 0;
    const auto &v = t->coverages[stoit];
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i].second;
    return v.size();
}

uint32_t covh_wants(const covh_estimator *est, size_t n_est) {
    uint32_t w = 0;
    for (size_t i = 0; i < n_est; i++) {
        if (est[i].kind == COVH_TRIMMED_MEAN || est[i].kind == COVH_PILEUP_COUNTS) w |= COV_WANT_HIST;
        if (est[i].kind == COVH_ANIR) w |= COV_WANT_IDENTITY;
    }
    return w;
}

// ---- trait MosdepthGenomeCoverageEstimator (estimators.rs:245-265) over device results: the stateful face a Rust
// `impl MosdepthGenomeCoverageEstimator for GpuEstimator` forwards to, one call per trait method.
struct covh_estimator_state {
    covh_estimator e;
    EntryAcc acc;
    u64 num_mapped_reads = 0;     // what num_mapped_reads() reports (estimators.rs:993-1010)
};
covh_estimator_state *covh_estimator_new(const covh_estimator *e) {
    if (!e) return nullptr;
    covh_estimator_state *s = new covh_estimator_state();
    s->e = *e; s->acc.reset();
    return s;
}
void covh_estimator_free(covh_estimator_state *s) { delete s; }
void covh_estimator_setup(covh_estimator_state *s) { s->acc.reset(); s->num_mapped_reads = 0; }          /* fn setup(&mut self) */
/* fn add_contig(&mut self, ups_and_downs, num_mapped_reads, total_mismatches, sum_identity): the delta array is replaced by the
 * contig's integer statistics from cov_finish (+ its histogram slice); num_mapped_reads / sum_identity are the values the scan
 * loop would have passed (n_primary / n_pass / n_nonsupp and the matching identity sum, depending on the loop). */
void covh_estimator_add_contig_stats(covh_estimator_state *s, const cov_contig_stats *stats, uint64_t target_len, const uint64_t *hist,
                                     uint64_t num_mapped_reads, double sum_identity) {
    s->acc.add_contig(*stats, target_len, s->e.contig_end_exclusion, num_mapped_reads, sum_identity, hist);
    s->acc.own_hist();   // exported path: the caller's `hist` need not outlive this call (the in-tree scan drivers keep the zero-copy view)
    // Mean/ReadCount/... accumulate, the histogram family assigns (estimators.rs:434): EntryAcc keeps the sum, the assignment
    // only matters to num_mapped_reads()
    if (s->e.kind == COVH_TRIMMED_MEAN || s->e.kind == COVH_PILEUP_COUNTS || s->e.kind == COVH_VARIANCE) s->num_mapped_reads = num_mapped_reads;
    else s->num_mapped_reads += num_mapped_reads;
}
float covh_estimator_calculate_coverage(covh_estimator_state *s, const uint64_t *unobserved_contig_lengths, size_t n) {
    return calculate(s->e, s->acc, unobserved_contig_lengths, n);
}
void covh_estimator_print_coverage(const covh_estimator_state *s, float coverage, covh_taker *t) { print_coverage(s->e, s->acc, coverage, *t); }
void covh_estimator_print_zero_coverage(const covh_estimator_state *s, covh_taker *t, uint64_t entry_length) { print_zero_coverage(s->e, *t, entry_length); }
covh_estimator_state *covh_estimator_copy(const covh_estimator_state *s) { return covh_estimator_new(&s->e); }   /* fn copy(&self): fresh state, same parameters */
uint64_t covh_estimator_num_mapped_reads(const covh_estimator_state *s) { return s->num_mapped_reads; }

float covh_calculate_coverage(const covh_estimator *e, const covh_entry *en, const uint64_t *unobs, size_t n) {
    EntryAcc a;
    a.win_len = en->win_len; a.win_sum_d = en->win_sum_d; a.win_sum_d2 = en->win_sum_d2; a.win_covered = en->win_covered;
    a.full_len = en->full_len; a.full_covered = en->full_covered; a.n_reads = en->n_reads; a.mismatches = en->mismatches;
    a.win_min_d = en->win_min_d; a.sum_identity = en->sum_identity;
    if (en->hist) a.hist.assign(en->hist, en->hist + en->hist_len);
    return calculate(*e, a, unobs, n);
}

// ------------------------------------------------------------------ contig.rs:13-253
extern "C++" {
namespace {
// A few persistent workers for loops over contigs whose iterations are independent (the float expressions of calculate_coverage
// are pure functions of one contig's integer statistics): 5 000 contigs x 4 estimators are 0.33 ms on one thread, a third of the
// kernels' time at BASELINE config 2.  Threads are started on first use and sleep on a condition variable in between.
class ContigWorkers {
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void(size_t, size_t)> fn_;
    size_t n_ = 0, parts_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    int busy_ = 0;
    bool stop_ = false;
    void drain() {
        for (;;) {
            const size_t p = next_.fetch_add(1);
            if (p >= parts_) break;
            fn_(n_ * p / parts_, n_ * (p + 1) / parts_);
        }
    }
    void worker(uint64_t seen) {        // seen = the generation current when the thread was created: a thread added later must not replay it
        for (;;) {
            { std::unique_lock<std::mutex> lk(m_); cv_.wait(lk, [&] { return stop_ || gen_ != seen; }); if (stop_) return; seen = gen_; }
            drain();
            { std::lock_guard<std::mutex> lk(m_); if (--busy_ == 0) done_.notify_all(); }
        }
    }
public:
    ~ContigWorkers() { { std::lock_guard<std::mutex> lk(m_); stop_ = true; } cv_.notify_all(); for (auto &t : th_) t.join(); }
    // fn(lo, hi) over [0, n) in `parts` ranges; the caller works too.  One run at a time (run_m_ serialises callers).
    std::mutex run_m_;
    void run(size_t n, std::function<void(size_t, size_t)> fn) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        size_t want = std::min<size_t>({(size_t)4, (size_t)hw, n / 512 + 1});
        { long long v; if (covknob::get("finalise_threads", v)) want = (size_t)std::max<long long>(1, v); }
        if (want <= 1) { fn(0, n); return; }
        std::lock_guard<std::mutex> rl(run_m_);
        while (th_.size() + 1 < want) { uint64_t g; { std::lock_guard<std::mutex> lk(m_); g = gen_; } th_.emplace_back([this, g] { worker(g); }); }
        { std::lock_guard<std::mutex> lk(m_); fn_ = std::move(fn); n_ = n; parts_ = want * 2; next_ = 0; busy_ = (int)th_.size(); gen_++; }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return busy_ == 0; });
    }
};
ContigWorkers &contig_workers() { static ContigWorkers w; return w; }
}  // namespace
}  // extern "C++"

int covh_contig_coverage(const covh_header *h, const covh_sample *samples, size_t n_samples, covh_taker *taker,
                         const covh_estimator *est, size_t n_est, int print_zero, covh_reads_mapped *rm_out) {
    return covh_contig_coverage_estimated(h, samples, n_samples, taker, est, n_est, print_zero, rm_out, nullptr);
}

// The same scan loop with calculate_coverage already done on the device for some or all samples (cov_set_estimators /
// cov_fetch_estimates: the same floats, bit for bit): what is left here is contig.rs:40-104's control flow — zero rows, ReadsMapped,
// the taker calls — in target order.
int covh_contig_coverage_estimated(const covh_header *h, const covh_sample *samples, size_t n_samples, covh_taker *taker,
                                   const covh_estimator *est, size_t n_est, int print_zero, covh_reads_mapped *rm_out,
                                   const float *const *estimates) {
    if (!check_excl(est, n_est)) { g_err = "estimators disagree on contig_end_exclusion"; return COV_ERR_INVALID_ARG; }
    if (estimates)
        for (size_t k = 0; k < n_est; k++)
            if (est[k].kind == COVH_PILEUP_COUNTS || est[k].kind == COVH_TPM) { g_err = "covh_contig_coverage_estimated: coverage histogram and TPM are evaluated on the host"; return COV_ERR_INVALID_ARG; }
    const u64 excl = session_excl(est, n_est);
    const u64 zero = 0;
    std::vector<float> cov(n_est);
    bool needs_acc_to_print = false;       // coverage_histogram prints the histogram itself: keeps the one-pass loop
    for (size_t k = 0; k < n_est; k++) needs_acc_to_print |= est[k].kind == COVH_PILEUP_COUNTS;
    std::vector<float> all;
    for (size_t si = 0; si < n_samples; si++) {
        const covh_sample &S = samples[si];
        taker->start_stoit(S.stoit_name);
        taker->reserve(h->n_targets, n_est, h->name_off[h->n_targets]);
        u64 mapped_total = 0;
        int64_t prev = -1;
        const float *ext = estimates ? estimates[si] : nullptr;
        if (ext || (!needs_acc_to_print && h->n_targets >= 1024)) {
            // phase 1 (contigs in parallel, or done on the device): the coverages; phase 2 (in order): zero rows, reads mapped, the taker
            if (!ext) all.resize((size_t)h->n_targets * n_est);
            if (!ext) contig_workers().run(h->n_targets, [&](size_t lo, size_t hi) {
                EntryAcc acc;
                for (size_t t = lo; t < hi; t++) {
                    const cov_contig_stats &s = S.stats[t];
                    if (s.n_pass == 0) continue;
                    acc.reset();
                    acc.add_contig(s, h->target_len[t], excl, s.n_primary, s.sum_identity_primary, S.hist);
                    for (size_t k = 0; k < n_est; k++) all[t * n_est + k] = calculate(est[k], acc, &zero, 1);
                }
            });
            auto zero_rows2 = [&](int64_t from, int64_t to) {
                for (int64_t t = from + 1; t < to; t++) {
                    taker->start_entry((size_t)t, target_name(h, (u32)t));
                    for (size_t k = 0; k < n_est; k++) print_zero_coverage(est[k], *taker, h->target_len[t]);
                    taker->finish_entry();
                }
            };
            if (print_zero && taker->names_bulk(h->names, h->name_off, h->n_targets)) {
                // every target is an entry of a cached taker: the (entry, coverage) pairs are written in place — the same pairs in the same
                // order as the calls below would append (zero rows through print_zero_coverage's values: the length for COVH_LENGTH, else 0)
                auto &v = taker->coverages[taker->cur_stoit_i];
                const size_t base = v.size();
                v.resize(base + (size_t)h->n_targets * n_est);
                std::pair<size_t, float> *o = v.data() + base;
                for (u32 t = 0; t < h->n_targets; t++, o += n_est) {
                    const cov_contig_stats &s = S.stats[t];
                    if (s.n_pass == 0) {
                        for (size_t k = 0; k < n_est; k++) o[k] = {t, est[k].kind == COVH_LENGTH ? (float)h->target_len[t] : 0.0f};
                        continue;
                    }
                    const float *c = ext ? ext + (size_t)t * n_est : &all[(size_t)t * n_est];
                    bool nonzero = false;
                    for (size_t k = 0; k < n_est; k++) { nonzero |= c[k] > 0.0f; o[k] = {t, c[k]}; }
                    if (nonzero) mapped_total += s.n_primary;           // :67-72
                }
                taker->cur_entry_i = h->n_targets ? h->n_targets - 1 : 0;
                if (rm_out) { rm_out[si].num_mapped_reads = mapped_total; rm_out[si].num_reads = S.num_detected_primary_alignments; }
                continue;
            }
            for (u32 t = 0; t < h->n_targets; t++) {
                const cov_contig_stats &s = S.stats[t];
                if (s.n_pass == 0) continue;
                if (print_zero) zero_rows2(prev, t);
                const float *c = ext ? ext + (size_t)t * n_est : &all[(size_t)t * n_est];
                bool nonzero = false;
                for (size_t k = 0; k < n_est; k++) nonzero |= c[k] > 0.0f;
                if (nonzero) mapped_total += s.n_primary;           // :67-72
                if (print_zero || nonzero) {
                    taker->start_entry(t, target_name(h, t));
                    for (size_t k = 0; k < n_est; k++) taker->add_single_coverage(c[k]);
                    taker->finish_entry();
                }
                prev = t;
            }
            if (print_zero) zero_rows2(prev, h->n_targets);
            if (rm_out) { rm_out[si].num_mapped_reads = mapped_total; rm_out[si].num_reads = S.num_detected_primary_alignments; }
            continue;
        }
        auto zero_rows = [&](int64_t from, int64_t to) {   // print_previous_zero_coverage_contigs, :255-277
            for (int64_t t = from + 1; t < to; t++) {
                taker->start_entry((size_t)t, target_name(h, (u32)t));
                for (size_t k = 0; k < n_est; k++) print_zero_coverage(est[k], *taker, h->target_len[t]);
                taker->finish_entry();
            }
        };
        EntryAcc acc;
        for (u32 t = 0; t < h->n_targets; t++) {
            const cov_contig_stats &s = S.stats[t];
            if (s.n_pass == 0) continue;   // no considered record: the scan never allocated this contig
            if (print_zero) zero_rows(prev, t);
            acc.reset();
            acc.add_contig(s, h->target_len[t], excl, s.n_primary, s.sum_identity_primary, S.hist);
            bool nonzero = false;
            for (size_t k = 0; k < n_est; k++) { cov[k] = calculate(est[k], acc, &zero, 1); nonzero |= cov[k] > 0.0f; }
            if (nonzero) mapped_total += s.n_primary;           // :67-72
            if (print_zero || nonzero) {
                taker->start_entry(t, target_name(h, t));
                for (size_t k = 0; k < n_est; k++) print_coverage(est[k], acc, cov[k], *taker);
                taker->finish_entry();
            }
            prev = t;
        }
        if (print_zero) zero_rows(prev, h->n_targets);
        if (rm_out) { rm_out[si].num_mapped_reads = mapped_total; rm_out[si].num_reads = S.num_detected_primary_alignments; }
    }
    return COV_OK;
}

// ------------------------------------------------------------------ genome.rs:17-322
int covh_genome_coverage_with_contig_names(const covh_header *h, const covh_sample *samples, size_t n_samples,
                                           const int32_t *genome_of_tid, const char *const *genome_names,
                                           size_t n_genomes, covh_taker *taker, int print_zero,
                                           const covh_estimator *est, size_t n_est, covh_reads_mapped *rm_out) {
    if (!check_excl(est, n_est)) { g_err = "estimators disagree on contig_end_exclusion"; return COV_ERR_INVALID_ARG; }
    const u64 excl = session_excl(est, n_est);
    std::vector<std::vector<u32>> refs(n_genomes);
    for (u32 t = 0; t < h->n_targets; t++) if (genome_of_tid[t] >= 0) refs[genome_of_tid[t]].push_back(t);
    std::vector<float> cov(n_est);
    for (size_t si = 0; si < n_samples; si++) {
        const covh_sample &S = samples[si];
        taker->start_stoit(S.stoit_name);
        std::vector<EntryAcc> acc(n_genomes);
        std::vector<u64> reads_in_genome(n_genomes, 0);
        bool any_seen = false;
        for (u32 t = 0; t < h->n_targets; t++) {
            const cov_contig_stats &s = S.stats[t];
            if (s.n_pass == 0) continue;
            any_seen = true;
            const int32_t g = genome_of_tid[t];
            if (g < 0) continue;                               // :170-171
            reads_in_genome[g] += s.n_pass;                    // :173-174 every record passing the flag filter
            acc[g].add_contig(s, h->target_len[t], excl, s.n_pass, s.sum_identity_nonsupp, S.hist);
        }
        u64 mapped_total = 0;
        if (!any_seen && S.num_detected_primary_alignments == 0) {
            // warn only (:230-234)
        } else {
            std::vector<u64> unobs;
            for (size_t gi = 0; gi < n_genomes; gi++) {
                unobs.clear();
                u64 genome_len = 0;
                for (u32 t : refs[gi]) { genome_len += h->target_len[t]; if (S.stats[t].n_pass == 0) unobs.push_back(h->target_len[t]); }
                bool nonzero = false;
                for (size_t k = 0; k < n_est; k++) { cov[k] = calculate(est[k], acc[gi], unobs.data(), unobs.size()); nonzero |= cov[k] > 0.0f; }
                if (nonzero) mapped_total += reads_in_genome[gi];
                if (print_zero || nonzero) {
                    taker->start_entry(gi, genome_names[gi]);
                    for (size_t k = 0; k < n_est; k++) {
                        if (cov[k] > 0.0f) print_coverage(est[k], acc[gi], cov[k], *taker);
                        else print_zero_coverage(est[k], *taker, genome_len);
                    }
                    taker->finish_entry();
                }
            }
        }
        if (rm_out) { rm_out[si].num_mapped_reads = mapped_total; rm_out[si].num_reads = S.num_detected_primary_alignments; }
    }
    return COV_OK;
}

// ------------------------------------------------------------------ genome.rs:419-929
namespace {
struct SepCtx {
    const covh_header *h; uint8_t split; bool single; bool err = false;
    std::string_view genome(u32 tid) {   // extract_genome, :799-805
        std::string_view n = target_name(h, tid);
        size_t p = n.find((char)split);
        if (p == std::string_view::npos) { err = true; return n; }
        return n.substr(0, p);
    }
};
struct Unobs { std::vector<u64> v; size_t first_tid = 0; };

void fill_backwards(SepCtx &c, u32 current_tid, std::string_view target, Unobs &u) {   // :807-853
    u.v.clear();
    if (current_tid == 0) { u.first_tid = 0; return; }
    u32 my = current_tid - 1;
    while (c.single || c.genome(my) == target) {
        u.v.push_back(c.h->target_len[my]);
        if (my == 0) { u.first_tid = 0; return; }
        my--;
    }
    u.first_tid = (size_t)my + 1;
}
void fill_backwards_to_last(SepCtx &c, u32 current_tid, u32 last_tid, std::string_view target, Unobs &u) {  // :477-499
    if (current_tid == 0) return;
    for (u32 my = last_tid + 1; my < current_tid; my++) {
        if (c.single || c.genome(my) == target) u.v.push_back(c.h->target_len[my]); else break;
    }
}
void fill_forwards(SepCtx &c, u32 current_tid, const std::optional<std::string_view> &target, Unobs &u) {   // :448-475
    if (!target) return;
    for (u32 my = current_tid + 1; my < c.h->n_targets; my++) {
        if (c.single || c.genome(my) == *target) u.v.push_back(c.h->target_len[my]); else break;
    }
}
// print_previous_zero_coverage_genomes2, :859-929
void zero_genomes2(SepCtx &c, const std::optional<std::string_view> &last_genome, std::string_view current_genome,
                   u32 current_tid, const covh_estimator *est, size_t n_est, covh_taker &t) {
    std::string_view my_current = current_genome;
    u32 tid = current_tid;
    std::vector<std::string_view> names; std::vector<size_t> first_tids; std::vector<u64> lens;
    u64 unobserved = 0;
    std::optional<u32> last_first;
    for (;;) {
        std::string_view g = c.genome(tid);
        if (last_genome && g == *last_genome) break;
        else if (g != my_current) {
            if (last_first) {
                if (!last_genome || g != *last_genome) { first_tids.push_back(*last_first); names.push_back(my_current); lens.push_back(unobserved); }
            }
            my_current = g; last_first = tid; unobserved = c.h->target_len[tid];
        } else if (g != current_genome) { last_first = tid; unobserved += c.h->target_len[tid]; }
        if (tid == 0) break;
        tid--;
    }
    if (last_first) { first_tids.push_back(*last_first); names.push_back(my_current); lens.push_back(unobserved); }
    for (size_t i = names.size(); i-- > 0;) {
        t.start_entry(first_tids[i], names[i]);
        for (size_t k = 0; k < n_est; k++) print_zero_coverage(est[k], t, lens[i]);
        t.finish_entry();
    }
}
}  // namespace

int covh_genome_coverage_separator(const covh_header *h, const covh_sample *samples, size_t n_samples,
                                   uint8_t split_char, covh_taker *taker, int print_zero,
                                   const covh_estimator *est, size_t n_est, int single_genome,
                                   covh_reads_mapped *rm_out) {
    if (!check_excl(est, n_est)) { g_err = "estimators disagree on contig_end_exclusion"; return COV_ERR_INVALID_ARG; }
    const u64 excl = session_excl(est, n_est);
    std::vector<float> cov(n_est);
    for (size_t si = 0; si < n_samples; si++) {
        const covh_sample &S = samples[si];
        taker->start_stoit(S.stoit_name);
        SepCtx cx{h, split_char, single_genome != 0};
        EntryAcc acc;
        Unobs u;
        bool doing_first = true;
        u32 last_tid = 0;
        std::optional<std::string_view> last_genome;
        u64 mapped_total = 0, n_in_genome = 0;
        auto add_last_contig = [&]() {
            // estimator.add_contig(&ups_and_downs, ...) for the contig the scan was in (last_tid); before the
            // first mapped record ups_and_downs is an empty Vec, which adds nothing
            if (doing_first) return;
            const cov_contig_stats &s = S.stats[last_tid];
            acc.add_contig(s, h->target_len[last_tid], excl, s.n_nonsupp, s.sum_identity_primary, S.hist);
        };
        // print_last_genomes, :331-416
        auto print_last_genomes = [&](std::string_view current_genome, u32 tid_to_print_zeros_to, bool had_contig) -> bool {
            if (had_contig) add_last_contig();
            bool positive = false;
            for (size_t k = 0; k < n_est; k++) { cov[k] = calculate(est[k], acc, u.v.data(), u.v.size()); positive |= cov[k] > 0.0f; }
            if ((print_zero || positive) && last_genome) {
                taker->start_entry(u.first_tid, *last_genome);
                for (size_t k = 0; k < n_est; k++) {
                    if (cov[k] > 0.0f) print_coverage(est[k], acc, cov[k], *taker);
                    else print_zero_coverage(est[k], *taker, 9);
                }
                taker->finish_entry();
            }
            acc.reset();
            if (print_zero && !cx.single) zero_genomes2(cx, last_genome, current_genome, tid_to_print_zeros_to, est, n_est, *taker);
            return positive;
        };
        for (u32 tid = 0; tid < h->n_targets; tid++) {
            const cov_contig_stats &s = S.stats[tid];
            if (s.n_pass == 0) continue;
            std::string_view current_genome = cx.single ? std::string_view() : cx.genome(tid);
            if (cx.err) { g_err = "Contig name does not contain split symbol, so cannot determine which genome it belongs to"; return COV_ERR_INVALID_ARG; }
            if (doing_first) {
                acc.reset();
                fill_backwards(cx, tid, current_genome, u);
                last_genome = current_genome;
                if (print_zero && !cx.single) zero_genomes2(cx, std::nullopt, current_genome, tid, est, n_est, *taker);
                doing_first = false;
            } else if (current_genome == *last_genome) {
                add_last_contig();
                fill_backwards_to_last(cx, tid, last_tid, current_genome, u);
            } else {
                fill_backwards_to_last(cx, tid, last_tid, *last_genome, u);
                if (print_last_genomes(current_genome, tid, true)) mapped_total += n_in_genome;
                n_in_genome = 0;
                last_genome = current_genome;
                fill_backwards(cx, tid, current_genome, u);
            }
            if (cx.err) { g_err = "Contig name does not contain split symbol, so cannot determine which genome it belongs to"; return COV_ERR_INVALID_ARG; }
            last_tid = tid;
            n_in_genome += s.n_nonsupp;    // :677-682
        }
        if (doing_first && S.num_detected_primary_alignments == 0) {
            // warn only (:731-735)
        } else {
            static const char g1[] = "genome1";
            const bool had_contig = !doing_first;
            if (cx.single) last_genome = std::string_view(g1, 7);   // :739-741
            fill_forwards(cx, last_tid, last_genome, u);
            if (print_last_genomes(std::string_view(), h->n_targets - 1, had_contig)) mapped_total += n_in_genome;
            if (cx.err) { g_err = "Contig name does not contain split symbol, so cannot determine which genome it belongs to"; return COV_ERR_INVALID_ARG; }
        }
        if (rm_out) { rm_out[si].num_mapped_reads = mapped_total; rm_out[si].num_reads = S.num_detected_primary_alignments; }
    }
    return COV_OK;
}

// ------------------------------------------------------------------ coverage_printer.rs
void covh_print_headers(covh_taker *t, int printer, const char *entry_type, const char *const *headers, size_t n) {
    if (printer == 0 || printer == 1) {   // :130-137
        t->text += "Sample\t"; t->text += entry_type;
        for (size_t i = 0; i < n; i++) { t->text += '\t'; t->text += headers[i]; }
        t->text += '\n';
    }
}

namespace {
bool contains(const int64_t *v, size_t n, size_t x) { for (size_t i = 0; i < n; i++) if ((size_t)v[i] == x) return true; return false; }
static std::string rstrip_cr(const std::string &s) { size_t n = s.size(); while (n && s[n - 1] == '\r') n--; return s.substr(0, n); }
static void put_name(std::string &o, std::string_view v) { size_t n = v.size(); while (n && v[n - 1] == '\r') n--; o.append(v.data(), n); }
double round4(double v) { return std::round(v) / 10000.0; }  // (x * 10000.0).round() / 10000.0 with x*10000 passed in

void print_sparse(covh_taker &t, const covh_reads_mapped *rm, const int64_t *norm, size_t n_norm, int64_t rpkm_col, int64_t tpm_col) {
    const size_t nc = t.num_coverages;
    size_t extra_cols = 0;
    for (auto n : t.entry_names) if (n) { const std::string_view v = n.view(); extra_cols = std::count(v.begin(), v.end(), '\t'); break; }
    auto all = iterate_cached(t);
    std::string &o = t.text;
    auto print_previous = [&](const std::vector<const EntryAndCoverages *> &rows, size_t si) {
        std::vector<std::optional<float>> mult(nc), totals(nc);
        for (size_t k = 0; k < n_norm; k++) {
            const size_t i = (size_t)norm[k];
            float tot = 0.0f;
            for (auto r : rows) tot += r->coverages[i];
            totals[i] = tot;
            if (rm) mult[i] = (float)rm[si].num_mapped_reads / (float)rm[si].num_reads;
        }
        if (tpm_col >= 0) { float tot = 0.0f; for (auto r : rows) tot += r->coverages[tpm_col]; totals[tpm_col] = tot; }
        const std::string &stoit = t.stoit_names[si];
        if (n_norm) {
            o += stoit; o += "\tunmapped"; o.append(extra_cols, '\t');
            for (size_t k = 0; k < n_norm; k++) {
                const size_t col = (size_t)norm[k];
                const size_t lo = k == 0 ? 0 : (size_t)norm[k - 1] + 1;
                for (size_t j = lo; j < col; j++) o += "\tNA";
                o += '\t'; o += f32s(100.0f * (1.0f - *mult[col]));
            }
            for (size_t j = (size_t)norm[n_norm - 1] + 1; j < nc; j++) o += "\tNA";
            o += '\n';
        }
        format_rows(o, rows.size(), stoit.size() + 24 + 12 * nc, [&](std::string &o, size_t k) {
            const EntryAndCoverages *r = rows[k];
            o += stoit; o += '\t'; put_name(o, t.entry_names[r->entry_index].view());
            for (size_t i = 0; i < nc; i++) {
                o += '\t';
                const float c = r->coverages[i];
                if (contains(norm, n_norm, i)) put_float(o, c * 100.0f * *mult[i] / *totals[i]);          // :285-286
                else if (rpkm_col == (int64_t)i) { const u64 n = rm[si].num_mapped_reads; put_float(o, n == 0 ? 0.0f : c / (float)n); }
                else if (tpm_col == (int64_t)i) {
                    const u64 n = rm[si].num_mapped_reads;
                    if (n == 0) put_float(o, 0.0);
                    else put_float(o, (double)std::exp(std::log(c) - std::log(*totals[i])) * (double)1000000);   // :320-323
                } else put_float(o, c);
            }
            o += '\n';
        });
    };
    std::vector<const EntryAndCoverages *> rows;
    size_t cur = 0;
    for (auto &e : all) {
        if (cur != e.stoit_index) { print_previous(rows, cur); rows.clear(); cur = e.stoit_index; }
        rows.push_back(&e);
    }
    if (!t.stoit_names.empty()) print_previous(rows, cur);
}

void print_dense(covh_taker &t, const char *entry_type, const char *const *headers, size_t n_headers, const covh_reads_mapped *rm,
                 size_t n_samples, const int64_t *norm, size_t n_norm, int64_t rpkm_col, int64_t tpm_col) {
    const size_t nc = t.num_coverages;
    std::string &o = t.text;
    o += entry_type;
    for (auto &s : t.stoit_names) for (size_t i = 0; i < n_headers; i++) { o += '\t'; o += s; o += ' '; o += headers[i]; }
    o += '\n';
    std::vector<float> mult;
    if (rm) for (size_t i = 0; i < n_samples; i++) mult.push_back((float)rm[i].num_mapped_reads / (float)rm[i].num_reads);
    if (n_norm) {
        o += "unmapped";
        o.append(std::count(entry_type, entry_type + strlen(entry_type), '\t'), '\t');
        for (size_t si = 0; si < t.stoit_names.size(); si++) {
            for (size_t k = 0; k < n_norm; k++) {
                const size_t col = (size_t)norm[k];
                const size_t lo = k == 0 ? 0 : (size_t)norm[k - 1] + 1;
                for (size_t j = lo; j < col; j++) o += "\tNA";
                o += '\t'; o += f32s(100.0f * (1.0f - mult[si]));
            }
            for (size_t j = (size_t)norm[n_norm - 1] + 1; j < nc; j++) o += "\tNA";
        }
        o += '\n';
    }
    {   // The usual table — every sample lists the same entries, in increasing order (one entry per reference with zero rows printed, or
        // the same mapped references in every sample) — read where the taker keeps it: CoverageTakerTypeIterator's merge
        // (iterate_cached below) yields exactly these rows in this order, at the cost of a small vector per row.
        const size_t ns = t.stoit_names.size();
        bool plain = ns > 0 && nc > 0;
        const size_t len0 = plain ? t.coverages[0].size() : 0;
        plain = plain && len0 % nc == 0;
        for (size_t si = 1; plain && si < ns; si++) plain = t.coverages[si].size() == len0;
        const size_t n_rows = plain ? len0 / nc : 0;
        for (size_t k = 0; plain && k < n_rows; k++) {
            const size_t e = t.coverages[0][k * nc].first;
            if (k && e <= t.coverages[0][(k - 1) * nc].first) plain = false;
            for (size_t si = 1; plain && si < ns; si++) plain = t.coverages[si][k * nc].first == e;
        }
        long long use_plain = 1;
        (void)covknob::get("printer_plain", use_plain);      // tests: 0 = always through the merge below
        if (plain && n_rows >= 4096 && use_plain) {
            std::vector<std::vector<std::optional<float>>> totals(ns, std::vector<std::optional<float>>(nc));
            for (size_t si = 0; si < ns; si++) {       // the same additions in the same order as the merge's (entry by entry, sample by sample: a total is per sample)
                auto addt = [&](size_t i) { auto &x = totals[si][i]; for (size_t k = 0; k < n_rows; k++) { const float c = t.coverages[si][k * nc + i].second; x = x ? *x + c : c; } };
                for (size_t k = 0; k < n_norm; k++) addt((size_t)norm[k]);
                if (tpm_col >= 0) addt((size_t)tpm_col);
            }
            format_rows(o, n_rows, 24 + 12 * nc * ns, [&](std::string &o, size_t k) {
                put_name(o, t.entry_names[t.coverages[0][k * nc].first].view());
                for (size_t si = 0; si < ns; si++)
                    for (size_t i = 0; i < nc; i++) {
                        o += '\t';
                        const float c = t.coverages[si][k * nc + i].second;
                        if (contains(norm, n_norm, i)) put_float(o, c / *totals[si][i] * 100.0f * mult[si]);   // :496-502
                        else if (rpkm_col == (int64_t)i) { const u64 n = rm[si].num_mapped_reads; put_float(o, n == 0 ? 0.0f : c / (float)n); }
                        else if (tpm_col == (int64_t)i) {
                            const u64 n = rm[si].num_mapped_reads;
                            put_float(o, n == 0 ? 0.0f : std::exp(std::log(c) - std::log(*totals[si][i])) * (float)1000000);  // :536-539
                        } else put_float(o, c);
                    }
                o += '\n';
            });
            return;
        }
    }
    auto all = iterate_cached(t);
    std::vector<std::vector<std::optional<float>>> totals(t.stoit_names.size(), std::vector<std::optional<float>>(nc));
    std::vector<std::vector<const EntryAndCoverages *>> by_stoit;
    for (auto &e : all) {
        auto addt = [&](size_t i) { auto &x = totals[e.stoit_index][i]; x = x ? *x + e.coverages[i] : e.coverages[i]; };
        for (size_t k = 0; k < n_norm; k++) addt((size_t)norm[k]);
        if (tpm_col >= 0) addt((size_t)tpm_col);
        if (by_stoit.size() <= e.stoit_index) by_stoit.emplace_back();
        by_stoit[e.stoit_index].push_back(&e);
    }
    if (by_stoit.empty()) return;
    for (size_t k = 0; k < by_stoit[0].size(); k++) {
        o += rstrip_cr(*t.entry_names[by_stoit[0][k]->entry_index]);
        for (size_t si = 0; si < by_stoit.size(); si++) {
            const EntryAndCoverages *e = by_stoit[si][k];
            for (size_t i = 0; i < e->coverages.size(); i++) {
                o += '\t';
                const float c = e->coverages[i];
                if (contains(norm, n_norm, i)) o += f32s(c / *totals[e->stoit_index][i] * 100.0f * mult[si]);   // :496-502
                else if (rpkm_col == (int64_t)i) { const u64 n = rm[si].num_mapped_reads; o += f32s(n == 0 ? 0.0f : c / (float)n); }
                else if (tpm_col == (int64_t)i) {
                    const u64 n = rm[si].num_mapped_reads;
                    o += f32s(n == 0 ? 0.0f : std::exp(std::log(c) - std::log(*totals[e->stoit_index][i])) * (float)1000000);  // :536-539
                } else o += f32s(c);
            }
        }
        o += '\n';
    }
}

void print_metabat(covh_taker &t) {   // coverage_printer.rs:57-119
    std::string &o = t.text;
    o += "contigName\tcontigLen\ttotalAvgDepth";
    for (auto &s : t.stoit_names) { o += '\t'; o += s; o += ".bam\t"; o += s; o += ".bam-var"; }
    o += '\n';
    auto all = iterate_cached(t);
    std::vector<std::vector<const EntryAndCoverages *>> by_stoit;
    for (auto &e : all) { if (by_stoit.size() <= e.stoit_index) by_stoit.emplace_back(); by_stoit[e.stoit_index].push_back(&e); }
    if (by_stoit.empty()) return;
    for (size_t k = 0; k < by_stoit[0].size(); k++) {
        float total = 0.0f;
        for (auto &st : by_stoit) total += st[k]->coverages[1];
        o += *t.entry_names[k]; o += '\t'; o += f32s(by_stoit[0][k]->coverages[0]); o += '\t';
        o += f64s(round4((double)total * 10000.0 / (double)t.coverages.size()));
        for (auto &st : by_stoit) {
            o += '\t'; o += f64s(round4((double)st[k]->coverages[1] * 10000.0));
            o += '\t'; o += f64s(round4((double)st[k]->coverages[2] * 10000.0));
        }
        o += '\n';
    }
}
}  // namespace

// ---- trait CoverageTaker (coverage_takers.rs:29-38) as plain calls, for hosts that drive a taker themselves
void covh_taker_start_stoit(covh_taker *t, const char *stoit_name) { t->start_stoit(stoit_name); }
void covh_taker_start_entry(covh_taker *t, size_t entry_order_id, const char *entry_name) { t->start_entry(entry_order_id, entry_name); }
void covh_taker_add_single_coverage(covh_taker *t, float coverage) { t->add_single_coverage(coverage); }
void covh_taker_add_coverage_entry(covh_taker *t, uint64_t num_reads, uint64_t num_bases) { t->add_coverage_entry(num_reads, num_bases); }
void covh_taker_finish_entry(covh_taker *t) { t->finish_entry(); }
int covh_taker_names_mismatch(const covh_taker *t) { return t->mismatch ? 1 : 0; }
// CoverageTakerTypeIterator (coverage_takers.rs:265-377) flattened: returns the number of (entry, stoit) items and fills up to
// `cap` of them; coverages holds num_coverages floats per item.
size_t covh_taker_iterate(const covh_taker *t, uint64_t *entry_index, uint64_t *stoit_index, float *coverages, size_t cap) {
    const auto all = iterate_cached(*t);
    for (size_t i = 0; i < all.size() && i < cap; i++) {
        if (entry_index) entry_index[i] = all[i].entry_index;
        if (stoit_index) stoit_index[i] = all[i].stoit_index;
        if (coverages) for (size_t k = 0; k < all[i].coverages.size(); k++) coverages[i * t->num_coverages + k] = all[i].coverages[k];
    }
    return all.size();
}

void covh_finalise_printing(covh_taker *t, int printer, const char *entry_type, const char *const *headers,
                            size_t n_headers, const covh_reads_mapped *rm, size_t n_samples,
                            const int64_t *norm, size_t n_norm, int64_t rpkm_col, int64_t tpm_col) {
    if (printer == 1) print_sparse(*t, rm, norm, n_norm, rpkm_col, tpm_col);
    else if (printer == 2) print_dense(*t, entry_type, headers, n_headers, rm, n_samples, norm, n_norm, rpkm_col, tpm_col);
    else if (printer == 3) print_metabat(*t);
}

}  // extern "C"

// =====================================================================================================================
// Per-gene coverage (--gff): src/genes.rs.  The pileup stays on the device (the depth of a contig comes through a
// callback: cov_copy_depth in the product, the oracle's depth in CPU tests); what genes.rs adds on top of the contig
// scan — GFF parsing (:42-161), gene resolution against the header (:346-419), per-gene window statistics over the
// contig's depth (:508-535) and per-gene read aggregates by leftmost position (:521-526) — is this host code, feeding the
// same calculate() / print_coverage() as the contig scan.
namespace {

struct GeneDef { std::string id, contig; u64 start, end; };

bool parse_u64(const std::string &s, u64 &v) {   // Rust u64::from_str: optional '+', decimal digits, no overflow
    size_t i = 0;
    if (i < s.size() && s[i] == '+') i++;
    if (i >= s.size()) return false;
    u64 x = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        const u64 d = (u64)(s[i] - '0');
        if (x > (0xffffffffffffffffull - d) / 10) return false;
        x = x * 10 + d;
    }
    v = x;
    return true;
}
std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    auto ws = [](unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); };
    while (a < b && ws((unsigned char)s[a])) a++;
    while (b > a && ws((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
bool gff_attribute(const std::string &attrs, const std::string &key, std::string &out) {   // genes.rs:144-161
    size_t p = 0;
    while (p <= attrs.size()) {
        size_t q = attrs.find(';', p);
        if (q == std::string::npos) q = attrs.size();
        const std::string entry = trim(attrs.substr(p, q - p));
        if (!entry.empty()) {
            if (entry.compare(0, key.size() + 1, key + "=") == 0) { out = trim(entry.substr(key.size() + 1)); return true; }
            if (entry.compare(0, key.size() + 1, key + " ") == 0) {
                std::string v = trim(entry.substr(key.size() + 1));
                size_t a = 0, b = v.size();
                while (a < b && v[a] == '"') a++;
                while (b > a && v[b - 1] == '"') b--;
                out = v.substr(a, b - a);
                return true;
            }
        }
        p = q + 1;
    }
    return false;
}

struct ResolvedGene { size_t entry_id; std::string name; u64 start, end; };

}  // namespace

struct covh_genes { std::vector<GeneDef> genes; };

extern "C" {

covh_genes *covh_genes_read_gff(const char *path, const char *feature_type, char *err, size_t errcap) {
    FILE *fh = fopen(path, "rb");
    if (!fh) {
        if (err && errcap) snprintf(err, errcap, "Failed to open GFF file %s", path);
        return nullptr;
    }
    std::string all;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fh)) > 0) all.append(buf, n);
    fclose(fh);
    covh_genes *g = new covh_genes();
    u64 auto_id = 0;
    size_t p = 0;
    while (p < all.size()) {
        size_t q = all.find('\n', p);
        if (q == std::string::npos) q = all.size();
        std::string line = all.substr(p, q - p);
        p = q + 1;
        {   // trim_end
            size_t b = line.size();
            while (b > 0 && (line[b - 1] == ' ' || ((unsigned char)line[b - 1] >= 9 && (unsigned char)line[b - 1] <= 13))) b--;
            line.resize(b);
        }
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> f;
        for (size_t a = 0;;) {
            const size_t t = line.find('\t', a);
            if (t == std::string::npos) { f.push_back(line.substr(a)); break; }
            f.push_back(line.substr(a, t - a));
            a = t + 1;
        }
        if (f.size() < 8) continue;
        if (feature_type && f[2] != feature_type) continue;
        u64 s1, e1;
        if (!parse_u64(f[3], s1) || !parse_u64(f[4], e1)) continue;
        if (s1 == 0 || e1 < s1) continue;
        const std::string attrs = f.size() > 8 ? f[8] : std::string();
        std::string id;
        bool have = false;
        for (const char *k : {"ID", "locus_tag", "gene_id", "Name", "gene", "Parent"}) {
            std::string v;
            if (gff_attribute(attrs, k, v) && !v.empty()) { id = v; have = true; break; }
        }
        if (!have) { auto_id++; id = f[0] + "_gene_" + std::to_string(auto_id); }
        g->genes.push_back(GeneDef{id, f[0], s1 - 1, e1});
    }
    return g;
}

covh_genes *covh_genes_from_arrays(const char *const *ids, const char *const *contigs, const uint64_t *start,
                                   const uint64_t *end, size_t n) {
    covh_genes *g = new covh_genes();
    for (size_t i = 0; i < n; i++) g->genes.push_back(GeneDef{ids[i], contigs[i], start[i], end[i]});
    return g;
}
size_t covh_genes_count(const covh_genes *g) { return g ? g->genes.size() : 0; }
void covh_genes_get(const covh_genes *g, size_t i, const char **id, const char **contig, uint64_t *start, uint64_t *end) {
    const GeneDef &d = g->genes[i];
    if (id) *id = d.id.c_str();
    if (contig) *contig = d.contig.c_str();
    if (start) *start = d.start;
    if (end) *end = d.end;
}
void covh_genes_free(covh_genes *g) { delete g; }

int covh_gene_coverage(const covh_header *h, const covh_genes *genes, const covh_genome_namer *namer, const char *stoit_name,
                       const cov_batch *rec, const cov_config *cfg, cov_session *device_session, covh_depth_fn depth_fn,
                       void *depth_ctx, uint64_t num_detected_primary_alignments, covh_taker *taker, const covh_estimator *est,
                       size_t n_est, int print_zero, covh_reads_mapped *rm_out) {
    if (!h || !genes || !rec || !cfg || (!depth_fn && !device_session) || !taker || (!est && n_est)) return COV_ERR_INVALID_ARG;
    if (!check_excl(est, n_est)) { g_err = "estimators disagree on contig_end_exclusion"; return COV_ERR_INVALID_ARG; }
    const u64 excl = session_excl(est, n_est);
    const u64 zero = 0;
    const u32 nT = h->n_targets;

    // ---- resolve_genes_against_header, genes.rs:346-419
    std::vector<std::vector<ResolvedGene>> by_tid(nT);
    {
        std::unordered_map<std::string_view, u32> name_to_tid;
        name_to_tid.reserve(nT * 2);
        for (u32 t = 0; t < nT; t++) name_to_tid[target_name(h, t)] = t;      // a repeated name keeps the last tid
        for (const GeneDef &g : genes->genes) {
            auto it = name_to_tid.find(std::string_view(g.contig));
            if (it == name_to_tid.end()) continue;
            const u32 tid = it->second;
            const u64 L = h->target_len[tid];
            const u64 s = std::min(g.start, L), e = std::min(g.end, L);
            if (s >= e) continue;
            std::string name = g.id + "\t" + g.contig;
            if (namer && namer->mode != 0) {
                if (namer->mode == 1) name += "\tgenome1";
                else if (namer->mode == 2) {
                    const size_t sp = g.contig.find((char)namer->separator);
                    if (sp == std::string::npos) continue;
                    name += "\t" + g.contig.substr(0, sp);
                } else {
                    const int32_t gi = namer->genome_of_tid ? namer->genome_of_tid[tid] : -1;
                    if (gi < 0) continue;
                    name += std::string("\t") + namer->genome_names[gi];
                }
            }
            by_tid[tid].push_back(ResolvedGene{0, std::move(name), s, e});
        }
        size_t next_id = 0;
        for (auto &v : by_tid) {
            std::stable_sort(v.begin(), v.end(), [](const ResolvedGene &a, const ResolvedGene &b) { return a.start < b.start; });
            for (auto &g : v) g.entry_id = next_id++;
        }
    }

    // ---- device path: every resolved gene is one interval, reduced on the GPU over the materialised depth
    std::vector<cov_interval_stats> dstats;
    std::vector<uint64_t> dhist;
    if (device_session) {
        std::vector<cov_interval> ivs;
        for (u32 t = 0; t < nT; t++)
            for (const ResolvedGene &g : by_tid[t]) { cov_interval iv; iv.tid = t; iv.pad = 0; iv.start = g.start; iv.end = g.end; ivs.push_back(iv); }
        bool want_hist = false;
        for (size_t k = 0; k < n_est; k++) want_hist |= est[k].kind == COVH_TRIMMED_MEAN || est[k].kind == COVH_PILEUP_COUNTS;
        dstats.resize(ivs.size());
        uint64_t htot = 0;
        int rc = (int)cov_interval_stats_compute(device_session, ivs.data(), ivs.size(), excl, want_hist, dstats.data(), &htot);
        if (rc == COV_OK && htot) { dhist.resize(htot); rc = (int)cov_fetch_interval_hist(device_session, dhist.data()); }
        if (rc != COV_OK) { g_err = cov_last_error(device_session); return rc; }
    }

    taker->start_stoit(stoit_name);
    auto zero_genes = [&](u32 tid) {   // emit_zero_coverage_genes, :554-567
        for (const ResolvedGene &g : by_tid[tid]) {
            taker->start_entry(g.entry_id, g.name);
            for (size_t k = 0; k < n_est; k++) print_zero_coverage(est[k], *taker, g.end - g.start);
            taker->finish_entry();
        }
    };

    std::vector<u64> starts, pprim, pmism;
    std::vector<double> pident;
    std::vector<int32_t> depth;
    std::vector<float> cov(n_est);
    EntryAcc acc;
    int rc_err = COV_OK;
    auto emit = [&](u32 tid) {   // emit_genes_for_contig, :462-552
        const auto &gl = by_tid[tid];
        if (gl.empty()) return;
        const u64 L = h->target_len[tid];
        if (!device_session) {
            depth.resize((size_t)L);
            const int rc = depth_fn(depth_ctx, tid, depth.data());
            if (rc != COV_OK) { if (!rc_err) rc_err = rc; return; }
        }
        for (const ResolvedGene &g : gl) {
            const u64 s = g.start, e = std::min(g.end, L);
            if (s >= e) continue;
            const u64 len = e - s;
            acc.reset();
            acc.full_len = len;
            if (device_session) {
                const cov_interval_stats &D = dstats[g.entry_id];
                acc.full_covered = D.full_covered;
                if (2 * excl < len) {
                    acc.win_len = len - 2 * excl;
                    acc.win_sum_d = D.win_sum_d; acc.win_sum_d2 = D.win_sum_d2; acc.win_covered = D.win_covered; acc.win_min_d = D.win_min_d;
                    if (D.hist_len && !dhist.empty()) acc.hist.assign(dhist.begin() + (ptrdiff_t)D.hist_off, dhist.begin() + (ptrdiff_t)(D.hist_off + D.hist_len));
                    else acc.hist_len_only = (u64)D.win_max_d + 1;
                }
            } else {
            for (u64 p = s; p < e; p++) acc.full_covered += depth[p] > 0;
            if (2 * excl < len) {                              // add_contig's window on the GENE's own array (:394-404)
                acc.win_len = len - 2 * excl;
                u32 mx = 0;
                for (u64 p = s + excl; p < e - excl; p++) {
                    const u32 d = (u32)depth[p];
                    acc.win_sum_d += d; acc.win_sum_d2 += (u64)d * d; acc.win_covered += d > 0;
                    acc.win_min_d = std::min(acc.win_min_d, d); mx = std::max(mx, d);
                }
                acc.hist.assign((size_t)mx + 1, 0);
                for (u64 p = s + excl; p < e - excl; p++) acc.hist[(u32)depth[p]]++;
            }
            }
            const size_t lo = std::lower_bound(starts.begin(), starts.end(), s) - starts.begin();
            const size_t hi = std::lower_bound(starts.begin(), starts.end(), e) - starts.begin();
            acc.n_reads = pprim[hi] - pprim[lo];
            acc.mismatches = pmism[hi] - pmism[lo];
            acc.sum_identity = pident[hi] - pident[lo];       // difference of the running f64 prefix sums, as :526 does
            bool nonzero = false;
            for (size_t k = 0; k < n_est; k++) { cov[k] = calculate(est[k], acc, &zero, 1); nonzero |= cov[k] > 0.0f; }
            if (print_zero || nonzero) {
                taker->start_entry(g.entry_id, g.name);
                for (size_t k = 0; k < n_est; k++) print_coverage(est[k], acc, cov[k], *taker);
                taker->finish_entry();
            }
        }
    };
    auto previous = [&](int64_t last, int64_t cur) {   // process_previous_genes, :421-460
        if (last != -2) emit((u32)last);
        if (print_zero) for (int64_t t = last == -2 ? 0 : last + 1; t < cur; t++) zero_genes((u32)t);
    };

    int64_t last_tid = -2;
    u64 mapped_total = 0;
    auto reset_contig = [&]() { starts.clear(); pprim.assign(1, 0); pmism.assign(1, 0); pident.assign(1, 0.0); };
    reset_contig();
    // Per-record work (filters, CIGAR walk, identity) is independent: blocks of records are classified by a few threads,
    // then consumed in file order (contig changes, prefix sums and emission are sequential, as in the reference).
    enum : uint8_t { SKIP = 0, TAKE = 1, NM_MISSING = 2, NM_BADTYPE = 3 };
    const u64 BLOCK = 1u << 20;
    std::vector<uint8_t> state(std::min<u64>(BLOCK, rec->n_records));
    std::vector<u64> mism(state.size());
    std::vector<double> ident(state.size());
    auto classify = [&](u64 i, u64 k) {
        const u32 flag = rec->flag[i];
        const bool supp = flag & 0x800u, sec = flag & 0x100u, unmapped = flag & 0x4u;
        state[k] = SKIP;
        u64 aligned = 0, indels = 0;
        auto walk = [&]() {
            for (u32 c = rec->cigar_off[i]; c < rec->cigar_off[i + 1]; c++) {
                const u32 op = rec->cigar[c] & 15u, len = rec->cigar[c] >> 4;
                if (op == 0 || op == 7 || op == 8) aligned += len;
                else if (op == 2 || op == 1) { aligned += len; indels += len; }
            }
        };
        bool walked = false;
        if (cfg->filter_single) {   // reader stage, single-read branch (filter.rs:88-116, 243-279), as k_prep applies it
            if (unmapped || (!cfg->include_supplementary && supp) || (!cfg->include_secondary && sec)) return;
            if (cfg->min_mapq != 255 && (rec->mapq[i] < cfg->min_mapq || rec->mapq[i] == 255)) return;
            if (rec->nm_kind[i] != COV_NM_UNSIGNED) { state[k] = rec->nm_kind[i] == COV_NM_ABSENT ? NM_MISSING : NM_BADTYPE; return; }
            walk(); walked = true;
            const u32 al = (u32)aligned;
            const float a = (float)al;
            if (!(al >= cfg->min_aligned_length && a / (float)rec->l_seq[i] >= cfg->min_aligned_percent &&
                  1.0f - (float)rec->nm[i] / a >= cfg->min_percent_identity)) return;
        }
        if ((!cfg->include_secondary && sec) || (!cfg->include_supplementary && supp) || (!cfg->include_improper_pairs && !(flag & 0x2u))) return;
        if (unmapped) return;
        if (!walked) walk();
        if (rec->nm_kind[i] != COV_NM_UNSIGNED) { state[k] = rec->nm_kind[i] == COV_NM_ABSENT ? NM_MISSING : NM_BADTYPE; return; }
        const u64 edit = rec->nm[i];
        const bool primary = !supp && !sec;
        state[k] = TAKE;
        mism[k] = edit > indels ? edit - indels : 0;                                   // saturating_sub, :296
        ident[k] = (primary && aligned > 0) ? ((double)aligned - (double)edit) / (double)aligned : 0.0;
    };
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    for (u64 b0 = 0; b0 < rec->n_records; b0 += BLOCK) {
        const u64 nb = std::min<u64>(BLOCK, rec->n_records - b0);
        if (nb >= (1u << 16) && hw > 1) {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < hw; t++)
                pool.emplace_back([&, t]() { for (u64 k = nb * t / hw, e = nb * (t + 1) / hw; k < e; k++) classify(b0 + k, k); });
            for (auto &th : pool) th.join();
        } else {
            for (u64 k = 0; k < nb; k++) classify(b0 + k, k);
        }
        for (u64 k = 0; k < nb; k++) {
            const u64 i = b0 + k;
            // a record that fails the NM requirement before its contig is reached still ends the scan there, like the panic
            if (state[k] == SKIP) continue;
            if (state[k] != TAKE) {
                // single-read mode meets nm() inside the filter, before the flag filter; otherwise after it: classify()
                // already ordered those tests, so any error state is the reference's panic at this record
                return state[k] == NM_MISSING ? COV_ERR_NM_MISSING : COV_ERR_NM_BADTYPE;
            }
            const int64_t tid = rec->tid[i];
            if (tid != last_tid) {
                if (tid < last_tid) { g_err = "BAM file appears to be unsorted. Input BAM files must be sorted by reference (i.e. by samtools sort)"; return COV_ERR_UNSORTED; }
                if (tid < 0 || (u64)tid >= nT) { g_err = "record refers to a reference id outside the header (Corrupt BAM file?)"; return COV_ERR_BAD_TID; }
                previous(last_tid, tid);
                if (rc_err) return rc_err;
                last_tid = tid;
                reset_contig();
            }
            const bool primary = !(rec->flag[i] & 0x900u);
            mapped_total += primary;
            starts.push_back((u64)(int64_t)rec->pos[i]);
            pprim.push_back(pprim.back() + (primary ? 1 : 0));
            pmism.push_back(pmism.back() + mism[k]);
            pident.push_back(pident.back() + ident[k]);
        }
    }
    previous(last_tid, nT);
    if (rc_err) return rc_err;
    if (rm_out) { rm_out->num_mapped_reads = mapped_total; rm_out->num_reads = num_detected_primary_alignments; }
    return COV_OK;
}

}  // extern "C"
