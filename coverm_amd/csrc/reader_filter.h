// ReferenceSortedBamFilter::read (src/filter.rs:84-228) as a state machine fed one record at a time: what `coverm filter` needs to run in
// bounded memory (csrc/host_bam.cpp covh_bam_filter_file streams the file window by window) and what covh_reader_filter_order runs over
// arrays (csrc/host_filter.cpp).  One implementation of the selection for both.
//
//   single branch (filter_single && !filter_pairs, :88-116): an unmapped record is returned when !filter_out; a record that passes
//     the flag test (:100-102) is returned iff single_read_passes_filter == filter_out; every other record is dropped;
//   pair branch (:117-228): an unmapped record is returned when !filter_out; secondary / supplementary records are dropped; a record
//     that is not a proper pair is dropped (filter_out) or returned (!filter_out); proper pairs are matched by name within one
//     reference — the first mate waits ("parked") until a record of its name arrives, the parked set is forgotten when the reference
//     changes — and the pair is returned, first mate then second, iff (its judgement == filter_out).
//
// The caller keeps whatever it needs of a parked record (its index, or its bytes) under the id it passed in; the machine keeps the name and
// the summary the judgement reads.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>

#include "../../include/coverm_host.h"

namespace covf {

// What the two judgements read of a record (filter.rs:243-336).  The aligned lengths are over the record's REAL CIGAR (htslib hands the
// reference the CG:B,I words of a long-CIGAR record): M I = X, with and without D.
struct RecSum {
    int32_t tid = -1, mtid = -1;
    uint16_t flag = 0;
    uint8_t mapq = 0, nm_kind = COV_NM_ABSENT;
    uint32_t nm = 0, l_seq = 0, aligned_with_del = 0, aligned_no_del = 0;
};

inline void aligned_lengths(const uint32_t *cigar, uint32_t n, uint32_t &with_del, uint32_t &no_del) {
    uint32_t a = 0, d = 0;
    for (uint32_t c = 0; c < n; c++) {
        const uint32_t op = cigar[c] & 15u, len = cigar[c] >> 4;
        if (op == 0 || op == 1 || op == 7 || op == 8) a += len;
        else if (op == 2) d += len;
    }
    no_del = a; with_del = a + d;
}

class ReaderFilter {
    const covh_pair_filter f_;
    const bool single_branch_, inc_supp_, inc_sec_, out_;
    int32_t cur_ = -1;                      // current_reference starts at -1 (filter.rs:76)
    struct Parked { uint64_t id; RecSum s; };
    std::unordered_map<std::string, Parked> first_;      // first_set: the parked records of the current reference by name

    bool nm(const RecSum &r, uint64_t &v) {              // nm(&record), lib.rs:138-158: the reference panics
        if (r.nm_kind == COV_NM_UNSIGNED) { v = r.nm; return true; }
        if (!err) err = r.nm_kind == COV_NM_ABSENT ? COV_ERR_NM_MISSING : COV_ERR_NM_BADTYPE;
        return false;
    }
    bool single_ok(const RecSum &r) {                    // filter.rs:243-279
        if (f_.min_mapq != 255 && (r.mapq < f_.min_mapq || r.mapq == 255)) return false;
        uint64_t e;
        if (!nm(r, e)) return false;
        const uint32_t al = r.aligned_with_del;
        return al >= f_.min_aligned_length_single && (float)al / (float)r.l_seq >= f_.min_aligned_percent_single &&
               1.0f - (float)e / (float)al >= f_.min_percent_identity_single;
    }
    bool pair_ok(const RecSum &r2, const RecSum &r1) {   // filter.rs:281-336
        if (f_.min_mapq != 255 && (r1.mapq < f_.min_mapq || r2.mapq < f_.min_mapq || r1.mapq == 255 || r2.mapq == 255)) return false;
        uint64_t e2, e1;
        if (!nm(r2, e2) || !nm(r1, e1)) return false;
        const uint64_t e = e2 + e1;
        const uint32_t al = r2.aligned_no_del + r1.aligned_no_del;
        return al >= f_.min_aligned_length_pair && (float)al / (float)((uint64_t)r1.l_seq + r2.l_seq) >= f_.min_aligned_percent_pair &&
               1.0f - (float)e / (float)al >= f_.min_percent_identity_pair;
    }

public:
    int err = 0;        // COV_ERR_NM_MISSING / COV_ERR_NM_BADTYPE met while judging
    enum Act { DROP, EMIT, PARK, EMIT_PAIR, DROP_PAIR };
    ReaderFilter(const covh_pair_filter &f, bool filter_pairs, bool include_supplementary, bool include_secondary, bool filter_out)
        : f_(f), single_branch_(f.filter_single && !filter_pairs), inc_supp_(include_supplementary), inc_sec_(include_secondary), out_(filter_out) {}
    size_t parked() const { return first_.size(); }

    // One record, in file order.  `self_id`: what the caller will know this record by should it be parked.  *forget is set when the parked
    // set was emptied before this record was looked at (the caller's copies of parked records can go).  EMIT_PAIR: *partner_id is the
    // parked first mate, to be returned before this record; DROP_PAIR: the pair was judged and goes, *partner_id is the parked mate the
    // caller can let go of.
    Act push(const RecSum &r, const char *name, size_t name_len, uint64_t self_id, uint64_t *partner_id, bool *forget) {
        *forget = false;
        const bool unmapped = r.flag & 0x4, supp = r.flag & 0x800, sec = r.flag & 0x100;
        if (single_branch_) {
            if (unmapped && !out_) return EMIT;
            const bool p1 = !unmapped && (inc_supp_ || !supp) && (inc_sec_ || !sec);
            return (p1 && single_ok(r) == out_) ? EMIT : DROP;
        }
        if (unmapped && !out_) return EMIT;
        if (r.flag & 0x900) return DROP;
        if (!(r.flag & 0x2)) return out_ ? DROP : EMIT;
        if (r.tid != cur_) { cur_ = r.tid; *forget = !first_.empty(); first_.clear(); }
        std::string q(name, name_len);
        auto it = first_.find(q);
        if (it == first_.end()) {
            if (r.mtid != cur_) return DROP;            // :164-171: only a record whose mate lies on this reference waits for it
            first_.emplace(std::move(q), Parked{self_id, r});
            return PARK;
        }
        const Parked p1 = it->second;
        first_.erase(it);
        const bool pass = (!f_.filter_single || (single_ok(p1.s) && single_ok(r))) && pair_ok(r, p1.s);
        *partner_id = p1.id;
        return pass == out_ ? EMIT_PAIR : DROP_PAIR;
    }
};

}  // namespace covf
