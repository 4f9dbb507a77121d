// covh_cli_main — `coverm contig` / `coverm genome` over --bam-files on the MI355X engine (the body of the coverm-amd binary).
//
// Mirrors the reference orchestrator for this path (src/bin/coverm.rs): FilterParameters::generate_from_clap
// :1659-1678 + doing_filtering :1695-1703, EstimatorsAndTaker::generate_from_clap :1315-1504, print_headers
// :1506-1519, run_contig :2088-2131, run_genome :1539-1628, parse_percentage :1296-1312, parse_separator
// :1522-1537; flag names and defaults from src/cli.rs (contig :2264-2582, genome :1669-2263).
// Everything else the reference binary does (mapping, indexing, filter/make/cluster subcommands) is out of scope.
//
// Ingest: a BAM is STREAMED (covh_bam_stream_*: windows of BGZF blocks inflated, parsed into page-locked SoA batches and
// pushed to HBM while later windows are still being inflated; host memory bounded) unless pair-mode filtering, --gff or SAM
// text input need the whole file on the host (covh_bam_open).  Each BAM brings its own header (contig.rs:29-32); only the
// cached taker insists that entry names agree between BAMs (coverage_takers.rs:140-148).
//
// Multi-GPU (--devices a,b,...; one process, one thread + session + stream reader per device):
//   * at least as many BAMs as devices: samples are dealt to devices (config 4 shape);
//   * fewer BAMs than devices: every BAM is cut into tid spans, one per device, each device inflating only its span
//     (config 5 shape); the per-contig result blocks meet on the first device through ONE RCCL gather (cov_gather).
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <stdexcept>
#include <cerrno>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/coverm_host.h"
#include "knobs.h"

namespace {

std::atomic<bool> g_skip_teardown{false};   // covh_cli_set_fast_exit: a process about to exit need not hand ~100 GB of HBM back allocation by allocation

struct Fatal : std::runtime_error { using std::runtime_error::runtime_error; };
struct SpanUnsorted : Fatal { using Fatal::Fatal; };      // a tid span met keys that decrease (covh_bam_gpu_ingest_span -2, the CPU span reader's same rule)
[[noreturn]] void die(const std::string &m) { throw Fatal(m); }

// clap's typed value parsers (u8 / u16 / u32 / u64 / f32 arguments of coverm.rs / cli.rs): a value that is not a number of the argument's
// type ends the run, it is not read as 0
uint64_t parse_uint(const std::string &opt, const char *v, uint64_t max) {
    char *end = nullptr;
    errno = 0;
    const unsigned long long x = (v && *v >= '0' && *v <= '9') ? strtoull(v, &end, 10) : 0ull;
    if (!end || *end || errno || x > max) die(std::string("invalid value '") + (v ? v : "") + "' for '" + opt + "'");
    return x;
}
float parse_f32(const std::string &opt, const char *v) {
    char *end = nullptr;
    const float x = (v && *v) ? strtof(v, &end) : 0.0f;
    if (!end || *end || end == v) die(std::string("invalid value '") + (v ? v : "") + "' for '" + opt + "'");
    return x;
}

float parse_percentage(const char *v, const char *opt = "percentage") {   // coverm.rs:1296-1312
    if (!v) return 0.0f;
    float p = parse_f32(opt, v);
    if (p >= 1.0f && p <= 100.0f) p /= 100.0f;
    else if (!(p >= 0.0f && p <= 100.0f)) die(std::string("Invalid alignment percentage: '") + v + "'");
    return p;
}

struct Args {
    std::string mode;
    std::vector<std::string> bams, methods;
    const char *min_covered_fraction = nullptr, *trim_min = "5", *trim_max = "95";
    uint64_t contig_end_exclusion = 75;
    std::string output_format = "dense", output_file, genome_definition, gff, gff_feature_type;
    bool have_gff_feature_type = false;
    bool no_zeros = false, proper_pairs_only = false, exclude_supplementary = false, include_secondary = false;
    bool single_genome = false, have_separator = false, no_stream = false;
    char separator = '~';
    uint32_t min_aligned_length = 0, min_aligned_length_pair = 0;
    const char *min_pid = nullptr, *min_aligned_pct = nullptr, *min_pid_pair = nullptr, *min_aligned_pct_pair = nullptr;
    int min_mapq = 255, threads = 1;
    std::vector<int> devices;
};

struct Filter {   // FilterParameters, coverm.rs:1648-1657
    bool improper = true, supp = true, sec = false;
    uint32_t len_single = 0, len_pair = 0;
    float pid_single = 0, pct_single = 0, pid_pair = 0, pct_pair = 0;
    int mapq = 255;
    bool doing_filtering() const {
        return pid_single > 0 || pid_pair > 0 || pct_single > 0 || mapq < 255 || pct_pair > 0 || len_single > 0 || len_pair > 0;
    }
    void mode(bool &fs, bool &fp) const {   // filter.rs:48-61
        const bool fs0 = len_single > 0 || pid_single > 0 || pct_single > 0;
        const bool fp0 = len_pair > 0 || pid_pair > 0 || pct_pair > 0;
        fs = fs0 || (!fp0 && mapq != 255);
        fp = fp0 || ((!fs || !improper) && mapq != 255);
    }
};

// One BAM's results plus the header they refer to.
struct Sample {
    std::string stoit, path;
    std::string names_blob; std::vector<uint32_t> name_off; std::vector<uint64_t> tlen;
    std::vector<int32_t> genome_of_tid;
    std::vector<cov_contig_stats> stats;
    std::vector<uint64_t> hist;
    std::vector<float> estimates;      // calculate_coverage of every contig, evaluated on the device (Run::dev_est): n_targets x estimators
    uint64_t prim = 0, n_records = 0;
    covh_reads_mapped gene_rm{0, 0};
    double t_open = 0, t_ingest = 0, t_finish = 0; uint64_t peak_bytes = 0; bool streamed = false, device_ingest = false;
    covh_header header() const { covh_header h; h.n_targets = (uint32_t)tlen.size(); h.names = names_blob.c_str(); h.name_off = name_off.data(); h.target_len = tlen.data(); return h; }
    std::string target_name(uint32_t t) const { return names_blob.substr(name_off[t], name_off[t + 1] - name_off[t]); }
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool timing_on() { return covh_timing_on() != 0; }
// COVERM_NO_GPU_INGEST: every file through the CPU readers (tests compare the two paths); COVERM_PAIR_ON_HOST: the pair-mode reader filter on the host
bool no_gpu_ingest() { static const bool v = getenv("COVERM_NO_GPU_INGEST") != nullptr; return v; }
bool pair_on_host() { static const bool v = getenv("COVERM_PAIR_ON_HOST") != nullptr; return v; }

struct HeaderAhead { covh_bam_header *hd = nullptr; std::string err; };

struct Run {
    std::mutex hdr_mutex;
    std::map<std::string, std::future<HeaderAhead>> hdr_ahead;   // headers of the first BAM files, read beside the runtime's start-up
    Args a;
    Filter f;
    bool contig = true, by_names = false, per_gene = false, fs = false, fp = false;
    std::vector<covh_estimator> est;
    bool dev_est = false;      // `coverm contig` with estimators the device evaluates (cov_set_estimators): no per-contig finalisation, no histogram fetch on the host
    uint32_t want = 0;
    cov_config cfg{};
    std::vector<std::string> genomes;
    std::unordered_map<std::string, int32_t> c2g;
    covh_genes *genes = nullptr;
    covh_taker *taker = nullptr;
    std::mutex taker_mutex;   // --gff writes entries while the sample is resident: one sample at a time
};

void check(cov_session *s, cov_status st) { if (st != COV_OK) die(cov_last_error(s)); }

std::string stoit_of(const std::string &path) {   // file stem, bam_generator.rs:358-365
    std::string p = path;
    const size_t sl = p.find_last_of('/');
    if (sl != std::string::npos) p = p.substr(sl + 1);
    const size_t dot = p.find_last_of('.');
    return dot == std::string::npos ? p : p.substr(0, dot);
}

void set_header(Sample &S, uint32_t nt, const std::function<const char *(uint32_t)> &name, const std::function<uint64_t(uint32_t)> &len) {
    S.names_blob.clear(); S.name_off.assign(1, 0); S.tlen.clear();
    for (uint32_t t = 0; t < nt; t++) { S.names_blob += name(t); S.name_off.push_back((uint32_t)S.names_blob.size()); S.tlen.push_back(len(t)); }
}

// tid -> genome table and participation mask of one BAM (genome.rs:51-67)
void genome_table(const Run &R, Sample &S, std::vector<uint8_t> &mask) {
    const uint32_t nt = (uint32_t)S.tlen.size();
    S.genome_of_tid.assign(nt, -1); mask.assign(nt, 0);
    uint32_t in = 0;
    for (uint32_t t = 0; t < nt; t++) {
        auto it = R.c2g.find(S.target_name(t));
        if (it != R.c2g.end()) { S.genome_of_tid[t] = it->second; mask[t] = 1; in++; }
    }
    if (!in) die("Error: There are no found reference sequences that are a part of a genome");
}

bool is_bgzf(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char m[2] = {0, 0};
    const size_t n = fread(m, 1, 2, f);
    fclose(f);
    return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
}

// device ingests that run at once: every device in span mode (nb < nd) and with at least as many files as devices
size_t span_mode_feeders(size_t nb, size_t nd) { return nb < nd ? nd : std::min(nb, nd); }

// What the host keeps of a finished session: the estimators' floats when the device evaluated them, else the histogram bins the host's
// calculate_coverage needs.
void fetch_results(Run &R, cov_session *s, Sample &S, const cov_summary &summ) {
    if (R.dev_est) {
        S.estimates.resize(S.tlen.size() * R.est.size());
        check(s, cov_fetch_estimates(s, S.estimates.data()));
    } else if (R.want & COV_WANT_HIST) { S.hist.resize(summ.hist_total); check(s, cov_fetch_hist(s, S.hist.data())); }
}

// Decode + push + finish of one BAM (or one tid span of it) on one session.  Leaves the session finished.
void ingest(Run &R, cov_session *s, Sample &S, int threads, uint32_t span_index, uint32_t span_count) {
    const Args &a = R.a;
    // pair-mode filtering (filter.rs:117-228) runs on the device over what the device ingest extracted (mate reference + read-name
    // hash per record, cov_pair_filter_apply); only when that path declines the file does the whole file come to the host
    const bool bgzf = is_bgzf(S.path);
    const bool pair_dev = R.fp && !a.no_stream && !R.per_gene && bgzf && !no_gpu_ingest() && !pair_on_host();
    const bool stream = !a.no_stream && (!R.fp || pair_dev) && !R.per_gene && bgzf;
    if (span_count > 1 && !stream) die("--devices with fewer BAM files than devices needs streamable input (BAM, no --gff)");
    S.stoit = stoit_of(S.path); S.streamed = stream && !R.fp;
    const double t0 = now();
    std::vector<uint8_t> mask;
    check(s, cov_reset(s));
    if (stream && !no_gpu_ingest()) {
        // ---- device ingest: the compressed file goes to HBM, the GPU inflates, finds the records and fills its own store
        char err[512] = {0};
        covh_bam_header *hd = nullptr;
        {   // the first files' headers were read while the HIP runtime came up
            std::unique_lock<std::mutex> lk(R.hdr_mutex);
            auto it = R.hdr_ahead.find(S.path);
            if (it != R.hdr_ahead.end()) {
                std::future<HeaderAhead> fut = std::move(it->second);
                R.hdr_ahead.erase(it);
                lk.unlock();
                HeaderAhead ha = fut.get();
                hd = ha.hd;
                if (!hd) snprintf(err, sizeof err, "%s", ha.err.c_str());
            }
        }
        if (!hd && !err[0]) hd = covh_bam_read_header(S.path.c_str(), err, sizeof err);
        if (!hd) die(err);
        struct HdFree { covh_bam_header *p; ~HdFree() { covh_bam_header_free(p); } } hdfree{hd};
        set_header(S, covh_bam_header_n_targets(hd), [&](uint32_t t) { return covh_bam_header_target_name(hd, t); },
                   [&](uint32_t t) { return covh_bam_header_target_len(hd, t); });
        check(s, cov_set_targets(s, (uint32_t)S.tlen.size(), S.tlen.data()));
        if (R.by_names) { genome_table(R, S, mask); check(s, cov_set_target_mask(s, mask.data())); }
        S.t_open = now() - t0;
        uint64_t nrec = 0; double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        check(s, cov_ingest_want_mates(s, R.fp ? 1 : 0));
        // (an assembly's statistics are 128 B x millions of contigs: the array is obtained and touched beside the ingest, not behind it)
        std::future<void> stats_ahead;
        if (S.tlen.size() >= 65536) stats_ahead = std::async(std::launch::async, [&S] { S.stats.resize(S.tlen.size()); });
        struct StatsWait { std::future<void> &f; ~StatsWait() { if (f.valid()) f.get(); } } stats_wait{stats_ahead};
        int rc = covh_bam_gpu_ingest_span(S.path.c_str(), threads, s, hd, 1, span_index, span_count, &nrec, tm, err, sizeof err);
        if (stats_ahead.valid()) stats_ahead.get();
        if (rc == -2 && span_count > 1) throw SpanUnsorted(err);
        if (rc < 0) die(err);
        uint64_t pair_prim = 0; double t_pair = 0;
        if (rc == 0 && R.fp) {     // the reader-stage pair filter, on the device
            const double tp0 = now();
            cov_pair_filter pf; memset(&pf, 0, sizeof pf);
            pf.filter_single = R.fs; pf.min_mapq = (uint8_t)R.f.mapq; pf.min_aligned_length_single = R.f.len_single;
            pf.min_percent_identity_single = R.f.pid_single; pf.min_aligned_percent_single = R.f.pct_single;
            pf.min_aligned_length_pair = R.f.len_pair; pf.min_percent_identity_pair = R.f.pid_pair; pf.min_aligned_percent_pair = R.f.pct_pair;
            uint64_t nsel = 0;
            const cov_status prc = cov_pair_filter_apply(s, &pf, &nsel, &pair_prim);
            if (prc == COV_ERR_INGEST_FALLBACK) { rc = 1; snprintf(err, sizeof err, "%s", cov_last_error(s)); }
            else check(s, prc);
            t_pair = now() - tp0;
            if (timing_on() && rc == 0) fprintf(stderr, "[coverm-amd] %s: pair filter on the device: %llu of %llu records selected, %.3fs\n", S.stoit.c_str(), (unsigned long long)nsel, (unsigned long long)nrec, t_pair);
        }
        if (rc == 0) {
            S.n_records = nrec; S.device_ingest = true;
            if (timing_on())
                fprintf(stderr, "[coverm-amd] %s span %u/%u: device ingest: first upload after %.3fs, file read %.3fs, staging waits %.3fs, header walk %.3fs, feed calls %.3fs, inflate tail + parse %.3fs, total %.3fs, %llu records, bytes from %s\n",
                        S.stoit.c_str(), span_index, span_count, tm[4], tm[0], tm[1], tm[5], tm[6], tm[2], tm[3], (unsigned long long)nrec,
                        tm[7] == 2 ? "the mapped file (registered up front)" : tm[7] == 1 ? "the mapped file" : tm[7] == 3 ? "staging slots (copied from the mapping)" : "staging slots (pread)");
            S.t_ingest = now() - t0;
            S.stats.resize(S.tlen.size());
            cov_summary summ;
            check(s, cov_finish(s, S.stats.data(), &summ));
            fetch_results(R, s, S, summ);
            S.prim = R.fp ? pair_prim : summ.num_detected_primary_alignments;      // filter.rs:129-131 counts every primary record of the input
            S.t_finish = now() - t0 - S.t_ingest;
            return;
        }
        if (timing_on()) fprintf(stderr, "[coverm-amd] %s: %s\n", S.stoit.c_str(), err);
        check(s, cov_reset(s));       // the CPU reader takes the file
        if (R.fp && span_count > 1) die(std::string("--devices with fewer BAM files than devices and a pair-mode filter needs the device ingest, which declined this file: ") + err);
    }
    if (stream && !R.fp) {
        char err[512] = {0};
        covh_bam_stream *st = covh_bam_stream_open(S.path.c_str(), threads, span_index, span_count, err, sizeof err);
        if (!st) die(err);
        struct Closer { covh_bam_stream *p; ~Closer() { covh_bam_stream_close(p); } } closer{st};
        set_header(S, covh_bam_stream_n_targets(st), [&](uint32_t t) { return covh_bam_stream_target_name(st, t); },
                   [&](uint32_t t) { return covh_bam_stream_target_len(st, t); });
        check(s, cov_set_targets(s, (uint32_t)S.tlen.size(), S.tlen.data()));
        if (R.by_names) { genome_table(R, S, mask); check(s, cov_set_target_mask(s, mask.data())); }
        S.t_open = now() - t0;
        cov_batch b;
        int rc;
        while ((rc = covh_bam_stream_next(st, &b)) == 1) check(s, cov_push_batch(s, &b));
        if (rc == -2 && span_count > 1) throw SpanUnsorted(covh_bam_stream_error(st));
        if (rc < 0) die(covh_bam_stream_error(st));
        S.n_records = covh_bam_stream_n_records(st);
        S.peak_bytes = covh_bam_stream_peak_bytes(st);
        if (timing_on()) {
            double t[5]; covh_bam_stream_timing(st, t);
            fprintf(stderr, "[coverm-amd] %s span %u/%u: stream read %.3fs inflate %.3fs parse %.3fs (coordinator waits: inflate %.3fs parse %.3fs), %llu records, buffers %.0f MB\n",
                    S.stoit.c_str(), span_index, span_count, t[0], t[1], t[2], t[3], t[4], (unsigned long long)S.n_records, S.peak_bytes / 1e6);
        }
        S.t_ingest = now() - t0;
        S.stats.resize(S.tlen.size());
        cov_summary summ;
        check(s, cov_finish(s, S.stats.data(), &summ));
        fetch_results(R, s, S, summ);
        S.prim = summ.num_detected_primary_alignments;
        S.t_finish = now() - t0 - S.t_ingest;
        return;
    }
    // ---- --gff over a BAM the device can ingest: the device inflates and parses (and applies a pair-mode filter), then the records
    // the gene driver needs on the host (per-read vectors, genes.rs:182-344) come BACK from the session's store — 24 B per record
    // + CIGAR words over PCIe instead of a whole-file decode on the host
    char err[512] = {0};
    cov_batch batch; memset(&batch, 0, sizeof batch);
    struct HostRecords { std::vector<int32_t> tid, pos; std::vector<uint16_t> flag; std::vector<uint8_t> mapq, nmk; std::vector<uint32_t> nm, lseq, coff, cig; } hr;
    bool have_records = false, prim_from_host = false;
    if (R.per_gene && bgzf && !a.no_stream && span_count == 1 && !no_gpu_ingest() && !getenv("COVERM_GENES_DECODE_ON_HOST") && !pair_on_host()) {
        covh_bam_header *hd = covh_bam_read_header(S.path.c_str(), err, sizeof err);
        if (!hd) die(err);
        struct HdFree { covh_bam_header *p; ~HdFree() { covh_bam_header_free(p); } } hdfree{hd};
        set_header(S, covh_bam_header_n_targets(hd), [&](uint32_t t) { return covh_bam_header_target_name(hd, t); },
                   [&](uint32_t t) { return covh_bam_header_target_len(hd, t); });
        check(s, cov_set_targets(s, (uint32_t)S.tlen.size(), S.tlen.data()));
        if (R.by_names) { genome_table(R, S, mask); check(s, cov_set_target_mask(s, mask.data())); }
        check(s, cov_ingest_want_mates(s, R.fp ? 1 : 0));
        uint64_t nrec = 0; double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int rc = covh_bam_gpu_ingest_span(S.path.c_str(), threads, s, hd, 1, 0, 1, &nrec, tm, err, sizeof err);
        if (rc < 0) die(err);
        S.n_records = nrec;
        if (rc == 0 && R.fp) {
            cov_pair_filter pf; memset(&pf, 0, sizeof pf);
            pf.filter_single = R.fs; pf.min_mapq = (uint8_t)R.f.mapq; pf.min_aligned_length_single = R.f.len_single;
            pf.min_percent_identity_single = R.f.pid_single; pf.min_aligned_percent_single = R.f.pct_single;
            pf.min_aligned_length_pair = R.f.len_pair; pf.min_percent_identity_pair = R.f.pid_pair; pf.min_aligned_percent_pair = R.f.pct_pair;
            uint64_t nsel = 0, prim = 0;
            const cov_status prc = cov_pair_filter_apply(s, &pf, &nsel, &prim);
            if (prc == COV_ERR_INGEST_FALLBACK) rc = 1;
            else { check(s, prc); S.prim = prim; prim_from_host = true; }
        }
        if (rc == 0) {
            uint64_t n = 0, nc = 0;
            check(s, cov_copy_records(s, nullptr, &n, &nc));
            hr.tid.resize(n); hr.pos.resize(n); hr.flag.resize(n); hr.mapq.resize(n); hr.nmk.resize(n); hr.nm.resize(n); hr.lseq.resize(n);
            hr.coff.resize(n + 1); hr.cig.resize(nc + 1);
            batch.tid = hr.tid.data(); batch.pos = hr.pos.data(); batch.flag = hr.flag.data(); batch.mapq = hr.mapq.data(); batch.nm = hr.nm.data();
            batch.nm_kind = hr.nmk.data(); batch.l_seq = hr.lseq.data(); batch.cigar_off = hr.coff.data(); batch.cigar = hr.cig.data(); batch.n_records = n;
            check(s, cov_copy_records(s, &batch, nullptr, nullptr));
            if (n == 0) hr.coff[0] = 0;
            have_records = true; S.device_ingest = true;
            if (timing_on()) fprintf(stderr, "[coverm-amd] %s: --gff over the device ingest: %llu records came back from the store\n", S.stoit.c_str(), (unsigned long long)n);
        } else {
            if (timing_on()) fprintf(stderr, "[coverm-amd] %s: %s\n", S.stoit.c_str(), err[0] ? err : cov_last_error(s));
            check(s, cov_reset(s)); S.prim = 0; prim_from_host = false;
        }
    }
    // ---- whole file on the host: SAM text, files the device ingest declined, --no-stream
    covh_bam *bam = nullptr;
    struct Closer { covh_bam *&p; ~Closer() { if (p) covh_bam_close(p); } } closer{bam};
    cov_batch selected; memset(&selected, 0, sizeof selected);
    struct Freer { cov_batch *b; ~Freer() { if (b->tid) covh_batch_free(b); } } freer{&selected};
    if (!have_records) {
        bam = covh_bam_open(S.path.c_str(), threads, R.fp ? 1 : 0, err, sizeof err);
        if (!bam) die(err);
        set_header(S, covh_bam_n_targets(bam), [&](uint32_t t) { return covh_bam_target_name(bam, t); }, [&](uint32_t t) { return covh_bam_target_len(bam, t); });
        if (R.by_names) genome_table(R, S, mask);
        covh_bam_batch(bam, &batch);
        S.n_records = batch.n_records;
        if (R.f.doing_filtering() && !(R.fs && !R.fp)) {
            for (uint64_t i = 0; i < batch.n_records; i++) if (!(batch.flag[i] & 0x900)) S.prim++;   // filter.rs:129-131
            prim_from_host = true;
            covh_pair_filter pf; memset(&pf, 0, sizeof pf);
            pf.filter_single = R.fs; pf.min_mapq = (uint8_t)R.f.mapq; pf.min_aligned_length_single = R.f.len_single;
            pf.min_percent_identity_single = R.f.pid_single; pf.min_aligned_percent_single = R.f.pct_single;
            pf.min_aligned_length_pair = R.f.len_pair; pf.min_percent_identity_pair = R.f.pid_pair; pf.min_aligned_percent_pair = R.f.pct_pair;
            uint64_t *order = nullptr, n_order = 0;
            const int prc = covh_pair_mode_order(&batch, covh_bam_mtid(bam), covh_bam_qname_off(bam), covh_bam_qnames(bam), &pf, threads, &order, &n_order);
            if (prc == COV_ERR_NM_MISSING) die("Mapping record encountered that does not have an 'NM' auxiliary tag in the SAM/BAM format");
            if (prc != COV_OK) die(prc == COV_ERR_NM_BADTYPE ? "Unexpected data type of NM aux tag" : "pair filter failed");
            const int src = covh_batch_select(&batch, order, n_order, threads, &selected);
            covh_free(order);
            if (src != COV_OK) die("pair filter: selection failed");
            batch = selected;
        }
    }
    const uint32_t nt = (uint32_t)S.tlen.size();
    S.t_open = now() - t0;
    if (!have_records) {
        check(s, cov_set_targets(s, nt, S.tlen.data()));
        if (R.by_names) check(s, cov_set_target_mask(s, mask.data()));
        check(s, cov_push_batch(s, &batch));
    }
    S.t_ingest = now() - t0;
    S.stats.resize(nt);
    cov_summary summ;
    check(s, cov_finish(s, S.stats.data(), &summ));
    fetch_results(R, s, S, summ);
    if (!prim_from_host) S.prim = summ.num_detected_primary_alignments;
    if (R.per_gene) {   // genes.rs:182-344: per-gene reductions over this sample's depth, while the session holds it
        std::lock_guard<std::mutex> lk(R.taker_mutex);
        const covh_header gh = S.header();
        covh_genome_namer nm; memset(&nm, 0, sizeof nm);
        std::vector<const char *> gn;
        for (auto &g : R.genomes) gn.push_back(g.c_str());
        if (!R.contig) {
            nm.mode = a.single_genome ? 1 : a.have_separator ? 2 : 3;
            nm.separator = (uint8_t)a.separator;
            nm.genome_of_tid = S.genome_of_tid.data(); nm.genome_names = gn.data();
        }
        auto depth_cb = [](void *ctx, uint32_t tid, int32_t *out) -> int { return (int)cov_copy_depth((cov_session *)ctx, tid, out); };
        const int grc = covh_gene_coverage(&gh, R.genes, &nm, S.stoit.c_str(), &batch, &R.cfg, getenv("COVERM_GENES_ON_HOST") ? nullptr : s, depth_cb, s,
                                           S.prim, R.taker, R.est.data(), R.est.size(), !a.no_zeros, &S.gene_rm);
        if (grc == COV_ERR_HIP || grc == COV_ERR_STATE) die(cov_last_error(s));
        if (grc != COV_OK) die(covh_last_error());
    }
    S.t_finish = now() - t0 - S.t_ingest;
}

// `coverm filter` (bin/coverm.rs:408-472): every input BAM through ReferenceSortedBamFilter into its output BAM.  Host code in the
// reference and here: the work is rewriting a BAM (inflate, select, deflate), which no pass over coverage touches.  The thresholds
// are FilterParameters::generate_from_clap's (coverm.rs:1659-1678); filter_out = !--inverse.
int run_filter(int argc, char **argv) {
    std::vector<std::string> in, out;
    Filter f;
    bool inverse = false, proper_pairs_only = false, exclude_supplementary = false, include_secondary = false;
    const char *pid = nullptr, *pct = nullptr, *pid_pair = nullptr, *pct_pair = nullptr;
    int threads = 1;
    auto collect = [&](int &i, std::vector<std::string> &dst) { while (i + 1 < argc && argv[i + 1][0] != '-') dst.push_back(argv[++i]); };
    for (int i = 2; i < argc; i++) {
        const std::string k = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) die("missing value for " + k); return argv[++i]; };
        if (k == "-b" || k == "--bam-files") collect(i, in);
        else if (k == "-o" || k == "--output-bam-files") collect(i, out);
        else if (k == "--inverse") inverse = true;
        else if (k == "--proper-pairs-only") proper_pairs_only = true;
        else if (k == "--exclude-supplementary") exclude_supplementary = true;
        else if (k == "--include-secondary") include_secondary = true;
        else if (k == "--min-read-aligned-length") f.len_single = (uint32_t)parse_uint(k, val(), 0xffffffffull);
        else if (k == "--min-read-percent-identity") pid = val();
        else if (k == "--min-read-aligned-percent") pct = val();
        else if (k == "--min-read-aligned-length-pair") f.len_pair = (uint32_t)parse_uint(k, val(), 0xffffffffull);
        else if (k == "--min-read-percent-identity-pair") pid_pair = val();
        else if (k == "--min-read-aligned-percent-pair") pct_pair = val();
        else if (k == "--min-mapq") f.mapq = (int)parse_uint(k, val(), 255);
        else if (k == "-t" || k == "--threads") threads = (int)parse_uint(k, val(), 65535);
        else if (k == "-v" || k == "--verbose" || k == "-q" || k == "--quiet") {}
        else die("unknown argument " + k);
    }
    if (in.empty()) die("--bam-files is required");
    if (in.size() != out.size()) die("The number of input BAM files must be the same as the number output");     // coverm.rs:422-425
    f.improper = !proper_pairs_only; f.supp = !exclude_supplementary; f.sec = include_secondary;
    f.pid_single = parse_percentage(pid, "--min-read-percent-identity"); f.pct_single = parse_percentage(pct, "--min-read-aligned-percent");
    f.pid_pair = parse_percentage(pid_pair, "--min-read-percent-identity-pair"); f.pct_pair = parse_percentage(pct_pair, "--min-read-aligned-percent-pair");
    bool fs = false, fp = false;
    f.mode(fs, fp);
    covh_pair_filter pf; memset(&pf, 0, sizeof pf);
    pf.filter_single = fs; pf.min_mapq = (uint8_t)f.mapq; pf.min_aligned_length_single = f.len_single; pf.min_percent_identity_single = f.pid_single;
    pf.min_aligned_percent_single = f.pct_single; pf.min_aligned_length_pair = f.len_pair; pf.min_percent_identity_pair = f.pid_pair;
    pf.min_aligned_percent_pair = f.pct_pair;
    for (size_t k = 0; k < in.size(); k++) {
        char err[512] = {0};
        uint64_t n_in = 0, n_out = 0;
        if (covh_bam_filter_file(in[k].c_str(), out[k].c_str(), &pf, fp ? 1 : 0, f.supp ? 1 : 0, f.sec ? 1 : 0, inverse ? 0 : 1, 6, std::max(1, threads), &n_in, &n_out, err,
                                 sizeof err) != 0)
            die(err);
        if (timing_on()) fprintf(stderr, "[coverm-amd] filter %s -> %s: %llu of %llu records\n", in[k].c_str(), out[k].c_str(), (unsigned long long)n_out, (unsigned long long)n_in);
    }
    return 0;
}

int run_cli(int argc, char **argv) {
    const double t_main0 = now();
    Run R;
    Args &a = R.a;
    if (argc >= 2 && !strcmp(argv[1], "filter")) return run_filter(argc, argv);
    if (argc < 2 || (strcmp(argv[1], "contig") && strcmp(argv[1], "genome"))) {
        fprintf(stderr, "usage: coverm-amd contig|genome -b <bam>... [-m <methods>...] [options]   (see src/cli.rs of CoverM for the flags; "
                        "engine flags: --device N | --devices a,b,..., --no-stream)\n"
                        "       coverm-amd filter -b <bam>... -o <bam>... [thresholds] [--inverse]\n");
        return 2;
    }
    a.mode = argv[1];
    auto collect = [&](int &i, std::vector<std::string> &dst) { while (i + 1 < argc && argv[i + 1][0] != '-') dst.push_back(argv[++i]); };
    for (int i = 2; i < argc; i++) {
        const std::string k = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) die("missing value for " + k); return argv[++i]; };
        if (k == "-b" || k == "--bam-files") collect(i, a.bams);
        else if (k == "-m" || k == "--methods") collect(i, a.methods);
        else if (k == "--min-covered-fraction") a.min_covered_fraction = val();
        else if (k == "--contig-end-exclusion") a.contig_end_exclusion = parse_uint(k, val(), ~0ull);
        else if (k == "--trim-min") a.trim_min = val();
        else if (k == "--trim-max") a.trim_max = val();
        else if (k == "--output-format") a.output_format = val();
        else if (k == "-o" || k == "--output-file") a.output_file = val();
        else if (k == "--no-zeros") a.no_zeros = true;
        else if (k == "--proper-pairs-only") a.proper_pairs_only = true;
        else if (k == "--exclude-supplementary") a.exclude_supplementary = true;
        else if (k == "--include-secondary") a.include_secondary = true;
        else if (k == "--min-read-aligned-length") a.min_aligned_length = (uint32_t)parse_uint(k, val(), 0xffffffffull);
        else if (k == "--min-read-percent-identity") a.min_pid = val();
        else if (k == "--min-read-aligned-percent") a.min_aligned_pct = val();
        else if (k == "--min-read-aligned-length-pair") a.min_aligned_length_pair = (uint32_t)parse_uint(k, val(), 0xffffffffull);
        else if (k == "--min-read-percent-identity-pair") a.min_pid_pair = val();
        else if (k == "--min-read-aligned-percent-pair") a.min_aligned_pct_pair = val();
        else if (k == "--min-mapq") a.min_mapq = (int)parse_uint(k, val(), 255);
        else if (k == "-s" || k == "--separator") { a.separator = val()[0]; a.have_separator = true; }
        else if (k == "--single-genome") a.single_genome = true;
        else if (k == "--genome-definition") a.genome_definition = val();
        else if (k == "--gff") a.gff = val();
        else if (k == "--gff-feature-type") { a.gff_feature_type = val(); a.have_gff_feature_type = true; }
        else if (k == "-t" || k == "--threads") a.threads = (int)parse_uint(k, val(), 65535);
        else if (k == "--device") { a.devices.assign(1, (int)parse_uint(k, val(), 1023)); }
        else if (k == "--devices") {   // 0,1,2 or 0-7
            a.devices.clear();
            std::string v = val();
            size_t p = 0;
            while (p < v.size()) {
                size_t c = v.find(',', p); if (c == std::string::npos) c = v.size();
                const std::string tok = v.substr(p, c - p);
                const size_t dash = tok.find('-');
                if (dash != std::string::npos && dash > 0) { for (int d = atoi(tok.substr(0, dash).c_str()); d <= atoi(tok.substr(dash + 1).c_str()); d++) a.devices.push_back(d); }
                else if (!tok.empty()) a.devices.push_back(atoi(tok.c_str()));
                p = c + 1;
            }
            if (a.devices.empty()) die("--devices needs a list such as 0,1,2,3 or 0-7");
        }
        else if (k == "--no-stream") a.no_stream = true;
        else if (k == "-v" || k == "--verbose" || k == "-q" || k == "--quiet") {}   // logging verbosity: nothing to tune here
        else die("unknown argument " + k);
    }
    if (a.bams.empty()) die("--bam-files is required (read mapping is out of scope for this engine)");
    if (a.devices.empty()) a.devices.push_back(0);
    R.contig = a.mode == "contig";
    const bool contig = R.contig;
    if (a.methods.empty()) a.methods.push_back(contig ? "mean" : "relative_abundance");   // cli.rs:2521, 2048
    if (!a.min_covered_fraction) a.min_covered_fraction = contig ? "0" : "10";            // cli.rs:2528, 2065

    // ---- EstimatorsAndTaker::generate_from_clap
    const float mcf = parse_percentage(a.min_covered_fraction, "--min-covered-fraction");
    const uint64_t excl = a.contig_end_exclusion;
    std::vector<covh_estimator> &est = R.est;
    std::vector<int64_t> norm;
    int64_t rpkm = -1, tpm = -1;
    int printer = 0, taker_kind = COVH_TAKER_STREAM;
    auto E = [&](int kind, float mf, uint64_t ex, float t0 = 0, float t1 = 0) {
        covh_estimator e; e.kind = kind; e.min_fraction_covered_bases = mf; e.contig_end_exclusion = ex;
        e.exclude_mismatches = 0; e.trim_min = t0; e.trim_max = t1; est.push_back(e);
    };
    Filter &f = R.f;
    f.improper = !a.proper_pairs_only; f.supp = !a.exclude_supplementary; f.sec = a.include_secondary;
    f.len_single = a.min_aligned_length; f.pid_single = parse_percentage(a.min_pid, "--min-read-percent-identity"); f.pct_single = parse_percentage(a.min_aligned_pct, "--min-read-aligned-percent");
    f.mapq = a.min_mapq; f.len_pair = a.min_aligned_length_pair; f.pid_pair = parse_percentage(a.min_pid_pair, "--min-read-percent-identity-pair");
    f.pct_pair = parse_percentage(a.min_aligned_pct_pair, "--min-read-aligned-percent-pair");
    const bool metabat = a.methods.size() == 1 && a.methods[0] == "metabat";
    for (auto &m : a.methods) if (m == "metabat" && a.methods.size() > 1) die("Cannot specify the metabat method with any other coverage methods");
    if (metabat) {
        E(COVH_LENGTH, 0, 0); E(COVH_MEAN, mcf, excl); E(COVH_VARIANCE, mcf, excl);
        taker_kind = COVH_TAKER_CACHED; printer = 3;
        f.pid_single = 0.97001f; f.improper = f.supp = f.sec = true;   // coverm.rs:1680-1693
    } else {
        for (size_t i = 0; i < a.methods.size(); i++) {
            const std::string &m = a.methods[i];
            if (m == "mean") E(COVH_MEAN, mcf, excl);
            else if (m == "coverage_histogram") E(COVH_PILEUP_COUNTS, mcf, excl);
            else if (m == "trimmed_mean") E(COVH_TRIMMED_MEAN, mcf, excl, parse_percentage(a.trim_min, "--trim-min"), parse_percentage(a.trim_max, "--trim-max"));
            else if (m == "covered_fraction") E(COVH_COVERED_FRACTION, mcf, 0);
            else if (m == "covered_bases") E(COVH_COVERED_BASES, mcf, 0);
            else if (m == "rpkm") { if (rpkm >= 0) die("The RPKM column cannot be specified more than once"); rpkm = (int64_t)i; E(COVH_RPKM, mcf, 0); }
            else if (m == "tpm") { if (tpm >= 0) die("The TPM column cannot be specified more than once"); tpm = (int64_t)i; E(COVH_TPM, mcf, 0); }
            else if (m == "variance") E(COVH_VARIANCE, mcf, excl);
            else if (m == "length") E(COVH_LENGTH, 0, 0);
            else if (m == "relative_abundance") { norm.push_back((int64_t)i); E(COVH_MEAN, mcf, excl); }
            else if (m == "count") E(COVH_READ_COUNT, 0, 0);
            else if (m == "reads_per_base") E(COVH_READS_PER_BASE, 0, 0);
            else if (m == "anir") E(COVH_ANIR, 0, 0);
            else die("unknown method " + m);
        }
        bool hist_method = false;
        for (auto &m : a.methods) hist_method |= m == "coverage_histogram";
        if (hist_method) {
            if (a.methods.size() > 1) die("Cannot specify the coverage_histogram method with any other coverage methods");
            taker_kind = COVH_TAKER_PILEUP; printer = 0;
        } else if (norm.empty() && rpkm < 0 && tpm < 0 && a.output_format == "sparse") { taker_kind = COVH_TAKER_STREAM; printer = 0; }
        else { taker_kind = COVH_TAKER_CACHED; printer = a.output_format == "sparse" ? 1 : 2; }
        if (mcf != 0.0f)
            for (auto &e : est)
                if (e.kind == COVH_READ_COUNT || e.kind == COVH_LENGTH || e.kind == COVH_READS_PER_BASE || e.kind == COVH_ANIR)
                    die("this coverage estimator cannot be used when --min-covered-fraction is > 0");
    }
    static const char *HDR[] = {"Mean", "Trimmed Mean", "Coverage\tBases", "Covered Fraction", "Covered Bases", "RPKM", "TPM",
                                "Variance", "Length", "Read Count", "Reads per base", "ANIr"};
    std::vector<std::string> headers;
    for (auto &e : est) {
        if (e.kind == COVH_PILEUP_COUNTS) { headers.push_back("Coverage"); headers.push_back("Bases"); }
        else headers.push_back(HDR[e.kind]);
    }
    for (int64_t i : norm) headers[(size_t)i] = "Relative Abundance (%)";
    std::vector<const char *> hptr;
    for (auto &h : headers) hptr.push_back(h.c_str());
    const char *entry_type = contig ? "Contig" : "Genome";
    R.per_gene = !a.gff.empty();                                    // coverm.rs:488-518, 1557-1590
    if (R.per_gene) entry_type = contig ? "Gene\tContig" : "Gene\tContig\tGenome";
    covh_taker *taker = R.taker = covh_taker_new(taker_kind, est.size());
    struct TakerFree { covh_taker *t; ~TakerFree() { covh_taker_free(t); } } taker_free{taker};
    covh_print_headers(taker, printer, entry_type, hptr.data(), hptr.size());

    if (R.per_gene) {
        if (a.methods.size() == 1 && a.methods[0] == "metabat") die("The metabat method cannot be used with --gff");
        char gerr[512] = {0};
        R.genes = covh_genes_read_gff(a.gff.c_str(), a.have_gff_feature_type ? a.gff_feature_type.c_str() : nullptr, gerr, sizeof gerr);
        if (!R.genes) die(gerr);
    }
    struct GenesFree { covh_genes *&g; ~GenesFree() { if (g) covh_genes_free(g); } } genes_free{R.genes};
    // ---- genome definition
    std::vector<std::string> &genomes = R.genomes;
    R.by_names = !contig && !a.have_separator && !a.single_genome;
    if (R.by_names) {
        if (a.genome_definition.empty()) die("genome mode over BAM files needs --separator, --single-genome or --genome-definition");
        FILE *fh = fopen(a.genome_definition.c_str(), "r");
        if (!fh) die("cannot open " + a.genome_definition);
        struct FClose { FILE *f; ~FClose() { fclose(f); } } fclose_{fh};
        std::vector<char> linebuf(1 << 16);
        char *line = linebuf.data();
        std::unordered_map<std::string, int32_t> gi;
        auto is_ws = [](unsigned char ch) { return ch == ' ' || (ch >= 9 && ch <= 13); };
        while (fgets(line, (int)linebuf.size(), fh)) {   // read_genome_definition_file, genome_parsing.rs:71-141
            std::string l(line);
            if (!l.empty() && l.back() == '\n') l.pop_back();
            if (!l.empty() && l.back() == '\r') l.pop_back();
            const size_t t = l.find('\t');
            if (t == std::string::npos || l.find('\t', t + 1) != std::string::npos)   // blank lines included (:116-124)
                die("The line \"" + l + "\" in the genome definition file is not a genome name and contig name separated by a tab");
            std::string g = l.substr(0, t);
            { size_t x = 0, y = g.size(); while (x < y && is_ws((unsigned char)g[x])) x++; while (y > x && is_ws((unsigned char)g[y - 1])) y--; g = g.substr(x, y - x); }
            size_t x = t + 1;
            while (x < l.size() && is_ws((unsigned char)l[x])) x++;
            size_t y = x;
            while (y < l.size() && !is_ws((unsigned char)l[y])) y++;
            if (x == y) die("Failed to split contig name by whitespace in genome definition file");
            const std::string c = l.substr(x, y - x);                                  // first token: comments after it are dropped
            auto it = gi.find(g);
            if (it == gi.end()) { it = gi.emplace(g, (int32_t)genomes.size()).first; genomes.push_back(g); }
            auto cit = R.c2g.find(c);
            if (cit != R.c2g.end() && cit->second != it->second) die("The contig name '" + c + "' was assigned to multiple genomes");
            if (cit == R.c2g.end()) R.c2g[c] = it->second;
        }
    }

    // ---- sessions: one per device, brought up in parallel
    R.want = covh_wants(est.data(), est.size());
    if (R.want & COV_WANT_IDENTITY)   // contig.rs:208 / genome.rs:724 use the primary-read sum, genome.rs:220 the not-supplementary one
        R.want |= R.by_names ? COV_WANT_IDENTITY_NONSUPP_ONLY : COV_WANT_IDENTITY_PRIMARY_ONLY;
    if (f.doing_filtering()) f.mode(R.fs, R.fp);
    cov_config &cfg = R.cfg; memset(&cfg, 0, sizeof cfg);
    cfg.include_improper_pairs = f.improper; cfg.include_supplementary = f.supp;
    cfg.include_secondary = f.sec; cfg.min_mapq = 255; cfg.contig_end_exclusion = excl; cfg.want = R.want;
    if (f.doing_filtering() && R.fs && !R.fp) {
        cfg.filter_single = 1; cfg.min_mapq = (uint8_t)f.mapq; cfg.min_aligned_length = f.len_single;
        cfg.min_percent_identity = f.pid_single; cfg.min_aligned_percent = f.pct_single;
    }
    const size_t nd = a.devices.size(), nb = a.bams.size();
    if (!a.no_stream && !R.per_gene && !no_gpu_ingest())
        for (size_t i = 0; i < std::min<size_t>(nb, std::max<size_t>(nd, 2)); i++) {      // (the file type is checked by covh_bam_read_header itself)
            const std::string path = a.bams[i];
            if (R.hdr_ahead.count(path) || !is_bgzf(path)) continue;
            R.hdr_ahead.emplace(path, std::async(std::launch::async, [path] {
                HeaderAhead h; char e[512] = {0};
                h.hd = covh_bam_read_header(path.c_str(), e, sizeof e);
                if (!h.hd) h.err = e;
                return h;
            }));
        }
    struct HdrDrain { Run &R; ~HdrDrain() { for (auto &kv : R.hdr_ahead) { HeaderAhead h = kv.second.get(); if (h.hd) covh_bam_header_free(h.hd); } } } hdr_drain{R};
    std::vector<cov_session *> sess(nd, nullptr);
    struct SessFree { std::vector<cov_session *> &v; ~SessFree() { if (!g_skip_teardown.load()) for (auto *s : v) if (s) cov_destroy(s); } } sess_free{sess};
    {
        std::vector<std::thread> th;
        std::vector<cov_status> rc(nd, COV_OK);
        std::vector<std::string> emsg(nd);
        std::mutex em;
        for (size_t d = 0; d < nd; d++)
            th.emplace_back([&, d] {
                cov_config c = cfg; c.device = a.devices[d];
                long long prep = 1;
                (void)covknob::get("ingest_prepare", prep);
                if (!no_gpu_ingest() && prep) c.want |= COV_WANT_INGEST;      // the ingest's streams and events come up beside the rest of the start-up
                rc[d] = cov_create(&c, &sess[d]);
                if (rc[d] != COV_OK) { std::lock_guard<std::mutex> lk(em); emsg[d] = cov_last_error(nullptr); }
            });
        for (auto &t : th) t.join();
        for (size_t d = 0; d < nd; d++) if (rc[d] != COV_OK) die(emsg[d]);
    }
    {
        // CoverageEstimator::calculate_coverage on the device for `coverm contig` (one entry per contig) when every estimator asked for is
        // one the device evaluates (all but TPM and the coverage histogram); COVERM_HOST_ESTIMATES=1 keeps the host's evaluation
        static_assert(sizeof(covh_estimator) == sizeof(cov_estimator) && offsetof(covh_estimator, trim_max) == offsetof(cov_estimator, trim_max), "estimator structs share one layout");
        bool ok = contig && !R.per_gene && !getenv("COVERM_HOST_ESTIMATES") && !est.empty() && est.size() <= COV_EST_MAX;
        for (const covh_estimator &e : est) ok = ok && e.kind != COVH_TPM && e.kind != COVH_PILEUP_COUNTS;
        R.dev_est = ok;
        if (ok) for (size_t d = 0; d < nd; d++) check(sess[d], cov_set_estimators(sess[d], reinterpret_cast<const cov_estimator *>(est.data()), (uint32_t)est.size()));
    }
    const double t_sessions = now();
    covh_bam_set_pinned(1);
    // (measured, profiles/r03_tail_variants.log: releasing the staging slots beside the last rounds shortens the exit by ~0.03 s and
    // lengthens the tail by as much — hipHostFree waits for the device — so it stays opt-in)
    covh_bam_set_release_staging(0);
    if (nb > 1) covh_bam_set_buffer_cache(1);
    covh_bam_set_concurrent_feeders((int)std::min(nd, span_mode_feeders(nb, nd)));      // > 2 at once: mapped files, registered up front (host memory traffic / 3)
    const bool timing = timing_on();
    std::vector<Sample> samples(nb);
    for (size_t i = 0; i < nb; i++) samples[i].path = a.bams[i];
    std::mutex err_mutex; std::string first_error;
    size_t n_errors = 0, n_span_unsorted = 0;      // a span-mode fallback is taken only when EVERY failed span failed on the order of its keys
    auto guarded = [&](auto fn) {
        try { fn(); }
        catch (const SpanUnsorted &e) { std::lock_guard<std::mutex> lk(err_mutex); n_errors++; n_span_unsorted++; if (first_error.empty()) first_error = e.what(); }
        catch (const Fatal &e) { std::lock_guard<std::mutex> lk(err_mutex); n_errors++; if (first_error.empty() || n_span_unsorted == n_errors - 1) first_error = e.what(); }
        catch (const std::exception &e) { std::lock_guard<std::mutex> lk(err_mutex); n_errors++; if (first_error.empty() || n_span_unsorted == n_errors - 1) first_error = e.what(); }
    };
    const bool span_mode = nd > 1 && nb < nd;
    if (!span_mode) {
        // samples dealt to devices; host threads shared between the concurrently decoded files
        const size_t lanes = std::min(nd, nb);
        // every device's reader gets at least six threads (four of them read): with -t divided evenly, eight devices under a 16-CPU
        // quota had two threads each and a reader that cannot fill its link; the readers block in pread and in slot waits most of
        // the time, so oversubscribing the CPUs costs less than starving a device
        const int thr = std::min(std::max(1, a.threads), std::max(6, a.threads / (int)lanes));
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (size_t d = 0; d < lanes; d++)
            th.emplace_back([&, d] {
                (void)cov_bind_thread_to_device_node(a.devices[d]);
                guarded([&] {
                    for (;;) {
                        const size_t bi = next.fetch_add(1);
                        if (bi >= nb) break;
                        { std::lock_guard<std::mutex> lk(err_mutex); if (!first_error.empty()) break; }
                        ingest(R, sess[d], samples[bi], thr, 0, 1);
                    }
                });
            });
        for (auto &t : th) t.join();
        if (!first_error.empty()) die(first_error);
    } else {
        // every BAM cut into nd tid spans; the per-contig result blocks meet on device 0 through one RCCL gather
        const int thr = std::min(std::max(1, a.threads), std::max(6, a.threads / (int)nd));      // (as above)
        for (size_t bi = 0; bi < nb; bi++) {
            std::vector<Sample> part(nd);
            std::vector<std::thread> th;
            for (size_t d = 0; d < nd; d++) {
                part[d].path = a.bams[bi];
                th.emplace_back([&, d] { (void)cov_bind_thread_to_device_node(a.devices[d]); guarded([&] { ingest(R, sess[d], part[d], thr, (uint32_t)d, (uint32_t)nd); }); });
            }
            for (auto &t : th) t.join();
            if (!first_error.empty()) {
                // A span trusts the file's order and refuses a file whose keys decrease anywhere — any record, also one the scan would
                // skip.  The reference only compares the tids of mapped records that passed the flag filters (contig.rs:118-132), so a
                // file it accepts can be refused here: such a file goes through ONE device whole, where cov_finish judges its order with
                // the reference's rule (and ends in the same error if it really is unsorted).
                if (n_span_unsorted != n_errors) die(first_error);      // some span failed for another reason: that error is the run's (first_error holds it)
                first_error.clear(); n_errors = n_span_unsorted = 0;
                if (timing) fprintf(stderr, "[coverm-amd] %s: keys decrease inside a span; the file goes through one device whole\n", a.bams[bi].c_str());
                ingest(R, sess[0], samples[bi], a.threads, 0, 1);
                continue;
            }
            Sample &S = samples[bi];
            S.stoit = part[0].stoit; S.names_blob = part[0].names_blob; S.name_off = part[0].name_off; S.tlen = part[0].tlen;
            S.genome_of_tid = part[0].genome_of_tid; S.streamed = true;
            const uint32_t nt = (uint32_t)S.tlen.size();
            check(sess[0], cov_gather(sess.data(), (uint32_t)nd, 0));
            S.stats.assign(nt, cov_contig_stats{});
            if (R.dev_est) S.estimates.assign((size_t)nt * est.size(), 0.0f);
            std::vector<cov_contig_stats> tmp(nt);
            for (size_t d = 0; d < nd; d++) {
                cov_summary summ;
                check(sess[0], cov_gathered(sess[0], (uint32_t)d, tmp.data(), &summ));
                S.prim += summ.num_detected_primary_alignments; S.n_records += summ.n_records;
                S.peak_bytes = std::max(S.peak_bytes, part[d].peak_bytes);
                S.t_ingest = std::max(S.t_ingest, part[d].t_ingest); S.t_finish = std::max(S.t_finish, part[d].t_finish);
                for (uint32_t t = 0; t < nt; t++) {
                    if (tmp[t].n_pass == 0) continue;
                    if (S.stats[t].n_pass != 0) die("internal error: contig " + S.target_name(t) + " was seen by two spans");
                    S.stats[t] = tmp[t];
                    if (R.dev_est) {                // the floats stay with the rank that evaluated them, like the histogram bins
                        std::copy(part[d].estimates.begin() + (size_t)t * est.size(), part[d].estimates.begin() + (size_t)(t + 1) * est.size(), S.estimates.begin() + (size_t)t * est.size());
                    } else if (R.want & COV_WANT_HIST) {   // histogram bins stay with the rank that built them: re-based into one array
                        S.stats[t].hist_off = S.hist.size();
                        S.hist.insert(S.hist.end(), part[d].hist.begin() + tmp[t].hist_off, part[d].hist.begin() + tmp[t].hist_off + tmp[t].hist_len);
                    }
                }
            }
        }
    }
    const double t_ingested = now();
    if (timing) {
        // where the resident set comes from (VmHWM = peak; RssShmem counts page-locked / device-visible mappings of the HIP runtime)
        if (FILE *ps = fopen("/proc/self/status", "r")) {
            char ln[256];
            while (fgets(ln, sizeof ln, ps))
                if (!strncmp(ln, "VmHWM", 5) || !strncmp(ln, "VmRSS", 5) || !strncmp(ln, "RssAnon", 7) || !strncmp(ln, "RssFile", 7) || !strncmp(ln, "RssShmem", 8)) {
                    ln[strcspn(ln, "\n")] = 0;
                    fprintf(stderr, "[coverm-amd] %s\n", ln);
                }
            fclose(ps);
        }
        // ... and its eight largest mappings (what the kernel has to take apart when the process ends)
        if (FILE *sm = fopen("/proc/self/smaps", "r")) {
            struct M { unsigned long long rss_kb, size_kb; std::string what; };
            std::vector<M> ms;
            char ln[512]; std::string cur; unsigned long long size_kb = 0;
            while (fgets(ln, sizeof ln, sm)) {
                unsigned long long a, b2, v;
                if (sscanf(ln, "%llx-%llx ", &a, &b2) == 2 && strchr(ln, '-') && strchr(ln, '-') < ln + 17) {
                    ln[strcspn(ln, "\n")] = 0;
                    const char *path = strchr(ln, '/'); const char *br = strchr(ln, '[');
                    cur = path ? path : br ? br : "(anonymous)"; size_kb = (b2 - a) >> 10;
                } else if (sscanf(ln, "Rss: %llu kB", &v) == 1) ms.push_back({v, size_kb, cur});
            }
            fclose(sm);
            std::sort(ms.begin(), ms.end(), [](const M &x, const M &y) { return x.rss_kb > y.rss_kb; });
            for (size_t i = 0; i < ms.size() && i < 8; i++)
                fprintf(stderr, "[coverm-amd] mapping %zu: %llu MB resident of %llu MB, %s\n", i, ms[i].rss_kb >> 10, ms[i].size_kb >> 10, ms[i].what.c_str());
        }
    }
    if (timing)
        for (auto &S : samples)
            fprintf(stderr, "[coverm-amd] sample %s: %s, open %.3fs, ingest (decode+push) %.3fs, finish+fetch %.3fs, %llu records, reader buffers %.0f MB\n", S.stoit.c_str(),
                    S.device_ingest ? "device ingest" : S.streamed ? "streamed" : "whole file", S.t_open, S.t_ingest, S.t_finish, (unsigned long long)S.n_records, S.peak_bytes / 1e6);

    // ---- scan drivers: one call per BAM, each with its own header (contig.rs:29-32)
    std::vector<covh_reads_mapped> rm(nb);
    for (size_t bi = 0; bi < nb; bi++) {
        const Sample &S = samples[bi];
        if (R.per_gene) { rm[bi] = S.gene_rm; continue; }
        const covh_header hdr = S.header();
        covh_sample hs; hs.stoit_name = S.stoit.c_str(); hs.stats = S.stats.data(); hs.hist = S.hist.empty() ? nullptr : S.hist.data();
        hs.num_detected_primary_alignments = S.prim;
        int rc;
        const float *ef = S.estimates.empty() ? nullptr : S.estimates.data();
        if (contig) rc = covh_contig_coverage_estimated(&hdr, &hs, 1, taker, est.data(), est.size(), !a.no_zeros, &rm[bi], R.dev_est ? &ef : nullptr);
        else if (a.have_separator || a.single_genome)
            rc = covh_genome_coverage_separator(&hdr, &hs, 1, (uint8_t)(a.single_genome ? '0' : a.separator), taker, !a.no_zeros, est.data(), est.size(),
                                                a.single_genome, &rm[bi]);
        else {
            std::vector<const char *> gn;
            for (auto &g : genomes) gn.push_back(g.c_str());
            rc = covh_genome_coverage_with_contig_names(&hdr, &hs, 1, S.genome_of_tid.data(), gn.data(), gn.size(), taker, !a.no_zeros, est.data(),
                                                        est.size(), &rm[bi]);
        }
        if (rc != COV_OK) die(covh_last_error());
        if (covh_taker_names_mismatch(taker))   // coverage_takers.rs:140-148
            die("Found a difference amongst the reference sets used for mapping. For this (non-streaming) usage of CoverM, all BAM files must have the "
                "same set of reference sequences.");
    }
    for (size_t i = 0; i < nb; i++)   // contig.rs:233-240
        fprintf(stderr, "[coverm-amd] In sample '%s', found %llu reads mapped out of %llu total (%.2f%%)\n", samples[i].stoit.c_str(),
                (unsigned long long)rm[i].num_mapped_reads, (unsigned long long)rm[i].num_reads,
                (double)(rm[i].num_mapped_reads * 100) / (double)rm[i].num_reads);
    const double t_scanned = now();
    covh_finalise_printing(taker, printer, entry_type, hptr.data(), hptr.size(), rm.data(), rm.size(), norm.data(), norm.size(), rpkm, tpm);
    const double t_printed = now();
    size_t len = 0;
    const char *txt = covh_taker_text(taker, &len);
    FILE *out = a.output_file.empty() || a.output_file == "-" ? stdout : fopen(a.output_file.c_str(), "w");
    if (!out) die("Failed to create output file: " + a.output_file);
    fwrite(txt, 1, len, out);
    if (out != stdout) fclose(out); else fflush(stdout);
    if (timing)
        fprintf(stderr, "[coverm-amd] main: arguments + device sessions %.3fs, samples %.3fs, scan drivers + table %.3fs (scan drivers %.3fs, printer %.3fs, %zu bytes written in %.3fs; process start-up and exit are outside)\n",
                t_sessions - t_main0, t_ingested - t_sessions, now() - t_ingested, t_scanned - t_ingested, t_printed - t_scanned, len, now() - t_printed);
    return 0;
}

}  // namespace

extern "C" void covh_cli_set_fast_exit(int on) { g_skip_teardown.store(on != 0); }

extern "C" int covh_cli_main(int argc, char **argv) {
    try {
        return run_cli(argc, argv);
    } catch (const Fatal &e) {
        fprintf(stderr, "[coverm-amd] ERROR: %s\n", e.what());
        return 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "[coverm-amd] ERROR: %s\n", e.what());
        return 1;
    }
}
