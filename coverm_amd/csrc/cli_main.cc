// coverm-amd — `coverm contig` / `coverm genome` over --bam-files on the MI355X engine.
//
// Mirrors the reference orchestrator for this path (src/bin/coverm.rs): FilterParameters::generate_from_clap
// :1659-1678 + doing_filtering :1695-1703, EstimatorsAndTaker::generate_from_clap :1315-1504, print_headers
// :1506-1519, run_contig :2088-2131, run_genome :1539-1628, parse_percentage :1296-1312, parse_separator
// :1522-1537; flag names and defaults from src/cli.rs (contig :2264-2582, genome :1669-2263).
// Everything else the reference binary does (mapping, indexing, filter/make/cluster subcommands) is out of scope.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/coverm_host.h"

namespace {

[[noreturn]] void die(const std::string &m) { fprintf(stderr, "[coverm-amd] ERROR: %s\n", m.c_str()); exit(1); }

float parse_percentage(const char *v) {   // coverm.rs:1296-1312
    if (!v) return 0.0f;
    float p = strtof(v, nullptr);
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
    bool single_genome = false, have_separator = false;
    char separator = '~';
    uint32_t min_aligned_length = 0, min_aligned_length_pair = 0;
    const char *min_pid = nullptr, *min_aligned_pct = nullptr, *min_pid_pair = nullptr, *min_aligned_pct_pair = nullptr;
    int min_mapq = 255, threads = 1, device = 0;
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

struct Sample {
    std::string stoit;
    std::vector<cov_contig_stats> stats;
    std::vector<uint64_t> hist;
    uint64_t prim = 0;
};

void check(cov_session *s, cov_status st) { if (st != COV_OK) die(cov_last_error(s)); }

}  // namespace

int main(int argc, char **argv) {
    Args a;
    if (argc < 2 || (strcmp(argv[1], "contig") && strcmp(argv[1], "genome"))) {
        fprintf(stderr, "usage: coverm-amd contig|genome -b <bam>... [-m <methods>...] [options]   (see src/cli.rs of CoverM for the flags)\n");
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
        else if (k == "--contig-end-exclusion") a.contig_end_exclusion = strtoull(val(), nullptr, 10);
        else if (k == "--trim-min") a.trim_min = val();
        else if (k == "--trim-max") a.trim_max = val();
        else if (k == "--output-format") a.output_format = val();
        else if (k == "-o" || k == "--output-file") a.output_file = val();
        else if (k == "--no-zeros") a.no_zeros = true;
        else if (k == "--proper-pairs-only") a.proper_pairs_only = true;
        else if (k == "--exclude-supplementary") a.exclude_supplementary = true;
        else if (k == "--include-secondary") a.include_secondary = true;
        else if (k == "--min-read-aligned-length") a.min_aligned_length = (uint32_t)strtoul(val(), nullptr, 10);
        else if (k == "--min-read-percent-identity") a.min_pid = val();
        else if (k == "--min-read-aligned-percent") a.min_aligned_pct = val();
        else if (k == "--min-read-aligned-length-pair") a.min_aligned_length_pair = (uint32_t)strtoul(val(), nullptr, 10);
        else if (k == "--min-read-percent-identity-pair") a.min_pid_pair = val();
        else if (k == "--min-read-aligned-percent-pair") a.min_aligned_pct_pair = val();
        else if (k == "--min-mapq") a.min_mapq = atoi(val());
        else if (k == "-s" || k == "--separator") { a.separator = val()[0]; a.have_separator = true; }
        else if (k == "--single-genome") a.single_genome = true;
        else if (k == "--genome-definition") a.genome_definition = val();
        else if (k == "--gff") a.gff = val();
        else if (k == "--gff-feature-type") { a.gff_feature_type = val(); a.have_gff_feature_type = true; }
        else if (k == "-t" || k == "--threads") a.threads = atoi(val());
        else if (k == "--device") a.device = atoi(val());
        else if (k == "-v" || k == "--verbose" || k == "-q" || k == "--quiet") {}   // logging verbosity: nothing to tune here
        else die("unknown argument " + k);
    }
    if (a.bams.empty()) die("--bam-files is required (read mapping is out of scope for this engine)");
    const bool contig = a.mode == "contig";
    if (a.methods.empty()) a.methods.push_back(contig ? "mean" : "relative_abundance");   // cli.rs:2521, 2048
    if (!a.min_covered_fraction) a.min_covered_fraction = contig ? "0" : "10";            // cli.rs:2528, 2065

    // ---- EstimatorsAndTaker::generate_from_clap
    const float mcf = parse_percentage(a.min_covered_fraction);
    const uint64_t excl = a.contig_end_exclusion;
    std::vector<covh_estimator> est;
    std::vector<int64_t> norm;
    int64_t rpkm = -1, tpm = -1;
    int printer = 0, taker_kind = COVH_TAKER_STREAM;
    auto E = [&](int kind, float mf, uint64_t ex, float t0 = 0, float t1 = 0) {
        covh_estimator e; e.kind = kind; e.min_fraction_covered_bases = mf; e.contig_end_exclusion = ex;
        e.exclude_mismatches = 0; e.trim_min = t0; e.trim_max = t1; est.push_back(e);
    };
    Filter f;
    f.improper = !a.proper_pairs_only; f.supp = !a.exclude_supplementary; f.sec = a.include_secondary;
    f.len_single = a.min_aligned_length; f.pid_single = parse_percentage(a.min_pid); f.pct_single = parse_percentage(a.min_aligned_pct);
    f.mapq = a.min_mapq; f.len_pair = a.min_aligned_length_pair; f.pid_pair = parse_percentage(a.min_pid_pair);
    f.pct_pair = parse_percentage(a.min_aligned_pct_pair);
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
            else if (m == "trimmed_mean") E(COVH_TRIMMED_MEAN, mcf, excl, parse_percentage(a.trim_min), parse_percentage(a.trim_max));
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
    const bool per_gene = !a.gff.empty();                           // coverm.rs:488-518, 1557-1590
    if (per_gene) entry_type = contig ? "Gene\tContig" : "Gene\tContig\tGenome";
    covh_taker *taker = covh_taker_new(taker_kind, est.size());
    covh_print_headers(taker, printer, entry_type, hptr.data(), hptr.size());

    covh_genes *genes = nullptr;
    if (per_gene) {
        if (a.methods.size() == 1 && a.methods[0] == "metabat") die("The metabat method cannot be used with --gff");
        char gerr[512] = {0};
        genes = covh_genes_read_gff(a.gff.c_str(), a.have_gff_feature_type ? a.gff_feature_type.c_str() : nullptr, gerr, sizeof gerr);
        if (!genes) die(gerr);
    }
    // ---- genome definition
    std::vector<std::string> genomes;
    std::unordered_map<std::string, int32_t> c2g;
    const bool by_names = !contig && !a.have_separator && !a.single_genome;
    if (by_names) {
        if (a.genome_definition.empty()) die("genome mode over BAM files needs --separator, --single-genome or --genome-definition");
        FILE *fh = fopen(a.genome_definition.c_str(), "r");
        if (!fh) die("cannot open " + a.genome_definition);
        char line[1 << 16];
        std::unordered_map<std::string, int32_t> gi;
        auto is_ws = [](unsigned char ch) { return ch == ' ' || (ch >= 9 && ch <= 13); };
        while (fgets(line, sizeof line, fh)) {   // read_genome_definition_file, genome_parsing.rs:71-141
            std::string l(line);
            if (!l.empty() && l.back() == '\n') l.pop_back();
            if (!l.empty() && l.back() == '\r') l.pop_back();
            const size_t t = l.find('\t');
            if (t == std::string::npos || l.find('\t', t + 1) != std::string::npos)   // blank lines included (:116-124)
                die("The line \"" + l + "\" in the genome definition file is not a genome name and contig name separated by a tab");
            std::string g = l.substr(0, t);
            { size_t a = 0, b = g.size(); while (a < b && is_ws((unsigned char)g[a])) a++; while (b > a && is_ws((unsigned char)g[b - 1])) b--; g = g.substr(a, b - a); }
            size_t a = t + 1;
            while (a < l.size() && is_ws((unsigned char)l[a])) a++;
            size_t b = a;
            while (b < l.size() && !is_ws((unsigned char)l[b])) b++;
            if (a == b) die("Failed to split contig name by whitespace in genome definition file");
            const std::string c = l.substr(a, b - a);                                  // first token: comments after it are dropped
            auto it = gi.find(g);
            if (it == gi.end()) { it = gi.emplace(g, (int32_t)genomes.size()).first; genomes.push_back(g); }
            auto cit = c2g.find(c);
            if (cit != c2g.end() && cit->second != it->second) die("The contig name '" + c + "' was assigned to multiple genomes");
            if (cit == c2g.end()) c2g[c] = it->second;
        }
        fclose(fh);
    }

    // ---- per BAM: decode (host threads), push, finish
    uint32_t want = covh_wants(est.data(), est.size());
    if (want & COV_WANT_IDENTITY)   // contig.rs:208 / genome.rs:724 use the primary-read sum, genome.rs:220 the not-supplementary one
        want |= by_names ? COV_WANT_IDENTITY_NONSUPP_ONLY : COV_WANT_IDENTITY_PRIMARY_ONLY;
    std::vector<Sample> samples(a.bams.size());
    std::vector<covh_reads_mapped> gene_rm;
    std::string names_blob; std::vector<uint32_t> name_off; std::vector<uint64_t> tlen;
    std::vector<int32_t> genome_of_tid;
    bool fs = false, fp = false;
    if (f.doing_filtering()) f.mode(fs, fp);
    cov_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.device = a.device; cfg.include_improper_pairs = f.improper; cfg.include_supplementary = f.supp;
    cfg.include_secondary = f.sec; cfg.min_mapq = 255; cfg.contig_end_exclusion = excl; cfg.want = want;
    if (f.doing_filtering() && fs && !fp) {
        cfg.filter_single = 1; cfg.min_mapq = (uint8_t)f.mapq; cfg.min_aligned_length = f.len_single;
        cfg.min_percent_identity = f.pid_single; cfg.min_aligned_percent = f.pct_single;
    }
    // The HIP runtime and the session come up on their own thread while the first file is being decoded, and a
    // decoder thread stays one file ahead of the GPU (the reference reads its BAMs strictly one after another,
    // contig.rs:29).
    cov_session *s = nullptr;
    cov_status create_rc = COV_OK;
    std::thread warm([&] { create_rc = cov_create(&cfg, &s); });
    struct Decoded { covh_bam *bam = nullptr; std::string err; };
    std::vector<Decoded> decoded(a.bams.size());
    std::mutex qm; std::condition_variable qcv;
    size_t produced = 0, consumed = 0;
    if (a.bams.size() > 1) covh_bam_set_buffer_cache(1);
    covh_bam_set_pinned(1);
    std::thread decoder([&] {
        for (size_t bi = 0; bi < a.bams.size(); bi++) {
            { std::unique_lock<std::mutex> lk(qm); qcv.wait(lk, [&] { return produced < consumed + 2; }); }
            char err[512] = {0};
            covh_bam *b = covh_bam_open(a.bams[bi].c_str(), a.threads, fp ? 1 : 0, err, sizeof err);
            { std::lock_guard<std::mutex> lk(qm); decoded[bi].bam = b; decoded[bi].err = err; produced = bi + 1; }
            qcv.notify_all();
            if (!b) return;
        }
    });
    decoder.detach();   // die() may exit while it is mid-file
    const bool timing = getenv("COVERM_CLI_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (size_t bi = 0; bi < a.bams.size(); bi++) {
        const double tw0 = now();
        { std::unique_lock<std::mutex> lk(qm); qcv.wait(lk, [&] { return produced > bi; }); }
        const double tw1 = now();
        covh_bam *bam = decoded[bi].bam;
        if (!bam) die(decoded[bi].err);
        const uint32_t nt = covh_bam_n_targets(bam);
        if (bi == 0) {
            name_off.push_back(0);
            for (uint32_t t = 0; t < nt; t++) { names_blob += covh_bam_target_name(bam, t); name_off.push_back((uint32_t)names_blob.size()); tlen.push_back(covh_bam_target_len(bam, t)); }
        } else if (nt != tlen.size()) die("all BAM files must have the same set of reference sequences");
        std::vector<uint8_t> mask;
        if (by_names) {
            genome_of_tid.assign(nt, -1); mask.assign(nt, 0);
            uint32_t in = 0;
            for (uint32_t t = 0; t < nt; t++) {
                auto it = c2g.find(covh_bam_target_name(bam, t));
                if (it != c2g.end()) { genome_of_tid[t] = it->second; mask[t] = 1; in++; }
            }
            if (!in) die("Error: There are no found reference sequences that are a part of a genome");
        }
        cov_batch batch; covh_bam_batch(bam, &batch);
        Sample &S = samples[bi];
        // stoit name = file stem (bam_generator.rs:358-365)
        { std::string p = a.bams[bi]; size_t sl = p.find_last_of('/'); if (sl != std::string::npos) p = p.substr(sl + 1);
          size_t dot = p.find_last_of('.'); S.stoit = dot == std::string::npos ? p : p.substr(0, dot); }
        cov_batch selected; memset(&selected, 0, sizeof selected);
        bool have_selected = false;
        bool prim_from_host = false;
        if (f.doing_filtering()) {
            if (!(fs && !fp)) {
                for (uint64_t i = 0; i < batch.n_records; i++) if (!(batch.flag[i] & 0x900)) S.prim++;   // filter.rs:129-131
                prim_from_host = true;
                covh_pair_filter pf; memset(&pf, 0, sizeof pf);
                pf.filter_single = fs; pf.min_mapq = (uint8_t)f.mapq; pf.min_aligned_length_single = f.len_single;
                pf.min_percent_identity_single = f.pid_single; pf.min_aligned_percent_single = f.pct_single;
                pf.min_aligned_length_pair = f.len_pair; pf.min_percent_identity_pair = f.pid_pair; pf.min_aligned_percent_pair = f.pct_pair;
                uint64_t *order = nullptr, n_order = 0;
                const int prc = covh_pair_mode_order(&batch, covh_bam_mtid(bam), covh_bam_qname_off(bam), covh_bam_qnames(bam), &pf,
                                                     a.threads, &order, &n_order);
                if (prc == COV_ERR_NM_MISSING) die("Mapping record encountered that does not have an 'NM' auxiliary tag in the SAM/BAM format");
                if (prc != COV_OK) die(prc == COV_ERR_NM_BADTYPE ? "Unexpected data type of NM aux tag" : "pair filter failed");
                if (covh_batch_select(&batch, order, n_order, a.threads, &selected) != COV_OK) die("pair filter: selection failed");
                covh_free(order);
                batch = selected; have_selected = true;
            }
        }
        if (bi == 0) {
            warm.join();
            if (create_rc != COV_OK) die(cov_last_error(nullptr));
        } else check(s, cov_reset(s));
        const double tw2 = now();
        check(s, cov_set_targets(s, nt, tlen.data()));
        if (by_names) check(s, cov_set_target_mask(s, mask.data()));
        check(s, cov_push_batch(s, &batch));
        const double tw3 = now();
        S.stats.resize(nt);
        cov_summary summ;
        check(s, cov_finish(s, S.stats.data(), &summ));
        if (want & COV_WANT_HIST) { S.hist.resize(summ.hist_total); check(s, cov_fetch_hist(s, S.hist.data())); }
        if (!prim_from_host) S.prim = summ.num_detected_primary_alignments;
        if (per_gene) {   // genes.rs:182-344: per-gene reductions over this sample's depth, while the session holds it
            covh_header gh; gh.n_targets = nt; gh.names = names_blob.c_str(); gh.name_off = name_off.data(); gh.target_len = tlen.data();
            covh_genome_namer nm; memset(&nm, 0, sizeof nm);
            std::vector<const char *> gn;
            for (auto &g : genomes) gn.push_back(g.c_str());
            if (!contig) {
                nm.mode = a.single_genome ? 1 : a.have_separator ? 2 : 3;
                nm.separator = (uint8_t)a.separator;
                nm.genome_of_tid = genome_of_tid.data(); nm.genome_names = gn.data();
            }
            gene_rm.resize(a.bams.size());
            auto depth_cb = [](void *ctx, uint32_t tid, int32_t *out) -> int { return (int)cov_copy_depth((cov_session *)ctx, tid, out); };
            const int grc = covh_gene_coverage(&gh, genes, &nm, S.stoit.c_str(), &batch, &cfg, getenv("COVERM_GENES_ON_HOST") ? nullptr : s, depth_cb, s, S.prim, taker, est.data(),
                                               est.size(), !a.no_zeros, &gene_rm[bi]);
            if (grc == COV_ERR_HIP || grc == COV_ERR_STATE) die(cov_last_error(s));
            if (grc != COV_OK) die(covh_last_error());
        }
        const double tw4 = now();
        if (have_selected) covh_batch_free(&selected);
        covh_bam_close(bam);
        if (timing) fprintf(stderr, "[coverm-amd] sample %zu: waited for decoder %.3fs, session ready %.3fs, push %.3fs, finish+fetch %.3fs, close %.3fs\n",
                            bi, tw1 - tw0, tw2 - tw1, tw3 - tw2, tw4 - tw3, now() - tw4);
        { std::lock_guard<std::mutex> lk(qm); consumed = bi + 1; }
        qcv.notify_all();
    }
    if (a.bams.empty()) warm.join();
    cov_destroy(s);

    covh_header hdr; hdr.n_targets = (uint32_t)tlen.size(); hdr.names = names_blob.c_str(); hdr.name_off = name_off.data(); hdr.target_len = tlen.data();
    std::vector<covh_sample> hs(samples.size());
    for (size_t i = 0; i < samples.size(); i++) {
        hs[i].stoit_name = samples[i].stoit.c_str(); hs[i].stats = samples[i].stats.data();
        hs[i].hist = samples[i].hist.empty() ? nullptr : samples[i].hist.data();
        hs[i].num_detected_primary_alignments = samples[i].prim;
    }
    std::vector<covh_reads_mapped> rm(samples.size());
    int rc;
    if (per_gene) { rm = gene_rm; rc = COV_OK; }
    else if (contig) rc = covh_contig_coverage(&hdr, hs.data(), hs.size(), taker, est.data(), est.size(), !a.no_zeros, rm.data());
    else if (a.have_separator || a.single_genome)
        rc = covh_genome_coverage_separator(&hdr, hs.data(), hs.size(), (uint8_t)(a.single_genome ? '0' : a.separator), taker,
                                            !a.no_zeros, est.data(), est.size(), a.single_genome, rm.data());
    else {
        std::vector<const char *> gn;
        for (auto &g : genomes) gn.push_back(g.c_str());
        rc = covh_genome_coverage_with_contig_names(&hdr, hs.data(), hs.size(), genome_of_tid.data(), gn.data(), gn.size(), taker,
                                                    !a.no_zeros, est.data(), est.size(), rm.data());
    }
    if (rc != COV_OK) die(covh_last_error());
    for (size_t i = 0; i < samples.size(); i++)   // contig.rs:233-240
        fprintf(stderr, "[coverm-amd] In sample '%s', found %llu reads mapped out of %llu total (%.2f%%)\n", samples[i].stoit.c_str(),
                (unsigned long long)rm[i].num_mapped_reads, (unsigned long long)rm[i].num_reads,
                (double)(rm[i].num_mapped_reads * 100) / (double)rm[i].num_reads);
    covh_finalise_printing(taker, printer, entry_type, hptr.data(), hptr.size(), rm.data(), rm.size(), norm.data(), norm.size(), rpkm, tpm);
    size_t len = 0;
    const char *txt = covh_taker_text(taker, &len);
    FILE *out = a.output_file.empty() || a.output_file == "-" ? stdout : fopen(a.output_file.c_str(), "w");
    if (!out) die("Failed to create output file: " + a.output_file);
    fwrite(txt, 1, len, out);
    if (out != stdout) fclose(out);
    covh_taker_free(taker);
    return 0;
}
