/* A plain C99 consumer of include/covermhip.h + include/coverm_host.h: what a Rust/C host links against.
 * Usage: consumer <fixture.bin>   (fixture = little-endian dump written by tests/test_c_consumer.py:
 *   u32 n_targets, u64 lens[n_targets], u64 n_records, then the nine cov_batch arrays, then u64 n_cigar)
 * Runs one contig through the engine (cov_* ABI) and the estimator trait exports (covh_estimator_*), prints
 * "<tid>\t<mean>\t<variance>\n" per contig with reads.  Without a usable GPU it reports cov_create's status and exits 3. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "covermhip.h"
#include "coverm_host.h"

_Static_assert(sizeof(cov_contig_stats) == 128, "cov_contig_stats is 128 bytes in the ABI");
_Static_assert(sizeof(cov_config) == 40, "cov_config is 40 bytes in the ABI");
_Static_assert(sizeof(cov_batch) == 80, "cov_batch is 80 bytes in the ABI");

static void *slurp(FILE *f, size_t n) { void *p = malloc(n ? n : 1); if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "short fixture\n"); exit(2); } return p; }

int main(int argc, char **argv) {
    if (cov_abi_version() != COVERMHIP_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    if (argc < 2) { printf("abi %d\n", cov_abi_version()); return 0; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t nt; uint64_t n, nc;
    if (fread(&nt, 4, 1, f) != 1) return 2;
    uint64_t *lens = (uint64_t *)slurp(f, 8ull * nt);
    if (fread(&n, 8, 1, f) != 1 || fread(&nc, 8, 1, f) != 1) return 2;
    cov_batch b;
    b.tid = (const int32_t *)slurp(f, 4 * n); b.pos = (const int32_t *)slurp(f, 4 * n); b.flag = (const uint16_t *)slurp(f, 2 * n);
    b.mapq = (const uint8_t *)slurp(f, n); b.nm = (const uint32_t *)slurp(f, 4 * n); b.nm_kind = (const uint8_t *)slurp(f, n);
    b.l_seq = (const uint32_t *)slurp(f, 4 * n); b.cigar_off = (const uint32_t *)slurp(f, 4 * (n + 1)); b.cigar = (const uint32_t *)slurp(f, 4 * nc);
    b.n_records = n;
    fclose(f);
    cov_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.include_improper_pairs = 1; cfg.include_supplementary = 1; cfg.min_mapq = 255; cfg.contig_end_exclusion = 75;
    cov_session *s = NULL;
    cov_status st = cov_create(&cfg, &s);
    if (st != COV_OK) { printf("cov_create: %d (%s)\n", (int)st, cov_last_error(NULL)); return 3; }
    if (cov_set_targets(s, nt, lens) != COV_OK || cov_push_batch(s, &b) != COV_OK) { fprintf(stderr, "%s\n", cov_last_error(s)); return 4; }
    cov_contig_stats *stats = (cov_contig_stats *)calloc(nt ? nt : 1, sizeof *stats);
    cov_summary summ;
    if (cov_finish(s, stats, &summ) != COV_OK) { fprintf(stderr, "%s\n", cov_last_error(s)); return 4; }
    covh_estimator pm = {COVH_MEAN, 0.0f, 75, 0, 0.0f, 0.0f}, pv = {COVH_VARIANCE, 0.0f, 75, 0, 0.0f, 0.0f};
    covh_estimator_state *em = covh_estimator_new(&pm), *ev = covh_estimator_new(&pv);
    const uint64_t zero = 0;
    for (uint32_t t = 0; t < nt; t++) {
        if (stats[t].n_pass == 0) continue;
        covh_estimator_setup(em); covh_estimator_setup(ev);
        covh_estimator_add_contig_stats(em, &stats[t], lens[t], NULL, stats[t].n_primary, stats[t].sum_identity_primary);
        covh_estimator_add_contig_stats(ev, &stats[t], lens[t], NULL, stats[t].n_primary, stats[t].sum_identity_primary);
        char a[64], c[64];
        covh_format_f32(covh_estimator_calculate_coverage(em, &zero, 1), a, sizeof a);
        covh_format_f32(covh_estimator_calculate_coverage(ev, &zero, 1), c, sizeof c);
        printf("%u\t%s\t%s\n", t, a, c);
    }
    printf("primary %llu\n", (unsigned long long)summ.num_detected_primary_alignments);
    covh_estimator_free(em); covh_estimator_free(ev);
    cov_destroy(s);
    return 0;
}
