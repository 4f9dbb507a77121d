#!/bin/bash
# Host layer (BAM reader, estimators/drivers/printers — the device-estimate path's taker included —, pair filter, gene driver, the
# orchestrator with its `filter` subcommand through the coverm-amd binary) under AddressSanitizer + UBSan: a host-only build of the four
# C++ files with stubs for the device ABI, swapped in for the CPU test run (253 tests, round 6).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cat > $T/stubs.cpp <<'EOS'
// Device ABI stubs for the host-only sanitizer build.  The coverage entry points fail (no device); the INGEST entry points are
// a CPU mock that checks what the host driver (covh_bam_gpu_ingest: reader thread, per-chunk header hop, block table, staging
// slots) hands over: every block must inflate to its ISIZE with its CRC-32 from exactly the bytes fed, in order and contiguous.
#include "covermhip.h"
#include <zlib.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>
struct cov_session { std::vector<uint8_t> comp; std::vector<cov_bgzf_block> blocks; uint64_t first = 0, fed_to = 0; std::string err; bool active = false; };
extern "C" {
int cov_abi_version(void) { return COVERMHIP_ABI_VERSION; }
int cov_mock_ingest_present(void) { return 1; }
int covh_timing_on(void) { static const int on = getenv("COVERM_CLI_TIMING") != nullptr; return on; }      // (lives in covermhip.hip in the real build)
// cov_host_free reports whether the pointer was one of cov_host_alloc's (callers free other pointers themselves)
static std::mutex g_hm; static std::set<void *> g_host;
void *cov_host_alloc(size_t n) { void *p = malloc(n ? n : 1); std::lock_guard<std::mutex> lk(g_hm); g_host.insert(p); return p; }
int cov_host_free(void *p) { { std::lock_guard<std::mutex> lk(g_hm); if (!g_host.erase(p)) return 0; } free(p); return 1; }
void cov_host_trim(void) {}
const char *cov_last_error(const cov_session *s) { return s ? s->err.c_str() : "stub"; }
cov_status cov_interval_stats_compute(cov_session *, const cov_interval *, uint64_t, uint64_t, int, cov_interval_stats *, uint64_t *) { return COV_ERR_HIP; }
cov_status cov_fetch_interval_hist(cov_session *, uint64_t *) { return COV_ERR_HIP; }
cov_status cov_create(const cov_config *, cov_session **out) {
    if (!getenv("COVERM_MOCK_INGEST")) return COV_ERR_HIP;
    *out = new cov_session(); return COV_OK;
}
void cov_destroy(cov_session *s) { delete s; }
cov_status cov_set_targets(cov_session *s, uint32_t, const uint64_t *) { return s ? COV_OK : COV_ERR_HIP; }
cov_status cov_set_target_mask(cov_session *, const uint8_t *) { return COV_ERR_HIP; }
cov_status cov_push_batch(cov_session *, const cov_batch *) { return COV_ERR_HIP; }
cov_status cov_push_batch_device(cov_session *, const cov_batch *) { return COV_ERR_HIP; }
cov_status cov_finish(cov_session *, cov_contig_stats *, cov_summary *) { return COV_ERR_HIP; }
cov_status cov_fetch_hist(cov_session *, uint64_t *) { return COV_ERR_HIP; }
cov_status cov_copy_depth(cov_session *, uint32_t, int32_t *) { return COV_ERR_HIP; }
cov_status cov_reset(cov_session *) { return COV_ERR_HIP; }
cov_status cov_kernel_ms(const cov_session *, cov_kernel_id, double *, uint32_t *) { return COV_ERR_HIP; }
cov_status cov_algorithmic_bytes(const cov_session *, uint64_t *) { return COV_ERR_HIP; }
cov_status cov_last_paths(const cov_session *, cov_path_counts *) { return COV_ERR_HIP; }
cov_status cov_ingest_begin(cov_session *s, uint64_t bytes, uint64_t first, int) {
    s->comp.assign(bytes, 0xAA); s->blocks.clear(); s->first = first; s->fed_to = 0; s->active = true; return COV_OK;
}
cov_status cov_ingest_span(cov_session *, int64_t, int64_t, int, int, uint64_t, uint64_t) { return COV_ERR_INVALID_ARG; }   // the mock checks whole files only
cov_status cov_ingest_slot_wait(cov_session *s, int slot) { return s && slot >= 0 && slot < COV_INGEST_SLOTS ? COV_OK : COV_ERR_INVALID_ARG; }
cov_status cov_ingest_feed(cov_session *s, int slot, const void *host, uint64_t off, uint64_t n, const cov_bgzf_block *b, uint32_t nb) {
    if (!s->active || slot < 0 || slot >= COV_INGEST_SLOTS || off != s->fed_to || off + n > s->comp.size()) { s->err = "mock: pieces out of order"; return COV_ERR_INVALID_ARG; }
    memcpy(s->comp.data() + off, host, n); s->fed_to = off + n;
    for (uint32_t i = 0; i < nb; i++) {
        const uint64_t prev_end = s->blocks.empty() ? 0 : s->blocks.back().in_off + s->blocks.back().in_len;
        const uint64_t prev_out = s->blocks.empty() ? 0 : s->blocks.back().out_off + s->blocks.back().isize;
        if (b[i].in_off < prev_end || b[i].in_off + b[i].in_len > s->fed_to || b[i].out_off != prev_out) { s->err = "mock: block table inconsistent"; return COV_ERR_INVALID_ARG; }
        s->blocks.push_back(b[i]);
    }
    return COV_OK;
}
cov_status cov_ingest_end(cov_session *s, uint64_t *n_records) {
    s->active = false;
    std::vector<uint8_t> infl;
    for (const cov_bgzf_block &b : s->blocks) {
        std::vector<uint8_t> out(b.isize ? b.isize : 1);
        z_stream z; memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) return COV_ERR_HIP;
        z.next_in = s->comp.data() + b.in_off; z.avail_in = b.in_len; z.next_out = out.data(); z.avail_out = b.isize;
        const int rc = inflate(&z, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && z.total_out == b.isize && z.avail_in == 0;
        inflateEnd(&z);
        if (!ok || (uint32_t)crc32(crc32(0L, Z_NULL, 0), out.data(), b.isize) != b.crc) { s->err = "mock: a block does not inflate to its ISIZE / CRC-32"; return COV_ERR_INGEST_FALLBACK; }
        infl.insert(infl.end(), out.begin(), out.begin() + b.isize);
    }
    uint64_t q = s->first, n = 0;
    while (q + 4 <= infl.size()) { uint32_t bs; memcpy(&bs, infl.data() + q, 4); if (q + 4 + bs > infl.size()) { s->err = "mock: truncated record"; return COV_ERR_INGEST_FALLBACK; } q += 4 + (uint64_t)bs; n++; }
    if (q != infl.size()) { s->err = "mock: trailing bytes"; return COV_ERR_INGEST_FALLBACK; }
    if (n_records) *n_records = n;
    return COV_OK;
}
cov_status cov_ingest_release(cov_session *) { return COV_OK; }
cov_status cov_ingest_abort(cov_session *s) { if (s) s->active = false; return COV_OK; }
int cov_bind_thread_to_device_node(int) { return -1; }
cov_status cov_reserve(cov_session *, uint64_t, uint64_t) { return COV_ERR_HIP; }
cov_status cov_ingest_copy_inflated(cov_session *, uint64_t, uint64_t, void *) { return COV_ERR_HIP; }
cov_status cov_copy_records(cov_session *, const cov_batch *, uint64_t *, uint64_t *) { return COV_ERR_HIP; }
cov_status cov_gather(cov_session *const *, uint32_t, uint32_t) { return COV_ERR_HIP; }
cov_status cov_gathered(cov_session *, uint32_t, cov_contig_stats *, cov_summary *) { return COV_ERR_HIP; }
cov_status cov_set_estimators(cov_session *, const cov_estimator *, uint32_t) { return COV_ERR_HIP; }
cov_status cov_fetch_estimates(cov_session *, float *) { return COV_ERR_HIP; }
uint32_t cov_store_spills(const cov_session *) { return 0; }
cov_status cov_ingest_want_mates(cov_session *, int) { return COV_OK; }
cov_status cov_pair_filter_apply(cov_session *, const cov_pair_filter *, uint64_t *, uint64_t *) { return COV_ERR_HIP; }
// registration of the mapped file: accepted (the mock's feed reads the mapping like any host memory), or refused to exercise the
// driver's switch to staging slots; every registered range must be unregistered exactly once, after the ingest ended
static std::set<void *> g_reg;
cov_status cov_host_register(cov_session *s, void *p, size_t n) {
    if (getenv("COVERM_MOCK_NO_REGISTER") || !p || !n || ((uintptr_t)p & 4095u)) { if (s) s->err = "mock: registration refused"; return COV_ERR_HIP; }
    std::lock_guard<std::mutex> lk(g_hm); if (!g_reg.insert(p).second) { s->err = "mock: range registered twice"; return COV_ERR_HIP; } return COV_OK;
}
cov_status cov_host_unregister(cov_session *s, void *p) {
    std::lock_guard<std::mutex> lk(g_hm);
    if (s->active || !g_reg.erase(p)) { s->err = "mock: unregister of an unknown range or during the ingest"; abort(); }
    return COV_OK;
}
}
EOS
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -I$R/include \
    $R/coverm_amd/csrc/host_bam.cpp $R/coverm_amd/csrc/host_coverage.cpp $R/coverm_amd/csrc/host_filter.cpp $R/coverm_amd/csrc/host_cli.cpp $T/stubs.cpp \
    -o $T/libcovermhip_asan.so -lz -lpthread -ldl
cp $R/coverm_amd/libcovermhip.so $T/real.so
trap 'cp $T/real.so $R/coverm_amd/libcovermhip.so' EXIT
cp $T/libcovermhip_asan.so $R/coverm_amd/libcovermhip.so
cd $R
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    COVERM_MOCK_INGEST=1 python -m pytest tests/test_bam_reader.py tests/test_host_golden.py tests/test_genes.py tests/test_ingest_driver_mock.py tests/test_host_estimated.py tests/test_takers_printers.py tests/test_printer_paths.py tests/test_filter_subcommand.py -q -m "not gpu" -p no:cacheprovider -k "not exports_every_declared_symbol"
