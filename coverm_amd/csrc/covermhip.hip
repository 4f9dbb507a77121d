// libcovermhip.so — session management and the C ABI declared in include/covermhip.h.
// Host code here only moves bytes and launches kernels; every statistic is computed on the device
// (pileup_kernels.hip.h).  There is deliberately no CPU fallback: without a usable HIP device every
// entry point fails with COV_ERR_HIP.
#include "pileup_kernels.hip.h"
#include "prep_lean.hip.h"
#include "ingest_kernels.hip.h"

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <chrono>
#include <cstdio>
#include <dlfcn.h>
#include <sched.h>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <future>
#include <thread>
#include <vector>

#include "../../include/covermhip.h"
#include "roctx_ranges.h"
#include "knobs.h"

using namespace covk;

// The C ABI structs are mirrored by hand elsewhere (numpy dtype in coverm_amd/native.py, #[repr(C)] in INTEGRATION.md):
// pin their layout here so that a change in include/covermhip.h cannot silently diverge from those mirrors.
static_assert(sizeof(cov_config) == 40 && offsetof(cov_config, min_aligned_length) == 12 && offsetof(cov_config, min_percent_identity) == 16 &&
              offsetof(cov_config, contig_end_exclusion) == 24 && offsetof(cov_config, want) == 32, "cov_config layout");
static_assert(sizeof(cov_batch) == 80 && offsetof(cov_batch, cigar) == 64 && offsetof(cov_batch, n_records) == 72, "cov_batch layout");
static_assert(sizeof(cov_contig_stats) == 128 && offsetof(cov_contig_stats, sum_identity_primary) == 40 && offsetof(cov_contig_stats, win_sum_d) == 56 &&
              offsetof(cov_contig_stats, first_record) == 88 && offsetof(cov_contig_stats, win_min_d) == 104 && offsetof(cov_contig_stats, hist_len) == 112 &&
              offsetof(cov_contig_stats, hist_off) == 120, "cov_contig_stats layout");
static_assert(sizeof(cov_summary) == 32 && offsetof(cov_summary, hist_total) == 24, "cov_summary layout");
static_assert(sizeof(cov_interval) == 24 && sizeof(cov_interval_stats) == 56, "interval struct layout");
static_assert(sizeof(cov_estimator) == sizeof(covk::DevEstimator) && offsetof(cov_estimator, contig_end_exclusion) == offsetof(covk::DevEstimator, excl) &&
              offsetof(cov_estimator, trim_max) == offsetof(covk::DevEstimator, trim_max) && COV_EST_MAX == covk::EST_MAX && COV_EST_ANIR == covk::EST_ANIR, "cov_estimator mirrors the device struct");

namespace {

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            char b_[512];                                                                     \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            s->err = b_;                                                                      \
            return COV_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n, hipStream_t st, size_t keep = 0) {
        if (n <= cap) return hipSuccess;
        size_t nc = std::max(n, cap + cap / 2);
        T *q = nullptr;
        hipError_t e = hipMalloc(&q, nc * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p) {
            e = hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
        }
        if (p) (void)hipFree(p);
        p = q; cap = nc;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

std::string g_create_error;

// COVERM_CLI_TIMING: the library's and the host layer's stamps on stderr (read once per process)
bool cov_timing_on() { static const bool on = getenv("COVERM_CLI_TIMING") != nullptr; return on; }

}  // namespace

extern "C" int covh_timing_on(void) { return cov_timing_on() ? 1 : 0; }

// Which inflate kernel runs, how many blocks make one round (= one launch = one window of the inflated stream) and the size of the carry area.
//   version 3 (default): k_inflate_wave, one wave per BGZF block; a round is as long as the parse wants it (windows bound the memory and set
//                        the fill and drain of the pipeline);
//   version 1 (COVERM_INFLATE_V=1): k_inflate, one LANE per block, private Huffman tables per lane in LDS — the second implementation the
//                        tests compare with; a launch is cut to exactly the blocks resident at once (a lane decodes a block serially).
struct InflateKernel { int version = 3; int lz_version = 2; u32 round_blocks = 0; u64 carry = 16ull << 20; u64 cwin = 0; u32 ext_parts = covi::EXT_PARTS; };

struct cov_session {
    cov_config cfg{};
    hipStream_t stream = nullptr;
    std::string err;
    int tile = 1024;  // bases per tile (one wave per tile: STREAM_TW)
    bool use_fast = true;  // k_pileup_fast + k_pileup_stream on the slow-tile list (default); COVERM_PILEUP=stream: k_pileup_stream alone
    int chunk_tiles = 8;   // consecutive tiles walked by one wave (swept in round 4: 4 -> 0.627 ms, 8 -> 0.596, 16 -> 0.642, 32 -> 0.866)
    int prep_kernel = 0;   // 0 = k_prep_lean (prep_lean.hip.h); COVERM_PREP_KERNEL=7 forces k_prep7s, the second implementation (tests)
    int est_lanes = -1;        // k_estimate_lanes (a lane per contig) from 65 536 contigs on; COVERM_EST_LANES=1 | 0 forces it on / off (tests)
    int fast_tables = 1;       // k_pileup_fast (one table of biased deltas: the default) or k_pileup_fast2t (two count tables; COVERM_FAST_TABLES=2)
    int n_cus = 256;

    // targets
    uint32_t n_targets = 0;
    std::vector<uint32_t> h_tlen;
    std::vector<uint32_t> h_tile_first;  // first tile of each contig (+ sentinel)
    uint32_t n_tiles = 0;
    DevBuf<u32> d_tlen, d_tile_contig, d_tile_start;
    DevBuf<u32> d_tile_first, d_tcnt, d_fov, d_tscan, d_ttop;   // TileIdx (pileup_kernels.hip.h)
    DevBuf<u32> d_slow_list;                                    // tiles k_ranges leaves to k_pileup_stream
    uint32_t tile_shift = 10;
    DevBuf<u32> d_cx_list, d_cx_cnt, d_cx_cur, d_cx_scan, d_cx_top;   // CxIdx: long-CIGAR buckets
    DevBuf<uint2> d_cx_runs;
    DevBuf<DevContig> d_ctg_scratch;   // cov_copy_depth works on a copy of the accumulators
    // per-interval statistics (cov_interval_stats_compute): depth of every target, materialised on demand
    DevBuf<int32_t> d_depth_all; DevBuf<u64> d_depth_off; bool depth_all_valid = false;
    DevBuf<DevInterval> d_iv; DevBuf<DevIntervalStats> d_ivst; DevBuf<unsigned long long> d_ivhist; uint64_t ivhist_total = 0;
    DevBuf<uint8_t> d_mask;
    std::vector<uint8_t> h_mask;       // host copy (convert_results lays the compact histogram out itself)
    bool have_mask = false;
    // results of the device pipeline live in ONE block [DevGlobal][DevContig x n_targets] (d_res): one DMA brings them to
    // the host, and cov_gather sends the same block over RCCL.  d_glob / d_ctg are views into it (never freed themselves).
    DevBuf<uint8_t> d_res;
    DevBuf<DevContig> d_ctg;
    DevBuf<DevGlobal> d_glob;
    DevBuf<uint4> d_desc;
    // cov_gather (root session): every rank's block, device and page-locked host copies
    DevBuf<uint8_t> d_gather; uint8_t *h_gather = nullptr; size_t h_gather_cap = 0; uint32_t gather_n = 0; size_t gather_block = 0;

    // record store (owned) or adopted device batch
    DevBuf<int32_t> s_tid, s_pos;
    DevBuf<uint16_t> s_flag;
    DevBuf<uint8_t> s_mapq, s_nmk;
    DevBuf<u32> s_nm, s_lseq, s_coff, s_cig;
    // mates of the records the device ingest extracted (cov_ingest_want_mates): next_refID + read-name hash, for cov_pair_filter_apply
    DevBuf<int32_t> s_mtid; DevBuf<u64> s_qh1; DevBuf<u32> s_qh2;
    bool want_mates = false;
    uint64_t mates_valid = 0;        // the mate columns describe records [0, mates_valid) of the store
    uint64_t n_records = 0, n_cigar = 0;
    bool adopted = false;
    cov_batch adopted_batch{};
    uint64_t adopted_ncig = 0;
    uint32_t adopted_cig_end = 0;

    // ---- bounded record store.  The reference holds ONE contig's state at a time and flushes when the tid changes (contig.rs:128-155), so it
    // has no size limit; the store here keeps as many contigs as fit under a cap.  When a push or an extraction would take it past the cap,
    // the pipeline runs over what the store holds, the contigs that are complete are kept on the host (`spill`), the records from the first
    // considered record of the contig in flight onwards move to the front, and the session goes on.  cov_finish merges.
    uint64_t cap_records = 0x80000000ull, cap_cigar = 0x80000000ull;      // COVERM_STORE_CAP_RECORDS / COVERM_STORE_CAP_CIGAR (tests); the hard limit stays 2^32 - 16 each
    struct Spill {
        bool active = false;
        std::vector<cov_contig_stats> stats; std::vector<DevContig> ctg; std::vector<uint8_t> have;     // per target: taken in an earlier spill
        std::vector<uint64_t> hist;                                       // their histogram bins, stats[c].hist_off indexes this
        uint64_t prim = 0, cons = 0, records = 0;                         // primaries / considered records / records that left the store
        int64_t inflight = -1;                                            // contig in flight at the last spill: a later considered contig below it = unsorted
        uint32_t count = 0;
        void clear() { active = false; stats.clear(); ctg.clear(); have.clear(); hist.clear(); prim = cons = records = 0; inflight = -1; count = 0; }
    } spill;
    // CoverageEstimator::calculate_coverage on the device (cov_set_estimators): parameters, device rows, page-locked copy of the last finish
    EstParams est{};
    float *h_estf = nullptr; bool est_valid = false;      // (views: device rows behind the result block in d_res, host rows behind it in h_res)
    std::vector<float> spill_est;           // rows of the contigs that left in a spill (n_targets x est.n, filled as they leave)
    std::vector<uint64_t> merged_hist;      // histogram of the last cov_finish when it merged spilled contigs (cov_fetch_hist serves it)
    bool merged_valid = false;
    uint64_t merged_records = 0;            // records of the whole sample after a merging finish (cov_gather sends it)
    DevBuf<uint8_t> d_spill_tmp;
    uint64_t ing_rec_spilled = 0;           // records of the running ingest that already left the store
    uint64_t spill_retry_at = 0;            // after a spill that could move nothing: no further attempt below this many records

    DevBuf<uint2> d_runs;
    DevBuf<PrepPartial> d_part;
    DevBuf<u32> d_gen_list;            // steps k_prep_lean leaves to k_prep_generic
    DevBuf<PrepArgs> d_prep_args;      // k_prep_lean's arguments in device memory (its rare branches read them there), written by k_init
    DevBuf<double> d_ident, d_identp;
    DevBuf<IdChunk> d_idch;
    int id_mode = 1;   // 1 = exact parallel identity sums, 0 = serial chain only (COVERM_IDENTITY=serial)
    hipStream_t side = nullptr; bool side_owned = false;     // k_identity overlaps k_ranges / k_pileup (created on first use, or the idle second stream of the ingest)
    hipEvent_t ev_prep_done = nullptr, ev_side_done = nullptr;
    DevBuf<u32> d_arena;
    DevBuf<u64> d_hist_top;            // k_hist_sum / k_hist_off: bins per block of 1024 contigs
    DevBuf<u64> d_chist;
    u64 *h_chist = nullptr; u64 h_chist_cap = 0, hist_prefetched = 0, last_chist_total = 0; bool hist_compacted = false, hist_fetch_seen = false;
    DevBuf<int32_t> d_depth;

    // device ingest (cov_ingest_*): compressed file and inflated stream in HBM, BGZF block table, record-boundary scratch
    DevBuf<uint8_t> g_scratch, g_carry;
    DevBuf<u32> g_crc_tab;             // k_crc32_wave's tables (crcw::build_tables), uploaded by the session's first cov_ingest_begin
    DevBuf<uint8_t> g_cwin[3];                             // compressed bytes, one buffer per round in flight (file offsets biased by the round's origin)
    DevBuf<uint8_t> g_win[3];                              // inflated windows (one k_inflate round each): [carry area | blocks]
    DevBuf<covi::BgzfBlock> g_blocks;
    DevBuf<u32> g_status;
    DevBuf<covi::SegInfo> g_seg[4];                        // per-window parse state: extract of window w runs up to three windows later
    DevBuf<u64> g_recbase[4], g_cigbase[4];
    // g_result: [3] inflate failures (u32) | extract failures (u32), [5] bytes in g_carry, [6] p0 of the window being parsed,
    // [7] key of the last record of the previous window (bit 63: there is one),
    // [8 + 8 * (w % 4) ..] result block of window w (k_bam_verify)
    DevBuf<u64> g_result;
    DevBuf<covi::tokpos_t> g_tok, g_tok2;
    DevBuf<u32> g_ntok, g_ntok2;
    hipStream_t ing_aux = nullptr;                         // k_lz_resolve / k_crc32 of round i run beside k_inflate of round i + 1
    hipStream_t ing_parse = nullptr;                       // record boundaries of window i run beside both
    hipStream_t ing_ext = nullptr;   // extraction of verified windows: the parse stream (a stream of its own made no difference, profiles/r03_tail_variants.log)
    hipEvent_t ing_inf_done[2] = {nullptr, nullptr}, ing_lz_done[2] = {nullptr, nullptr};
    hipEvent_t ing_ver_done[4] = {}, ing_ext_done[3] = {}, ing_cdone[3] = {};
    uint64_t ing_round_start = 0, ing_prev_end = 0, ing_ccap = 0;   // accumulating round: file offset its buffer starts at; end of the last payload seen; bytes a round may span
    uint32_t ing_round_n = 0;                              // blocks of the accumulating round
    int64_t ing_copy_cleared = -1;                         // highest round whose compressed buffer the copy stream may already write
    long long ing_key_lo = 0, ing_key_hi = 0x80000000ll;   // cov_ingest_span: records outside [key_lo, key_hi) are not taken
    bool ing_search_first = false, ing_open_end = false, ing_fed_any = false;
    uint64_t ing_span_lo = 0, ing_span_hi = 0;             // file bytes the span covers (store size estimate)
    uint64_t ing_tail_key = ~0ull;                         // key of the record cut by the end of the last window (~0: none)
    u64 *h_winres = nullptr;                               // page-locked: 4 x 8 words, the windows' result blocks
    struct WinInfo { u64 N = 0; u32 n_seg = 0; u64 comp_end = 0; };
    WinInfo ing_win[4];
    uint32_t ing_batch = 0; int ing_check_crc = 1;         // ing_batch = rounds (= windows) launched so far
    uint32_t ing_extracted = 0;                            // windows whose records are in the store (or skipped after a failure)
    uint64_t ing_rec_total = 0, ing_cig_total = 0;         // records / CIGAR words extracted so far (not committed before cov_ingest_end)
    uint32_t ing_fail = 0;                                 // sticky status bits of k_bam_verify
    uint64_t ing_fail_dbg[3] = {0, 0, 0};
    uint64_t ing_first_record = 0;
    uint64_t ing_launched = 0;   // blocks already handed to k_inflate
    covi::BgzfBlock *h_blocks = nullptr; size_t h_blocks_cap = 0;   // page-locked mirror of the block table (async uploads read from it)
    uint64_t ing_comp = 0, ing_infl = 0, ing_blocks = 0;
    bool ing_active = false;
    InflateKernel ing_K;
    hipStream_t ing_copy = nullptr;
    std::future<hipError_t> ing_prep;          // COV_WANT_INGEST: the ingest's streams, events and fixed tables, being created since cov_create (ingest_prep_wait)
    // (ONE upload stream.  Its H2D copies are 0.595 ms per 32 MiB piece = 56 GB/s with a ~38 us gap between two of them: the link is busy
    // 91 % of the time the file streams, profiles/r06_ingest_copy_trace.json.  A second stream — the halves of a piece on two DMA engines in
    // round 3, whole pieces in turn in round 6 — is SLOWER both ways: 0.82 s of ingest against 0.50 s at 200 M reads,
    // profiles/r06_copy_streams_ab_200M.json, profiles/r03_copy_queues_50M.log.)
    hipEvent_t ing_ev[COV_INGEST_SLOTS] = {}, ing_fed = nullptr;
    double ing_s_alloc = 0;     // host seconds inside device allocations of the ingest
    double ing_s_part[4] = {0, 0, 0, 0};     // ... inside ingest_drain / launch_round / the upload calls / event waits (COVERM_CLI_TIMING)

    // results of the last finish
    bool finished = false;
    // result staging in page-locked memory: [DevGlobal][DevContig x n_targets], one DMA pair per finish
    uint8_t *h_res = nullptr; size_t h_res_cap = 0;
    std::future<std::pair<uint8_t *, size_t>> h_res_prep;      // an assembly's result staging (hundreds of MB of page-locked memory: ~0.17 s per GiB) being obtained since cov_set_targets
    DevContig *h_ctg = nullptr;
    DevGlobal h_glob{};
    bool last_gen_all = false;         // the last finish ran k_prep_generic over every step (cov_last_paths)
    uint64_t algo_bytes = 0;

    hipEvent_t ev[COV_K_COUNT][2] = {};
    hipEvent_t ev_begin[COV_K_COUNT] = {};      // the event a group's time is measured from (its own, or the previous group's end)
    int ev_fresh = -1;                          // group whose end event is the last thing on the stream (-1: something else followed)
    float k_ms[COV_K_COUNT] = {};
    uint32_t k_launches[COV_K_COUNT] = {};
};

cov_status spill_store(cov_session *s, bool &progress);      // bounded record store (defined behind finish_once)

namespace {

size_t result_block_bytes(u32 n_targets) { return sizeof(DevGlobal) + (size_t)std::max<u32>(n_targets, 1) * sizeof(DevContig); }
// Page-locked staging for the results of a sample with very many references: the block is [DevGlobal][DevContig x n][floats x n x n_est] —
// 350 MB at 2 M contigs and four estimators, 0.06 s of hipHostMalloc that used to sit in the sample's one cov_finish.  cov_set_targets starts
// it on a helper thread (the ingest or the pushes pass meanwhile); cov_finish takes what the helper got, or allocates as before.
static size_t result_host_bytes(const cov_session *s, u32 nT) {
    return result_block_bytes(nT) + (size_t)std::max<u32>(nT, 1) * std::max<u32>(s->est.n, 1u) * sizeof(float);
}
static void result_host_take(cov_session *s) {
    if (!s->h_res_prep.valid()) return;
    const std::pair<uint8_t *, size_t> r = s->h_res_prep.get();
    if (!r.first) return;
    if (r.second > s->h_res_cap) {
        if (s->h_res) (void)hipHostFree(s->h_res);
        s->h_res = r.first; s->h_res_cap = r.second;
    } else (void)hipHostFree(r.first);
}

// [DevGlobal][DevContig x n_targets] (the block cov_gather sends) and, behind it, room for the estimators' floats (COV_EST_MAX per target):
// both come to the host in one copy
hipError_t bind_result_block(cov_session *s, u32 n_targets) {
    hipError_t e = s->d_res.reserve(result_block_bytes(n_targets) + (size_t)std::max<u32>(n_targets, 1) * COV_EST_MAX * sizeof(float), s->stream);
    if (e != hipSuccess) return e;
    s->d_glob.p = reinterpret_cast<DevGlobal *>(s->d_res.p); s->d_glob.cap = 1;
    s->d_ctg.p = reinterpret_cast<DevContig *>(s->d_res.p + sizeof(DevGlobal)); s->d_ctg.cap = std::max<u32>(n_targets, 1);
    return hipSuccess;
}

Records records_of(const cov_session *s) {
    Records r{};
    if (s->adopted) {
        const cov_batch &b = s->adopted_batch;
        r.tid = b.tid; r.pos = b.pos; r.flag = b.flag; r.mapq = b.mapq; r.nm = b.nm; r.nm_kind = b.nm_kind;
        r.l_seq = b.l_seq; r.cigar_off = b.cigar_off; r.cigar = b.cigar;
    } else {
        r.tid = s->s_tid.p; r.pos = s->s_pos.p; r.flag = s->s_flag.p; r.mapq = s->s_mapq.p; r.nm = s->s_nm.p;
        r.nm_kind = s->s_nmk.p; r.l_seq = s->s_lseq.p; r.cigar_off = s->s_coff.p; r.cigar = s->s_cig.p;
    }
    r.n = (u32)s->n_records;
    r.cigar_end = s->adopted ? s->adopted_cig_end : (u32)s->n_cigar;
    return r;
}

__global__ void k_rebase_offsets(u32 *off, u32 n, u32 sub, u32 add) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) off[i] = off[i] - sub + add;
}

// records [lo, n) with neither the secondary nor the supplementary bit (what num_detected_primary_alignments counts, bam_generator.rs:114-118)
__global__ void k_count_primary(const uint16_t *__restrict__ flag, u32 lo, u32 n, unsigned long long *out) {
    u32 cnt = 0;
    for (u64 i = (u64)lo + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) cnt += (flag[i] & 0x900u) ? 0u : 1u;
    cnt = wave_sum_u32(cnt);
    if ((threadIdx.x & 63u) == 0u && cnt) atomicAdd(out, (unsigned long long)cnt);
}

cov_status append(cov_session *s, const cov_batch *b, bool from_device) {
    const uint64_t n = b->n_records;
    if (n == 0) return COV_OK;
    const hipMemcpyKind kind = from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    u32 off0 = 0, offn = 0;
    if (from_device) {
        HIPCHK(hipMemcpyAsync(&off0, b->cigar_off, 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(&offn, b->cigar_off + n, 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    } else {
        off0 = b->cigar_off[0]; offn = b->cigar_off[n];
    }
    const uint64_t ncig = (uint64_t)offn - off0;
    if (s->n_records && (s->n_records + n > s->cap_records || s->n_cigar + ncig > s->cap_cigar) && s->n_records >= s->spill_retry_at) {
        // bounded store: the contigs that are complete leave for the host, the contig in flight moves to the front (contig.rs:128-155)
        bool progress = false;
        const cov_status sp = spill_store(s, progress);
        if (sp != COV_OK) return sp;
        // one reference already fills the store: a spill is a whole pass over it that moves nothing — not again before the store has doubled
        s->spill_retry_at = progress ? 0 : 2 * s->n_records;
    }
    if (s->n_records + n >= 0xfffffff0ull) { s->err = "more than 2^32 records of one reference (or of one batch) in the record store"; return COV_ERR_INVALID_ARG; }
    if (s->n_cigar + ncig >= 0xfffffff0ull) { s->err = "more than 2^32 CIGAR words of one reference (or of one batch) in the record store"; return COV_ERR_INVALID_ARG; }
    const uint64_t R = s->n_records, N = R + n;
    hipStream_t st = s->stream;
    HIPCHK(s->s_tid.reserve(N, st, R)); HIPCHK(s->s_pos.reserve(N, st, R)); HIPCHK(s->s_flag.reserve(N, st, R));
    HIPCHK(s->s_mapq.reserve(N, st, R)); HIPCHK(s->s_nmk.reserve(N, st, R)); HIPCHK(s->s_nm.reserve(N, st, R));
    HIPCHK(s->s_lseq.reserve(N, st, R)); HIPCHK(s->s_coff.reserve(N + 1, st, R + 1));
    HIPCHK(s->s_cig.reserve(s->n_cigar + ncig + 1, st, s->n_cigar));
    HIPCHK(hipMemcpyAsync(s->s_tid.p + R, b->tid, n * 4, kind, st));
    HIPCHK(hipMemcpyAsync(s->s_pos.p + R, b->pos, n * 4, kind, st));
    HIPCHK(hipMemcpyAsync(s->s_flag.p + R, b->flag, n * 2, kind, st));
    HIPCHK(hipMemcpyAsync(s->s_mapq.p + R, b->mapq, n, kind, st));
    HIPCHK(hipMemcpyAsync(s->s_nmk.p + R, b->nm_kind, n, kind, st));
    HIPCHK(hipMemcpyAsync(s->s_nm.p + R, b->nm, n * 4, kind, st));
    HIPCHK(hipMemcpyAsync(s->s_lseq.p + R, b->l_seq, n * 4, kind, st));
    HIPCHK(hipMemcpyAsync(s->s_coff.p + R, b->cigar_off, (n + 1) * 4, kind, st));
    if (ncig) HIPCHK(hipMemcpyAsync(s->s_cig.p + s->n_cigar, b->cigar + off0, ncig * 4, kind, st));
    if (off0 != (u32)s->n_cigar) {
        const u32 cnt = (u32)(n + 1);
        hipLaunchKernelGGL(k_rebase_offsets, dim3((cnt + 255) / 256), dim3(256), 0, st, s->s_coff.p + R, cnt, off0,
                           (u32)s->n_cigar);
        HIPCHK(hipGetLastError());
    }
    s->n_records = N; s->n_cigar += ncig;
    s->finished = false;
    if (!from_device) HIPCHK(hipStreamSynchronize(st));   // contract: host arrays are free to change on return
    return COV_OK;
}

void timing_events(cov_session *s) {
    if (s->ev[0][0]) return;
    for (int k = 0; k < COV_K_COUNT; k++)
        for (int j = 0; j < 2; j++) (void)hipEventCreate(&s->ev[k][j]);
}
// Timing brackets of the kernel groups.  An event record costs the stream ~5 us (profiles/r05b_bench_kernel_trace: the gaps of a step sat
// exactly where its eleven records were), so a group that begins where the previous one ended takes that group's end event for its
// beginning: six records per finish instead of eleven.
void time_begin(cov_session *s, int k) {
    if (s->ev_fresh >= 0) { s->ev_begin[k] = s->ev[s->ev_fresh][1]; return; }
    (void)hipEventRecord(s->ev[k][0], s->stream); s->ev_begin[k] = s->ev[k][0];
}
void time_end(cov_session *s, int k) { (void)hipEventRecord(s->ev[k][1], s->stream); s->k_launches[k]++; s->ev_fresh = k; }

// k_pileup_fast over every tile (it skips the ones k_ranges flagged TILE_F_SLOW)
template <bool H, int TABLES>
void launch_fast_v(cov_session *s, const PileupArgs &a, u32 n_tiles) {
    const auto kern = TABLES == 1 ? &k_pileup_fast<H> : &k_pileup_fast2t<H>;
    const size_t smem = pileup_fast_smem_bytes(H, TABLES == 1 ? FAST_HB : FAST_HB7, TABLES);
    // the dynamic-LDS limit and the occupancy are per-device facts, and span mode launches from one thread per device: cached per
    // device id, in atomics (two threads racing for the same device compute the same value)
    static std::atomic<int> occ_dev[64];
    std::atomic<int> &occ_slot = occ_dev[(unsigned)s->cfg.device & 63u];
    int occ = occ_slot.load(std::memory_order_relaxed);
    const u32 chunk = (u32)s->chunk_tiles;
    const u32 n_chunks = (n_tiles + chunk - 1) / chunk;
    if (!occ) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), 256, smem) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / smem));
        }
        occ = nb; occ_slot.store(nb, std::memory_order_relaxed);
    }
    const u32 wg_per_cu = 8u * (u32)occ;
    const u32 grid = std::max(1u, std::min((n_chunks + 3) / 4, (u32)s->n_cus * wg_per_cu));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s->stream, a, n_tiles, chunk);
}
template <bool H>
void launch_fast_t(cov_session *s, const PileupArgs &a, u32 n_tiles) {
    if (s->fast_tables == 2) launch_fast_v<H, 2>(s, a, n_tiles);
    else launch_fast_v<H, 1>(s, a, n_tiles);
}

template <bool H, bool W>
void launch_stream_t(cov_session *s, const PileupArgs &a, u32 n_tiles, bool slow_list_only = false) {
    const size_t smem = pileup_stream_smem_bytes();
    static std::atomic<int> occ_dev[64];       // per device id, as in launch_fast_t
    std::atomic<int> &occ_slot = occ_dev[(unsigned)s->cfg.device & 63u];
    int occ = occ_slot.load(std::memory_order_relaxed);
    const u32 chunk = (u32)s->chunk_tiles;
    const u32 n_chunks = (n_tiles + chunk - 1) / chunk;
    // Grid = 8x the resident workgroups (registers, not LDS, bound residency): each wave still walks several chunks
    // and keeps its sums in registers across them, but the hardware dispatcher hands out the 8 successive "rounds",
    // which evens out the heavy-tailed per-chunk cost far better than a purely persistent grid (measured, 50 M reads:
    // 1.06 ms at 1x, 0.905 ms at 8x, 0.956 ms at one chunk per wave).
    if (!occ) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pileup_stream<H, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_pileup_stream<H, W>), 256, smem) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / smem));
        }
        occ = nb; occ_slot.store(nb, std::memory_order_relaxed);
    }
    const u32 wg_per_cu = 8u * (u32)occ;
    if (slow_list_only) {   // the listed tiles only, one per wave step; the count lives on the device (usually 0: waves exit at once)
        hipLaunchKernelGGL((k_pileup_stream<H, W>), dim3((u32)s->n_cus), dim3(256), smem, s->stream, a, 0u, 1u,
                           (const u32 *)s->d_slow_list.p, (const u32 *)&s->d_glob.p->n_slow);
        return;
    }
    const u32 grid = std::max(1u, std::min((n_chunks + 3) / 4, (u32)s->n_cus * wg_per_cu));
    hipLaunchKernelGGL((k_pileup_stream<H, W>), dim3(grid), dim3(256), smem, s->stream, a, n_tiles, chunk, (const u32 *)nullptr,
                       (const u32 *)nullptr);
}
// dispatches to the configured pileup kernel
template <bool H, bool W>
void launch_any_pileup(cov_session *s, const PileupArgs &a, u32 n_tiles) {
    if (s->use_fast && !W) {
        launch_fast_t<H>(s, a, n_tiles);
        launch_stream_t<H, W>(s, a, n_tiles, true);
    } else launch_stream_t<H, W>(s, a, n_tiles);
}

PileupArgs pileup_args(cov_session *s) {
    PileupArgs a{};
    a.tile_contig = s->d_tile_contig.p; a.tile_start = s->d_tile_start.p; a.desc = s->d_desc.p;
    a.runs = s->d_runs.p; a.cx_runs = s->d_cx_runs.p; a.r = records_of(s); a.ctg = s->d_ctg.p; a.g = s->d_glob.p;
    a.hist_arena = s->d_arena.p; a.excl = s->cfg.contig_end_exclusion; a.depth_out = nullptr; a.tile_base = 0;
    return a;
}

}  // namespace

extern "C" {

int cov_abi_version(void) { return COVERMHIP_ABI_VERSION; }

const char *cov_last_error(const cov_session *s) { return s ? s->err.c_str() : g_create_error.c_str(); }

// The part of cov_ingest_begin that does not depend on the file: three streams (a hardware queue with its 173 MB save area each), two dozen
// events, the page-locked result words and block-table mirror, k_crc32_wave's tables — ~50 ms that used to pass between cov_ingest_begin and
// the first piece's upload (tools/r06/call39.sh: "first upload after 0.053 s").  With COV_WANT_INGEST cov_create starts it on a helper thread
// as soon as the runtime is up, beside its own stream and whatever the caller does next (targets, estimators, the file's header).
static hipError_t ingest_prepare_(cov_session *s) {
    hipError_t e = hipSetDevice(s->cfg.device);
    if (e != hipSuccess || s->ing_copy) return e;
#define PREP(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
    PREP(hipStreamCreateWithFlags(&s->ing_copy, hipStreamNonBlocking));
    for (int k = 0; k < COV_INGEST_SLOTS; k++) PREP(hipEventCreateWithFlags(&s->ing_ev[k], hipEventDisableTiming));
    PREP(hipEventCreateWithFlags(&s->ing_fed, hipEventDisableTiming));
    PREP(hipStreamCreateWithFlags(&s->ing_aux, hipStreamNonBlocking));
    // (the boundary search on the LZ / CRC stream — the runtime's four hardware queues put the two on one queue anyway — was measured:
    // ingest 0.650 s against 0.612 s, the extraction of window w then also waits behind the LZ of w + 1, and the exit is no shorter:
    // profiles/r04_e2e_200M_runs_*.log)
    PREP(hipStreamCreateWithFlags(&s->ing_parse, hipStreamNonBlocking));
    s->ing_ext = s->ing_parse;
    for (int k = 0; k < 2; k++) { PREP(hipEventCreateWithFlags(&s->ing_inf_done[k], hipEventDisableTiming)); PREP(hipEventCreateWithFlags(&s->ing_lz_done[k], hipEventDisableTiming)); }
    for (int k = 0; k < 4; k++) PREP(hipEventCreateWithFlags(&s->ing_ver_done[k], hipEventDisableTiming));
    for (int k = 0; k < 3; k++) { PREP(hipEventCreateWithFlags(&s->ing_ext_done[k], hipEventDisableTiming)); PREP(hipEventCreateWithFlags(&s->ing_cdone[k], hipEventDisableTiming)); }
    PREP(hipHostMalloc((void **)&s->h_winres, 4 * 8 * sizeof(u64), hipHostMallocDefault));
    {   // page-locked mirror of the block table: room for a 25 GB file of ordinary ~20 KB blocks (cov_ingest_begin replaces it for a larger one)
        const size_t nc = (size_t)3 << 19;
        PREP(hipHostMalloc((void **)&s->h_blocks, nc * sizeof(covi::BgzfBlock), hipHostMallocDefault));
        s->h_blocks_cap = nc;
    }
    {
        static const std::vector<u32> crc_tables = [] { std::vector<u32> t(covi::crcw::TABLE_WORDS); covi::crcw::build_tables(t.data()); return t; }();
        PREP(s->g_crc_tab.reserve(covi::crcw::TABLE_WORDS, s->ing_copy));
        PREP(hipMemcpyAsync(s->g_crc_tab.p, crc_tables.data(), covi::crcw::TABLE_WORDS * sizeof(u32), hipMemcpyHostToDevice, s->ing_copy));
        PREP(hipStreamSynchronize(s->ing_copy));
    }
#undef PREP
    return hipSuccess;
}
// Everything that looks at the ingest's streams outside cov_ingest_begin waits for the helper first.
static hipError_t ingest_prep_wait(cov_session *s) { return s->ing_prep.valid() ? s->ing_prep.get() : hipSuccess; }

cov_status cov_create(const cov_config *cfg, cov_session **out) {
    if (!cfg || !out) { g_create_error = "null argument"; return COV_ERR_INVALID_ARG; }
    *out = nullptr;
    const bool timing = cov_timing_on();
    const auto tc0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) { if (timing) fprintf(stderr, "[covermhip] cov_create: %s at %.4fs\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count()); };
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    stamp("hipGetDeviceCount (runtime initialised)");
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no usable HIP device: ") + hipGetErrorString(e) +
                         " (the engine has no CPU fallback)";
        return COV_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return COV_ERR_INVALID_ARG; }
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return COV_ERR_HIP; }
    cov_session *s = new cov_session();
    s->cfg = *cfg;
    {
        const char *mode = getenv("COVERM_PILEUP");      // "stream": k_pileup_stream over every tile (the second kernel behind k_pileup_fast, alone: tests)
        if (mode && !strcmp(mode, "stream")) s->use_fast = false;
        s->tile = STREAM_TW;
        stamp("hipSetDevice");
        int cus = 0;      // (hipGetDeviceProperties fills a kilobyte of fields from many driver queries; one attribute is all that is needed)
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) s->n_cus = cus;
        stamp("device attribute");
    }
    if (cfg->want & COV_WANT_INGEST) s->ing_prep = std::async(std::launch::async, [s] { return ingest_prepare_(s); });
    if (const char *el = getenv("COVERM_EST_LANES")) s->est_lanes = atoi(el) ? 1 : 0;
    if (const char *ft = getenv("COVERM_FAST_TABLES")) s->fast_tables = atoi(ft) == 2 ? 2 : 1;
    if (const char *pk = getenv("COVERM_PREP_KERNEL")) s->prep_kernel = atoi(pk) == 7 ? 7 : 0;
    if (const char *im = getenv("COVERM_IDENTITY")) s->id_mode = strcmp(im, "serial") ? 1 : 0;
    { long long v; if (covknob::get("store_cap_records", v) && v >= 1) s->cap_records = std::min<uint64_t>((uint64_t)v, 0xfffffff0ull); }
    { long long v; if (covknob::get("store_cap_cigar", v) && v >= 1) s->cap_cigar = std::min<uint64_t>((uint64_t)v, 0xfffffff0ull); }
    // (every stream costs ~6 ms to create and as much again when the process ends, tools/ubench/exit_probe: the side stream of the
    // identity kernels is created by the first cov_finish that wants it, and borrows the ingest's second stream when there is one)
    e = hipEventCreateWithFlags(&s->ev_prep_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_side_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); (void)ingest_prep_wait(s); delete s; return COV_ERR_HIP; }
    stamp("streams and events");      // (the kernels' timing events are created by the first cov_finish: a run that ingests a file needs them half a second later)
    e = bind_result_block(s, 1);
    if (e != hipSuccess) { g_create_error = std::string("hipMalloc: ") + hipGetErrorString(e); cov_destroy(s); return COV_ERR_HIP; }
    stamp("first hipMalloc");
    *out = s;
    return COV_OK;
}

static void ingest_free_buffers(cov_session *s) {
    s->g_scratch.release(); s->g_carry.release(); s->g_blocks.release(); s->g_status.release();
    s->g_crc_tab.release();
    for (int k = 0; k < 3; k++) { s->g_win[k].release(); s->g_cwin[k].release(); }
    for (int k = 0; k < 4; k++) { s->g_seg[k].release(); s->g_recbase[k].release(); s->g_cigbase[k].release(); }
    s->g_tok.release(); s->g_ntok.release(); s->g_tok2.release(); s->g_ntok2.release();
}

void cov_destroy(cov_session *s) {
    if (!s) return;
    (void)ingest_prep_wait(s);
    (void)hipSetDevice(s->cfg.device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    s->d_tlen.release(); s->d_tile_contig.release(); s->d_tile_start.release(); s->d_mask.release();
    s->d_tile_first.release(); s->d_tcnt.release(); s->d_fov.release(); s->d_tscan.release(); s->d_ttop.release(); s->d_slow_list.release();
    s->d_ctg_scratch.release(); s->d_depth_all.release(); s->d_depth_off.release(); s->d_iv.release(); s->d_ivst.release(); s->d_ivhist.release();
    s->d_cx_list.release(); s->d_cx_cnt.release(); s->d_cx_cur.release(); s->d_cx_scan.release(); s->d_cx_top.release(); s->d_cx_runs.release();
    if (s->side) {     // before ing_aux goes: `side` may be that very stream (finish_once borrows the idle ingest stream)
        (void)hipStreamSynchronize(s->side);
        if (s->side_owned) (void)hipStreamDestroy(s->side);
        s->side = nullptr; s->side_owned = false;
    }
    if (s->ing_aux) (void)hipStreamSynchronize(s->ing_aux);
    if (s->ing_parse) (void)hipStreamSynchronize(s->ing_parse);
    if (s->ing_ext) (void)hipStreamSynchronize(s->ing_ext);
    ingest_free_buffers(s);
    s->g_result.release();
    if (s->ing_aux) (void)hipStreamDestroy(s->ing_aux);
    if (s->ing_parse) (void)hipStreamDestroy(s->ing_parse);
    for (int k = 0; k < 2; k++) { if (s->ing_inf_done[k]) (void)hipEventDestroy(s->ing_inf_done[k]); if (s->ing_lz_done[k]) (void)hipEventDestroy(s->ing_lz_done[k]); }
    for (int k = 0; k < 4; k++) if (s->ing_ver_done[k]) (void)hipEventDestroy(s->ing_ver_done[k]);
    for (int k = 0; k < 3; k++) { if (s->ing_ext_done[k]) (void)hipEventDestroy(s->ing_ext_done[k]); if (s->ing_cdone[k]) (void)hipEventDestroy(s->ing_cdone[k]); }
    if (s->h_winres) (void)hipHostFree(s->h_winres);
    s->h_winres = nullptr;
    if (s->h_blocks) (void)hipHostFree(s->h_blocks);
    s->h_blocks = nullptr;
    if (s->ing_copy) { (void)hipStreamSynchronize(s->ing_copy); (void)hipStreamDestroy(s->ing_copy); }
    for (int k = 0; k < COV_INGEST_SLOTS; k++) if (s->ing_ev[k]) (void)hipEventDestroy(s->ing_ev[k]);
    if (s->ing_fed) (void)hipEventDestroy(s->ing_fed);
    s->d_res.release(); s->d_ctg.p = nullptr; s->d_glob.p = nullptr; s->d_desc.release(); s->d_gather.release();
    if (s->h_gather) (void)hipHostFree(s->h_gather);
    s->h_gather = nullptr;
    result_host_take(s);
    if (s->h_res) (void)hipHostFree(s->h_res);
    s->h_res = nullptr; s->h_res_cap = 0; s->h_ctg = nullptr;
    if (s->h_chist) (void)hipHostFree(s->h_chist);
    s->h_chist = nullptr; s->h_chist_cap = 0;
    s->h_estf = nullptr; s->d_spill_tmp.release();
    s->s_tid.release(); s->s_pos.release(); s->s_flag.release(); s->s_mapq.release(); s->s_nmk.release();
    s->s_nm.release(); s->s_lseq.release(); s->s_coff.release(); s->s_cig.release();
    s->s_mtid.release(); s->s_qh1.release(); s->s_qh2.release();
    s->d_runs.release(); s->d_part.release(); s->d_prep_args.release(); s->d_gen_list.release(); s->d_hist_top.release(); s->d_ident.release(); s->d_identp.release(); s->d_idch.release();
    if (s->ev_prep_done) (void)hipEventDestroy(s->ev_prep_done);
    if (s->ev_side_done) (void)hipEventDestroy(s->ev_side_done); s->d_arena.release(); s->d_chist.release(); s->d_depth.release();
    for (int k = 0; k < COV_K_COUNT; k++)
        for (int j = 0; j < 2; j++) if (s->ev[k][j]) (void)hipEventDestroy(s->ev[k][j]);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

cov_status cov_set_targets(cov_session *s, uint32_t n_targets, const uint64_t *target_len) {
    if (!s || (!target_len && n_targets)) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    s->n_targets = n_targets;
    s->h_tlen.resize(n_targets);
    s->h_tile_first.assign((size_t)n_targets + 1, 0);
    uint64_t nt = 0;
    for (uint32_t c = 0; c < n_targets; c++) {
        if (target_len[c] > 0x7fffffffull) { s->err = "target length exceeds BAM's i32 range"; return COV_ERR_INVALID_ARG; }
        s->h_tlen[c] = (uint32_t)target_len[c];
        s->h_tile_first[c] = (uint32_t)nt;
        nt += (target_len[c] + (uint64_t)s->tile - 1) / (uint64_t)s->tile;
    }
    if (nt >= 0x7fffffffull) { s->err = "too many tiles"; return COV_ERR_INVALID_ARG; }
    s->h_tile_first[n_targets] = (uint32_t)nt;
    s->n_tiles = (uint32_t)nt;
    std::vector<u32> tc(nt), ts(nt);
    for (uint32_t c = 0; c < n_targets; c++) {
        u32 k = s->h_tile_first[c];
        for (uint64_t p = 0; p < target_len[c]; p += (uint64_t)s->tile, k++) { tc[k] = c; ts[k] = (u32)p; }
    }
    HIPCHK(s->d_tlen.reserve(std::max<size_t>(1, n_targets), s->stream));
    HIPCHK(s->d_tile_contig.reserve(std::max<size_t>(1, nt), s->stream));
    HIPCHK(s->d_tile_start.reserve(std::max<size_t>(1, nt), s->stream));
    HIPCHK(s->d_desc.reserve(std::max<size_t>(2, 2 * nt), s->stream));
    HIPCHK(s->d_tile_first.reserve((size_t)n_targets + 1, s->stream));
    HIPCHK(s->d_tcnt.reserve(std::max<size_t>(1, nt), s->stream)); HIPCHK(s->d_fov.reserve(std::max<size_t>(1, nt), s->stream));
    HIPCHK(s->d_tscan.reserve(std::max<size_t>(1, nt), s->stream)); HIPCHK(s->d_ttop.reserve(nt / 1024 + 2, s->stream));
    HIPCHK(s->d_slow_list.reserve(std::max<size_t>(1, nt), s->stream));
    HIPCHK(s->d_cx_cnt.reserve(std::max<size_t>(1, nt), s->stream)); HIPCHK(s->d_cx_cur.reserve(std::max<size_t>(1, nt), s->stream));
    HIPCHK(s->d_cx_scan.reserve(std::max<size_t>(1, nt), s->stream)); HIPCHK(s->d_cx_top.reserve(nt / 1024 + 2, s->stream));
    HIPCHK(hipMemcpyAsync(s->d_tile_first.p, s->h_tile_first.data(), ((size_t)n_targets + 1) * 4, hipMemcpyHostToDevice, s->stream));
    s->tile_shift = 0;
    while ((1u << s->tile_shift) < (uint32_t)s->tile) s->tile_shift++;
    HIPCHK(bind_result_block(s, n_targets));
    if (n_targets) HIPCHK(hipMemcpyAsync(s->d_tlen.p, s->h_tlen.data(), (size_t)n_targets * 4, hipMemcpyHostToDevice, s->stream));
    if (nt) {
        HIPCHK(hipMemcpyAsync(s->d_tile_contig.p, tc.data(), nt * 4, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->d_tile_start.p, ts.data(), nt * 4, hipMemcpyHostToDevice, s->stream));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    s->have_mask = false;
    s->finished = false;
    if (n_targets >= 65536u && !s->h_res_prep.valid() && result_host_bytes(s, n_targets) > s->h_res_cap) {
        const size_t need = result_host_bytes(s, n_targets);
        const int dev = s->cfg.device;
        s->h_res_prep = std::async(std::launch::async, [need, dev]() -> std::pair<uint8_t *, size_t> {
            uint8_t *p = nullptr;
            if (hipSetDevice(dev) != hipSuccess || hipHostMalloc((void **)&p, need, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return {nullptr, 0}; }
            return {p, need};
        });
    }
    return COV_OK;
}

cov_status cov_set_target_mask(cov_session *s, const uint8_t *mask) {
    if (!s) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    if (!mask) { s->have_mask = false; s->h_mask.clear(); return COV_OK; }
    s->h_mask.assign(mask, mask + s->n_targets);
    HIPCHK(s->d_mask.reserve(std::max<size_t>(1, s->n_targets), s->stream));
    if (s->n_targets) HIPCHK(hipMemcpyAsync(s->d_mask.p, mask, s->n_targets, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->have_mask = true;
    s->finished = false;
    return COV_OK;
}

cov_status cov_push_batch(cov_session *s, const cov_batch *b) {
    if (!s || !b) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    if (s->ing_active) { const cov_status a = cov_ingest_abort(s); if (a != COV_OK) return a; }   // an ingest left open: its queued work may still write the store
    if (s->adopted) {  // materialise the adopted device batch into the owned store first
        cov_batch ab = s->adopted_batch;
        s->adopted = false; s->n_records = 0; s->n_cigar = 0;
        cov_status st = append(s, &ab, true);
        if (st) return st;
    }
    return append(s, b, false);
}

cov_status cov_push_batch_device(cov_session *s, const cov_batch *b) {
    if (!s || !b) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    if (s->n_records == 0 && !s->adopted) {
        if (b->n_records >= 0xfffffff0ull) { s->err = "more than 2^32 records in one session"; return COV_ERR_INVALID_ARG; }
        u32 o0 = 0, o1 = 0;
        if (b->n_records) {
            HIPCHK(hipMemcpy(&o0, b->cigar_off, 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(&o1, b->cigar_off + b->n_records, 4, hipMemcpyDeviceToHost));
        }
        s->adopted = true; s->adopted_batch = *b; s->n_records = b->n_records;
        s->adopted_ncig = (uint64_t)o1 - o0;
        s->adopted_cig_end = o1;
        s->finished = false;
        return COV_OK;
    }
    if (s->adopted) {
        cov_batch ab = s->adopted_batch;
        s->adopted = false; s->n_records = 0; s->n_cigar = 0;
        cov_status st = append(s, &ab, true);
        if (st) return st;
    }
    return append(s, b, true);
}

// Abandons an ingest between cov_ingest_begin and cov_ingest_end: everything queued on the ingest streams (uploads, inflate
// rounds, boundary search, extraction) is waited for, then the session is as if the ingest had never begun — the record
// store holds what it held before (extraction only ever writes behind n_records) and may be pushed to again.
cov_status cov_ingest_abort(cov_session *s) {
    if (!s) return COV_ERR_INVALID_ARG;
    if (!s->ing_active) return COV_OK;
    HIPCHK(hipSetDevice(s->cfg.device));
    s->ing_active = false;
    for (hipStream_t st : {s->ing_copy, s->stream, s->ing_aux, s->ing_parse, s->ing_ext})
        if (st) HIPCHK(hipStreamSynchronize(st));
    s->ing_rec_total = s->ing_cig_total = 0; s->ing_round_n = 0; s->ing_fail = 0;
    s->ing_extracted = s->ing_batch;
    if (s->ing_rec_spilled) {     // the store no longer holds what it held before the ingest: only cov_reset brings the session back
        s->ing_rec_spilled = 0;
        s->err = "cov_ingest_abort: part of the file had already left the bounded record store; cov_reset the session";
        return COV_ERR_STATE;
    }
    return COV_OK;
}

cov_status cov_reset(cov_session *s) {
    if (!s) return COV_ERR_INVALID_ARG;
    if (s->ing_active) {
        // an ingest that had already spilled part of its file: the abort drains the device and says "cov_reset the session" (COV_ERR_STATE) —
        // this IS that reset, so the status is expected here and everything below still has to be cleared
        const bool spilled = s->ing_rec_spilled != 0;
        const cov_status a = cov_ingest_abort(s);
        if (a != COV_OK && !(spilled && a == COV_ERR_STATE)) return a;
        if (a != COV_OK) s->err.clear();
    }
    s->adopted = false; s->n_records = 0; s->n_cigar = 0; s->finished = false; s->depth_all_valid = false; s->mates_valid = 0;
    s->spill.clear(); s->merged_valid = false; s->merged_hist.clear(); s->ing_rec_spilled = 0; s->spill_retry_at = 0; s->spill_est.clear(); s->est_valid = false;
    return COV_OK;
}

// Host side of a finished pipeline: error checks in file order, sortedness, DevContig -> cov_contig_stats.  Used for the
// session's own results and for the blocks cov_gather collected from other ranks.
// `min_ok_tid` / `rec_base`: the session's own results after a spill (bounded record store) — a considered contig below the contig that
// was in flight at the last spill breaks the reference's order rule exactly like an interleaving inside one pass, and record indices
// count from the first record of the sample, not of the store.
static cov_status convert_results(cov_session *s, const DevGlobal &G, const DevContig *ctg, uint64_t n_records, cov_contig_stats *stats,
                                  cov_summary *summary, int64_t min_ok_tid = -1, uint64_t rec_base = 0) {
    const u32 nT = s->n_targets;
    const bool want_hist = s->cfg.want & COV_WANT_HIST;
    if (G.internal_error) { s->err = "internal error: depth exceeded its proven bound"; return COV_ERR_STATE; }
    // errors in file order (the reference panics at the first offending record)
    uint64_t err_rec = ~0ull; int err_code = 0;
    if (G.first_error != ~0ull) { err_rec = G.first_error >> 8; err_code = (int)(G.first_error & 0xff); }
    // The two passes below walk every contig; above 64 k contigs (assemblies: 10^5 - 10^7 of them) they run on a few threads — each chunk
    // first leaves what the chunks behind it need from it (its largest last record, its histogram bins), then every chunk runs the
    // serial loop from the prefix of the chunks in front of it.  (One thread until round 6: 2 M contigs cost ~50 ms per finish.)
    const u32 n_chunks = nT >= 65536u ? std::min<u32>(8u, std::max<u32>(1u, std::thread::hardware_concurrency())) : 1u;
    const u32 per_chunk = (nT + n_chunks - 1) / std::max<u32>(n_chunks, 1u);
    auto chunked = [&](auto fn) {
        if (n_chunks <= 1) { fn(0u, 0u, nT); return; }
        std::vector<std::thread> th;
        for (u32 k = 1; k < n_chunks; k++) th.emplace_back([&, k] { fn(k, std::min(nT, k * per_chunk), std::min(nT, (k + 1) * per_chunk)); });
        fn(0u, 0u, std::min(nT, per_chunk));
        for (auto &t : th) t.join();
    };
    const bool want_hist_bins = want_hist;
    const u64 excl = s->cfg.contig_end_exclusion;
    std::vector<uint64_t> ch_last(n_chunks, 0), ch_bins(n_chunks, 0), ch_unsorted(n_chunks, ~0ull);
    std::vector<uint8_t> ch_any(n_chunks, 0);
    if (n_chunks > 1)
        chunked([&](u32 k, u32 lo, u32 hi) {
            uint64_t last = 0, bins = 0; bool any = false;
            for (u32 c = lo; c < hi; c++) {
                const DevContig &C = ctg[c];
                if (C.n_pass == 0) continue;
                last = any ? std::max<uint64_t>(last, C.last_rec) : C.last_rec; any = true;
                if (want_hist_bins) {
                    const u64 L = s->h_tlen[c];
                    if (2 * excl < L && (!s->have_mask || s->h_mask[c])) bins += C.max_d + 1u;
                }
            }
            ch_last[k] = last; ch_bins[k] = bins; ch_any[k] = any;
        });
    // sortedness: the considered records of successive touched contigs must not interleave
    // (contig.rs:129-132 panics when a considered record has tid < the previous considered tid)
    {
        chunked([&](u32 k, u32 lo, u32 hi) {
            uint64_t prev_last = 0; bool any = false; uint64_t unsorted_at = ~0ull;
            for (u32 j = 0; j < k; j++) if (ch_any[j]) { prev_last = any ? std::max(prev_last, ch_last[j]) : ch_last[j]; any = true; }
            for (u32 c = lo; c < hi; c++) {
                const DevContig &C = ctg[c];
                if (C.n_pass == 0) continue;
                if ((int64_t)c < min_ok_tid) unsorted_at = std::min<uint64_t>(unsorted_at, C.first_rec);      // tid < last_tid across a spill (contig.rs:129-132)
                if (any && C.first_rec < prev_last) unsorted_at = std::min<uint64_t>(unsorted_at, std::max<uint64_t>(C.first_rec, 0));
                prev_last = any ? std::max<uint64_t>(prev_last, C.last_rec) : C.last_rec;
                any = true;
            }
            ch_unsorted[k] = unsorted_at;
        });
        uint64_t unsorted_at = ~0ull;
        for (u32 k = 0; k < n_chunks; k++) unsorted_at = std::min(unsorted_at, ch_unsorted[k]);
        if (unsorted_at != ~0ull && unsorted_at <= err_rec) {
            s->err = "BAM file appears to be unsorted. Input BAM files must be sorted by reference (i.e. by samtools sort)";
            return COV_ERR_UNSORTED;
        }
    }
    if (err_code) {
        char b[256];
        const char *what = err_code == 2 ? "Mapping record encountered that does not have an 'NM' auxiliary tag in the SAM/BAM format"
                         : err_code == 3 ? "Unexpected data type of NM aux tag"
                         : err_code == 4 ? "aligned block starts at or beyond the end of its reference sequence"
                         : err_code == 7 ? "record refers to a reference id outside the header (Corrupt BAM file?)"
                                         : "invalid CIGAR operation";
        snprintf(b, sizeof b, "%s (record %llu)", what, (unsigned long long)(err_rec + rec_base));
        s->err = b;
        return (cov_status)err_code;
    }
    uint64_t hist_total = 0;
    std::vector<uint64_t> ch_total(n_chunks, 0);
    chunked([&](u32 k, u32 lo, u32 hi) {
        uint64_t hist_run = 0;
        for (u32 j = 0; j < k; j++) hist_run += ch_bins[j];
        for (u32 c = lo; c < hi; c++) {
            const DevContig &C = ctg[c];
            cov_contig_stats &o = stats[c];
            memset(&o, 0, sizeof o);
            o.n_primary = C.n_primary; o.n_pass = C.n_pass; o.n_nonsupp = C.n_nonsupp;
            o.sum_nm = C.sum_nm; o.sum_indel = C.sum_indel;
            o.sum_identity_primary = C.id_primary; o.sum_identity_nonsupp = C.id_nonsupp;
            o.win_sum_d = C.sum_d; o.win_sum_d2 = C.sum_d2; o.win_covered = C.cov_win; o.full_covered = C.cov_full;
            o.first_record = C.first_rec; o.last_record = C.last_rec;
            if (C.n_pass) { o.first_record += rec_base; o.last_record += rec_base; }
            const u64 L = s->h_tlen[c];
            const u64 win_len = 2 * excl < L ? L - 2 * excl : 0;
            if (C.n_pass && win_len) {
                o.win_max_d = C.max_d;
                o.win_min_d = (C.proc_win < win_len || C.min_d == 0xffffffffu) ? 0u : C.min_d;
            }
            if (want_hist) {      // the compact histogram's layout (what k_hist_off<1> computes on the device when the bins are compacted there)
                const bool live = C.n_pass != 0 && (!s->have_mask || s->h_mask[c]);
                o.hist_len = (live && win_len) ? C.max_d + 1u : 0u; o.hist_off = hist_run; hist_run += o.hist_len;
            }
        }
        ch_total[k] = hist_run;
    });
    hist_total = ch_total[n_chunks - 1];
    if (summary) {
        uint64_t prim = 0, cons = 0;
        for (u32 k = 0; k < COUNTER_SLOTS * 8; k++) { prim += G.prim_slots[k]; cons += G.cons_slots[k]; }
        summary->num_detected_primary_alignments = prim;
        summary->n_records = n_records;
        summary->n_considered = cons;
        summary->hist_total = hist_total;
    }
    return COV_OK;
}

static cov_status finish_once(cov_session *s, cov_contig_stats *stats, cov_summary *summary, bool &again);

// The pipeline over what the store holds.  The buckets of long-CIGAR records are sized optimistically: a pass that finds them too
// small has already computed the exact need, grows the buffer and runs once more (at most once per growth of the workload).
static cov_status finish_store(cov_session *s, cov_contig_stats *stats, cov_summary *summary) {
    bool again = false;
    cov_status st = finish_once(s, stats, summary, again);
    if (st == COV_OK && again) st = finish_once(s, stats, summary, again);
    if (st == COV_OK && again) { s->err = "internal error: bucket sizing did not converge"; return COV_ERR_STATE; }
    return st;
}

static cov_status merge_spilled(cov_session *s, cov_contig_stats *stats, cov_summary *summary);

cov_status cov_finish(cov_session *s, cov_contig_stats *stats, cov_summary *summary) {
    covr::Range rr("cov_finish");
    if (s) s->merged_valid = false;
    cov_summary local;
    cov_status st = finish_store(s, stats, (s && s->spill.active && !summary) ? &local : summary);
    if (st == COV_OK && s->spill.active) st = merge_spilled(s, stats, summary ? summary : &local);
    return st;
}

static cov_status finish_once(cov_session *s, cov_contig_stats *stats, cov_summary *summary, bool &again) {
    again = false;
    if (!s || (!stats && s->n_targets)) return COV_ERR_INVALID_ARG;
    s->depth_all_valid = false;
    HIPCHK(hipSetDevice(s->cfg.device));
    timing_events(s);
    hipStream_t st = s->stream;
    const u32 nT = s->n_targets;
    const u32 R = (u32)s->n_records;
    const bool want_hist = s->cfg.want & COV_WANT_HIST, want_id = s->cfg.want & COV_WANT_IDENTITY;
    for (int k = 0; k < COV_K_COUNT; k++) { s->k_launches[k] = 0; s->k_ms[k] = 0.f; }
    bool compacted = false;      // the compact histogram was built by this pass (else: by the first cov_fetch_hist)
    s->est_valid = false;
    s->ev_fresh = -1;

    HIPCHK(s->d_runs.reserve(std::max<size_t>(1, R), st));
    // k_prep geometry: 4096 records per workgroup for short reads (k_prep_lean: each of the four waves walks 16 steps of 64 consecutive records;
    // k_prep7s: 16 passes of 256); one step / pass when CIGARs are long, where the work per record is large and records are few (long-read
    // mappings), so that every CU gets workgroups
    const uint64_t ncig_all = s->adopted ? s->adopted_ncig : s->n_cigar;
    const bool long_cigars = R && ncig_all / R >= 16;
    const int prep_kernel = s->prep_kernel;
    const int prep_passes = long_cigars ? 1 : 2 * PREP_PASSES, prep_b = 1;
    const u32 prep_chunk = (u32)(256 * prep_b * prep_passes);
    const u32 prep_grid = (R + prep_chunk - 1) / prep_chunk;
    HIPCHK(s->d_part.reserve((size_t)prep_grid + 2, st));
    if (want_id) {
        if (!(s->cfg.want & COV_WANT_IDENTITY_PRIMARY_ONLY)) HIPCHK(s->d_ident.reserve(std::max<size_t>(1, R), st));
        if (!(s->cfg.want & COV_WANT_IDENTITY_NONSUPP_ONLY)) HIPCHK(s->d_identp.reserve(std::max<size_t>(1, R), st));
        HIPCHK(s->d_idch.reserve((size_t)R / ID_CH + 2, st));
    }
    if (want_hist) HIPCHK(s->d_arena.reserve((size_t)R + nT + 1, st));

    TileIdx ti{};
    ti.tile_first = s->d_tile_first.p; ti.tcnt = s->d_tcnt.p; ti.fov = s->d_fov.p; ti.shift = s->tile_shift; ti.n_tiles = s->n_tiles;
    CxIdx cx{};
    {
        const uint64_t ncig_now = s->adopted ? s->adopted_ncig : s->n_cigar;
        HIPCHK(s->d_cx_list.reserve((size_t)(ncig_now / CX_MIN_OPS) + 64, st));
        cx.list = s->d_cx_list.p; cx.list_cap = (u32)std::min<uint64_t>(ncig_now / CX_MIN_OPS + 64, 0xffffffffull);
        cx.cnt = s->d_cx_cnt.p; cx.cur = s->d_cx_cur.p; cx.cscan = s->d_cx_scan.p; cx.ctop = s->d_cx_top.p;
        cx.runs = s->d_cx_runs.p; cx.runs_cap = s->d_cx_runs.cap;
    }
    const Records r = records_of(s);
    const uint8_t *mask = s->have_mask ? s->d_mask.p : nullptr;
    FilterCfg f{};
    f.include_improper_pairs = s->cfg.include_improper_pairs; f.include_supplementary = s->cfg.include_supplementary;
    f.include_secondary = s->cfg.include_secondary; f.filter_single = s->cfg.filter_single;
    f.min_mapq = s->cfg.min_mapq; f.min_aligned_length = s->cfg.min_aligned_length;
    f.min_percent_identity = s->cfg.min_percent_identity; f.min_aligned_percent = s->cfg.min_aligned_percent;
    // identity streams actually needed (NULL = skipped by k_prep and the k_id_* kernels)
    double *idp = (want_id && !(s->cfg.want & COV_WANT_IDENTITY_NONSUPP_ONLY)) ? s->d_identp.p : nullptr;
    double *idn = (want_id && !(s->cfg.want & COV_WANT_IDENTITY_PRIMARY_ONLY)) ? s->d_ident.p : nullptr;
    HIPCHK(s->d_prep_args.reserve(1, st));
    HIPCHK(s->d_gen_list.reserve((size_t)R / 64 + 2, st));
    PrepArgs pa{};
    {
        PrepHot &ph = pa.hot;
        ph.tid = r.tid; ph.pos = r.pos; ph.flag = r.flag; ph.nmk = r.nm_kind; ph.nm = r.nm; ph.coff = r.cigar_off; ph.cigar = r.cigar; ph.mapq = r.mapq; ph.lseq = r.l_seq;
        ph.runs = s->d_runs.p; ph.tcnt = ti.tcnt; ph.fov = ti.fov; ph.identp = idp; ph.identn = idn;
        ph.n = r.n; ph.cigar_end = r.cigar_end; ph.n_targets = nT; ph.shift = ti.shift; ph.steps = (u32)prep_chunk / 256u;
        ph.gate_mask = 0x4u | (f.include_secondary ? 0u : 0x100u) | (f.include_supplementary ? 0u : 0x800u) | (f.include_improper_pairs ? 0u : 0x2u);
        ph.p1_mask = 0x4u | (f.include_secondary ? 0u : 0x100u) | (f.include_supplementary ? 0u : 0x800u);
        ph.min_mapq = f.min_mapq; ph.min_aligned_length = f.min_aligned_length; ph.min_percent_identity = f.min_percent_identity; ph.min_aligned_percent = f.min_aligned_percent;
        PrepCold &pc = pa.cold;
        pc.ctg = s->d_ctg.p; pc.g = s->d_glob.p; pc.part = s->d_part.p; pc.cx_list = cx.list; pc.mask = mask; pc.tlen = s->d_tlen.p; pc.tile_first = s->d_tile_first.p;
        pc.cx_list_cap = cx.list_cap; pc.gen_list = s->d_gen_list.p;
    }
    hipLaunchKernelGGL(k_init, dim3((std::max(std::max(nT, COUNTER_SLOTS * 8), s->n_tiles) + 255) / 256), dim3(256), 0, st, s->d_ctg.p,
                       nT, s->d_glob.p, ti, cx, s->d_prep_args.p, pa);
    HIPCHK(hipGetLastError());

    if (R) {
        time_begin(s, COV_K_PREP);
        // k_prep_lean's loop takes the steps of 64 records that lie inside one contig; a sample with fewer than ~128 records per contig has few
        // of those, and k_prep_generic then walks every step itself (no list, no partial records for k_post_prep to add)
        const u32 n_steps = (R + 63u) / 64u;
        const bool gen_all = prep_kernel != 7 && (uint64_t)R < (uint64_t)nT * 128u;
        s->last_gen_all = gen_all;
        const u32 gen_grid = gen_all ? std::min<u32>((u32)s->n_cus * 64u, (n_steps + 3u) / 4u) : std::min<u32>((u32)s->n_cus * 8u, (n_steps + 3u) / 4u);      // waves stride over the steps
#define COV_LAUNCH_PREP(ID, FI, MA)                                                                                                       \
        do {                                                                                                                              \
            if (prep_kernel == 7)                                                                                                         \
                hipLaunchKernelGGL((k_prep7s<ID, FI, MA>), dim3(prep_grid), dim3(256), 0, st, r, s->d_tlen.p, nT, mask, f, s->d_ctg.p,   \
                                   s->d_glob.p, s->d_runs.p, idp, idn, s->d_part.p, ti, prep_passes, prep_b, cx.list, cx.list_cap);      \
            else if (gen_all)                                                                                                             \
                hipLaunchKernelGGL((k_prep_generic<ID, FI, MA>), dim3(gen_grid), dim3(256), 0, st, (const PrepArgs *)s->d_prep_args.p, n_steps); \
            else {                                                                                                                        \
                hipLaunchKernelGGL((k_prep_lean<ID, FI, MA>), dim3(prep_grid), dim3(256), 0, st, pa.hot, (const PrepArgs *)s->d_prep_args.p); \
                hipLaunchKernelGGL((k_prep_generic<ID, FI, MA>), dim3(gen_grid), dim3(256), 0, st, (const PrepArgs *)s->d_prep_args.p, 0u); \
            }                                                                                                                             \
        } while (0)
        {
            const int key = (want_id ? 4 : 0) | (s->cfg.filter_single ? 2 : 0) | (mask != nullptr ? 1 : 0);
            switch (key) {
            case 0: COV_LAUNCH_PREP(false, false, false); break;
            case 1: COV_LAUNCH_PREP(false, false, true); break;
            case 2: COV_LAUNCH_PREP(false, true, false); break;
            case 3: COV_LAUNCH_PREP(false, true, true); break;
            case 4: COV_LAUNCH_PREP(true, false, false); break;
            case 5: COV_LAUNCH_PREP(true, false, true); break;
            case 6: COV_LAUNCH_PREP(true, true, false); break;
            default: COV_LAUNCH_PREP(true, true, true); break;
            }
        }
#undef COV_LAUNCH_PREP
        if (nT) {   // what depends on k_prep alone, one launch: the workgroups' partial counters to their contigs (the identity kernels read them),
                    // and the long-CIGAR bucket counts per tile (waves stride over the RW_BUCKET list; nothing to do for short reads)
            const u32 n_red = gen_all ? 0u : (nT + 3u) / 4u, cx_grid0 = (u32)s->n_cus * 8u;
            hipLaunchKernelGGL(k_post_prep, dim3(n_red + cx_grid0), dim3(256), 0, st, r, s->d_tlen.p, s->d_glob.p, cx, ti, s->d_ctg.p, nT, (const PrepPartial *)s->d_part.p,
                               prep_grid, prep_chunk, n_red);
        }
        time_end(s, COV_K_PREP);
        HIPCHK(hipGetLastError());
        if (want_id && nT) {   // depends only on k_prep: run beside k_ranges / k_pileup
            if (!s->side) {
                (void)ingest_prep_wait(s);
                if (s->ing_aux && !s->ing_active) s->side = s->ing_aux;      // idle since cov_ingest_end
                else { HIPCHK(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking)); s->side_owned = true; }
            }
            HIPCHK(hipEventRecord(s->ev_prep_done, st));
            HIPCHK(hipStreamWaitEvent(s->side, s->ev_prep_done, 0));
            (void)hipEventRecord(s->ev[COV_K_IDENTITY][0], s->side);
            if (s->id_mode) {
                const u32 nch = (R + ID_CH - 1) / ID_CH;
                hipLaunchKernelGGL(k_id_approx, dim3(nch), dim3(256), 0, s->side, idp, idn, R, s->d_idch.p);
                hipLaunchKernelGGL(k_id_predict, dim3(nT), dim3(64), 0, s->side, s->d_ctg.p, nT, idp, idn, s->d_idch.p);
                hipLaunchKernelGGL(k_id_exact, dim3(nch), dim3(256), 0, s->side, idp, idn, R, s->d_idch.p);
                hipLaunchKernelGGL(k_id_combine, dim3(nT), dim3(64), 0, s->side, s->d_ctg.p, nT, idp, idn, r.tid, s->d_idch.p);
            } else
                hipLaunchKernelGGL(k_identity, dim3(nT), dim3(64), 0, s->side, s->d_ctg.p, nT, idp, idn, r.tid);
            (void)hipEventRecord(s->ev[COV_K_IDENTITY][1], s->side);
            s->k_launches[COV_K_IDENTITY]++;
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(s->ev_side_done, s->side));   // joined just before the results are copied back
            s->ev_fresh = -1;      // the host worked between the groups (a stream created on first use, five launches on it): the next group's
                                   // time starts at its own event, not at k_prep's end (smoke() used to print k_ranges 7.8 ms for 80 k records)
        }
    }
    if (R && s->n_tiles) {
        time_begin(s, COV_K_RANGES);
        const u32 n_blocks = (s->n_tiles + 1023u) / 1024u;
        const u32 cx_grid = (u32)s->n_cus * 8u;     // waves stride over the RW_BUCKET list; nothing to do for short reads
        ScanPair sp{};
        sp.cnt[0] = s->d_tcnt.p; sp.scan[0] = s->d_tscan.p; sp.top[0] = s->d_ttop.p; sp.total[0] = nullptr;
        sp.cnt[1] = s->d_cx_cnt.p; sp.scan[1] = s->d_cx_scan.p; sp.top[1] = s->d_cx_top.p; sp.total[1] = &s->d_glob.p->cx_total;
        hipLaunchKernelGGL(k_tile_scan1, dim3(2u * n_blocks), dim3(1024), 0, st, sp, s->n_tiles, n_blocks);
        hipLaunchKernelGGL(k_tile_scan2, dim3(2), dim3(1024), 0, st, sp, n_blocks);
        hipLaunchKernelGGL((k_cx_expand<true>), dim3(cx_grid), dim3(256), 0, st, r, s->d_tlen.p, s->d_glob.p, cx, ti);
        u32 *slow_list = s->use_fast ? s->d_slow_list.p : nullptr;
        if (want_hist)
            hipLaunchKernelGGL((k_ranges<true>), dim3((s->n_tiles + 255) / 256), dim3(256), 0, st, s->d_tile_contig.p,
                               s->d_tile_start.p, s->n_tiles, s->d_tlen.p, mask, s->d_ctg.p, s->d_desc.p, ti, s->d_tscan.p,
                               s->d_ttop.p, cx, s->d_glob.p, slow_list);
        else
            hipLaunchKernelGGL((k_ranges<false>), dim3((s->n_tiles + 255) / 256), dim3(256), 0, st, s->d_tile_contig.p,
                               s->d_tile_start.p, s->n_tiles, s->d_tlen.p, mask, s->d_ctg.p, s->d_desc.p, ti, s->d_tscan.p,
                               s->d_ttop.p, cx, s->d_glob.p, slow_list);
        time_end(s, COV_K_RANGES);
        HIPCHK(hipGetLastError());
        if (want_hist) {
            time_begin(s, COV_K_HIST);
            const u32 hb = (nT + 1023u) / 1024u;
            HIPCHK(s->d_hist_top.reserve(hb, st));
            hipLaunchKernelGGL((k_hist_sum<0>), dim3(hb), dim3(1024), 0, st, (const DevContig *)s->d_ctg.p, nT, s->d_tlen.p, mask, (u64)s->cfg.contig_end_exclusion, s->d_hist_top.p);
            hipLaunchKernelGGL((k_hist_off<0>), dim3(hb), dim3(1024), 0, st, s->d_ctg.p, nT, s->d_tlen.p, mask, (u64)s->cfg.contig_end_exclusion, (const u64 *)s->d_hist_top.p, s->d_glob.p);
            hipLaunchKernelGGL(k_zero_u32, dim3(2048), dim3(256), 0, st, s->d_arena.p, &s->d_glob.p->hist_cap_total);
            time_end(s, COV_K_HIST);
            HIPCHK(hipGetLastError());
        }
        PileupArgs a = pileup_args(s);
        time_begin(s, COV_K_PILEUP);
        if (want_hist) launch_any_pileup<true, false>(s, a, s->n_tiles);
        else launch_any_pileup<false, false>(s, a, s->n_tiles);
        time_end(s, COV_K_PILEUP);
        HIPCHK(hipGetLastError());
        if (want_hist) {
            compacted = s->est.n == 0 || s->hist_fetch_seen;
            s->hist_prefetched = 0;
            if (compacted) {
                HIPCHK(s->d_chist.reserve((size_t)R + nT + 1, st));
                time_begin(s, COV_K_HIST_COMPACT);
                const u32 hb = (nT + 1023u) / 1024u;
                hipLaunchKernelGGL((k_hist_sum<1>), dim3(hb), dim3(1024), 0, st, (const DevContig *)s->d_ctg.p, nT, s->d_tlen.p, mask, (u64)s->cfg.contig_end_exclusion, s->d_hist_top.p);
                hipLaunchKernelGGL((k_hist_off<1>), dim3(hb), dim3(1024), 0, st, s->d_ctg.p, nT, s->d_tlen.p, mask, (u64)s->cfg.contig_end_exclusion, (const u64 *)s->d_hist_top.p, s->d_glob.p);
                hipLaunchKernelGGL(k_hist_compact, dim3((nT + 3u) / 4u), dim3(256), 0, st, s->d_ctg.p, nT, s->d_tlen.p, (u64)s->cfg.contig_end_exclusion, s->d_arena.p, s->d_chist.p);
                time_end(s, COV_K_HIST_COMPACT);
                HIPCHK(hipGetLastError());
                // (only for a caller that fetched the histogram after the finish before this one: a caller that never does pays no copy)
                const u64 guess = s->hist_fetch_seen ? std::min<u64>({s->last_chist_total + s->last_chist_total / 16, (u64)R + nT + 1, (u64)(64u << 20) / 8}) : 0;
                if (guess) {
                    if (guess > s->h_chist_cap) {
                        if (s->h_chist) (void)hipHostFree(s->h_chist);
                        s->h_chist = nullptr; s->h_chist_cap = 0;
                        HIPCHK(hipHostMalloc((void **)&s->h_chist, (size_t)guess * 8, hipHostMallocDefault));
                        s->h_chist_cap = guess;
                    }
                    HIPCHK(hipMemcpyAsync(s->h_chist, s->d_chist.p, (size_t)guess * 8, hipMemcpyDeviceToHost, st));
                    s->hist_prefetched = guess;
                }
            }
            s->hist_fetch_seen = false;
        }
    }
    // (with a target mask the entries are genomes, aggregated on the host: masked-out contigs have no bins in the arena, and nobody may fetch
    // per-contig floats — no k_estimate launch, no floats in the copy)
    const size_t block = result_block_bytes(nT), nf = s->have_mask ? 0 : (size_t)nT * s->est.n;
    {
        result_host_take(s);
        const size_t need = result_host_bytes(s, nT);
        if (need > s->h_res_cap) {
            if (s->h_res) (void)hipHostFree(s->h_res);
            s->h_res = nullptr; s->h_res_cap = 0;
            const size_t cap = need < ((size_t)64 << 20) ? need + need / 2 : need;      // (room to grow for small blocks only)
            HIPCHK(hipHostMalloc((void **)&s->h_res, cap, hipHostMallocDefault));
            s->h_res_cap = cap;
        }
        s->h_ctg = (DevContig *)(s->h_res + sizeof(DevGlobal));
        s->h_estf = reinterpret_cast<float *>(s->h_res + block);
    }
    if (want_id && R && nT) { HIPCHK(hipStreamWaitEvent(st, s->ev_side_done, 0)); s->ev_fresh = -1; }      // (the wait is not the next group's time)
    if (nf) {      // CoverageEstimator::calculate_coverage of every contig (k_init left n_pass = 0 everywhere when nothing ran: rows of zeros)
        time_begin(s, COV_K_ESTIMATE);
        if (s->est_lanes < 0 ? nT >= 65536u : s->est_lanes == 1)      // an assembly: a lane per contig (pileup_kernels.hip.h, estimate_body)
            hipLaunchKernelGGL(k_estimate_lanes, dim3((nT + 255) / 256), dim3(256), 0, st, (const DevContig *)s->d_ctg.p, nT, (const u32 *)s->d_tlen.p, (u64)s->cfg.contig_end_exclusion,
                               (const u32 *)s->d_arena.p, s->est, reinterpret_cast<float *>(s->d_res.p + block));
        else
            hipLaunchKernelGGL(k_estimate, dim3((nT + 3) / 4), dim3(256), 0, st, (const DevContig *)s->d_ctg.p, nT, (const u32 *)s->d_tlen.p, (u64)s->cfg.contig_end_exclusion,
                               (const u32 *)s->d_arena.p, s->est, reinterpret_cast<float *>(s->d_res.p + block));
        time_end(s, COV_K_ESTIMATE);
        HIPCHK(hipGetLastError());
    }
    // one copy: the result block and, behind it, the estimators' floats
    HIPCHK(hipMemcpyAsync(s->h_res, s->d_res.p, sizeof(DevGlobal) + (size_t)nT * sizeof(DevContig) + (nf ? (block - sizeof(DevGlobal) - (size_t)nT * sizeof(DevContig)) + nf * sizeof(float) : 0),
                          hipMemcpyDeviceToHost, st));
    // (c) a finish is a millisecond: the host looks for its end itself for a while before it asks to be woken
    {
        const auto t_spin = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
            (void)hipGetLastError();
            if (std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(4)) break;
            __builtin_ia32_pause();
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    memcpy(&s->h_glob, s->h_res, sizeof(DevGlobal));
    if (s->h_glob.cx_total > s->d_cx_runs.cap) {
        const uint64_t need = s->h_glob.cx_total + s->h_glob.cx_total / 8 + 1024;
        s->d_cx_runs.release();
        HIPCHK(s->d_cx_runs.reserve((size_t)need, st));
        again = true;
        return COV_OK;
    }
    for (int k = 0; k < COV_K_COUNT; k++)
        if (s->k_launches[k]) (void)hipEventElapsedTime(&s->k_ms[k], k == COV_K_IDENTITY ? s->ev[k][0] : s->ev_begin[k], s->ev[k][1]);

    // algorithmic bytes: each record's SoA fields and CIGAR words are needed once; results written once
    {
        const uint64_t ncig = s->adopted ? s->adopted_ncig : s->n_cigar;
        s->algo_bytes = (uint64_t)R * 24 + ncig * 4 + (uint64_t)nT * sizeof(DevContig);
    }

    const cov_status cst = convert_results(s, s->h_glob, s->h_ctg, R, stats, summary, s->spill.inflight, s->spill.records);
    if (cst != COV_OK) return cst;
    s->last_chist_total = 0;
    if (want_hist) for (u32 c = 0; c < nT; c++) s->last_chist_total += stats[c].hist_len;      // (the host laid the compact histogram out: convert_results)
    s->hist_compacted = compacted;
    s->est_valid = s->est.n != 0 && !s->have_mask;
    s->finished = true;
    return COV_OK;
}


// ---- bounded record store: spill and merge (see cov_session::spill)
static cov_status fetch_chunk_hist(cov_session *s, uint64_t *hist);

extern "C++" {
namespace {
template <typename T>
cov_status move_front(cov_session *s, T *p, size_t from, size_t n) {
    if (!n || !from) return COV_OK;
    hipStream_t st = s->stream;
    if (n <= from) { HIPCHK(hipMemcpyAsync(p, p + from, n * sizeof(T), hipMemcpyDeviceToDevice, st)); return COV_OK; }      // disjoint
    HIPCHK(s->d_spill_tmp.reserve(n * sizeof(T), st));
    HIPCHK(hipMemcpyAsync(s->d_spill_tmp.p, p + from, n * sizeof(T), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(p, s->d_spill_tmp.p, n * sizeof(T), hipMemcpyDeviceToDevice, st));
    return COV_OK;
}

// Runs the pipeline over the store, keeps every contig but the one in flight, moves the records from that contig's first considered
// record onwards to the front.  The contig in flight = the last considered contig (after the order check: the highest tid with a
// considered record); every considered record behind its first one is its own (anything else would have failed the order check),
// so the contigs in front of it are complete — the reference would have flushed them at the tid change (contig.rs:128-155).
// progress = false: nothing could leave (one contig fills the store, or the store is an adopted batch / carries mate columns).
cov_status spill_store_impl(cov_session *s, bool &progress) {
    progress = false;
    if (s->adopted || s->n_records == 0 || s->want_mates || s->n_targets == 0) return COV_OK;
    covr::Range rr("bounded store: spill");
    const u32 nT = s->n_targets;
    const bool want_hist = s->cfg.want & COV_WANT_HIST;
    std::vector<cov_contig_stats> st(nT);
    cov_summary sm{};
    cov_status rc = finish_store(s, st.data(), &sm);
    if (rc != COV_OK) return rc;
    const u64 R = s->n_records;
    cov_session::Spill &S = s->spill;
    int64_t cstar = -1; u64 best = 0;
    for (u32 c = 0; c < nT; c++)
        if (st[c].n_pass && (cstar < 0 || st[c].last_record >= best)) { cstar = c; best = st[c].last_record; }
    const u64 keep_from = cstar >= 0 ? st[cstar].first_record - S.records : R;      // (record indices come back counted from the sample's first record)
    if (keep_from == 0) { s->finished = false; return COV_OK; }
    if (!S.active) { S.stats.assign(nT, cov_contig_stats{}); S.ctg.assign(nT, DevContig{}); S.have.assign(nT, 0); S.active = true; }
    std::vector<uint64_t> h;
    if (want_hist && sm.hist_total) { h.resize(sm.hist_total); rc = fetch_chunk_hist(s, h.data()); if (rc != COV_OK) return rc; }
    for (u32 c = 0; c < nT; c++) {
        if ((int64_t)c == cstar || !st[c].n_pass) continue;
        if (S.have[c]) { s->err = "BAM file appears to be unsorted. Input BAM files must be sorted by reference (i.e. by samtools sort)"; return COV_ERR_UNSORTED; }
        S.stats[c] = st[c]; S.ctg[c] = s->h_ctg[c]; S.have[c] = 1;
        if (s->est.n) {
            if (s->spill_est.size() != (size_t)nT * s->est.n) s->spill_est.assign((size_t)nT * s->est.n, 0.0f);
            memcpy(&s->spill_est[(size_t)c * s->est.n], s->h_estf + (size_t)c * s->est.n, s->est.n * sizeof(float));
        }
        if (want_hist && st[c].hist_len) {
            S.stats[c].hist_off = S.hist.size();
            S.hist.insert(S.hist.end(), h.begin() + st[c].hist_off, h.begin() + st[c].hist_off + st[c].hist_len);
        }
    }
    hipStream_t q = s->stream;
    u64 prim_keep = 0;
    u32 c0 = (u32)s->n_cigar;
    if (keep_from < R) {
        unsigned long long *ctr = reinterpret_cast<unsigned long long *>(&s->d_glob.p->pad2[1]);
        HIPCHK(hipMemsetAsync(ctr, 0, 8, q));
        hipLaunchKernelGGL(k_count_primary, dim3((u32)std::min<u64>((R - keep_from + 255) / 256, (u64)s->n_cus * 8)), dim3(256), 0, q, (const uint16_t *)s->s_flag.p, (u32)keep_from, (u32)R, ctr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&prim_keep, ctr, 8, hipMemcpyDeviceToHost, q));
        HIPCHK(hipMemcpyAsync(&c0, s->s_coff.p + keep_from, 4, hipMemcpyDeviceToHost, q));
        HIPCHK(hipStreamSynchronize(q));
    }
    S.prim += sm.num_detected_primary_alignments - prim_keep;
    S.cons += sm.n_considered - (cstar >= 0 ? st[cstar].n_pass : 0);
    S.records += keep_from;
    if (cstar > S.inflight) S.inflight = cstar;
    S.count++;
    const u64 n_keep = R - keep_from, cig_keep = s->n_cigar - c0;
    cov_status m = COV_OK;
    if ((m = move_front(s, s->s_tid.p, keep_from, n_keep)) || (m = move_front(s, s->s_pos.p, keep_from, n_keep)) || (m = move_front(s, s->s_flag.p, keep_from, n_keep)) ||
        (m = move_front(s, s->s_mapq.p, keep_from, n_keep)) || (m = move_front(s, s->s_nmk.p, keep_from, n_keep)) || (m = move_front(s, s->s_nm.p, keep_from, n_keep)) ||
        (m = move_front(s, s->s_lseq.p, keep_from, n_keep)) || (m = move_front(s, s->s_coff.p, keep_from, n_keep + 1)) || (m = move_front(s, s->s_cig.p, c0, cig_keep)))
        return m;
    if (n_keep && c0) {
        hipLaunchKernelGGL(k_rebase_offsets, dim3((u32)((n_keep + 1 + 255) / 256)), dim3(256), 0, q, s->s_coff.p, (u32)(n_keep + 1), c0, 0u);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(q));
    s->n_records = n_keep; s->n_cigar = cig_keep;
    s->finished = false; s->depth_all_valid = false;
    if (cov_timing_on())
        fprintf(stderr, "[covermhip] bounded store: spill %u, %llu records left the store (%llu stay: contig %lld in flight), %llu since the sample began\n", S.count,
                (unsigned long long)keep_from, (unsigned long long)n_keep, (long long)cstar, (unsigned long long)S.records);
    progress = true;
    return COV_OK;
}
}  // namespace
cov_status spill_store(cov_session *s, bool &progress) { return spill_store_impl(s, progress); }
}  // extern "C++"

// cov_finish after one or more spills: `stats` / `summary` hold the pass over what the store still held; the contigs that left
// earlier come back from the host, the histogram is re-based into one array (cov_fetch_hist serves it), and the device's result
// block is rewritten as the merged one so that cov_gather sends the whole sample.
static cov_status merge_spilled(cov_session *s, cov_contig_stats *stats, cov_summary *summary) {
    cov_session::Spill &S = s->spill;
    const u32 nT = s->n_targets;
    const bool want_hist = s->cfg.want & COV_WANT_HIST;
    std::vector<uint64_t> h;
    if (want_hist && summary->hist_total) { h.resize(summary->hist_total); const cov_status rc = fetch_chunk_hist(s, h.data()); if (rc != COV_OK) return rc; }
    s->merged_hist = S.hist;
    std::vector<DevContig> m(nT);
    u32 rank = 0;
    for (u32 c = 0; c < nT; c++) {
        if (S.have[c]) {
            if (stats[c].n_pass) { s->err = "BAM file appears to be unsorted. Input BAM files must be sorted by reference (i.e. by samtools sort)"; return COV_ERR_UNSORTED; }
            stats[c] = S.stats[c]; m[c] = S.ctg[c];
            if (s->est.n && s->est_valid && s->spill_est.size() == (size_t)nT * s->est.n)
                memcpy(s->h_estf + (size_t)c * s->est.n, &s->spill_est[(size_t)c * s->est.n], s->est.n * sizeof(float));
        } else {
            m[c] = s->h_ctg[c];
            if (want_hist && stats[c].hist_len) {
                const u64 off = s->merged_hist.size();
                s->merged_hist.insert(s->merged_hist.end(), h.begin() + stats[c].hist_off, h.begin() + stats[c].hist_off + stats[c].hist_len);
                stats[c].hist_off = off;
            }
        }
        m[c].chist_off = stats[c].hist_off; m[c].hist_len = stats[c].hist_len;
        // the block a peer converts (cov_gathered) judges the order from first_rec / last_rec: the merged block carries the contigs' ranks
        if (m[c].n_pass) { m[c].first_rec = m[c].last_rec = rank++; }
    }
    summary->num_detected_primary_alignments += S.prim; summary->n_records += S.records; summary->n_considered += S.cons;
    summary->hist_total = s->merged_hist.size();
    s->merged_valid = true; s->merged_records = summary->n_records;
    DevGlobal g = s->h_glob;
    g.prim_slots[0] += S.prim; g.cons_slots[0] += S.cons; g.chist_total = s->merged_hist.size();
    HIPCHK(hipSetDevice(s->cfg.device));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(s->d_res.p, &g, sizeof g, hipMemcpyHostToDevice));
    if (nT) HIPCHK(hipMemcpy(s->d_res.p + sizeof(DevGlobal), m.data(), (size_t)nT * sizeof(DevContig), hipMemcpyHostToDevice));
    return COV_OK;
}


cov_status cov_reserve(cov_session *s, uint64_t n_records, uint64_t n_cigar) {
    if (!s) return COV_ERR_INVALID_ARG;
    if (n_records >= 0xfffffff0ull || n_cigar >= 0xfffffff0ull) { s->err = "more than 2^32 records in one session"; return COV_ERR_INVALID_ARG; }
    HIPCHK(hipSetDevice(s->cfg.device));
    hipStream_t st = s->stream;
    const uint64_t R = s->adopted ? 0 : s->n_records, C = s->adopted ? 0 : s->n_cigar;
    HIPCHK(s->s_tid.reserve(n_records, st, R)); HIPCHK(s->s_pos.reserve(n_records, st, R)); HIPCHK(s->s_flag.reserve(n_records, st, R));
    HIPCHK(s->s_mapq.reserve(n_records, st, R)); HIPCHK(s->s_nmk.reserve(n_records, st, R)); HIPCHK(s->s_nm.reserve(n_records, st, R));
    HIPCHK(s->s_lseq.reserve(n_records, st, R)); HIPCHK(s->s_coff.reserve(n_records + 1, st, R ? R + 1 : 0));
    HIPCHK(s->s_cig.reserve(n_cigar + 1, st, C));
    return COV_OK;
}

// ---- cov_gather: ONE RCCL gather of every rank's result block to the root rank's device (north star: "a single RCCL gather
// of per-contig results over xGMI").  librccl is bound at run time (dlopen), so single-GPU users need no RCCL at all.
namespace {
struct Rccl {
    typedef int (*init_all_t)(void **, int, const int *);
    typedef int (*gather_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef int (*grp_t)();
    typedef const char *(*errstr_t)(int);
    init_all_t init_all = nullptr; gather_t gather = nullptr; grp_t group_start = nullptr, group_end = nullptr; errstr_t errstr = nullptr;
    bool ok = false;
    Rccl() {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        init_all = (init_all_t)dlsym(h, "ncclCommInitAll"); gather = (gather_t)dlsym(h, "ncclGather");
        group_start = (grp_t)dlsym(h, "ncclGroupStart"); group_end = (grp_t)dlsym(h, "ncclGroupEnd");
        errstr = (errstr_t)dlsym(h, "ncclGetErrorString");
        ok = init_all && gather && group_start && group_end;
    }
};
Rccl *rccl_lib() { static Rccl r; return &r; }
std::mutex g_comm_mutex;
std::map<std::vector<int>, std::vector<void *>> g_comms;   // communicators per device list, kept for the process lifetime
}  // namespace

cov_status cov_gather(cov_session *const *sessions, uint32_t n, uint32_t root) {
    if (!sessions || n == 0 || root >= n) return COV_ERR_INVALID_ARG;
    cov_session *s = sessions[root];
    const u32 nT = s->n_targets;
    std::vector<int> devs(n);
    bool distinct = true;
    for (u32 i = 0; i < n; i++) {
        if (!sessions[i] || !sessions[i]->finished || sessions[i]->n_targets != nT) { s->err = "cov_gather: every session must be finished over the same targets"; return COV_ERR_STATE; }
        devs[i] = sessions[i]->cfg.device;
        for (u32 j = 0; j < i; j++) if (devs[j] == devs[i]) distinct = false;
    }
    const size_t block = result_block_bytes(nT);
    HIPCHK(hipSetDevice(s->cfg.device));
    HIPCHK(s->d_gather.reserve((size_t)n * block, s->stream));
    if ((size_t)n * block > s->h_gather_cap) {
        if (s->h_gather) (void)hipHostFree(s->h_gather);
        s->h_gather = nullptr; s->h_gather_cap = 0;
        HIPCHK(hipHostMalloc((void **)&s->h_gather, (size_t)n * block, hipHostMallocDefault));
        s->h_gather_cap = (size_t)n * block;
    }
    for (u32 i = 0; i < n; i++) {   // the record count of each rank travels in its block (a padding word of DevGlobal)
        HIPCHK(hipSetDevice(devs[i]));
        const u64 nr = sessions[i]->merged_valid ? sessions[i]->merged_records : sessions[i]->n_records;
        HIPCHK(hipMemcpyAsync(&sessions[i]->d_glob.p->pad2[0], &nr, 8, hipMemcpyHostToDevice, sessions[i]->stream));
        HIPCHK(hipStreamSynchronize(sessions[i]->stream));
    }
    if (distinct && (n > 1 || getenv("COVERM_FORCE_RCCL") != nullptr)) {
        Rccl &R = *rccl_lib();
        if (!R.ok) { s->err = "cov_gather: librccl.so.1 not available"; return COV_ERR_HIP; }
        std::vector<void *> comms;
        {
            std::lock_guard<std::mutex> lk(g_comm_mutex);
            auto it = g_comms.find(devs);
            if (it == g_comms.end()) {
                std::vector<void *> c(n, nullptr);
                const int rc = R.init_all(c.data(), (int)n, devs.data());
                if (rc != 0) { s->err = std::string("ncclCommInitAll: ") + (R.errstr ? R.errstr(rc) : "error"); return COV_ERR_HIP; }
                it = g_comms.emplace(devs, c).first;
            }
            comms = it->second;
        }
        int rc = R.group_start();
        for (u32 i = 0; i < n && rc == 0; i++) {
            (void)hipSetDevice(devs[i]);
            rc = R.gather(sessions[i]->d_res.p, i == root ? s->d_gather.p : nullptr, block, /* ncclUint8 */ 1, (int)root, comms[i], sessions[i]->stream);
        }
        const int rc2 = R.group_end();
        if (rc != 0 || rc2 != 0) { s->err = std::string("ncclGather: ") + (R.errstr ? R.errstr(rc ? rc : rc2) : "error"); return COV_ERR_HIP; }
        for (u32 i = 0; i < n; i++) { HIPCHK(hipSetDevice(devs[i])); HIPCHK(hipStreamSynchronize(sessions[i]->stream)); }
    } else {
        // one device given more than once (functional checks on a single-GPU box): RCCL refuses two ranks on one device,
        // so the blocks move by plain device copies instead
        for (u32 i = 0; i < n; i++) {
            HIPCHK(hipSetDevice(devs[i]));
            HIPCHK(hipStreamSynchronize(sessions[i]->stream));
        }
        HIPCHK(hipSetDevice(s->cfg.device));
        for (u32 i = 0; i < n; i++)    // on the root's stream: ordered before the copy to the host below (a device-to-device hipMemcpyPeer may return early)
            HIPCHK(hipMemcpyPeerAsync(s->d_gather.p + (size_t)i * block, s->cfg.device, sessions[i]->d_res.p, devs[i], block, s->stream));
    }
    HIPCHK(hipSetDevice(s->cfg.device));
    HIPCHK(hipMemcpyAsync(s->h_gather, s->d_gather.p, (size_t)n * block, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->gather_n = n; s->gather_block = block;
    return COV_OK;
}

cov_status cov_gathered(cov_session *root, uint32_t rank, cov_contig_stats *stats, cov_summary *summary) {
    if (!root || !root->h_gather || rank >= root->gather_n || (!stats && root->n_targets)) return COV_ERR_INVALID_ARG;
    const uint8_t *blk = root->h_gather + (size_t)rank * root->gather_block;
    DevGlobal G; memcpy(&G, blk, sizeof G);
    return convert_results(root, G, reinterpret_cast<const DevContig *>(blk + sizeof(DevGlobal)), G.pad2[0], stats, summary);
}


// ---------------------------------------------------------------------------------------------- device ingest (covermhip.h cov_ingest_*)
static_assert(sizeof(cov_bgzf_block) == sizeof(covi::BgzfBlock) && offsetof(cov_bgzf_block, in_len) == offsetof(covi::BgzfBlock, in_len) &&
              offsetof(cov_bgzf_block, out_off) == offsetof(covi::BgzfBlock, out_off), "cov_bgzf_block mirrors the device struct");

constexpr int INF1_LB = 7, INF1_DB = 6;       // k_inflate's primary tables: 7 + 6 bits = 28 KiB per wave, five waves per CU (the fastest measured)
// Blocks per k_inflate_wave round = per window.  81 920 from round 3 to the middle of round 6 (round 4's sweep: smaller windows lost to the fixed
// host cost per window); with the feed at the link's rate and round 5-6's faster kernels the sweep was repeated at 200 M reads,
// alternating: ingest 0.478-0.485 s with 61 440 blocks against 0.487-0.494 s with 81 920, 0.484-0.494 s with 40 960, 0.496-0.505 s with 122 880
// (profiles/r06_round_blocks_ab_200M.json) — what the device still has to do when the last byte has arrived is about one window's
// processing, 32 ms instead of 44.  61 440 is also exactly 15 launches' worth of resident waves on 256 CUs (16 per CU).
constexpr u32 WAVE_ROUND_BLOCKS = 61440;
// Decided once per ingest (cov_ingest_begin) and kept in the session: the switches are read there, never in a launch path.
static InflateKernel choose_inflate_kernel(cov_session *s) {
    InflateKernel K;
    const char *ve = getenv("COVERM_INFLATE_V");
    K.version = ve && atoi(ve) == 1 ? 1 : 3;
    if (const char *lz = getenv("COVERM_LZ_V")) K.lz_version = atoi(lz) == 1 ? 1 : 2;      // 1: k_lz_resolve (rounds through global memory), 2: k_lz_stage (batches staged in LDS)
    if (K.version == 1) {
        int per_cu = 0;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&covi::k_inflate<INF1_LB, INF1_DB, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)covi::inflate_smem_bytes(INF1_LB, INF1_DB));
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, covi::k_inflate<INF1_LB, INF1_DB, false>, 64, covi::inflate_smem_bytes(INF1_LB, INF1_DB));
        if (per_cu <= 0) per_cu = 2;
        K.round_blocks = (u32)s->n_cus * (u32)per_cu * 64u;
    } else K.round_blocks = WAVE_ROUND_BLOCKS;
    { long long v; if (covknob::get("ingest_round_blocks", v) && v >= 64) K.round_blocks = (u32)v / 64u * 64u; }   // tests: many small windows
    { long long v; if (covknob::get("ingest_carry_kb", v) && v >= 1) K.carry = (u64)v << 10; }
    K.carry = (K.carry + 63u) & ~63ull;
    // compressed bytes one round may span: 32 KiB per block on average (BGZF blocks of BAM files compress to 15-25 KiB); a round of
    // less compressible blocks simply closes earlier
    K.cwin = std::max<u64>((u64)K.round_blocks * 32768u, 1ull << 20);
    { long long v; if (covknob::get("ingest_cwin_kb", v) && v >= 256) K.cwin = (u64)v << 10; }
    return K;
}
static inline const InflateKernel &inflate_kernel(cov_session *s) { return s->ing_K; }

cov_status cov_ingest_begin(cov_session *s, uint64_t compressed_bytes, uint64_t first_record_offset, int check_crc) {
    if (!s) return COV_ERR_INVALID_ARG;
    covr::Range rr("ingest: buffers, streams, events (cov_ingest_begin)");
    HIPCHK(hipSetDevice(s->cfg.device));
    HIPCHK(ingest_prep_wait(s));
    if (!s->ing_copy) HIPCHK(ingest_prepare_(s));
    if (s->adopted) {  // materialise an adopted device batch into the owned store first
        cov_batch ab = s->adopted_batch;
        s->adopted = false; s->n_records = 0; s->n_cigar = 0;
        cov_status a = append(s, &ab, true);
        if (a) return a;
    }
    HIPCHK(hipStreamSynchronize(s->stream));     // the record store is about to be written from the parse stream
    s->ing_batch = 0; s->ing_extracted = 0; s->ing_rec_total = s->ing_cig_total = 0; s->ing_fail = 0; s->ing_rec_spilled = 0;
    s->ing_K = choose_inflate_kernel(s);      // (also sets the kernel's dynamic-LDS limit on this session's device)
    s->ing_first_record = first_record_offset;
    s->ing_check_crc = check_crc ? 1 : 0;
    HIPCHK(s->g_result.reserve(8 + 4 * 8, s->stream));
    HIPCHK(s->g_carry.reserve(inflate_kernel(s).carry, s->stream));
    if (!s->g_crc_tab.p) {      // (cov_ingest_release between two files gave the tables back with the buffers)
        std::vector<u32> t(covi::crcw::TABLE_WORDS);
        covi::crcw::build_tables(t.data());
        HIPCHK(s->g_crc_tab.reserve(covi::crcw::TABLE_WORDS, s->stream));
        HIPCHK(hipMemcpy(s->g_crc_tab.p, t.data(), covi::crcw::TABLE_WORDS * sizeof(u32), hipMemcpyHostToDevice));
    }
    HIPCHK(s->g_blocks.reserve(compressed_bytes / 8192 + 1024, s->stream));
    HIPCHK(s->g_status.reserve(compressed_bytes / 8192 + 1024, s->stream));
    if (compressed_bytes / 16384 + 1024 > s->h_blocks_cap) {     // page-locked mirror of the block table: sized once for ordinary ~20 KB blocks (it grows if they are smaller)
        if (s->h_blocks) { (void)hipHostFree(s->h_blocks); s->h_blocks = nullptr; s->h_blocks_cap = 0; }
        const size_t nc = compressed_bytes / 16384 + 1024;
        HIPCHK(hipHostMalloc((void **)&s->h_blocks, nc * sizeof(covi::BgzfBlock), hipHostMallocDefault));
        s->h_blocks_cap = nc;
    }
    HIPCHK(hipMemsetAsync(s->g_result.p, 0, (8 + 4 * 8) * sizeof(u64), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->ing_comp = compressed_bytes; s->ing_infl = 0; s->ing_blocks = 0; s->ing_launched = 0; s->ing_active = true;
    s->ing_round_start = 0; s->ing_prev_end = 0; s->ing_round_n = 0; s->ing_copy_cleared = -1;
    s->ing_key_lo = 0; s->ing_key_hi = 0x80000000ll; s->ing_search_first = false; s->ing_open_end = false; s->ing_fed_any = false;
    s->ing_span_lo = 0; s->ing_span_hi = compressed_bytes; s->ing_tail_key = ~0ull;
    s->ing_ccap = std::min<u64>(inflate_kernel(s).cwin, compressed_bytes + 65536u + 128u);     // a small file is one buffer: the byte rule never fires
    s->ing_s_alloc = 0; for (double &x : s->ing_s_part) x = 0;
    return COV_OK;
}

cov_status cov_ingest_span(cov_session *s, int64_t key_lo, int64_t key_hi, int search_first_record, int open_end, uint64_t file_lo, uint64_t file_hi) {
    if (!s || !s->ing_active || s->ing_fed_any || key_lo < 0 || key_hi < key_lo || file_hi < file_lo || file_hi > s->ing_comp) return COV_ERR_INVALID_ARG;
    s->ing_key_lo = key_lo; s->ing_key_hi = key_hi; s->ing_search_first = search_first_record != 0; s->ing_open_end = open_end != 0;
    s->ing_span_lo = file_lo; s->ing_span_hi = file_hi;
    return COV_OK;
}

// Records of the windows whose boundaries are verified go into the store: all windows up to `must_upto` (waiting for their
// k_bam_verify if need be), later ones only if their result is already here.
static cov_status ingest_drain_(cov_session *s, int64_t must_upto);
struct PartTimer { double &acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); ~PartTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } };
static cov_status ingest_drain(cov_session *s, int64_t must_upto) { PartTimer t{s->ing_s_part[0]}; covr::Range rr("ingest: verify + extract windows"); return ingest_drain_(s, must_upto); }
static cov_status ingest_drain_(cov_session *s, int64_t must_upto) {
    hipStream_t ps = s->ing_ext;     // the host has seen the window's verification finish: nothing on the device to wait for
    while (s->ing_extracted < s->ing_batch) {
        const u32 w = s->ing_extracted, q = w & 3u;
        if ((int64_t)w > must_upto) { if (hipEventQuery(s->ing_ver_done[q]) != hipSuccess) { (void)hipGetLastError(); break; } }   // not ready is not an error
        else { PartTimer pt{s->ing_s_part[3]}; HIPCHK(hipEventSynchronize(s->ing_ver_done[q])); }
        const u64 *res = s->h_winres + 8 * q;
        const u64 nrec = res[0], ncig = res[1];
        u32 st = (u32)res[2];
        s->ing_tail_key = res[7];
        const bool span = s->ing_key_lo > 0 || s->ing_key_hi < 0x80000000ll || s->ing_search_first || s->ing_open_end;
        if (!span && !s->want_mates) st &= ~64u;       // whole file: cov_finish reports disorder in file order, beside the other per-record errors
        if (st && !s->ing_fail) { s->ing_fail = st; s->ing_fail_dbg[0] = res[4]; s->ing_fail_dbg[1] = res[5]; s->ing_fail_dbg[2] = res[6]; }
        u64 R = s->n_records + s->ing_rec_total, Cg = s->n_cigar + s->ing_cig_total;
        if (!s->ing_fail && nrec && R && !s->want_mates && (R + nrec > s->cap_records || Cg + ncig > s->cap_cigar)) {
            // bounded store: what has been extracted so far becomes the store's content, the pipeline runs over it, the complete contigs
            // leave for the host and the contig in flight moves to the front; this window's records follow it
            HIPCHK(hipStreamSynchronize(ps));
            const u32 end_off = (u32)Cg;
            HIPCHK(hipMemcpyAsync(s->s_coff.p + R, &end_off, sizeof end_off, hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
            s->n_records = R; s->n_cigar = Cg;
            s->ing_rec_spilled += s->ing_rec_total; s->ing_rec_total = 0; s->ing_cig_total = 0;
            bool progress = false;
            const cov_status sp = spill_store(s, progress);
            if (sp != COV_OK) return sp;
            R = s->n_records; Cg = s->n_cigar;
        }
        if (!s->ing_fail && (R + nrec >= 0xfffffff0ull || Cg + ncig >= 0xfffffff0ull)) s->ing_fail = 16u;
        if (!s->ing_fail && nrec) {
            u64 Nn = R + nrec, Cn = Cg + ncig + 1;
            // first of several windows (it ends in front of the span's end): size the store for the whole file at once.  (Not "several
            // windows launched so far": with a fast inflate kernel window 0 is verified before window 1 is launched, the store was then
            // sized for one window, and every later window grew it — allocate, copy, drain the device, free: 0.8 s of stalls at 100 M reads,
            // profiles/r04_timeline_store_regrowth.txt.)
            if (w == 0 && s->ing_win[0].comp_end && s->ing_win[0].comp_end + 65536u < s->ing_span_hi) {
                const double scale = (double)(s->ing_span_hi - s->ing_span_lo) / (double)std::max<u64>(1, s->ing_win[0].comp_end - std::min<u64>(s->ing_win[0].comp_end, s->ing_span_lo)) * 1.1;
                Nn = std::max<u64>(Nn, R + (u64)((double)nrec * scale) + 1024); Cn = std::max<u64>(Cn, Cg + (u64)((double)ncig * scale) + 1024);
                // (never beyond the store's cap: what would lie behind it is spilled before it is written)
                Nn = std::max<u64>(R + nrec, std::min<u64>(Nn, std::min<u64>(s->cap_records + 1024, 0xfffffff0ull)));
                Cn = std::max<u64>(Cg + ncig + 1, std::min<u64>(Cn, std::min<u64>(s->cap_cigar + 1024, 0xfffffff0ull)));
            }
            const auto ta0 = std::chrono::steady_clock::now();
            HIPCHK(s->s_tid.reserve(Nn, ps, R)); HIPCHK(s->s_pos.reserve(Nn, ps, R)); HIPCHK(s->s_flag.reserve(Nn, ps, R));
            HIPCHK(s->s_mapq.reserve(Nn, ps, R)); HIPCHK(s->s_nmk.reserve(Nn, ps, R)); HIPCHK(s->s_nm.reserve(Nn, ps, R));
            HIPCHK(s->s_lseq.reserve(Nn, ps, R)); HIPCHK(s->s_coff.reserve(Nn + 1, ps, R + 1));
            HIPCHK(s->s_cig.reserve(Cn, ps, Cg));
            if (s->want_mates) { HIPCHK(s->s_mtid.reserve(Nn, ps, R)); HIPCHK(s->s_qh1.reserve(Nn, ps, R)); HIPCHK(s->s_qh2.reserve(Nn, ps, R)); }
            s->ing_s_alloc += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta0).count();
            covi::RecStore RS{};
            if (s->want_mates) { RS.mtid = s->s_mtid.p; RS.qh1 = s->s_qh1.p; RS.qh2 = s->s_qh2.p; }
            RS.tid = s->s_tid.p; RS.pos = s->s_pos.p; RS.flag = s->s_flag.p; RS.mapq = s->s_mapq.p; RS.nm_kind = s->s_nmk.p; RS.nm = s->s_nm.p;
            RS.l_seq = s->s_lseq.p; RS.cigar_off = s->s_coff.p; RS.cigar = s->s_cig.p; RS.rec0 = R; RS.cig0 = Cg;
            covi::BamScan S{};
            S.u = s->g_win[w % 3u].p; S.N = s->ing_win[q].N; S.p0 = s->g_result.p + 6; S.seg_bytes = 32768; S.n_seg = s->ing_win[q].n_seg;
            S.n_ref = (int)s->n_targets; S.ref_len = s->d_tlen.p; S.final = 0; S.key_lo = s->ing_key_lo; S.key_hi = s->ing_key_hi; S.search_first = 0;
            hipLaunchKernelGGL(covi::k_bam_extract, dim3((S.n_seg * s->ing_K.ext_parts + 63) / 64), dim3(64), 0, ps, S, (const covi::SegInfo *)s->g_seg[q].p, (const u64 *)s->g_recbase[q].p,
                               (const u64 *)s->g_cigbase[q].p, RS, reinterpret_cast<u32 *>(s->g_result.p + 3) + 1, s->ing_K.ext_parts);
            HIPCHK(hipGetLastError());
            s->ing_rec_total += nrec; s->ing_cig_total += ncig;
        }
        HIPCHK(hipEventRecord(s->ing_ext_done[w % 3u], ps));
        s->ing_extracted++;
    }
    return COV_OK;
}

// One round = one window: k_inflate over the next `n` blocks fed but not yet launched (main stream), k_lz_resolve + k_crc32
// behind it (aux stream, beside the next round's k_inflate), then the window's record boundaries (parse stream).
// A lane decodes a whole block serially and every block takes about the same time T, so a launch costs T per started ROUND
// of resident waves: launches are cut to exactly the number of blocks the device holds at once (a launch of 1.05 rounds
// costs 2 T — measured: 46 ms per launch of ~51 k blocks against 23.5 ms per round of 49 152).
// Windows bound the memory: the inflated stream of a 200 M-read BAM is 62 GB, and device allocations cost ~30 ms per GB.
static cov_status launch_round_(cov_session *s, uint64_t n64, bool final);
static cov_status launch_round(cov_session *s, uint64_t n64, bool final) { PartTimer t{s->ing_s_part[1]}; covr::Range rr("ingest: inflate round launch"); return launch_round_(s, n64, final); }
static cov_status launch_round_(cov_session *s, uint64_t n64, bool final) {
    const uint64_t b0 = s->ing_launched;
    n64 = std::min<uint64_t>(n64, s->ing_blocks - b0);
    const InflateKernel &K = inflate_kernel(s);
    const u32 n = (u32)n64, w = s->ing_batch, q = w & 3u;
    if (w >= 3) { const cov_status d = ingest_drain(s, (int64_t)w - 3); if (d != COV_OK) return d; }   // this window's buffer and parse state are free again
    const int bb = (int)(w & 1u);
    DevBuf<covi::tokpos_t> &tokb = bb ? s->g_tok2 : s->g_tok;
    DevBuf<u32> &ntokb = bb ? s->g_ntok2 : s->g_ntok;
    DevBuf<uint8_t> &win = s->g_win[w % 3u];
    // sized for a full round at once (no regrowth between rounds) — or for the blocks a SMALL file can be expected to have: a 200 MB file does not
    // need three 4 GB windows and two 2.7 GB token lists, and a process that starts right after another one has given tens of GB back waits
    // for the driver in its large allocations (0.06 s for these in a 2 M-read run, tools/r06/call43.sh).  The estimate is one block per
    // 4 KiB of compressed bytes; a file of smaller blocks takes the regrowth path below.
    const size_t est_blocks = (size_t)((s->ing_span_hi - std::min(s->ing_span_lo, s->ing_span_hi)) / 4096u + 256u);
    const size_t full = std::max<size_t>(n, std::min<size_t>(K.round_blocks, est_blocks));
    const size_t win_bytes = (size_t)K.carry + full * 65536u + 64u;
    const u32 seg_cap = (u32)((win_bytes + 32767u) / 32768u);
    {
        // A buffer that has to GROW may still be in use: drain the device first.  First-time allocations need no such thing —
        // and must not wait, or the first four windows (three window buffers, four parse-state sets) would run one after the other.
        const size_t scr = (full + 63) / 64 * 64u * covi::INF_SCRATCH_BYTES;
        auto grows = [](size_t need, size_t cap) { return cap != 0 && need > cap; };
        if (grows(full * covi::INF_TOK_CAP, tokb.cap) || grows(full, ntokb.cap) || grows(scr, s->g_scratch.cap) || grows(win_bytes, win.cap) ||
            grows(seg_cap, s->g_seg[q].cap)) {
            HIPCHK(hipStreamSynchronize(s->ing_aux)); HIPCHK(hipStreamSynchronize(s->ing_parse)); HIPCHK(hipStreamSynchronize(s->ing_ext)); HIPCHK(hipStreamSynchronize(s->stream));
        }
        const auto ta0 = std::chrono::steady_clock::now();
        HIPCHK(s->g_scratch.reserve(scr, s->stream));
        HIPCHK(tokb.reserve(full * covi::INF_TOK_CAP, s->stream));
        HIPCHK(ntokb.reserve(full, s->stream));
        HIPCHK(win.reserve(win_bytes, s->stream));
        HIPCHK(s->g_seg[q].reserve(seg_cap, s->stream)); HIPCHK(s->g_recbase[q].reserve(seg_cap, s->stream)); HIPCHK(s->g_cigbase[q].reserve(seg_cap, s->stream));
        s->ing_s_alloc += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta0).count();
    }
    const u64 out_off0 = n ? s->h_blocks[b0].out_off : s->ing_infl;
    const u64 wbytes = n ? s->h_blocks[b0 + n - 1].out_off + s->h_blocks[b0 + n - 1].isize - out_off0 : 0;
    uint8_t *out_bias = win.p + K.carry - out_off0;     // blocks carry absolute offsets of the inflated stream
    if (w == 0 && s->ing_first_record > wbytes && !s->ing_fail) s->ing_fail = 32u;    // the BAM header does not end inside the first window
    if (n) {
        HIPCHK(hipStreamWaitEvent(s->stream, s->ing_fed, 0));
        // token buffers alternate: this k_inflate may not start before the k_lz_resolve that read the same buffer two rounds ago is done;
        // the window buffer is free once the extraction of the window three rounds ago is done
        if (w >= 2) HIPCHK(hipStreamWaitEvent(s->stream, s->ing_lz_done[bb], 0));
        if (w >= 3) HIPCHK(hipStreamWaitEvent(s->stream, s->ing_ext_done[w % 3u], 0));
        const u32 grid = (n + 63u) / 64u;
        const uint8_t *comp_bias = s->g_cwin[w % 3u].p - s->ing_round_start;     // blocks carry absolute file offsets
        if (K.version == 3)
            hipLaunchKernelGGL(covi::k_inflate_wave, dim3(n), dim3(64), 0, s->stream, comp_bias, (const covi::BgzfBlock *)(s->g_blocks.p + b0), n, out_bias,
                               tokb.p, ntokb.p, s->g_status.p + b0, reinterpret_cast<u32 *>(s->g_result.p + 3), 0u);
        else
            hipLaunchKernelGGL((covi::k_inflate<INF1_LB, INF1_DB, false>), dim3(grid), dim3(64), covi::inflate_smem_bytes(INF1_LB, INF1_DB), s->stream, comp_bias,
                               (const covi::BgzfBlock *)(s->g_blocks.p + b0), n, out_bias, s->g_scratch.p, tokb.p, ntokb.p,
                               s->g_status.p + b0, reinterpret_cast<u32 *>(s->g_result.p + 3));
        HIPCHK(hipEventRecord(s->ing_inf_done[bb], s->stream));
        HIPCHK(hipEventRecord(s->ing_cdone[w % 3u], s->stream));      // this round's compressed buffer may be overwritten (three rounds on)
        HIPCHK(hipStreamWaitEvent(s->ing_aux, s->ing_inf_done[bb], 0));
        if (K.lz_version == 1)
            hipLaunchKernelGGL(covi::k_lz_resolve, dim3((n + 3u) / 4u), dim3(256), 0, s->ing_aux, (const covi::BgzfBlock *)(s->g_blocks.p + b0), n, out_bias,
                               (const covi::tokpos_t *)tokb.p, (const u32 *)ntokb.p);
        else
            hipLaunchKernelGGL(covi::k_lz_stage, dim3((n + 3u) / 4u), dim3(256), 0, s->ing_aux, (const covi::BgzfBlock *)(s->g_blocks.p + b0), n, out_bias,
                               (const covi::tokpos_t *)tokb.p, (const u32 *)ntokb.p);
        // the window's bytes are final once the matches are resolved: the boundary search (parse stream) starts here, beside the
        // CRC-32 pass, whose verdict is only looked at in cov_ingest_end (the aux stream is in order, so CRC(w) is done before LZ(w + 1)
        // and with it before anything may overwrite window w)
        HIPCHK(hipEventRecord(s->ing_lz_done[bb], s->ing_aux));
        if (s->ing_check_crc) {
            if (K.version == 3)      // a wave per block, coalesced; resident workgroups stride over the window's blocks
                hipLaunchKernelGGL(covi::k_crc32_wave, dim3(std::min<u32>((n + 7u) / 8u, (u32)s->n_cus * 4u)), dim3(512), 0, s->ing_aux,
                                   (const covi::BgzfBlock *)(s->g_blocks.p + b0), n, (const uint8_t *)out_bias, s->g_status.p + b0,
                                   reinterpret_cast<u32 *>(s->g_result.p + 3), (const u32 *)s->g_crc_tab.p);
            else                     // the lane-per-block combination keeps the lane-per-block CRC
                hipLaunchKernelGGL(covi::k_crc32, dim3((n + 255u) / 256u), dim3(256), 0, s->ing_aux, (const covi::BgzfBlock *)(s->g_blocks.p + b0), n,
                                   (const uint8_t *)out_bias, s->g_status.p + b0, reinterpret_cast<u32 *>(s->g_result.p + 3));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamWaitEvent(s->ing_parse, s->ing_lz_done[bb], 0));
    }
    // ---- the window's records: tail of the previous window in front, boundaries, counts, the new tail
    covi::BamScan S{};
    S.u = win.p; S.N = K.carry + wbytes; S.p0 = s->g_result.p + 6; S.seg_bytes = 32768; S.n_seg = (u32)((S.N + S.seg_bytes - 1) / S.seg_bytes);
    S.n_ref = (int)s->n_targets; S.ref_len = s->d_tlen.p; S.final = (final && !s->ing_open_end) ? 1 : 0;
    S.key_lo = s->ing_key_lo; S.key_hi = s->ing_key_hi; S.search_first = (w == 0 && s->ing_search_first) ? 1 : 0;
    s->ing_win[q].N = S.N; s->ing_win[q].n_seg = S.n_seg; s->ing_win[q].comp_end = n ? s->h_blocks[b0 + n - 1].in_off + s->h_blocks[b0 + n - 1].in_len + 8 : s->ing_comp;
    hipStream_t ps = s->ing_parse;
    if (w >= 3) HIPCHK(hipStreamWaitEvent(ps, s->ing_ext_done[w % 3u], 0));     // the carried tail goes in front of a buffer whose records must be out
    hipLaunchKernelGGL(covi::k_carry_in, dim3(1), dim3(1024), 0, ps, win.p, K.carry, (const uint8_t *)s->g_carry.p, (const u64 *)(s->g_result.p + 5),
                       w == 0 ? s->ing_first_record : 0ull, s->g_result.p + 6);
    hipLaunchKernelGGL(covi::k_bam_find, dim3((S.n_seg + 3) / 4), dim3(256), 0, ps, S, s->g_seg[q].p);
    hipLaunchKernelGGL(covi::k_bam_hop, dim3((S.n_seg + 63) / 64), dim3(64), 0, ps, S, s->g_seg[q].p);
    hipLaunchKernelGGL(covi::k_bam_verify, dim3(1), dim3(1024), 0, ps, S, s->g_seg[q].p, s->g_recbase[q].p, s->g_cigbase[q].p, s->g_result.p + 8 + 8 * q,
                       s->g_carry.p, (u64)K.carry, s->g_result.p + 5, s->g_result.p + 7);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s->h_winres + 8 * q, s->g_result.p + 8 + 8 * q, 8 * sizeof(u64), hipMemcpyDeviceToHost, ps));
    HIPCHK(hipEventRecord(s->ing_ver_done[q], ps));
    s->ing_launched += n;
    s->ing_batch++;
    return ingest_drain(s, -1);      // whatever is verified already
}

cov_status cov_ingest_slot_wait(cov_session *s, int slot) {
    if (!s || slot < 0 || slot >= COV_INGEST_SLOTS || !s->ing_ev[slot]) return COV_ERR_INVALID_ARG;      // (also valid after cov_ingest_end: the slot's last upload)
    HIPCHK(hipSetDevice(s->cfg.device));
    HIPCHK(hipEventSynchronize(s->ing_ev[slot]));
    return COV_OK;
}

// Bytes [a, b) of the file (present in the staging piece that starts at file offset `piece_off`) go to the compressed buffer of
// round r, whose first byte is file offset `origin`.
static cov_status ingest_copy_range(cov_session *s, u32 r, u64 origin, u64 a, u64 b, const uint8_t *piece, u64 piece_off) {
    if (b <= a) return COV_OK;
    PartTimer pt{s->ing_s_part[2]};
    DevBuf<uint8_t> &cw = s->g_cwin[r % 3u];
    const size_t need = (size_t)s->ing_ccap + 2 * 65536u + 256u;
    if (cw.cap < need) {
        if (cw.cap) { HIPCHK(hipStreamSynchronize(s->ing_copy)); HIPCHK(hipStreamSynchronize(s->stream)); }      // replaced while possibly in use
        const auto ta0 = std::chrono::steady_clock::now();
        HIPCHK(cw.reserve(need, s->stream));
        s->ing_s_alloc += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta0).count();
    }
    if ((int64_t)r > s->ing_copy_cleared) {     // first bytes of this round: the k_inflate that read this buffer three rounds ago must be done
        if (r >= 3) HIPCHK(hipStreamWaitEvent(s->ing_copy, s->ing_cdone[r % 3u], 0));
        s->ing_copy_cleared = (int64_t)r;
    }
    if (a < origin || b - origin > cw.cap) { s->err = "cov_ingest_feed: internal error, bytes outside their round's buffer"; return COV_ERR_STATE; }
    const u64 len = b - a;
    HIPCHK(hipMemcpyAsync(cw.p + (a - origin), piece + (a - piece_off), len, hipMemcpyHostToDevice, s->ing_copy));
    return COV_OK;
}

// Table entries [from, ing_blocks) follow the bytes on the copy stream; ing_fed marks "everything fed so far is on the device"
// (recorded even without new entries: the bytes in front of it may complete a round).
static cov_status ingest_upload_table(cov_session *s, u64 from) {
    if (from < s->ing_blocks)
        HIPCHK(hipMemcpyAsync(s->g_blocks.p + from, s->h_blocks + from, (size_t)(s->ing_blocks - from) * sizeof(covi::BgzfBlock), hipMemcpyHostToDevice, s->ing_copy));
    HIPCHK(hipEventRecord(s->ing_fed, s->ing_copy));
    return COV_OK;
}

cov_status cov_ingest_feed(cov_session *s, int slot, const void *host_bytes, uint64_t file_offset, uint64_t n_bytes,
                           const cov_bgzf_block *blocks, uint32_t n_blocks) {
    if (!s || !s->ing_active || slot < 0 || slot >= COV_INGEST_SLOTS || (n_bytes && !host_bytes) || (n_blocks && !blocks)) return COV_ERR_INVALID_ARG;
    if (file_offset + n_bytes > s->ing_comp) { s->err = "cov_ingest_feed: bytes beyond the size given to cov_ingest_begin"; return COV_ERR_INVALID_ARG; }
    covr::Range rr("ingest: window upload (cov_ingest_feed)");
    HIPCHK(hipSetDevice(s->cfg.device));
    { const cov_status d = ingest_drain(s, -1); if (d != COV_OK) return d; }      // extraction of whatever got verified meanwhile
    const InflateKernel &K = inflate_kernel(s);
    // block table: page-locked mirror (the async upload reads it later), then the device copy
    if (s->ing_blocks + n_blocks > s->h_blocks_cap) {
        const size_t nc = std::max<size_t>(s->ing_blocks + n_blocks, s->h_blocks_cap * 2 + 4096);
        covi::BgzfBlock *nb = nullptr;
        HIPCHK(hipHostMalloc((void **)&nb, nc * sizeof(covi::BgzfBlock), hipHostMallocDefault));
        HIPCHK(hipStreamSynchronize(s->ing_copy));      // earlier uploads still read the old mirror
        if (s->h_blocks) { memcpy(nb, s->h_blocks, s->ing_blocks * sizeof(covi::BgzfBlock)); (void)hipHostFree(s->h_blocks); }
        s->h_blocks = nb; s->h_blocks_cap = nc;
    }
    if (s->ing_blocks + n_blocks > s->g_blocks.cap || s->ing_blocks + n_blocks > s->g_status.cap) {
        HIPCHK(hipStreamSynchronize(s->ing_copy)); HIPCHK(hipStreamSynchronize(s->ing_aux)); HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(s->g_blocks.reserve(s->ing_blocks + n_blocks, s->stream, s->ing_blocks));
        HIPCHK(s->g_status.reserve(s->ing_blocks + n_blocks, s->stream, s->ing_blocks));
    }
    // A round (= the blocks of one k_inflate launch, decoded into one window) closes when it holds as many blocks as the device
    // keeps resident, or when the next block might no longer fit its compressed buffer.  The decision only looks at where the
    // previous payload ended, so the bytes of a block that is still incomplete at the end of a piece already go to the buffer of
    // the round the block will join.
    if (!s->ing_fed_any) { s->ing_fed_any = true; s->ing_round_start = file_offset; s->ing_prev_end = file_offset; }     // a span starts somewhere inside the file
    const uint8_t *piece = (const uint8_t *)host_bytes;
    const u64 piece_end = file_offset + n_bytes;
    u64 cursor = file_offset, tbl_from = s->ing_blocks;
    // (Short first rounds — an eighth, a quarter, half a window, so that the device starts on 0.2 GB instead of 1.8 — were measured on one box,
    // alternating: ingest 0.726 s against 0.677 s with full rounds from the start, profiles/r04_ramp_ab_200M.log.  Removed.)
    // (Short LAST rounds — a quarter of the blocks over the file's last 450 / 900 / 1800 MB, to shorten what the device still has to do when the
    // last byte has arrived — measured in round 6: ingest 0.498 / 0.505 / 0.500 s against 0.494 s with full rounds to the end; the device is
    // only just faster than the link, so the lag behind the feed does not shrink with the rounds and their launches wait for the drains.
    // profiles/r06_tail_rounds_ab_200M.json.  Removed.)
    auto round_full = [&](u64 proxy) { return s->ing_round_n > 0 && (s->ing_round_n >= K.round_blocks || proxy + 65536u + 64u - s->ing_round_start > s->ing_ccap); };
    for (uint32_t i = 0; i < n_blocks; i++) {
        const cov_bgzf_block &b = blocks[i];
        if (b.in_off < s->ing_prev_end || b.in_off + b.in_len > piece_end || b.isize > 65536u || b.in_len > 65536u || b.out_off != s->ing_infl) {
            s->err = "cov_ingest_feed: block outside the bytes fed so far, out of order, or not contiguous in the inflated stream"; return COV_ERR_INVALID_ARG;
        }
        const u64 proxy = s->ing_prev_end;
        if (round_full(proxy)) {
            if (proxy > cursor) { const cov_status c = ingest_copy_range(s, s->ing_batch, s->ing_round_start, cursor, proxy, piece, file_offset); if (c != COV_OK) return c; cursor = proxy; }
            { const cov_status c = ingest_upload_table(s, tbl_from); if (c != COV_OK) return c; tbl_from = s->ing_blocks; }
            const cov_status lrc = launch_round(s, s->ing_round_n, false);
            if (lrc != COV_OK) return lrc;
            s->ing_round_start = proxy; s->ing_round_n = 0;
        }
        covi::BgzfBlock d; d.in_off = b.in_off; d.out_off = b.out_off; d.in_len = b.in_len; d.isize = b.isize; d.crc = b.crc; d.pad = 0;
        s->h_blocks[s->ing_blocks] = d;
        s->ing_blocks++; s->ing_round_n++;
        s->ing_prev_end = b.in_off + b.in_len; s->ing_infl = b.out_off + b.isize;
    }
    {   // what is left of the piece: the rest of the current round, and the beginning of a block that may open the next one
        const u64 proxy = s->ing_prev_end;
        if (round_full(proxy)) {
            const u64 cut = std::min(std::max(cursor, proxy), piece_end);
            cov_status c = ingest_copy_range(s, s->ing_batch, s->ing_round_start, cursor, cut, piece, file_offset);
            if (c == COV_OK) c = ingest_copy_range(s, s->ing_batch + 1u, proxy, cut, piece_end, piece, file_offset);
            if (c != COV_OK) return c;
        } else {
            const cov_status c = ingest_copy_range(s, s->ing_batch, s->ing_round_start, cursor, piece_end, piece, file_offset);
            if (c != COV_OK) return c;
        }
    }
    { const cov_status c = ingest_upload_table(s, tbl_from); if (c != COV_OK) return c; }
    HIPCHK(hipEventRecord(s->ing_ev[slot], s->ing_copy));
    return COV_OK;
}

static cov_status ingest_end_(cov_session *s, uint64_t *n_records_out);
cov_status cov_ingest_end(cov_session *s, uint64_t *n_records_out) {
    const cov_status rc = ingest_end_(s, n_records_out);
    if (s && s->ing_rec_spilled) {
        if (rc == COV_OK && n_records_out) *n_records_out += s->ing_rec_spilled;
        if (rc == COV_ERR_INGEST_FALLBACK) {     // "nothing was appended" no longer holds: part of the file is already in the spilled results
            s->err += " (after part of the file had left the bounded record store: it cannot be handed to the CPU reader on this session any more)";
            return COV_ERR_STATE;
        }
    }
    return rc;
}
static cov_status ingest_end_(cov_session *s, uint64_t *n_records_out) {
    if (!s || !s->ing_active) return COV_ERR_INVALID_ARG;
    covr::Range rr("ingest: last rounds + parse (cov_ingest_end)");
    s->ing_active = false;
    HIPCHK(hipSetDevice(s->cfg.device));
    if (n_records_out) *n_records_out = 0;
    const InflateKernel &K = inflate_kernel(s);
    { const cov_status lrc = launch_round(s, s->ing_round_n, true); if (lrc != COV_OK) return lrc; }     // the last round: whatever the open one holds
    { const cov_status d = ingest_drain(s, (int64_t)s->ing_batch); if (d != COV_OK) return d; }
    u64 glob[8];
    HIPCHK(hipStreamSynchronize(s->ing_ext));
    HIPCHK(hipStreamSynchronize(s->ing_aux));        // the CRC pass of the last window runs beside its boundary search: its failure count is read below
    HIPCHK(hipMemcpyAsync(glob, s->g_result.p, sizeof glob, hipMemcpyDeviceToHost, s->ing_parse));
    HIPCHK(hipStreamSynchronize(s->ing_parse));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (cov_timing_on())
        fprintf(stderr, "[covermhip] ingest: %llu blocks in %u windows of %u, device allocations %.3fs; host time in drain %.3fs (waiting for a verification %.3fs), in launches %.3fs (drains inside included), in upload calls %.3fs\n",
                (unsigned long long)s->ing_blocks, s->ing_batch, K.round_blocks, s->ing_s_alloc, s->ing_s_part[0], s->ing_s_part[3], s->ing_s_part[1], s->ing_s_part[2]);
    const u32 inflate_fail = (u32)(glob[3] & 0xffffffffu);
    if (inflate_fail) { s->err = "device ingest: " + std::to_string(inflate_fail) + " BGZF blocks failed to inflate or their CRC-32 (handing the file to the CPU reader)"; return COV_ERR_INGEST_FALLBACK; }
    if (s->ing_fail) {
        const u32 f = s->ing_fail;
        const bool span = s->ing_key_lo > 0 || s->ing_key_hi < 0x80000000ll || s->ing_search_first || s->ing_open_end;
        if ((f & 64u) && !span) {     // only looked at because mates were asked for: the device pair filter takes "same tid" for "same run of one reference"
            s->err = "device ingest: record keys (tid) decrease somewhere in the file; the pair filter of such a file runs on the host (handing the file to the CPU reader)";
            return COV_ERR_INGEST_FALLBACK;
        }
        if (f & 64u) {    // contig.rs:129-132: the reference stops at the first record whose tid is lower than its predecessor's
            s->err = "BAM file appears to be unsorted. Input BAM files must be sorted by reference (i.e. by samtools sort)";
            return COV_ERR_UNSORTED;
        }
        s->err = (f & 16u) ? "more than 2^32 records or CIGAR words of one reference in the record store"
               : (f & 32u) ? "device ingest: the BAM header is longer than the first window of the inflated stream (handing the file to the CPU reader)"
               : (f & 4u) ? "device ingest: a record keeps its CIGAR in CG:B,I (handing the file to the CPU reader)"
               : (f & 8u) ? "device ingest: a record larger than the carry buffer between windows (handing the file to the CPU reader)"
               : (f & 2u) ? "device ingest: truncated BAM record"
               : "device ingest: record boundaries did not verify at segment " + std::to_string(s->ing_fail_dbg[0]) + " (start " + std::to_string(s->ing_fail_dbg[1]) +
                 ", landed " + std::to_string(s->ing_fail_dbg[2]) + "; handing the file to the CPU reader)";
        return (f & 16u) ? COV_ERR_INVALID_ARG : COV_ERR_INGEST_FALLBACK;
    }
    if ((u32)(glob[3] >> 32)) { s->err = "device ingest: corrupt BAM record"; return COV_ERR_INGEST_FALLBACK; }
    if (s->ing_open_end && s->ing_tail_key != ~0ull && (long long)s->ing_tail_key < s->ing_key_hi) {
        s->err = "device ingest: the bytes fed for this span end inside one of its own records (handing the span to the CPU reader)"; return COV_ERR_INGEST_FALLBACK;
    }
    if (s->ing_open_end) {
        // an open end is only trusted when a record from beyond the span was actually seen (complete, or cut by the end of the bytes):
        // otherwise the span's last records may lie behind the bytes that were fed
        const bool seen_beyond = ((glob[7] >> 63) && (long long)(u32)glob[7] >= s->ing_key_hi) || (s->ing_tail_key != ~0ull && (long long)s->ing_tail_key >= s->ing_key_hi);
        if (!seen_beyond) { s->err = "device ingest: the bytes fed for this span end before a record of the next span (handing the span to the CPU reader)"; return COV_ERR_INGEST_FALLBACK; }
    }
    if (s->ing_rec_total) {
        const u64 Nn = s->n_records + s->ing_rec_total, Cn = s->n_cigar + s->ing_cig_total;
        const u32 end_off = (u32)Cn;
        HIPCHK(hipMemcpyAsync(s->s_coff.p + Nn, &end_off, sizeof end_off, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        if (s->want_mates && s->mates_valid == s->n_records) s->mates_valid = Nn;     // the columns cover the whole store as long as every file came through here
        s->n_records = Nn; s->n_cigar = Cn;
        s->finished = false;
    }
    if (n_records_out) *n_records_out = s->ing_rec_total;
    return COV_OK;
}

cov_status cov_ingest_want_mates(cov_session *s, int on) {
    if (!s || s->ing_active) return COV_ERR_INVALID_ARG;
    s->want_mates = on != 0;
    return COV_OK;
}

// ---------------------------------------------------------------------------------------------- reader-stage pair filter (pair_kernels.hip.h)
static_assert(sizeof(cov_pair_filter) == sizeof(covp::PairFilter) && offsetof(cov_pair_filter, min_aligned_length_pair) == offsetof(covp::PairFilter, min_aligned_length_pair) &&
              offsetof(cov_pair_filter, min_mapq) == offsetof(covp::PairFilter, min_mapq), "cov_pair_filter mirrors the device struct");

cov_status cov_pair_filter_apply(cov_session *s, const cov_pair_filter *f, uint64_t *n_selected, uint64_t *n_primary) {
    if (!s || !f) return COV_ERR_INVALID_ARG;
    covr::Range rr("pair filter (cov_pair_filter_apply)");
    if (s->adopted || s->ing_active) { s->err = "cov_pair_filter_apply: needs the session's own record store, filled by the device ingest"; return COV_ERR_STATE; }
    if (s->spill.active) { s->err = "cov_pair_filter_apply: part of the sample's records already left the bounded record store"; return COV_ERR_STATE; }
    if (s->mates_valid != s->n_records) { s->err = "cov_pair_filter_apply: the store holds records without mate columns (cov_ingest_want_mates before every ingest, no cov_push_batch)"; return COV_ERR_STATE; }
    HIPCHK(hipSetDevice(s->cfg.device));
    hipStream_t st = s->stream;
    const u32 R = (u32)s->n_records;
    if (n_selected) *n_selected = 0;
    if (n_primary) *n_primary = 0;
    s->finished = false; s->depth_all_valid = false;
    if (R == 0) return COV_OK;
    covp::PairCols C{};
    C.tid = s->s_tid.p; C.flag = s->s_flag.p; C.mapq = s->s_mapq.p; C.nm_kind = s->s_nmk.p; C.nm = s->s_nm.p; C.l_seq = s->s_lseq.p;
    C.cigar_off = s->s_coff.p; C.cigar = s->s_cig.p; C.mtid = s->s_mtid.p; C.qh1 = s->s_qh1.p; C.qh2 = s->s_qh2.p;
    covp::PairFilter F; memcpy(&F, f, sizeof F);
    // chunks of whole references (a pair never spans two), about T records each: the table of a chunk stays small enough for the
    // last-level cache, where the scattered atomics of the join are served
    u32 T = 4u << 20;
    { long long v; if (covknob::get("pair_chunk", v) && v >= 1024) T = (u32)std::min<long long>(v, 1ll << 30); }
    const u32 n_chunks = (u32)(((u64)R + T - 1) / T);
    DevBuf<u32> d_cuts, d_partner, d_bsum, d_order, d_cnt;       // d_cnt: [0] members listed by k_pair_collect, [1] table overflow, [2 + c] entries of chunk c with more than two records
    DevBuf<u64> d_w;        // [0] primaries, [1] first error, [2] selected slots, [3] selected CIGAR words
    DevBuf<covp::PairEntry> d_tab;
    struct Rel { DevBuf<u32> &a, &b, &c, &d, &e; DevBuf<u64> &w; DevBuf<covp::PairEntry> &t; ~Rel() { a.release(); b.release(); c.release(); d.release(); e.release(); w.release(); t.release(); } }
        rel{d_cuts, d_partner, d_bsum, d_order, d_cnt, d_w, d_tab};
    HIPCHK(d_cuts.reserve((size_t)n_chunks + 1, st)); HIPCHK(d_w.reserve(4, st)); HIPCHK(d_cnt.reserve((size_t)n_chunks + 2, st));
    if (d_partner.reserve(R, st) != hipSuccess) { (void)hipGetLastError(); s->err = "device pair filter: no device memory for the partner column (handing the file to the CPU reader)"; return COV_ERR_INGEST_FALLBACK; }
    const u64 w_init[4] = {0ull, ~0ull, 0ull, 0ull};
    HIPCHK(hipMemcpyAsync(d_w.p, w_init, sizeof w_init, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_cnt.p, 0, ((size_t)n_chunks + 2) * sizeof(u32), st));
    HIPCHK(hipMemsetAsync(d_partner.p, 0xff, (size_t)R * sizeof(u32), st));
    hipLaunchKernelGGL(covp::k_count_primary, dim3(std::min<u32>((R + 255u) / 256u, 4096u)), dim3(256), 0, st, (const uint16_t *)s->s_flag.p, R, d_w.p);
    hipLaunchKernelGGL(covp::k_pair_cuts, dim3((n_chunks + 1 + 255) / 256), dim3(256), 0, st, (const int32_t *)s->s_tid.p, R, T, n_chunks, d_cuts.p);
    HIPCHK(hipGetLastError());
    std::vector<u32> cuts((size_t)n_chunks + 1);
    HIPCHK(hipMemcpyAsync(cuts.data(), d_cuts.p, cuts.size() * sizeof(u32), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    u32 biggest = 0;
    for (u32 c = 0; c < n_chunks; c++) biggest = std::max(biggest, cuts[c + 1] > cuts[c] ? cuts[c + 1] - cuts[c] : 0u);
    u64 cap = 1024;
    while (cap < 2ull * biggest) cap <<= 1;
    // (the kernels carry the table size as 32 bits: a table of 2^32 entries would read as 0.)  A reference with that many records, or a
    // table that does not fit the device beside the store, goes to the host's pair filter: the store is still untouched here.
    auto too_big = [&](const char *what) {
        (void)hipGetLastError();
        s->err = std::string("device pair filter: ") + what + " (handing the file to the CPU reader)";
        return COV_ERR_INGEST_FALLBACK;
    };
    if (cap > (1ull << 31)) return too_big("more than 2^30 records of one reference");
    if (d_tab.reserve((size_t)cap, st) != hipSuccess) return too_big("the join table of the largest reference does not fit the device's memory");
    auto chunk_table = [&](u32 r0, u32 r1) -> u64 {        // this chunk's table (a prefix of the buffer) built: its size
        u64 cc = 1024;
        while (cc < 2ull * (r1 - r0)) cc <<= 1;
        (void)hipMemsetAsync(d_tab.p, 0, (size_t)cc * sizeof(covp::PairEntry), st);
        hipLaunchKernelGGL(covp::k_pair_insert, dim3((r1 - r0 + 255u) / 256u), dim3(256), 0, st, C, r0, r1, d_tab.p, (u32)(cc - 1), d_cnt.p);
        return cc;
    };
    for (u32 c = 0; c < n_chunks; c++) {
        const u32 r0 = cuts[c], r1 = cuts[c + 1];
        if (r1 <= r0) continue;
        const u64 cc = chunk_table(r0, r1);
        hipLaunchKernelGGL(covp::k_pair_resolve, dim3((u32)((cc + 255) / 256)), dim3(256), 0, st, C, (const covp::PairEntry *)d_tab.p, (u32)cc, F, d_partner.p, d_cnt.p + 2 + c,
                           d_w.p + 1, (u32)COV_ERR_NM_MISSING, (u32)COV_ERR_NM_BADTYPE);
    }
    HIPCHK(hipGetLastError());
    {   // read names that occur more than twice among the eligible records of a reference: the reference's serial sequence, replayed
        std::vector<u32> hcnt((size_t)n_chunks + 2);
        HIPCHK(hipMemcpyAsync(hcnt.data(), d_cnt.p, hcnt.size() * sizeof(u32), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (hcnt[1]) { s->err = "cov_pair_filter_apply: internal error, hash table overflow"; return COV_ERR_STATE; }
        constexpr u32 CAP = 1u << 20;
        DevBuf<covp::MultiRec> d_list; DevBuf<uint2> d_pairs;
        struct Rel2 { DevBuf<covp::MultiRec> &a; DevBuf<uint2> &b; ~Rel2() { a.release(); b.release(); } } rel2{d_list, d_pairs};
        for (u32 c = 0; c < n_chunks; c++) {
            if (!hcnt[2 + c]) continue;
            const u32 r0 = cuts[c], r1 = cuts[c + 1];
            HIPCHK(d_list.reserve(CAP, st));
            HIPCHK(hipMemsetAsync(d_cnt.p, 0, sizeof(u32), st));
            const u64 cc = chunk_table(r0, r1);
            hipLaunchKernelGGL(covp::k_pair_collect, dim3((r1 - r0 + 255u) / 256u), dim3(256), 0, st, C, r0, r1, (const covp::PairEntry *)d_tab.p, (u32)(cc - 1), d_list.p, CAP, d_cnt.p);
            HIPCHK(hipGetLastError());
            u32 n_list = 0;
            HIPCHK(hipMemcpyAsync(&n_list, d_cnt.p, sizeof n_list, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (n_list > CAP) {
                s->err = "pair filter: more than 2^20 records carry a read name that occurs more than twice among the primary proper-pair records of one reference "
                         "(handing the file to the CPU reader)";
                return COV_ERR_INGEST_FALLBACK;
            }
            std::vector<covp::MultiRec> list(n_list);
            HIPCHK(hipMemcpyAsync(list.data(), d_list.p, (size_t)n_list * sizeof(covp::MultiRec), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            std::sort(list.begin(), list.end(), [](const covp::MultiRec &x, const covp::MultiRec &y) { return x.entry != y.entry ? x.entry < y.entry : x.i < y.i; });
            std::vector<uint2> pairs;
            for (size_t a = 0; a < list.size();) {       // filter.rs:163-187 over one name's records, in file order
                size_t b = a; bool parked = false; u32 first = 0;
                for (; b < list.size() && list[b].entry == list[a].entry; b++) {
                    if (parked) { pairs.push_back(make_uint2(first, list[b].i)); parked = false; }
                    else if (list[b].parks) { parked = true; first = list[b].i; }
                }
                a = b;
            }
            if (!pairs.empty()) {
                HIPCHK(d_pairs.reserve(pairs.size(), st));
                HIPCHK(hipMemcpyAsync(d_pairs.p, pairs.data(), pairs.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(covp::k_pair_judge_list, dim3((u32)((pairs.size() + 255) / 256)), dim3(256), 0, st, C, (const uint2 *)d_pairs.p, (u32)pairs.size(), F, d_partner.p,
                                   d_w.p + 1, (u32)COV_ERR_NM_MISSING, (u32)COV_ERR_NM_BADTYPE);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(st));          // `pairs` is read by the copy until here
            }
        }
    }
    // ---- the reference's output order: pairs by their second record, first record then second
    const u32 nb = (R + covp::SCAN_BLOCK - 1) / covp::SCAN_BLOCK;
    HIPCHK(d_bsum.reserve((size_t)nb + 1, st));
    const covp::PairSlots slots{d_partner.p};
    hipLaunchKernelGGL((covp::k_scan_sums<covp::PairSlots>), dim3(nb), dim3(256), 0, st, slots, R, d_bsum.p);
    hipLaunchKernelGGL(covp::k_scan_offsets, dim3(1), dim3(1024), 0, st, d_bsum.p, nb, d_w.p + 2);
    HIPCHK(hipGetLastError());
    u64 hw[4];
    HIPCHK(hipMemcpyAsync(hw, d_w.p, sizeof hw, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (hw[1] != ~0ull) {
        const u32 code = (u32)(hw[1] & 0xffu);
        s->err = code == (u32)COV_ERR_NM_MISSING ? "Mapping record encountered that does not have an 'NM' auxiliary tag in the SAM/BAM format" : "Unexpected data type of NM aux tag";
        return (cov_status)code;
    }
    const u64 n_sel = hw[2];
    if (n_primary) *n_primary = hw[0];
    if (n_selected) *n_selected = n_sel;
    if (n_sel == 0) { s->n_records = 0; s->n_cigar = 0; s->mates_valid = 0; return COV_OK; }
    const u32 S = (u32)n_sel;
    HIPCHK(d_order.reserve(S, st));
    const covp::PairEmit emit{d_partner.p, d_order.p};
    hipLaunchKernelGGL((covp::k_scan_apply<covp::PairSlots, covp::PairEmit>), dim3(nb), dim3(256), 0, st, slots, emit, R, (const u32 *)d_bsum.p);
    // ---- the selected store: CIGAR offsets by a second scan, whose consumer moves the records
    const u32 nb2 = (S + covp::SCAN_BLOCK - 1) / covp::SCAN_BLOCK;
    HIPCHK(d_bsum.reserve((size_t)std::max(nb, nb2) + 1, st));      // (never grows: S <= R)
    const covp::SelCigarLen clen{d_order.p, s->s_coff.p};
    hipLaunchKernelGGL((covp::k_scan_sums<covp::SelCigarLen>), dim3(nb2), dim3(256), 0, st, clen, S, d_bsum.p);
    hipLaunchKernelGGL(covp::k_scan_offsets, dim3(1), dim3(1024), 0, st, d_bsum.p, nb2, d_w.p + 3);
    HIPCHK(hipGetLastError());
    u64 n_cig_sel = 0;
    HIPCHK(hipMemcpyAsync(&n_cig_sel, d_w.p + 3, sizeof n_cig_sel, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    DevBuf<int32_t> n_tid, n_pos; DevBuf<uint16_t> n_flag; DevBuf<uint8_t> n_mapq, n_nmk; DevBuf<u32> n_nm, n_lseq, n_coff, n_cig;
    struct Rel3 { DevBuf<int32_t> &a, &b; DevBuf<uint16_t> &c; DevBuf<uint8_t> &d, &e; DevBuf<u32> &f, &g, &h, &i;
                  ~Rel3() { a.release(); b.release(); c.release(); d.release(); e.release(); f.release(); g.release(); h.release(); i.release(); } }
        rel3{n_tid, n_pos, n_flag, n_mapq, n_nmk, n_nm, n_lseq, n_coff, n_cig};      // (after the swap below: the old store)
    HIPCHK(n_tid.reserve(S, st)); HIPCHK(n_pos.reserve(S, st)); HIPCHK(n_flag.reserve(S, st)); HIPCHK(n_mapq.reserve(S, st)); HIPCHK(n_nmk.reserve(S, st));
    HIPCHK(n_nm.reserve(S, st)); HIPCHK(n_lseq.reserve(S, st)); HIPCHK(n_coff.reserve((size_t)S + 1, st)); HIPCHK(n_cig.reserve((size_t)n_cig_sel + 1, st));
    covp::SelGather G{};
    G.order = d_order.p;
    G.src = covp::Store{s->s_tid.p, s->s_pos.p, s->s_flag.p, s->s_mapq.p, s->s_nmk.p, s->s_nm.p, s->s_lseq.p, s->s_coff.p, s->s_cig.p};
    G.dst = covp::Store{n_tid.p, n_pos.p, n_flag.p, n_mapq.p, n_nmk.p, n_nm.p, n_lseq.p, n_coff.p, n_cig.p};
    hipLaunchKernelGGL((covp::k_scan_apply<covp::SelCigarLen, covp::SelGather>), dim3(nb2), dim3(256), 0, st, clen, G, S, (const u32 *)d_bsum.p);
    HIPCHK(hipGetLastError());
    const u32 end_off = (u32)n_cig_sel;
    HIPCHK(hipMemcpyAsync(n_coff.p + S, &end_off, sizeof end_off, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    std::swap(s->s_tid, n_tid); std::swap(s->s_pos, n_pos); std::swap(s->s_flag, n_flag); std::swap(s->s_mapq, n_mapq); std::swap(s->s_nmk, n_nmk);
    std::swap(s->s_nm, n_nm); std::swap(s->s_lseq, n_lseq); std::swap(s->s_coff, n_coff); std::swap(s->s_cig, n_cig);
    s->n_records = S; s->n_cigar = n_cig_sel; s->mates_valid = 0;      // the mate columns still describe the unselected store: spent
    return COV_OK;
}

// Test hook: the owned record store copied back to host arrays (cov_batch of writable pointers sized by the caller:
// n_records entries, n_records + 1 cigar offsets, n_cigar words; *n_cigar receives the word count when words are not wanted).
cov_status cov_copy_records(cov_session *s, const cov_batch *host, uint64_t *n_records, uint64_t *n_cigar) {
    if (!s || s->adopted) return COV_ERR_STATE;
    if (s->spill.active) { s->err = "cov_copy_records: part of the sample's records already left the bounded record store"; return COV_ERR_STATE; }
    HIPCHK(hipSetDevice(s->cfg.device));
    if (n_records) *n_records = s->n_records;
    if (n_cigar) *n_cigar = s->n_cigar;
    if (!host) return COV_OK;
    const uint64_t n = s->n_records;
    hipStream_t st = s->stream;
    if (n) {
        HIPCHK(hipMemcpyAsync((void *)host->tid, s->s_tid.p, n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((void *)host->pos, s->s_pos.p, n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((void *)host->flag, s->s_flag.p, n * 2, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((void *)host->mapq, s->s_mapq.p, n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((void *)host->nm, s->s_nm.p, n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((void *)host->nm_kind, s->s_nmk.p, n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((void *)host->l_seq, s->s_lseq.p, n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync((void *)host->cigar_off, s->s_coff.p, (n + 1) * 4, hipMemcpyDeviceToHost, st));
        if (s->n_cigar) HIPCHK(hipMemcpyAsync((void *)host->cigar, s->s_cig.p, s->n_cigar * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return COV_OK;
}

cov_status cov_ingest_release(cov_session *s) {
    if (!s) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->ing_aux) HIPCHK(hipStreamSynchronize(s->ing_aux));
    if (s->ing_parse) HIPCHK(hipStreamSynchronize(s->ing_parse));
    if (s->ing_ext) HIPCHK(hipStreamSynchronize(s->ing_ext));
    ingest_free_buffers(s);
    return COV_OK;
}

// Test hook: the inflated stream of the last ingest (bytes [offset, offset + n) copied to `out`) — kept only when the whole file
// was a single window.
cov_status cov_ingest_copy_inflated(cov_session *s, uint64_t offset, uint64_t n, void *out) {
    if (!s || offset + n > s->ing_infl || (n && !out)) return COV_ERR_INVALID_ARG;
    if (s->ing_batch != 1) { s->err = "cov_ingest_copy_inflated: the file was parsed in several windows, the stream is gone"; return COV_ERR_STATE; }
    HIPCHK(hipSetDevice(s->cfg.device));
    HIPCHK(hipMemcpyAsync(out, s->g_win[0].p + inflate_kernel(s).carry + offset, n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return COV_OK;
}

cov_status cov_fetch_hist(cov_session *s, uint64_t *hist) {
    if (!s || !s->finished || !(s->cfg.want & COV_WANT_HIST)) return COV_ERR_STATE;
    if (s->merged_valid) {      // a finish that merged spilled contigs (bounded record store): the bins were re-based on the host
        s->hist_fetch_seen = true;
        if (s->merged_hist.empty()) return COV_OK;
        if (!hist) return COV_ERR_INVALID_ARG;
        memcpy(hist, s->merged_hist.data(), s->merged_hist.size() * 8);
        return COV_OK;
    }
    return fetch_chunk_hist(s, hist);
}

// the compact histogram of the last pass over the store
static cov_status fetch_chunk_hist(cov_session *s, uint64_t *hist) {
    if (!s || !s->finished || !(s->cfg.want & COV_WANT_HIST)) return COV_ERR_STATE;
    HIPCHK(hipSetDevice(s->cfg.device));
    const uint64_t total = s->last_chist_total;
    s->hist_fetch_seen = true;
    if (total == 0) return COV_OK;
    if (!hist) return COV_ERR_INVALID_ARG;
    if (!s->hist_compacted) {      // the finish left the bins in the arena (its estimators were evaluated on the device): lay them out and compact them now
        HIPCHK(s->d_chist.reserve(total, s->stream));
        const uint8_t *mask = s->have_mask ? (const uint8_t *)s->d_mask.p : (const uint8_t *)nullptr;
        const u32 hb = (s->n_targets + 1023u) / 1024u;
        HIPCHK(s->d_hist_top.reserve(hb, s->stream));
        hipLaunchKernelGGL((k_hist_sum<1>), dim3(hb), dim3(1024), 0, s->stream, (const DevContig *)s->d_ctg.p, s->n_targets, s->d_tlen.p, mask, (u64)s->cfg.contig_end_exclusion, s->d_hist_top.p);
        hipLaunchKernelGGL((k_hist_off<1>), dim3(hb), dim3(1024), 0, s->stream, s->d_ctg.p, s->n_targets, s->d_tlen.p, mask, (u64)s->cfg.contig_end_exclusion, (const u64 *)s->d_hist_top.p, s->d_glob.p);
        hipLaunchKernelGGL(k_hist_compact, dim3((s->n_targets + 3u) / 4u), dim3(256), 0, s->stream, s->d_ctg.p, s->n_targets, s->d_tlen.p,
                           (u64)s->cfg.contig_end_exclusion, s->d_arena.p, s->d_chist.p);
        HIPCHK(hipGetLastError());
        s->hist_prefetched = 0; s->hist_compacted = true;
    }
    const u64 have = std::min<u64>(total, s->hist_prefetched);      // arrived with cov_finish's own synchronisation
    if (have) memcpy(hist, s->h_chist, (size_t)have * 8);
    if (have < total) {
        HIPCHK(hipMemcpyAsync(hist + have, s->d_chist.p + have, (size_t)(total - have) * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return COV_OK;
}

cov_status cov_copy_depth(cov_session *s, uint32_t tid, int32_t *depth_out) {
    if (!s || !s->finished) return COV_ERR_STATE;
    if (s->spill.active) { s->err = "cov_copy_depth: part of the sample's records already left the bounded record store"; return COV_ERR_STATE; }
    if (tid >= s->n_targets || !depth_out) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    const u32 L = s->h_tlen[tid];
    if (L == 0) return COV_OK;
    HIPCHK(s->d_depth.reserve(L, s->stream));
    HIPCHK(hipMemsetAsync(s->d_depth.p, 0, (size_t)L * 4, s->stream));
    PileupArgs a = pileup_args(s);
    // depth materialisation must not disturb the accumulated statistics: run against a scratch copy
    DevBuf<DevContig> &scratch = s->d_ctg_scratch;   // kept for the next call: per-gene coverage asks for every contig
    HIPCHK(scratch.reserve(s->n_targets, s->stream));
    HIPCHK(hipMemcpyAsync(scratch.p, s->d_ctg.p, (size_t)s->n_targets * sizeof(DevContig), hipMemcpyDeviceToDevice, s->stream));
    a.ctg = scratch.p;
    a.depth_out = s->d_depth.p;
    a.tile_base = s->h_tile_first[tid];
    const u32 grid = s->h_tile_first[tid + 1] - s->h_tile_first[tid];
    launch_any_pileup<false, true>(s, a, grid);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(depth_out, s->d_depth.p, (size_t)L * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return COV_OK;
}

cov_status cov_interval_stats_compute(cov_session *s, const cov_interval *iv, uint64_t n, uint64_t excl, int want_hist,
                                      cov_interval_stats *out, uint64_t *hist_total) {
    if (!s || !s->finished) return COV_ERR_STATE;
    if (s->spill.active) { s->err = "cov_interval_stats_compute: part of the sample's records already left the bounded record store (per-gene coverage needs them all: raise COVERM_STORE_CAP_*)"; return COV_ERR_STATE; }
    if ((n && (!iv || !out))) return COV_ERR_INVALID_ARG;
    if (hist_total) *hist_total = 0;
    s->ivhist_total = 0;
    if (n == 0) return COV_OK;
    for (uint64_t i = 0; i < n; i++)
        if (iv[i].tid >= s->n_targets || iv[i].start >= iv[i].end || iv[i].end > s->h_tlen[iv[i].tid]) {
            s->err = "interval outside its target"; return COV_ERR_INVALID_ARG;
        }
    HIPCHK(hipSetDevice(s->cfg.device));
    hipStream_t st = s->stream;
    if (!s->depth_all_valid) {   // depth of every target, once per finish
        std::vector<u64> off((size_t)s->n_targets + 1, 0);
        for (u32 c = 0; c < s->n_targets; c++) off[c + 1] = off[c] + s->h_tlen[c];
        HIPCHK(s->d_depth_off.reserve(off.size(), st));
        HIPCHK(hipMemcpyAsync(s->d_depth_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, st));
        HIPCHK(s->d_depth_all.reserve(std::max<size_t>(1, off.back()), st));
        HIPCHK(hipMemsetAsync(s->d_depth_all.p, 0, (size_t)off.back() * 4, st));
        if (s->n_records && s->n_tiles) {
            PileupArgs a = pileup_args(s);
            HIPCHK(s->d_ctg_scratch.reserve(s->n_targets, st));   // the accumulated statistics must stay as they are
            HIPCHK(hipMemcpyAsync(s->d_ctg_scratch.p, s->d_ctg.p, (size_t)s->n_targets * sizeof(DevContig), hipMemcpyDeviceToDevice, st));
            a.ctg = s->d_ctg_scratch.p;
            a.depth_out = s->d_depth_all.p; a.depth_off = s->d_depth_off.p; a.tile_base = 0;
            launch_any_pileup<false, true>(s, a, s->n_tiles);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(st));   // `off` is read by the copy above
        s->depth_all_valid = true;
    }
    static_assert(sizeof(DevInterval) == sizeof(cov_interval) && sizeof(DevIntervalStats) == sizeof(cov_interval_stats), "ABI structs mirror the device ones");
    HIPCHK(s->d_iv.reserve(n, st)); HIPCHK(s->d_ivst.reserve(n, st));
    HIPCHK(hipMemcpyAsync(s->d_iv.p, iv, n * sizeof(cov_interval), hipMemcpyHostToDevice, st));
    const u32 grid = (u32)std::max<uint64_t>(1, std::min<uint64_t>((n + 3) / 4, (uint64_t)s->n_cus * 16));
    hipLaunchKernelGGL(k_interval_stats, dim3(grid), dim3(256), 0, st, s->d_depth_all.p, s->d_depth_off.p, s->d_iv.p, n, excl, s->d_ivst.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, s->d_ivst.p, n * sizeof(cov_interval_stats), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    uint64_t tot = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (want_hist) { out[i].hist_off = tot; tot += out[i].hist_len; }
        else out[i].hist_len = 0;
    }
    if (want_hist && tot) {
        HIPCHK(s->d_ivhist.reserve(tot, st));
        HIPCHK(hipMemsetAsync(s->d_ivhist.p, 0, tot * 8, st));
        HIPCHK(hipMemcpyAsync(s->d_ivst.p, out, n * sizeof(cov_interval_stats), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_interval_hist, dim3(grid), dim3(256), 0, st, s->d_depth_all.p, s->d_depth_off.p, s->d_iv.p, n, excl, s->d_ivst.p,
                           s->d_ivhist.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
    }
    s->ivhist_total = want_hist ? tot : 0;
    if (hist_total) *hist_total = s->ivhist_total;
    return COV_OK;
}

cov_status cov_fetch_interval_hist(cov_session *s, uint64_t *hist) {
    if (!s || !s->finished) return COV_ERR_STATE;
    if (s->ivhist_total == 0) return COV_OK;
    if (!hist) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    HIPCHK(hipMemcpyAsync(hist, s->d_ivhist.p, s->ivhist_total * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return COV_OK;
}

cov_status cov_kernel_ms(const cov_session *s, cov_kernel_id k, double *ms_total, uint32_t *launches) {
    if (!s || k < 0 || k >= COV_K_COUNT) return COV_ERR_INVALID_ARG;
    if (ms_total) *ms_total = s->k_ms[k];
    if (launches) *launches = s->k_launches[k];
    return COV_OK;
}

cov_status cov_last_paths(const cov_session *s, cov_path_counts *out) {
    if (!s || !out) return COV_ERR_INVALID_ARG;
    out->listed_steps = s->h_glob.n_gen; out->generic_only = s->last_gen_all ? 1u : 0u;
    out->slow_tiles = s->h_glob.n_slow; out->bucket_records = s->h_glob.n_cx;
    return COV_OK;
}

uint32_t cov_store_spills(const cov_session *s) { return s ? s->spill.count : 0u; }

cov_status cov_set_estimators(cov_session *s, const cov_estimator *est, uint32_t n_est) {
    if (!s || (n_est && !est)) return COV_ERR_INVALID_ARG;
    if (n_est > COV_EST_MAX) { s->err = "cov_set_estimators: more than COV_EST_MAX estimators"; return COV_ERR_INVALID_ARG; }
    if (s->spill.active) { s->err = "cov_set_estimators: set them before the sample's first records (part of it already left the bounded record store)"; return COV_ERR_STATE; }
    for (uint32_t k = 0; k < n_est; k++) {
        const int kind = est[k].kind;
        if (kind < 0 || kind > COV_EST_ANIR || kind == COV_EST_TPM || kind == COV_EST_PILEUP_COUNTS) {
            s->err = "cov_set_estimators: this estimator is evaluated on the host (TPM: f64 exp / ln of the host's libm; coverage histogram: prints the histogram itself)";
            return COV_ERR_INVALID_ARG;
        }
        if (kind == COV_EST_TRIMMED_MEAN && !(s->cfg.want & COV_WANT_HIST)) { s->err = "cov_set_estimators: a trimmed mean needs COV_WANT_HIST"; return COV_ERR_INVALID_ARG; }
        if (kind == COV_EST_ANIR && (!(s->cfg.want & COV_WANT_IDENTITY) || (s->cfg.want & COV_WANT_IDENTITY_NONSUPP_ONLY))) {
            s->err = "cov_set_estimators: ANIr needs COV_WANT_IDENTITY with the primary-read sum"; return COV_ERR_INVALID_ARG;
        }
    }
    s->est = EstParams{};
    for (uint32_t k = 0; k < n_est; k++) memcpy(&s->est.e[k], &est[k], sizeof(cov_estimator));
    s->est.n = n_est;
    s->est_valid = false;
    return COV_OK;
}

cov_status cov_fetch_estimates(cov_session *s, float *out) {
    if (!s || !s->finished || !s->est_valid) { if (s) s->err = "cov_fetch_estimates: cov_set_estimators, then cov_finish"; return COV_ERR_STATE; }
    if (s->have_mask) { s->err = "cov_fetch_estimates: with a target mask the entries are genomes: aggregate on the host"; return COV_ERR_STATE; }
    const size_t nf = (size_t)s->n_targets * s->est.n;
    if (nf && !out) return COV_ERR_INVALID_ARG;
    if (nf) memcpy(out, s->h_estf, nf * sizeof(float));
    return COV_OK;
}

cov_status cov_algorithmic_bytes(const cov_session *s, uint64_t *bytes) {
    if (!s || !bytes) return COV_ERR_INVALID_ARG;
    *bytes = s->algo_bytes;
    return COV_OK;
}

// ---- pooled page-locked host memory (covermhip.h: cov_host_alloc / cov_host_free / cov_host_trim)
namespace {
struct HostPool {
    std::mutex m;
    std::map<void *, size_t> live;          // handed out: ptr -> block size
    std::multimap<size_t, void *> parked;   // free blocks by size
    static size_t round_up(size_t b) { const size_t g = (size_t)16 << 20; return b < g ? g : (b + g - 1) / g * g; }
};
HostPool g_host_pool;
}  // namespace

void *cov_host_alloc(size_t bytes) {
    const size_t want = HostPool::round_up(bytes);
    {
        std::lock_guard<std::mutex> lk(g_host_pool.m);
        auto it = g_host_pool.parked.lower_bound(want);
        if (it != g_host_pool.parked.end() && it->first <= 2 * want) {
            void *p = it->second; const size_t sz = it->first;
            g_host_pool.parked.erase(it);
            g_host_pool.live[p] = sz;
            return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(g_host_pool.m);
    g_host_pool.live[p] = want;
    return p;
}

int cov_host_free(void *p) {
    if (!p) return 0;
    std::lock_guard<std::mutex> lk(g_host_pool.m);
    auto it = g_host_pool.live.find(p);
    if (it == g_host_pool.live.end()) return 0;
    g_host_pool.parked.emplace(it->second, p);
    g_host_pool.live.erase(it);
    return 1;
}

// File pages mapped by the caller (mmap of the BAM) made readable by the session's device, so that the DMA engine takes the
// compressed bytes straight from the page cache: no staging copy by the CPU (measured on the lease box, tools/ubench/io_probe:
// 57 GB/s, the link's rate, against 42 GB/s through 64 MiB staging slots filled by 16 threads).
cov_status cov_host_register(cov_session *s, void *p, size_t bytes) {
    if (!s || !p || !bytes) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); s->err = std::string("hipHostRegister: ") + hipGetErrorString(e); return COV_ERR_HIP; }
    return COV_OK;
}
cov_status cov_host_unregister(cov_session *s, void *p) {
    if (!s || !p) return COV_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(s->cfg.device));
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); s->err = std::string("hipHostUnregister: ") + hipGetErrorString(e); return COV_ERR_HIP; }
    return COV_OK;
}

// The calling thread (and every thread it creates afterwards: readers, pools) runs on the CPUs of the NUMA node the device hangs
// off, so that staging buffers, the page-cache copies into them and the DMA reads stay on one memory controller (lease box, 200 M
// reads: 0.87-0.88 s on the device's node, 0.99-1.00 s on the other one, 0.90-0.93 s unbound; profiles/r03_reader_sweep_200M.log).
// Returns the node, or -1 when nothing was changed (one node, no sysfs entry, affinity not permitted, COVERM_NUMA_BIND=0).
int cov_bind_thread_to_device_node(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    if (node < 0) return -1;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return -1;
    char list[4096] = {0};
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return -1;
    cpu_set_t want, have;
    CPU_ZERO(&want);
    for (char *p = list; *p;) {          // "0-63,128-191"
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET((int)c, &want);
        p = (*end == ',') ? end + 1 : end;
        if (*end != ',') break;
    }
    if (sched_getaffinity(0, sizeof have, &have) != 0) return -1;
    cpu_set_t both;
    CPU_AND(&both, &want, &have);
    if (CPU_COUNT(&both) == 0 || CPU_COUNT(&both) == CPU_COUNT(&have)) return -1;      // nothing to narrow
    if (sched_setaffinity(0, sizeof both, &both) != 0) return -1;
    return node;
}

void cov_host_trim(void) {
    std::multimap<size_t, void *> drop;
    { std::lock_guard<std::mutex> lk(g_host_pool.m); drop.swap(g_host_pool.parked); }
    for (auto &kv : drop) (void)hipHostFree(kv.second);
}

}  // extern "C"
