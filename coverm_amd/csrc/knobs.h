// Sizes that tests shrink — the bounded store's caps, the ingest's window, the readers' windows — come in through ONE environment
// variable, COVERM_KNOBS="name=value,name=value" (e.g. "store_cap_records=50000,ingest_round_blocks=128"), looked up where a session, an
// ingest or a reader is set up and never in a launch path.  A name that is absent leaves the built-in size in place.  The names:
//   store_cap_records, store_cap_cigar      cov_create: the bounded record store (covermhip.hip)
//   ingest_round_blocks, ingest_carry_kb, ingest_cwin_kb      cov_ingest_begin: blocks per inflate window, carry buffer, compressed window
//   pair_chunk                              cov_pair_filter: records per table chunk
//   ingest_piece_kb                         covh_bam_ingest_device: bytes per staging piece
//   stream_window_kb, filter_window_kb      the streamed CPU reader's and `coverm-amd filter`'s compressed windows
//   finalise_threads                        the host estimators' thread count
//   printer_plain                           0: the dense printer always goes through CoverageTakerTypeIterator's merge (tests compare the two)
//   ingest_zap, ingest_prepare              how the mapped file's pages leave the page table (host_bam.cpp); 0: the ingest's streams are created in cov_ingest_begin
#pragma once
#include <cstdlib>
#include <cstring>

namespace covknob {
inline bool get(const char *name, long long &out) {
    const char *e = getenv("COVERM_KNOBS");
    if (!e) return false;
    const size_t ln = strlen(name);
    for (const char *p = e; *p;) {
        const char *end = strchr(p, ',');
        if (!end) end = p + strlen(p);
        if ((size_t)(end - p) > ln + 1 && !strncmp(p, name, ln) && p[ln] == '=') { out = atoll(p + ln + 1); return true; }
        p = *end ? end + 1 : end;
    }
    return false;
}
}  // namespace covknob
