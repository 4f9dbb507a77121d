#!/bin/bash
# Host layer (BAM reader, estimators/drivers/printers, pair filter, gene driver) under AddressSanitizer + UBSan:
# a host-only build of the three C++ files with stubs for the device ABI, swapped in for the CPU test run.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cat > $T/stubs.cpp <<'EOS'
#include "covermhip.h"
extern "C" {
int cov_abi_version(void) { return COVERMHIP_ABI_VERSION; }
void *cov_host_alloc(size_t) { return nullptr; }
int cov_host_free(void *) { return 0; }
void cov_host_trim(void) {}
const char *cov_last_error(const cov_session *) { return "stub"; }
cov_status cov_interval_stats_compute(cov_session *, const cov_interval *, uint64_t, uint64_t, int, cov_interval_stats *, uint64_t *) { return COV_ERR_HIP; }
cov_status cov_fetch_interval_hist(cov_session *, uint64_t *) { return COV_ERR_HIP; }
cov_status cov_create(const cov_config *, cov_session **) { return COV_ERR_HIP; }
void cov_destroy(cov_session *) {}
cov_status cov_set_targets(cov_session *, uint32_t, const uint64_t *) { return COV_ERR_HIP; }
cov_status cov_set_target_mask(cov_session *, const uint8_t *) { return COV_ERR_HIP; }
cov_status cov_push_batch(cov_session *, const cov_batch *) { return COV_ERR_HIP; }
cov_status cov_push_batch_device(cov_session *, const cov_batch *) { return COV_ERR_HIP; }
cov_status cov_finish(cov_session *, cov_contig_stats *, cov_summary *) { return COV_ERR_HIP; }
cov_status cov_fetch_hist(cov_session *, uint64_t *) { return COV_ERR_HIP; }
cov_status cov_copy_depth(cov_session *, uint32_t, int32_t *) { return COV_ERR_HIP; }
cov_status cov_reset(cov_session *) { return COV_ERR_HIP; }
cov_status cov_kernel_ms(const cov_session *, cov_kernel_id, double *, uint32_t *) { return COV_ERR_HIP; }
cov_status cov_algorithmic_bytes(const cov_session *, uint64_t *) { return COV_ERR_HIP; }
}
EOS
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -I$R/include \
    $R/coverm_amd/csrc/host_bam.cpp $R/coverm_amd/csrc/host_coverage.cpp $R/coverm_amd/csrc/host_filter.cpp $T/stubs.cpp \
    -o $T/libcovermhip_asan.so -lz -lpthread -ldl
cp $R/coverm_amd/libcovermhip.so $T/real.so
trap 'cp $T/real.so $R/coverm_amd/libcovermhip.so' EXIT
cp $T/libcovermhip_asan.so $R/coverm_amd/libcovermhip.so
cd $R
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_bam_reader.py tests/test_host_golden.py tests/test_genes.py -q -m "not gpu" -p no:cacheprovider
