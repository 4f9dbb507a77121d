// coverm-amd — `coverm contig` / `coverm genome` over --bam-files on the MI355X engine.
// The orchestrator lives in libcovermhip.so (csrc/host_cli.cpp, covh_cli_main) so that other hosts can call it too.
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>

#include "../../include/coverm_host.h"

static double wall_now() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

int main(int argc, char **argv) {
    const bool timing = covh_timing_on() != 0;
    if (timing) fprintf(stderr, "[coverm-amd] wall clock at main(): %.3f\n", wall_now());
    const bool fast = getenv("COVERM_NO_FAST_EXIT") == nullptr;     // profilers and sanitizers want the ordinary exit path
    covh_cli_set_fast_exit(fast ? 1 : 0);   // the process ends right after the table is written: skip freeing device memory piecemeal
    const int rc = covh_cli_main(argc, argv);
    if (timing) fprintf(stderr, "[coverm-amd] wall clock at exit: %.3f\n", wall_now());
    fflush(stdout); fflush(stderr);
    if (fast) _exit(rc);
    return rc;
}
