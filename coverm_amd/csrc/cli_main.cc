// coverm-amd — `coverm contig` / `coverm genome` over --bam-files on the MI355X engine.
// The orchestrator lives in libcovermhip.so (csrc/host_cli.cpp, covh_cli_main) so that other hosts can call it too.
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

#include "../../include/coverm_host.h"

int main(int argc, char **argv) {
    const bool fast = getenv("COVERM_NO_FAST_EXIT") == nullptr;     // profilers and sanitizers want the ordinary exit path
    covh_cli_set_fast_exit(fast ? 1 : 0);   // the process ends right after the table is written: skip freeing device memory piecemeal
    const int rc = covh_cli_main(argc, argv);
    fflush(stdout); fflush(stderr);
    if (fast) _exit(rc);
    return rc;
}
