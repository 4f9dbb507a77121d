// coverm-amd — `coverm contig` / `coverm genome` over --bam-files on the MI355X engine.
// The orchestrator lives in libcovermhip.so (csrc/host_cli.cpp, covh_cli_main) so that other hosts can call it too.
#include <cstdio>
#include <unistd.h>

#include "../../include/coverm_host.h"

int main(int argc, char **argv) {
    covh_cli_set_fast_exit(1);      // the process ends right after the table is written: skip freeing device memory piecemeal
    const int rc = covh_cli_main(argc, argv);
    fflush(stdout); fflush(stderr);
    _exit(rc);
}
