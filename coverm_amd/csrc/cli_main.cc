// coverm-amd — `coverm contig` / `coverm genome` over --bam-files on the MI355X engine.
// The orchestrator lives in libcovermhip.so (csrc/host_cli.cpp, covh_cli_main) so that other hosts can call it too.
#include "../../include/coverm_host.h"

int main(int argc, char **argv) { return covh_cli_main(argc, argv); }
