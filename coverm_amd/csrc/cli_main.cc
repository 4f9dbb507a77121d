// coverm-amd — `coverm contig` / `coverm genome` over --bam-files on the MI355X engine.
// The orchestrator lives in libcovermhip.so (csrc/host_cli.cpp, covh_cli_main) so that other hosts can call it too.
//
// How the process ends.  A GPU process does not end when its work does: the kernel takes the runtime's queues, ~30 GB of device mappings
// and the page-locked staging slots apart before the parent's wait() returns, 0.10-0.18 s on the lease boxes (tools/ubench/exit_probe,
// profiles/r03_exit_probe.log; tools/r06/queues_probe.py, profiles/r06_queues_probe.json) — a fifth of a 200 M-read run.  None of that
// is the caller's business once the table is written, so by default the work runs in a CHILD forked before the runtime is touched:
// when covh_cli_main has returned and stdout / stderr are flushed, the child hands its exit code to the launcher through a pipe, lets go of
// its standard streams and ends by itself; the launcher exits with that code at once.  The command returns when its output is complete;
// the driver's teardown finishes a moment later in the orphan (which the pid namespace's init reaps).  COVERM_NO_FAST_EXIT=1 keeps
// everything in one process with the ordinary exit path (profilers and sanitizers want that; bench.py reports the end-to-end time both ways).
// In a loop of runs with no pause the orphan's teardown overlaps the next run's start instead of coming back as its cost: twenty runs of a
// 2 M-read file take 6.4 s this way and 8.9 s as one process each (profiles/r06_back_to_back_runs.log).
#include <cerrno>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/coverm_host.h"

static double wall_now() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

int main(int argc, char **argv) {
    const bool timing = covh_timing_on() != 0;
    if (timing) fprintf(stderr, "[coverm-amd] wall clock at main(): %.3f\n", wall_now());
    const bool fast = getenv("COVERM_NO_FAST_EXIT") == nullptr;
    int fds[2] = {-1, -1};
    pid_t child = -1;
    const pid_t launcher = getpid();
    if (fast && pipe(fds) == 0) {
        fflush(stdout); fflush(stderr);
        child = fork();
        if (child < 0) { close(fds[0]); close(fds[1]); fds[0] = fds[1] = -1; }      // no fork: one process, as with COVERM_NO_FAST_EXIT
    }
    if (child > 0) {      // the launcher: waits for the child's verdict, not for the child
        close(fds[1]);
        int rc = 0;
        ssize_t got;
        do got = read(fds[0], &rc, sizeof rc); while (got < 0 && errno == EINTR);
        if (got == (ssize_t)sizeof rc) _exit(rc);
        int st = 0;       // the child ended without a verdict (killed, crashed): its status is this command's
        while (waitpid(child, &st, 0) < 0 && errno == EINTR) { }
        _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0));
    }
    if (child == 0) {
        close(fds[0]);
        (void)prctl(PR_SET_PDEATHSIG, SIGKILL);      // a launcher that is killed takes the work with it
        if (getppid() != launcher) _exit(1);         // (it already was, between fork and prctl)
    }
    covh_cli_set_fast_exit(fast ? 1 : 0);   // the process ends right after the table is written: skip freeing device memory piecemeal
    const int rc = covh_cli_main(argc, argv);
    if (timing) fprintf(stderr, "[coverm-amd] wall clock at exit: %.3f\n", wall_now());
    fflush(stdout); fflush(stderr);
    if (child == 0) {
        (void)prctl(PR_SET_PDEATHSIG, 0);
        ssize_t put;
        do put = write(fds[1], &rc, sizeof rc); while (put < 0 && errno == EINTR);
        close(fds[1]);
        close(0); close(1); close(2);     // whoever reads this command's output sees its end when the launcher goes, not when the teardown is done
    }
    if (fast) _exit(rc);
    return rc;
}
