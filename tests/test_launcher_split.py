"""coverm-amd's launcher / child split (csrc/cli_main.cc): by default the work runs in a child forked before the HIP runtime is touched and
the command returns the child's exit code as soon as its output is complete; COVERM_NO_FAST_EXIT=1 keeps one process.  Either way the
caller must see the same exit code, the same standard streams (complete when the command returns) and the same files.  CPU-only: the
`filter` subcommand needs no device (tests/test_filter_subcommand.py pins its output on the reference's goldens)."""
import os
import signal
import subprocess
import time

import pytest

from oracle import bamio
from tests.binary import BIN

RAW = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raw")
SRC = os.path.join(RAW, "2seqs.bad_read.1.bam")
pytestmark = pytest.mark.skipif(not (os.path.exists(BIN) and os.path.exists(SRC)), reason="needs the built binary and the raw fixtures")
MODES = [{}, {"COVERM_NO_FAST_EXIT": "1"}]


def run(args, env):
    return subprocess.run([BIN] + args, capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))


def test_same_exit_code_streams_and_files_either_way(tmp_path):
    got = []
    for k, env in enumerate(MODES):
        out = str(tmp_path / ("out%d.bam" % k))
        ok = run(["filter", "-b", SRC, "-o", out, "--min-read-percent-identity-pair", "0.99", "--proper-pairs-only"], env)
        bad = run(["filter", "-b", str(tmp_path / "nope.bam"), "-o", out + ".x"], env)
        usage = run(["filter", "-b", SRC, SRC, "-o", out + ".y"], env)
        b = bamio.read_alignment_file(out)
        got.append((ok.returncode, ok.stdout, ok.stderr, bad.returncode, bad.stderr, usage.returncode, usage.stderr,
                    b.n_records, b.qname, b.flag.tolist(), b.pos.tolist()))
    assert got[0] == got[1]
    assert got[0][0] == 0 and got[0][3] != 0 and "Unable to find BAM file" in got[0][4] and got[0][5] != 0


def test_the_timing_lines_say_when_the_work_ended(tmp_path):
    """COVERM_CLI_TIMING prints the wall clock at main() and at the end of the work; the command itself returns right behind the second
    stamp in the default mode (the launcher does not wait for the child to be taken apart)."""
    out = str(tmp_path / "o.bam")
    t0 = time.time()
    r = run(["filter", "-b", SRC, "-o", out], {"COVERM_CLI_TIMING": "1"})
    t1 = time.time()
    assert r.returncode == 0
    st = [float(l.split()[-1]) for l in r.stderr.splitlines() if "wall clock at" in l]
    assert len(st) == 2 and t0 - 0.05 <= st[0] <= st[1] <= t1 + 0.05


def test_a_child_that_dies_is_this_commands_failure(tmp_path):
    """The launcher reports the child's fate when no exit code arrives through the pipe: a child killed by a signal makes the command fail
    with 128 + the signal, as a shell would report it."""
    big = str(tmp_path / "big.bam")
    # a FIFO as input: the child blocks opening it, which leaves time to find and kill it
    fifo = str(tmp_path / "in.bam")
    os.mkfifo(fifo)
    p = subprocess.Popen([BIN, "filter", "-b", fifo, "-o", big], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    child = None
    for _ in range(200):
        time.sleep(0.02)
        kids = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
        if kids:
            child = int(kids[0])
            break
    assert child is not None, "the launcher has no child"
    os.kill(child, signal.SIGKILL)
    rc = p.wait(timeout=30)
    p.stdout.close(); p.stderr.close()
    assert rc == 128 + signal.SIGKILL


def test_a_killed_launcher_takes_the_work_with_it(tmp_path):
    fifo = str(tmp_path / "in.bam")
    os.mkfifo(fifo)
    p = subprocess.Popen([BIN, "filter", "-b", fifo, "-o", str(tmp_path / "o.bam")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    child = None
    for _ in range(200):
        time.sleep(0.02)
        kids = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
        if kids:
            child = int(kids[0])
            break
    assert child is not None
    p.kill()
    p.wait(timeout=30)
    for _ in range(200):      # PR_SET_PDEATHSIG: the child goes too
        alive = subprocess.run(["ps", "-o", "stat=", "-p", str(child)], capture_output=True, text=True).stdout.strip()
        if not alive or alive.startswith("Z"):
            break
        time.sleep(0.02)
    assert not alive or alive.startswith("Z"), "the child outlived its killed launcher"
