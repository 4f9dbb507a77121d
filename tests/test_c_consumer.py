"""A C99 program compiled against include/*.h and linked with libcovermhip.so (the way a Rust / C host would bind the ABI):
layout asserts, every declared symbol resolvable, and — on a GPU — one reference golden reproduced from C."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from tests.fixtures import load_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "coverm_amd")


def build_consumer(tmp_path):
    exe = str(tmp_path / "consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + INC, os.path.join(ROOT, "tests", "c", "consumer.c"),
                           "-L" + LIBDIR, "-lcovermhip", "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def test_c_consumer_compiles_links_and_every_declared_symbol_exists(tmp_path):
    exe = build_consumer(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "abi 1"
    # every function declared in the two headers is exported by the library
    decl = set()
    for h in ("covermhip.h", "coverm_host.h"):
        txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, h)).read(), flags=re.S)
        decl |= set(re.findall(r"\b(cov_[a-z0-9_]+|covh_[a-z0-9_]+)\s*\(", txt))
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libcovermhip.so")], capture_output=True, text=True).stdout
    exported = set(l.split()[-1] for l in syms.splitlines() if l.strip())
    types = {"covh_depth_fn"}
    missing = sorted(d for d in decl - types if d not in exported)
    assert not missing, missing


@pytest.mark.gpu
def test_c_consumer_reproduces_reference_golden(tmp_path):
    """contig.rs:493-510 (7seqs, contig_end_exclusion 75): mean 1.4117647 / 1.2435294, variance 1.3049262 / 0.6862065 — from C."""
    exe = build_consumer(tmp_path)
    b = load_fixture("7seqs.reads_for_seq1_and_seq2.bam")
    fx = str(tmp_path / "fixture.bin")
    with open(fx, "wb") as fh:
        fh.write(struct.pack("<I", len(b.ref_lens)))
        fh.write(np.asarray(b.ref_lens, np.uint64).tobytes())
        fh.write(struct.pack("<QQ", b.n_records, int(b.cigar_off[-1])))
        for a, dt in ((b.tid, np.int32), (b.pos, np.int32), (b.flag, np.uint16), (b.mapq, np.uint8), (b.nm, np.uint32), (b.nm_kind, np.uint8),
                      (b.l_seq, np.uint32), (b.cigar_off, np.uint32), (b.cigar, np.uint32)):
            fh.write(np.ascontiguousarray(a, dt).tobytes())
    out = subprocess.run([exe, fx], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split("\t") for l in out.stdout.splitlines() if "\t" in l]
    assert rows == [["2", "1.4117647", "1.3049262"], ["5", "1.2435294", "0.6862065"]], out.stdout     # genome2~seq1, genome5~seq2
    assert "primary 24" in out.stdout
