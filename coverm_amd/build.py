"""Builds the native libraries of coverm_amd in-tree (so that they travel with gpurun snapshots).

    python -m coverm_amd.build            # libcovermhip.so for gfx950 (hipcc cross-compiles without a GPU)

No JIT cache, no pip install: the .so lives next to this file and is what tests and bench.py load.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcovermhip.so")
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP engine cannot be built (there is no CPU fallback)")


def sources():
    hip = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    cpp = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".cpp")]
    hdr = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hdr.append(os.path.join(HERE, "..", "include", "covermhip.h"))
    return hip, cpp, hdr


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    hip, cpp, hdr = sources()
    if not os.path.exists(CLI) or os.path.getmtime(os.path.join(CSRC, "cli_main.cc")) > os.path.getmtime(CLI):
        return True
    return any(os.path.getmtime(p) > t for p in hip + cpp + hdr)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hip, cpp, _ = sources()
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-Wno-unused-function", "-Wno-pass-failed", "-I" + os.path.join(HERE, "..", "include")]
    for f in cpp:
        cmd += ["-x", "c++", f]
    for f in hip:
        cmd += ["-x", "hip", f]
    cmd += ["-o", LIB + ".tmp", "-lz", "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    build_cli(verbose)
    return LIB


CLI = os.path.join(HERE, "coverm-amd")


def build_cli(verbose=False):
    """The standalone C++ CLI (csrc/cli_main.cc) linked against the in-tree libcovermhip.so."""
    cmd = [_hipcc(), "-O2", "-std=c++17", "-x", "c++", os.path.join(CSRC, "cli_main.cc"), "-I" + os.path.join(HERE, "..", "include"),
           "-L" + HERE, "-lcovermhip", "-Wl,-rpath,$ORIGIN", "-o", CLI + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(CLI + ".tmp", CLI)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
