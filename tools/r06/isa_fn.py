"""Pull one kernel's instructions out of hipcc -S output and count them by class (round 6's instruction diet works from these listings).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude --cuda-device-only -S -o /tmp/isa/dev.s -x hip coverm_amd/csrc/covermhip.hip
    python tools/r06/isa_fn.py /tmp/isa/dev.s 'k_prep8sILb0ELb0ELb0' [--dump out.s]
"""
import re
import sys
from collections import Counter


def extract(path, pat):
    out, on = [], False
    for ln in open(path):
        if not on:
            if ln.startswith("_Z") and pat in ln and ln.rstrip().endswith(tuple(["EEv" + x for x in ()]) or ":") or (ln.startswith("_Z") and pat in ln and ":" in ln.split(";")[0]):
                on = True
            continue
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            break
        out.append(ln.rstrip("\n"))
    return out


def classify(op):
    if op.startswith(("v_cmp", "v_cmpx")):
        return "valu"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio")):
        return "wait/misc"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = extract(path, pat)
    ins = []
    for ln in lines:
        s = ln.split(";")[0].strip()
        if not s or s.startswith(".") and not s.endswith(":"):
            continue
        if s.endswith(":"):
            ins.append(("label", s))
            continue
        ins.append((classify(s.split()[0]), s))
    c = Counter(k for k, _ in ins)
    print("static:", dict(c))
    if "--dump" in sys.argv:
        with open(sys.argv[sys.argv.index("--dump") + 1], "w") as f:
            for k, s in ins:
                f.write(("%s\n" % s) if k == "label" else ("    %-10s %s\n" % (k, s)))


main()
