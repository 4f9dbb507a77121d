import os, subprocess, sys
sys.path.insert(0, ".")
from coverm_amd import bam as cbam, synth
ref = synth.make_reference(90, 8_000_000, seed=51, min_len=1500, max_len=500_000)
batch = synth.make_reads(ref, 200_000, seed=52)
p = "/tmp/s.bam"
cbam.write_bam(p, ref.names, ref.lengths, batch, with_seq=2)
BIN = "coverm_amd/coverm-amd"
env = dict(os.environ, COVERM_KNOBS="stream_window_kb=512", COVERM_CLI_TIMING="1")
base = subprocess.run([BIN, "contig", "-b", p, "-t", "4", "-m", "mean", "trimmed_mean", "variance", "count", "anir"], capture_output=True, text=True, env=env)
print("base rc", base.returncode, base.stderr[-300:])
for devs in ("0,0", "0,0,0", "0,0,0", "0,0,0"):
    for extra_env in ({}, {"COVERM_NO_GPU_INGEST": "1"}):
        r = subprocess.run([BIN, "contig", "-b", p, "-t", "6", "--devices", devs, "-m", "mean", "trimmed_mean", "variance", "count", "anir"], capture_output=True, text=True, env=dict(env, **extra_env))
        same = r.stdout == base.stdout
        print(devs, extra_env, "rc", r.returncode, "same", same)
        if not same:
            print(r.stderr[-1500:])
            a, b = base.stdout.splitlines(), r.stdout.splitlines()
            bad = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
            print("first differing rows", bad[:5], "of", len(a), len(b))
            for i in bad[:3]:
                print(" base:", a[i]); print(" got :", b[i])
