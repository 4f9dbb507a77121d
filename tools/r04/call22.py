"""Fuzz parity cases of tests/test_gpu_abi_parity.py beyond the suite's seeds, called directly (no pytest, no torch import): the default
pileup (k_pileup_fast7), then the six-wave kernel; stops by the clock."""
import os, sys, time
t0 = time.time()
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
out = open(os.path.join(R, "gpurun_out", "r04_call22.log"), "w")
def say(*a):
    print("%6.1fs" % (time.time() - t0), *a, flush=True); print("%6.1fs" % (time.time() - t0), *a, file=out, flush=True)
import tests.test_gpu_abi_parity as T
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 9.0
for label, env, share in (("k_pileup_fast7 (default)", None, 0.65), ("k_pileup_fast (COVERM_FAST_WAVES=6)", "6", 1.0)):
    if env: os.environ["COVERM_FAST_WAVES"] = env
    n = bad = 0
    seed = 48
    while time.time() - t0 < budget * share:
        try:
            T.test_fuzz_small_cases(seed)
        except Exception as e:
            bad += 1; say("seed", seed, "FAILED:", str(e)[:200].replace("\n", " | "))
        n += 1; seed += 1
    say("%s: seeds 48..%d, %d cases, %d failures" % (label, seed - 1, n, bad))
