"""Runs the GPU fuzz parity case of tests/test_gpu_abi_parity.py over many more seeds (not part of the suite: ~13 s for 400)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, importlib.util
try:
    import torch; torch.cuda.is_available() and torch.cuda.init()
except Exception: pass
spec = importlib.util.spec_from_file_location("tp", os.path.join(sys.path[0], "tests/test_gpu_abi_parity.py"))
tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
bad = 0
for seed in range(48, 448):
    try:
        tp.test_fuzz_small_cases(seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300].replace("\n", " | "))
print("done, failures:", bad)
