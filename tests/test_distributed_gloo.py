"""N > 1 path on CPU: world_size 2, gloo.  (The same worker runs with nccl on a multi-GPU node.)"""
import os
import subprocess
import sys

import numpy as np

from coverm_amd import distributed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tid_range_shards_cover_and_balance():
    lens = np.asarray([10, 10, 80, 5, 5, 40, 50], dtype=np.int64)
    for world in (1, 2, 3, 8):
        sh = distributed.tid_range_shards(lens, world)
        assert sh[0][0] == 0 and sh[-1][1] == len(lens)
        assert all(a[1] == b[0] for a, b in zip(sh[:-1], sh[1:]))
    assert distributed.tid_range_shards(lens, 2) == [(0, 3), (3, 7)]


def test_two_rank_gloo_sharding():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "tests", "dist_worker.py"), "gloo"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_OK world=2 backend=gloo" in p.stdout
    assert "WAIT_OK rank=1" in p.stdout        # coverm_amd.distributed.wait_for_root: rank 1 waited on the host for rank 0


import pytest


@pytest.mark.gpu
def test_two_rank_nccl_sharding_when_two_gpus_are_visible():
    """The same worker over RCCL (backend nccl), one rank per GPU.  The lease boxes of the build environment expose ONE GPU, so
    this skips there — loudly: the N > 1 RCCL path then stays unmeasured by the test-suite (the driver's 8-GPU bench runs it)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("only %d GPU visible: tests/dist_worker.py nccl needs 2 (RCCL N > 1 path NOT exercised here)" % n)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tests", "dist_worker.py"), "nccl"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_OK world=2 backend=nccl" in p.stdout
