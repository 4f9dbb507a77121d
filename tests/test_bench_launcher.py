"""`python bench.py --gpus N` must BE an N-rank run (the flag was parsed and never read, so a driver that ran it the
way it runs `--gpus 1` got an N = 1 line).  The reference this replaces processes BAM files one after the other in one process
(/root/reference src/contig.rs:22); here the samples are dealt to N ranks, one per GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "COVERM_BENCH_SHARE_GPU")}
    e.update(kw)
    return e


def test_relaunch_command_shape():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.relaunch_cmd(4, ["--gpus", "4", "--steps", "3"], port=29777)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "3"]


def test_world_size_must_agree_with_gpus_flag():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "mislabelled" in p.stderr


def test_more_ranks_than_gpus_fails_loudly():
    """On this CPU-only container: no GPU at all.  On a 1-GPU box the gpu test below covers `--gpus 2` without the sharing switch."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: --gpus 2 is a valid request here")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--reads", "200000", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "GPU" in p.stderr or "MI355X" in p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_launches_two_ranks_by_itself():
    """COVERM_BENCH_SHARE_GPU=1: both ranks on device 0, gloo exchange — a functional check of the launcher and of the N > 1 code path,
    never a measurement (the line says so in config.sharding)."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--reads", "200000", "--contigs", "200", "--bp", "20000000", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=_env(COVERM_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["samples"] == 2 and out["config"]["rccl_ranks"] == 2
    assert "FUNCTIONAL CHECK" in out["config"]["sharding"]


@pytest.mark.gpu
def test_bench_gpus_2_on_one_gpu_without_the_switch_is_an_error():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--reads", "200000", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0
    assert "only 1 GPU" in p.stderr
