"""`python bench.py --gpus N` must start its own N ranks when no launcher did (the driver's scaling run invokes it
exactly so).  Without a GPU the ranks cannot compute, so this checks the launch itself: two children come up with
WORLD_SIZE=2, fail loudly for want of a GPU, and the parent exits non-zero instead of reporting n_gpus=1.
(The computing version of this test is tests/test_gpu_bench.py::test_bench_two_ranks.)"""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="launch-failure path needs a box without a GPU")
def test_gpus_flag_spawns_ranks_and_fails_loudly_without_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--res", "24", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no fake single-rank result line
    assert "2-rank launch failed" in r.stderr
    assert "local_rank: 1" in r.stderr                                        # a second rank really existed


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
