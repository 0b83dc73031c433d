"""bench.py contract: one JSON line with the fields the driver reads (tiny grid so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--res", "48", "--steps", "2", "--warmup", "1"],
                         check=True, capture_output=True, text=True, timeout=600, cwd=ROOT).stdout
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "points/s"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(d["value"] - 49 ** 3 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["mesh_matches_gpu"] is True and c["value"] > 0


def test_bench_two_ranks():
    """--gpus 2 with no launcher around it: bench.py starts two ranks itself (gloo here: both ranks share the one
    GPU of the test box; RCCL needs one GPU per rank) and rank 0 reports n_gpus=2 on one z-slab-sharded grid."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["DISN_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--res", "40", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["sdf_res"] == 40 and d["config"]["points"] == 41 ** 3
    assert d["config"]["mesh_triangles"] > 0 and d["scaling"] == "weak"


@pytest.mark.parametrize("res", [40, 12])
def test_bench_eight_ranks(res):
    """The driver's scaling run: `bench.py --gpus 8` (eight z-slabs, here rehearsed over gloo on the one GPU of the test
    box with a small --res).  sdf_res 40: five voxel layers per rank, halo planes exchanged; sdf_res 12: some ranks own
    a single plane, so every rank recomputes its halo instead."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["DISN_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--res", str(res), "--steps", "1",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["sdf_res"] == res and d["config"]["points"] == (res + 1) ** 3
    assert d["config"]["mesh_triangles"] > 0 and d["scaling"] == "weak" and d["config"]["stitch"] == "placed"
    if res == 40:
        assert d["config"]["per_gpu_points_evaluated"] == 5 * 41 * 41          # rank 0: exactly its own five planes

