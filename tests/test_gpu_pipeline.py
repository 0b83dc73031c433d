"""End-to-end pipeline on the GPU: features -> SDF grid -> mesh, single rank vs the oracle,
and two ranks (z-slabs, gloo rehearsal on one GPU) vs the single rank."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import mc, net

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_pipeline_matches_oracle(ctx):
    from disn_amd import synth
    from disn_amd.pipeline import MeshPipeline, median_iso
    maps, gf = net.synthetic_features(seed=8)
    w = synth.decoder_weights(seed=8)
    ctx.set_weights(w)
    feats = [torch.from_numpy(m).cuda() for m in maps]
    g = torch.from_numpy(gf).cuda()
    pipe = MeshPipeline(ctx)
    pipe.run(feats, g, synth.GT_TRANS_MAT, 24, want_mesh=False)
    iso = median_iso(pipe.last_sdf)
    verts, tris = pipe.run(feats, g, synth.GT_TRANS_MAT, 24, iso)
    want = net.sdf_grid(24, synth.GT_TRANS_MAT, maps, gf, w).reshape(25, 25, 25)
    got = pipe.last_sdf.cpu().numpy()
    assert float(np.abs(got - want).max()) <= 1e-4
    ref = mc.marching_cubes(got, iso=iso)          # identical grid -> topology bit-exact, vertices equal
    assert np.array_equal(tris.cpu().numpy(), ref["tris"])
    assert np.array_equal(verts.cpu().numpy(), ref["verts"])
    assert ref["n_tri"] > 100


@pytest.mark.parametrize("world,halo", [(2, "exchange"), (3, "exchange"), (3, "recompute")])
def test_slab_sharded_pipeline_equals_single_rank(world, halo, tmp_path):
    """torchrun, `world` ranks on GPU 0, gloo backend (the box has one GPU; RCCL needs one device per
    rank).  Exercises MeshPipeline's broadcast / halo all-gather / count all-gather / send-recv / device sort."""
    out = str(tmp_path / "r")
    env = dict(os.environ, DISN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", DISN_HALO=halo)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + world), os.path.join(ROOT, "tests", "dist_pipeline_worker.py"), out]
    subprocess.run(cmd, check=True, env=env, timeout=300, cwd=ROOT)
    z = np.load(out + ".npz")
    assert bool(z["tris_equal"]) and bool(z["verts_equal"]) and int(z["n_tris"]) > 100
    assert str(z["stitch"]) == "placed"      # every rank sorted its own triangles; rank 0 only placed the pieces
    # rank 0 of a 31-plane grid: with the exchange it evaluates only its own planes, the halo comes from rank 1
    own = (30 * 1) // world
    assert int(z["planes"]) == own + 2
    assert int(z["points"]) == (own if halo == "exchange" else own + 2) * 31 * 31
