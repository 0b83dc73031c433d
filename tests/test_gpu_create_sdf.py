"""Batch driver on the GPU: files land where the reference puts them, each object's own sdf_params box
is honoured, and a view's mesh equals the oracle's mesh of the oracle-checked grid."""
import os

import numpy as np
import pytest
import torch

from oracle import mc, net

pytestmark = pytest.mark.gpu


def _read_obj(path):
    v, f = [], []
    for l in open(path):
        if l.startswith("v "):
            v.append([float(x) for x in l.split()[1:]])
        elif l.startswith("f "):
            f.append([int(x) - 1 for x in l.split()[1:]])
    return np.array(v), np.array(f, dtype=np.int64)


def test_batch_cli_layout_and_boxes(tmp_path):
    from disn_amd import create_sdf
    log_dir = str(tmp_path / "exp")
    assert create_sdf.main(["--img_feat_twostream", "--sdf_res", "24", "--synthetic", "3", "--log_dir", log_dir,
                            "--store_dist", "--random_init"]) == 0
    with pytest.raises(SystemExit) as e:      # no checkpoint and no --random_init: loud failure, nothing written
        create_sdf.main(["--img_feat_twostream", "--sdf_res", "8", "--synthetic", "1", "--log_dir", str(tmp_path / "x")])
    assert "no checkpoint" in str(e.value.code) and not os.path.exists(str(tmp_path / "x"))
    views = list(create_sdf.synthetic_views(3))
    root = create_sdf.result_obj_path(log_dir, 24, 0.0, False)
    for view in views:
        path = create_sdf.obj_file_name(root, view["cat_id"], view["obj_nm"], view["view_id"])
        assert os.path.exists(path), path
        v, f = _read_obj(path)
        assert len(f) > 50
        lo, hi = view["sdf_params"][:3], view["sdf_params"][3:]
        assert bool((v >= lo - 1e-5).all()) and bool((v <= hi + 1e-5).all())
    # the non-unit box is really used: its mesh touches neither +-1 nor stays inside a smaller default
    v2, _ = _read_obj(create_sdf.obj_file_name(root, views[2]["cat_id"], views[2]["obj_nm"], views[2]["view_id"]))
    assert float(np.abs(v2).max()) < 1.0
    # --store_dist keeps the .dist with the object's bbox in its header (reference test/create_sdf.py:292-303)
    from disn_amd import api
    dist = create_sdf.obj_file_name(root, views[2]["cat_id"], views[2]["obj_nm"], views[2]["view_id"])[:-4] + ".dist"
    res, bbox, vals = api.read_dist(dist)
    assert res == 24 and np.allclose(bbox, views[2]["sdf_params"]) and vals.size == 25 ** 3


def test_batch_mesher_matches_oracle_on_a_non_unit_box(ctx, tmp_path):
    """same features + weights through oracle/net.py (grid over the object's box) and oracle/mc_oracle.c"""
    from disn_amd import create_sdf, synth
    from disn_amd.encoder import Vgg16Encoder
    res = 16
    wdec = synth.decoder_weights(seed=3)
    ctx.set_weights(wdec)
    enc = Vgg16Encoder(ctx, synth.vgg_weights(seed=3))
    view = list(create_sdf.synthetic_views(3))[2]
    gfeat, feats = enc(view["img"])
    params = tuple(float(x) for x in view["sdf_params"])
    ref = net.sdf_grid(res, view["trans_mat"], [f.cpu().numpy() for f in feats], gfeat.cpu().numpy(), wdec, sdf_params=params)
    iso = float(np.median(ref))
    m = create_sdf.BatchMesher(ctx, enc, res, iso, str(tmp_path), store_dist=True)
    path = m.submit(view)
    m.finish()
    from disn_amd import api
    _, bbox, vals = api.read_dist(path[:-4] + ".dist")
    assert np.allclose(bbox, params)
    assert float(np.abs(vals.reshape(-1) - ref.reshape(-1)).max()) < 1e-4          # north_star tolerance, after /10
    # topology: the oracle's MC of the GPU's own grid (identical grid -> bit-exact faces)
    want = mc.marching_cubes(vals.reshape(res + 1, res + 1, res + 1), iso=iso, bbox=params)
    v, f = _read_obj(path)
    assert np.array_equal(f, want["tris"])
    assert float(np.abs(v - want["verts"]).max()) < 1e-5 * max(1.0, float(np.abs(want["verts"]).max()))


def test_two_lanes_write_the_same_files_as_one(ctx, tmp_path):
    """BatchMesher with a second context / HIP stream / worker thread on the same GPU (views in flight overlap):
    byte-identical OBJ files and the stats in submission order; a failing view surfaces at finish()."""
    from disn_amd import create_sdf, synth
    from disn_amd.encoder import Vgg16Encoder
    wdec, wvgg = synth.decoder_weights(seed=5), synth.vgg_weights(seed=5)
    ctx.set_weights(wdec)
    enc = Vgg16Encoder(ctx, wvgg)
    views = list(create_sdf.synthetic_views(7))
    one = create_sdf.BatchMesher(ctx, enc, 40, 0.0, str(tmp_path / "one"), median_iso=True)
    paths1 = [one.submit(v) for v in views]
    stats1 = one.finish()
    lane2 = create_sdf.second_lane(ctx, wdec, wvgg)
    two = create_sdf.BatchMesher(ctx, enc, 40, 0.0, str(tmp_path / "two"), median_iso=True, extra_lanes=[lane2])
    paths2 = [two.submit(v) for v in views[:4]]
    two.drain()                                    # usable again after a drain
    paths2 += [two.submit(v) for v in views[4:]]
    stats2 = two.finish()
    assert [s[0] for s in stats2] == paths2 and len(stats2) == 7
    for a, b, s1, s2 in zip(paths1, paths2, stats1, stats2):
        assert os.path.basename(a) == os.path.basename(b) and s1[1:3] == s2[1:3] and s1[2] > 500
        assert open(a, "rb").read() == open(b, "rb").read()
    bad = create_sdf.BatchMesher(ctx, enc, 40, 0.0, str(tmp_path / "bad"), extra_lanes=[lane2])
    broken = dict(views[0], img=np.zeros((137, 137), np.float32))     # not HxWx3
    bad.submit(broken)
    with pytest.raises(Exception):
        bad.finish()
    lane2[0].close()
