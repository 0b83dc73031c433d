"""Checkpoint reader against bundles written in the published format.
PARITY UNPINNED: no released checkpoint is in the reference tree (see disn_amd/tf_bundle.py)."""
import os

import numpy as np
import pytest

from disn_amd import synth, tf_bundle
from tf_bundle_writer import write_bundle


def _decoder_bundle(tmp_path, seed=0):
    w = synth.decoder_weights(seed)
    t = {"global_step": np.array(1234, np.int64), "lr": np.float32(1e-4).reshape(())}
    for k, a in w.items():
        s, l, kind = k.split("_")
        name = "%s/%s/%s" % (tf_bundle._SCOPE[s], tf_bundle._LAYER[l], "weights" if kind == "w" else "biases")
        t[name] = a.reshape((1, 1) + a.shape) if kind == "w" else a
    prefix = str(tmp_path / "model.ckpt")
    write_bundle(prefix, t)
    open(str(tmp_path / "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt"\nall_model_checkpoint_paths: "model.ckpt"\n')
    return w, t, prefix


def test_roundtrip_and_name_map(tmp_path):
    w, t, prefix = _decoder_bundle(tmp_path)
    shards, entries = tf_bundle.read_index(prefix)
    assert shards == 1 and set(entries) == set(t)
    assert entries["sdfprediction_imgfeat/fold2/conv1/weights"]["shape"] == (1, 1, 1984, 512)
    back = tf_bundle.read_bundle(prefix)
    assert all(np.array_equal(back[k], t[k]) and back[k].dtype == t[k].dtype for k in t)
    assert tf_bundle.latest_checkpoint(str(tmp_path)) == prefix
    got = tf_bundle.decoder_weights(prefix)
    assert set(got) == set(w) and all(np.array_equal(got[k], w[k]) for k in w)
    assert len(tf_bundle.read_bundle(prefix, r"fold1/conv1/")) == 4


def test_vgg_and_camera_maps(tmp_path):
    from disn_amd import camera
    vw, cw = synth.vgg_weights(1), camera.random_weights(1)
    t = {"cameraprediction/" + k: v for k, v in cw.items()}
    for k, a in vw.items():
        base, kind = k[:-2], k[-1]
        scope = "vgg_16/%s" % base if base.startswith("fc") else "vgg_16/conv%s/%s" % (base[4], base)
        t[scope + ("/weights" if kind == "w" else "/biases")] = a.transpose(2, 3, 1, 0) if kind == "w" else a
    prefix = str(tmp_path / "cam.ckpt")
    write_bundle(prefix, t)
    got = tf_bundle.vgg_weights(prefix)
    assert all(np.array_equal(got[k], vw[k]) for k in vw)
    gc = tf_bundle.camera_weights(prefix)
    assert all(np.array_equal(gc[k], cw[k]) for k in cw)


def test_errors(tmp_path):
    with pytest.raises(tf_bundle.BundleError):
        tf_bundle.read_index(str(tmp_path / "missing"))
    p = str(tmp_path / "junk")
    open(p + ".index", "wb").write(b"\x00" * 100)
    with pytest.raises(tf_bundle.BundleError, match="magic"):
        tf_bundle.read_index(p)
    _, _, prefix = _decoder_bundle(tmp_path)
    with pytest.raises(tf_bundle.BundleError, match="no tensor"):
        tf_bundle.read_bundle(prefix, ["nope"])
    os.truncate(prefix + ".data-00000-of-00001", 1000)
    with pytest.raises(tf_bundle.BundleError):
        tf_bundle.read_bundle(prefix)
    assert tf_bundle.latest_checkpoint(str(tmp_path / "nowhere")) is None
