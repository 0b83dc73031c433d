"""Checkpoint reader against bundles written in the published format.
PARITY UNPINNED: no released checkpoint is in the reference tree (see disn_amd/tf_bundle.py)."""
import os

import numpy as np
import pytest

from disn_amd import synth, tf_bundle
from tf_bundle_writer import crc32c_py, write_bundle


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


def test_crc32c_known_answers():
    """CRC-32C check value of "123456789" (0xE3069283, the polynomial's published known answer), the all-zero and
    all-ones 32-byte vectors of RFC 3720 B.4, and the library against a plain-Python bytewise implementation."""
    assert tf_bundle.crc32c(b"123456789") == 0xE3069283 == crc32c_py(b"123456789")
    assert tf_bundle.crc32c(bytes(32)) == 0x8A9136AA
    assert tf_bundle.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tf_bundle.crc32c(bytes(range(32))) == 0x46DD794E
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 63, 1000, 4099):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tf_bundle.crc32c(b) == crc32c_py(b)
        assert tf_bundle.crc32c(b[n // 3:], tf_bundle.crc32c(b[:n // 3])) == crc32c_py(b)     # continuation
    assert tf_bundle.crc_mask(0) == 0xa282ead8


def test_corrupt_and_truncated_checkpoints_are_errors(tmp_path):
    """PARITY UNPINNED (no released checkpoint offline): the bundle is written by tests/tf_bundle_writer.py.  A flipped
    byte in a tensor, a flipped byte in an index block and a truncated data file must all fail loudly."""
    w, t, prefix = _decoder_bundle(tmp_path)
    data, index = prefix + ".data-00000-of-00001", prefix + ".index"
    good_d, good_i = open(data, "rb").read(), open(index, "rb").read()
    _, entries = tf_bundle.read_index(prefix)
    e = entries["sdfprediction/fold2/conv2/weights"]
    bad = bytearray(good_d)
    bad[e["offset"] + e["size"] // 2] ^= 0x10
    open(data, "wb").write(bytes(bad))
    with pytest.raises(tf_bundle.BundleError, match="checksum"):
        tf_bundle.decoder_weights(prefix)
    assert len(tf_bundle.read_bundle(prefix, ["sdfprediction/fold1/conv1/weights"])) == 1    # other tensors still load
    open(data, "wb").write(good_d[:e["offset"] + e["size"] - 4])                              # truncated inside the tensor
    with pytest.raises(tf_bundle.BundleError):
        tf_bundle.read_bundle(prefix, ["sdfprediction/fold2/conv2/weights"])
    open(data, "wb").write(good_d)
    bad = bytearray(good_i)
    bad[20] ^= 0x01                                                                           # inside the first data block
    open(index, "wb").write(bytes(bad))
    with pytest.raises(tf_bundle.BundleError, match="checksum"):
        tf_bundle.read_index(prefix)
    open(index, "wb").write(good_i[:len(good_i) // 2])
    with pytest.raises(tf_bundle.BundleError):
        tf_bundle.read_index(prefix)
    open(index, "wb").write(good_i)
    got = tf_bundle.decoder_weights(prefix)
    assert all(np.array_equal(got[k], w[k]) for k in w)
