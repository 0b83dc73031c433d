"""CPU checks of the network restatement (oracle/net.py).  PARITY UNPINNED for
the TensorFlow kernels (see oracle/net.py header): these tests pin the
documented semantics against independent
formulations, not against TensorFlow."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import net


def test_grid_points_layout():
    # reference demo/demo.py:297-305: x fastest, z slowest, endpoints exact
    p = net.grid_points(4)
    assert p.shape == (125, 3) and p.dtype == np.float32
    assert np.array_equal(p[0], [-1, -1, -1]) and np.array_equal(p[-1], [1, 1, 1])
    assert np.array_equal(p[1], [-0.5, -1, -1]) and np.array_equal(p[5], [-1, -0.5, -1])
    assert np.array_equal(p[25], [-1, -1, -0.5])


def test_projection_of_origin():
    # the demo matrix projects the origin to about (70.7, 70.8)
    xy = net.get_img_points(np.zeros((1, 3), np.float32), net.GT_TRANS_MAT)
    assert np.allclose(xy, [[70.695, 70.841]], atol=2e-3)
    far = net.get_img_points(np.array([[50.0, 50, -50]], np.float32), net.GT_TRANS_MAT)
    assert far.min() >= 0 and far.max() <= 136      # clamp, models/model_normalization.py:249-251


def test_resize_bilinear_legacy_rule():
    rng = np.random.default_rng(0)
    f = rng.standard_normal((14, 14, 4)).astype(np.float32)
    o = net.resize_bilinear_tf1(f, 137, 137)
    assert np.array_equal(o[0, 0], f[0, 0])                      # src = dst*scale, no half-pixel shift
    s = np.float32(14) / np.float32(137)
    src = np.float32(100) * s
    lo = int(np.floor(src)); l = src - np.float32(lo)
    want = f[0, lo, 0] + (f[0, lo + 1, 0] - f[0, lo, 0]) * l
    assert o[0, 100, 0] == np.float32(want)
    assert np.array_equal(o[136, 136], o[136, 136])
    # last rows clamp hi index to in-1: src=136*14/137=13.9 -> lo=13, hi=13
    assert np.allclose(o[136, 0], f[13, 0])
    # identity when sizes match
    assert np.array_equal(net.resize_bilinear_tf1(f, 14, 14), f)


def test_resampler_equals_grid_sample_align_corners():
    # a3 == F.grid_sample(bilinear, zeros, align_corners=True), g = 2x/(W-1) - 1
    rng = np.random.default_rng(1)
    img = rng.standard_normal((137, 137, 8)).astype(np.float32)
    xy = rng.uniform(0, 136, (500, 2)).astype(np.float32)
    xy[:4] = [[0, 0], [136, 136], [136, 0], [12, 136]]
    ours = net.resampler(img, xy)
    t = torch.from_numpy(img).permute(2, 0, 1)[None]
    g = torch.from_numpy(2 * xy / 136 - 1)[None, :, None, :]
    ref = F.grid_sample(t, g, mode="bilinear", padding_mode="zeros", align_corners=True)[0, :, :, 0].T.numpy()
    assert np.allclose(ours, ref, atol=2e-5)
    assert np.array_equal(ours[0], img[0, 0]) and np.array_equal(ours[1], img[136, 136])


def test_decoder_f32_close_to_f64_and_golden():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "net_points64.npz"))
    maps, gf = net.synthetic_features(seed=5)
    w = net.xavier_decoder_weights(seed=5, bias_scale=0.05)
    f32 = net.sdf_points(z["pts"], z["pts"], net.GT_TRANS_MAT, maps, gf, w)
    assert np.allclose(f32, z["pred_f32"], atol=1e-5)
    assert np.abs(f32 - z["pred_f64"]).max() < 1e-4


def test_sdf_grid_scaling_and_shape():
    maps, gf = net.synthetic_features(seed=2)
    w = net.xavier_decoder_weights(seed=2)
    g = net.sdf_grid(2, net.GT_TRANS_MAT, maps, gf, w)
    assert g.shape == (27,) and g.dtype == np.float32
    pts = net.grid_points(2)
    raw = net.sdf_points(pts, pts, net.GT_TRANS_MAT, maps, gf, w)
    assert np.array_equal(g, (raw.astype(np.float64) / 10.0).astype(np.float32))   # demo/demo.py:335,351


def test_vgg_shapes():
    w = net.vgg16_random_weights(seed=0)
    emb, feats = net.vgg16_forward(net.synthetic_image(0), w)
    assert emb.shape == (1024,)
    assert [f.shape for f in feats] == [(224, 224, 64), (112, 112, 128), (56, 56, 256), (28, 28, 512), (14, 14, 512)]


def test_camera_head_matches_restatement_and_geometry():
    # product module (torch, device-agnostic) vs numpy restatement; rotation is orthonormal
    import torch
    from disn_amd import camera
    w = camera.random_weights(seed=3)
    rng = np.random.default_rng(3)
    emb = rng.standard_normal(1024).astype(np.float32)
    got = camera.pred_trans_mat(torch.from_numpy(emb), w)[0].numpy()
    want = net.cam_pred_trans_mat(emb, w)
    assert got.shape == (4, 3) and np.allclose(got, want, rtol=1e-4, atol=1e-4)
    R = camera.rotation_from_ortho6d(torch.from_numpy(rng.standard_normal((5, 6)).astype(np.float32)))
    eye = torch.eye(3).expand(5, 3, 3)
    assert torch.allclose(R.transpose(1, 2) @ R, eye, atol=1e-5)
    assert torch.allclose(torch.linalg.det(R), torch.ones(5), atol=1e-5)
    # with scale=1, R=I, t=offset the origin lands at the principal point region of the 137x137 image
    K = camera.K_INTRINSIC
    rt = np.concatenate([np.eye(3, dtype=np.float32), camera.TRANSLATION_OFFSET[None]], 0)
    uvw = np.array([0, 0, 0, 1], np.float32) @ (rt @ K.T)
    assert np.allclose(uvw[:2] / uvw[2], [68.3, 68.7], atol=0.3)
