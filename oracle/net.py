"""CPU restatement of the DISN two-stream SDF network path (SURVEY.md s8a rows
a0-a7, a10) -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: the reference implementation is a TensorFlow-1 graph
(models/model_normalization.py, models/sdfnet.py); TensorFlow is not installed
and cannot be installed here, the released checkpoints are not in the tree and
the reference holds no golden output for this path.  The
third-party kernels (tf.image.resize_bilinear, tf.contrib.resampler,
slim vgg_16; TensorFlow 1.10, README.md:29) are restated from their documented
semantics at the reference's call sites.  Each function cites the reference
lines it follows.  float32 throughout unless `dtype=torch.float64` is asked
for (used by tests to measure both the oracle's and the kernels' error).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

IMG_H = IMG_W = 137                      # FLAGS.img_h / img_w, reference demo/demo.py:43-44
FEAT_CH = (64, 128, 256, 512, 512)       # conv1_2 .. conv5_3
FEAT_HW = (224, 112, 56, 28, 14)
GT_TRANS_MAT = np.array([[-68.453156, 5.5086656, -0.37556022],
                         [-17.138561, -84.685486, -0.250198],
                         [-47.284092, -3.6569588, 0.2493176],
                         [101.133705, 101.34268, 1.4305686]], dtype=np.float32)  # reference demo/demo.py:272-276

DECODER_SHAPES = [  # (name, Cin, Cout) in the order of include/disn.h DisnDecoderWeights
    ("g_f1c1", 3, 64), ("g_f1c2", 64, 256), ("g_f1c3", 256, 512),
    ("g_f2c1", 1536, 512), ("g_f2c2", 512, 256), ("g_f2c5", 256, 1),
    ("l_f1c1", 3, 64), ("l_f1c2", 64, 256), ("l_f1c3", 256, 512),
    ("l_f2c1", 1984, 512), ("l_f2c2", 512, 256), ("l_f2c5", 256, 1),
]


def grid_points(res, sdf_params=(-1, -1, -1, 1, 1, 1)):
    """a0 -- reference demo/demo.py:296-305: linspace per axis, meshgrid(z,y,x,'ij'),
    xyz interleaved, x fastest, cast to float32.  Returns (P,3) float32."""
    sp = np.asarray(sdf_params)
    x_ = np.linspace(sp[0], sp[3], num=res + 1)
    y_ = np.linspace(sp[1], sp[4], num=res + 1)
    z_ = np.linspace(sp[2], sp[5], num=res + 1)
    z, y, x = np.meshgrid(z_, y_, x_, indexing="ij")
    return np.stack((x, y, z), axis=3).astype(np.float32).reshape(-1, 3)


def get_img_points(pts, trans_mat):
    """a1 -- reference models/model_normalization.py:241-251."""
    p = np.asarray(pts, np.float32)
    t = np.asarray(trans_mat, np.float32).reshape(4, 3)
    # [x y z 1] @ T, float32, accumulated in k order
    acc = p[:, 0:1] * t[0] + p[:, 1:2] * t[1]
    acc = acc + p[:, 2:3] * t[2]
    acc = acc + np.float32(1.0) * t[3]
    xy = acc[:, :2] / acc[:, 2:3]
    return np.minimum(np.float32(136.0), np.maximum(np.float32(0.0), xy)).astype(np.float32)


def resize_bilinear_tf1(feat, oh, ow):
    """a2 -- tf.image.resize_bilinear(feat,(oh,ow)), align_corners=False, TF1
    legacy kernel (no half-pixel centres); call sites reference
    models/model_normalization.py:72,171-183.  feat: (H,W,C) array."""
    f = np.asarray(feat, np.float32)
    ih, iw, _ = f.shape

    def axis(isz, osz):
        scale = np.float32(isz) / np.float32(osz)
        src = np.arange(osz, dtype=np.float32) * scale
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, isz - 1)
        lerp = (src - lo.astype(np.float32)).astype(np.float32)
        return lo, hi, lerp

    ylo, yhi, yl = axis(ih, oh)
    xlo, xhi, xl = axis(iw, ow)
    tl = f[ylo][:, xlo]
    tr = f[ylo][:, xhi]
    bl = f[yhi][:, xlo]
    br = f[yhi][:, xhi]
    xl = xl[None, :, None]
    yl = yl[:, None, None]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return (top + (bot - top) * yl).astype(np.float32)


def resampler(img, xy):
    """a3 -- tf.contrib.resampler.resampler(img[1,H,W,C], xy[1,N,2]); call sites
    reference models/model_normalization.py:172-184.  img (H,W,C), xy (N,2)."""
    im = np.asarray(img, np.float32)
    h, w, c = im.shape
    x = np.asarray(xy[:, 0], np.float32)
    y = np.asarray(xy[:, 1], np.float32)
    fx = np.floor(x)
    fy = np.floor(y)
    cx = fx + 1
    cy = fy + 1
    dx = (cx - x).astype(np.float32)[:, None]
    dy = (cy - y).astype(np.float32)[:, None]

    def px(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = im[np.clip(yy, 0, h - 1).astype(np.int64), np.clip(xx, 0, w - 1).astype(np.int64)]
        return np.where(ok[:, None], v, np.float32(0))

    one = np.float32(1)
    out = (dx * dy * px(fy, fx) + (one - dx) * (one - dy) * px(cy, cx)
           + dx * (one - dy) * px(cy, fx) + (one - dx) * dy * px(fy, cx))
    inside = (x > -1) & (y > -1) & (x < w) & (y < h)
    return np.where(inside[:, None], out, np.float32(0)).astype(np.float32)


def sample_point_features(feat_maps, xy):
    """a2+a3+a4 -- reference models/model_normalization.py:171-190: five maps
    resized to 137x137, resampled at xy, concatenated conv1..conv5 (1472 ch)."""
    return np.concatenate([resampler(resize_bilinear_tf1(f, IMG_H, IMG_W), xy) for f in feat_maps], axis=1)


def xavier_decoder_weights(seed=0, bias_scale=0.0):
    """Random-init decoder weights as the reference would create them:
    xavier-uniform weights, zero biases (reference utils/tf_util.py:40-44,173-174).
    bias_scale>0 draws non-zero biases so tests exercise the bias path."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, cin, cout in DECODER_SHAPES:
        lim = math.sqrt(6.0 / (cin + cout))
        w[name + "_w"] = ((torch.rand(cin, cout, generator=g) * 2 - 1) * lim).numpy().astype(np.float32)
        w[name + "_b"] = (torch.randn(cout, generator=g) * bias_scale).numpy().astype(np.float32)
    return w


def _conv1x1(x, w, b, relu=True):
    """tf_util.conv2d 1x1 VALID + bias (+ReLU) -- reference utils/tf_util.py:154-184."""
    y = x @ w + b
    return torch.relu(y) if relu else y


def decoder(pts_rot, global_feat, point_img_feat, w, dtype=torch.float32):
    """a5+a6+a7(sum) -- reference models/sdfnet.py:69-92 (global stream),
    models/sdfnet.py:171-190 (local stream), models/model_normalization.py:204.
    Returns raw pred_sdf (N,) (not divided by 10)."""
    t = {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in w.items()}
    p = torch.from_numpy(np.asarray(pts_rot, np.float32)).to(dtype)
    gf = torch.from_numpy(np.asarray(global_feat, np.float32)).to(dtype).reshape(1, -1)
    pf = torch.from_numpy(np.asarray(point_img_feat, np.float32)).to(dtype)
    n = p.shape[0]

    def fold1(s):
        h = _conv1x1(p, t[s + "_f1c1_w"], t[s + "_f1c1_b"])
        h = _conv1x1(h, t[s + "_f1c2_w"], t[s + "_f1c2_b"])
        return _conv1x1(h, t[s + "_f1c3_w"], t[s + "_f1c3_b"])

    def fold2(s, cat):
        h = _conv1x1(cat, t[s + "_f2c1_w"], t[s + "_f2c1_b"])
        h = _conv1x1(h, t[s + "_f2c2_w"], t[s + "_f2c2_b"])
        return _conv1x1(h, t[s + "_f2c5_w"], t[s + "_f2c5_b"], relu=False)

    g = fold2("g", torch.cat([fold1("g"), gf.expand(n, -1)], dim=1))
    l = fold2("l", torch.cat([fold1("l"), pf], dim=1))
    return (g + l).reshape(-1).to(torch.float64 if dtype == torch.float64 else torch.float32).numpy()


def sdf_points(pts, pts_rot, trans_mat, feat_maps, global_feat, w, chunk=65536, dtype=torch.float32):
    """One `sess.run(pred_sdf)` over arbitrary points (reference demo/demo.py:317-326)."""
    out = []
    maps137 = [resize_bilinear_tf1(f, IMG_H, IMG_W) for f in feat_maps]
    for s in range(0, len(pts), chunk):
        xy = get_img_points(pts[s:s + chunk], trans_mat)
        feat = np.concatenate([resampler(m, xy) for m in maps137], axis=1)
        out.append(decoder(pts_rot[s:s + chunk], global_feat, feat, w, dtype=dtype))
    return np.concatenate(out)


def sdf_grid(res, trans_mat, feat_maps, global_feat, w, sdf_params=(-1, -1, -1, 1, 1, 1), dtype=torch.float32):
    """a0-a7 end to end -- reference demo/demo.py:292-335: returns the float32
    values to_binary would pack: float32(float64(pred)/10), (res+1)^3, x fastest."""
    pts = grid_points(res, sdf_params)
    pred = sdf_points(pts, pts, trans_mat, feat_maps, global_feat, w, dtype=dtype)
    return (pred.astype(np.float64) / 10.0).astype(np.float32)


# ---- a10: VGG-16 encoder (once per image; not the hot path) -----------------
VGG_CFG = [(2, 64), (2, 128), (3, 256), (3, 512), (3, 512)]


def vgg16_random_weights(seed=0, num_classes=1024):
    """slim vgg_16 variable shapes (mirror: reference models/CNN/vgg.py:182-218),
    xavier-uniform conv weights (slim default initializer), zero biases."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    cin = 3

    def xavier(shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(*shape, generator=g) * 2 - 1) * lim

    for bi, (n, cout) in enumerate(VGG_CFG, 1):
        for li in range(1, n + 1):
            w["conv%d_%d_w" % (bi, li)] = xavier((cout, cin, 3, 3), cin * 9, cout * 9)
            w["conv%d_%d_b" % (bi, li)] = torch.zeros(cout)
            cin = cout
    w["fc6_w"] = xavier((4096, 512, 7, 7), 512 * 49, 4096 * 49)
    w["fc6_b"] = torch.zeros(4096)
    w["fc7_w"] = xavier((4096, 4096, 1, 1), 4096, 4096)
    w["fc7_b"] = torch.zeros(4096)
    w["fc8_w"] = xavier((num_classes, 4096, 1, 1), 4096, num_classes)
    w["fc8_b"] = torch.zeros(num_classes)
    return w


def vgg16_forward(img, w):
    """a10 -- reference models/model_normalization.py:65-78: img (137,137,3) float32
    in [0,1] (BGR order, no mean subtraction, demo/demo.py:263-264) -> legacy
    resize to 224 -> vgg_16 (is_training=False, spatial_squeeze=False).
    Returns (embedding[1024], [conv1_2, conv2_2, conv3_3, conv4_3, conv5_3] as HWC)."""
    x = resize_bilinear_tf1(img, 224, 224)
    x = torch.from_numpy(x).permute(2, 0, 1)[None]
    feats = []
    for bi, (n, _) in enumerate(VGG_CFG, 1):
        for li in range(1, n + 1):
            x = F.relu(F.conv2d(x, w["conv%d_%d_w" % (bi, li)], w["conv%d_%d_b" % (bi, li)], padding=1))
        feats.append(x[0].permute(1, 2, 0).contiguous().numpy())
        x = F.max_pool2d(x, 2, 2)
    x = F.relu(F.conv2d(x, w["fc6_w"], w["fc6_b"]))   # 7x7 VALID
    x = F.relu(F.conv2d(x, w["fc7_w"], w["fc7_b"]))   # dropout = identity at inference
    x = F.conv2d(x, w["fc8_w"], w["fc8_b"])
    return x.reshape(-1).numpy(), feats


def synthetic_image(seed=0):
    """SURVEY.md s8d synthetic input: smooth sinusoid pattern, uint8 137x137x3 -> /255 float32."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.arange(IMG_H) / IMG_H, np.arange(IMG_W) / IMG_W, indexing="ij")
    img = np.zeros((IMG_H, IMG_W, 3), np.float64)
    for ch in range(3):
        acc = 0
        for _ in range(3):
            f = rng.uniform(0.5, 4.0, size=2)
            ph = rng.uniform(0, 1)
            acc = acc + np.sin(2 * np.pi * (f[0] * u + f[1] * v + ph))
        img[:, :, ch] = np.clip(0.5 + 0.5 * acc / 3.0, 0, 1)
    return (np.round(img * 255).astype(np.uint8).astype(np.float32) / np.float32(255.0))


def synthetic_features(seed=0, scale=1.0):
    """Stand-in VGG end points (post-ReLU statistics: non-negative, sparse) for
    tests that do not want to pay for a VGG forward."""
    rng = np.random.default_rng(seed)
    maps = []
    for hw, c in zip(FEAT_HW, FEAT_CH):
        m = rng.standard_normal((hw, hw, c)).astype(np.float32)
        maps.append(np.maximum(m, 0) * np.float32(scale))
    g = rng.standard_normal(1024).astype(np.float32)
    return maps, g


# ---- camera estimator (once per image; reference demo/demo.py:195-258) ---------
def cam_pred_trans_mat(embedding, w):
    """numpy restatement of cam_est/model_cam_old.py:82-106 + models/posenet_old.py:104-142.
    embedding [1024], w: dict '<head>/fc<i>/{weights,biases}'.  Returns pred_trans_mat (4,3)."""
    e = np.asarray(embedding, np.float32).reshape(1, 1024)

    def head(name):
        h = e
        for li in (1, 2, 3):
            h = h @ w["%s/fc%d/weights" % (name, li)] + w["%s/fc%d/biases" % (name, li)]
            if li < 3:
                h = np.maximum(h, 0)
        return h[0]

    def norm(v):
        return v / max(np.sqrt((v * v).sum()), 1e-8)

    scale = head("scale")[0]
    p = head("ortho6d")
    x = norm(p[0:3])
    z = norm(np.cross(x, p[3:6]))
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1)                               # columns x, y, z
    t = head("translation") + np.array([-0.00193892, 0.00169222, 1.3949631], np.float32)
    RT = np.concatenate([(scale * np.eye(3, dtype=np.float32)) @ R, t[None, :]], axis=0)
    K = np.array([[149.84375, 0., 68.5], [0., 149.84375, 68.5], [0., 0., 1.]], dtype=np.float32)
    return (RT @ K.T).astype(np.float32)
