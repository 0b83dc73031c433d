"""Seeded synthetic inputs for benchmarks, smoke tests and the demo when no
checkpoint is given: random-init weights exactly as the
reference would create them (xavier-uniform weights, zero biases; reference
utils/tf_util.py:40-44,173-174) and a smooth synthetic 137x137 view."""
import math

import numpy as np

DECODER_SHAPES = [
    ("g_f1c1", 3, 64), ("g_f1c2", 64, 256), ("g_f1c3", 256, 512),
    ("g_f2c1", 1536, 512), ("g_f2c2", 512, 256), ("g_f2c5", 256, 1),
    ("l_f1c1", 3, 64), ("l_f1c2", 64, 256), ("l_f1c3", 256, 512),
    ("l_f2c1", 1984, 512), ("l_f2c2", 512, 256), ("l_f2c5", 256, 1),
]
VGG_CFG = [(2, 64), (2, 128), (3, 256), (3, 512), (3, 512)]
# fixed ground-truth camera of the demo (reference demo/demo.py:272-276)
GT_TRANS_MAT = np.array([[-68.453156, 5.5086656, -0.37556022],
                         [-17.138561, -84.685486, -0.250198],
                         [-47.284092, -3.6569588, 0.2493176],
                         [101.133705, 101.34268, 1.4305686]], dtype=np.float32)
CATEGORIES = ["02691156", "02828884", "02933112", "02958343", "03001627", "03211117", "03636649",
              "03691459", "04090263", "04256520", "04379243", "04401088", "04530566"]  # preprocessing/info.json:3-17


def _xavier(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def decoder_weights(seed=0):
    rng = np.random.default_rng(seed)
    w = {}
    for name, cin, cout in DECODER_SHAPES:
        w[name + "_w"] = _xavier(rng, (cin, cout), cin, cout)
        w[name + "_b"] = np.zeros(cout, np.float32)
    return w


def vgg_weights(seed=0, num_classes=1024):
    """slim vgg_16 variable set (mirror: reference models/CNN/vgg.py:182-218), OIHW."""
    rng = np.random.default_rng(seed + 1000)
    w = {}
    cin = 3
    for bi, (n, cout) in enumerate(VGG_CFG, 1):
        for li in range(1, n + 1):
            w["conv%d_%d_w" % (bi, li)] = _xavier(rng, (cout, cin, 3, 3), cin * 9, cout * 9)
            w["conv%d_%d_b" % (bi, li)] = np.zeros(cout, np.float32)
            cin = cout
    w["fc6_w"] = _xavier(rng, (4096, 512, 7, 7), 512 * 49, 4096 * 49)
    w["fc6_b"] = np.zeros(4096, np.float32)
    w["fc7_w"] = _xavier(rng, (4096, 4096, 1, 1), 4096, 4096)
    w["fc7_b"] = np.zeros(4096, np.float32)
    w["fc8_w"] = _xavier(rng, (num_classes, 4096, 1, 1), 4096, num_classes)
    w["fc8_b"] = np.zeros(num_classes, np.float32)
    return w


def image(seed=0):
    """Smooth sinusoid pattern, uint8 137x137x3 -> /255 float32 (the reference
    feeds img/255 in BGR order with no mean subtraction, demo/demo.py:263-264)."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.arange(137) / 137.0, np.arange(137) / 137.0, indexing="ij")
    img = np.zeros((137, 137, 3), np.float64)
    for ch in range(3):
        acc = 0
        for _ in range(3):
            f = rng.uniform(0.5, 4.0, size=2)
            ph = rng.uniform(0, 1)
            acc = acc + np.sin(2 * np.pi * (f[0] * u + f[1] * v + ph))
        img[:, :, ch] = np.clip(0.5 + 0.5 * acc / 3.0, 0, 1)
    return np.round(img * 255).astype(np.uint8).astype(np.float32) / np.float32(255.0)
