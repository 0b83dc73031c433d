"""Once-per-image VGG-16 encoder (NOT the hot path) through the C ABI.

Reference: models/model_normalization.py:65-78 (137->224 legacy bilinear
resize, slim vgg_16 with num_classes=1024, is_training=False,
spatial_squeeze=False; mirror models/CNN/vgg.py:182-218).  Returns the 1024-d
embedding and the five end points conv1_2..conv5_3 as NHWC device tensors in
the layout disn_sdf_set_image expects.  The forward pass is disn_vgg_forward
(csrc/vgg.hip): exact fp32 implicit-GEMM convolutions with a fixed summation
order, so the same image gives the same features bit for bit on every run;
torch only holds the device buffers."""
import ctypes

import numpy as np
import torch

from . import _lib
from .api import DisnError, FEAT_SHAPES, _stream
from .synth import VGG_CFG


class Vgg16Encoder:
    """weights: dict conv<b>_<l>_w (OIHW, as tf_bundle.vgg_weights / synth.vgg_weights give them), conv<b>_<l>_b,
    fc6/fc7/fc8 _w (OIHW) / _b.  slot 0 = SDF encoder, 1 = camera-estimator encoder (one weight set per slot and
    context; creating another encoder on the same slot replaces the weights)."""

    _owner = {}    # (ctx handle, slot) -> id of the encoder whose weights the slot holds

    def __init__(self, ctx, weights, slot=0):
        self.ctx, self.slot = ctx, slot
        self._hwio = []
        conv_w, conv_b = [], []
        for bi, (n, _) in enumerate(VGG_CFG, 1):
            for li in range(1, n + 1):
                conv_w.append(self._as_hwio(weights["conv%d_%d_w" % (bi, li)]))
                conv_b.append(np.ascontiguousarray(weights["conv%d_%d_b" % (bi, li)], dtype=np.float32))
        fc_w = [self._as_hwio(weights["fc%d_w" % i]) for i in (6, 7, 8)]
        fc_b = [np.ascontiguousarray(weights["fc%d_b" % i], dtype=np.float32) for i in (6, 7, 8)]
        if fc_w[0].shape != (7, 7, 512, 4096) or fc_w[1].shape != (1, 1, 4096, 4096) or fc_w[2].shape != (1, 1, 4096, 1024):
            raise DisnError("Vgg16Encoder: unexpected fc shapes %s" % ([a.shape for a in fc_w],))
        self._host = (conv_w, conv_b, fc_w, fc_b)
        self._load()

    @staticmethod
    def _as_hwio(w):
        a = np.asarray(w, dtype=np.float32)
        return np.ascontiguousarray(a.transpose(2, 3, 1, 0))     # OIHW -> HWIO (the TF variable layout)

    def _load(self):
        conv_w, conv_b, fc_w, fc_b = self._host
        st = _lib.DisnVggWeights()
        fp = ctypes.POINTER(ctypes.c_float)
        for i in range(13):
            st.conv_w[i] = conv_w[i].ctypes.data_as(fp)
            st.conv_b[i] = conv_b[i].ctypes.data_as(fp)
        for i in range(3):
            st.fc_w[i] = fc_w[i].ctypes.data_as(fp)
            st.fc_b[i] = fc_b[i].ctypes.data_as(fp)
        self.ctx._check(self.ctx.lib.disn_vgg_set_weights(self.ctx._h, self.slot, ctypes.byref(st)), "disn_vgg_set_weights")
        Vgg16Encoder._owner[(self.ctx._h.value, self.slot)] = id(self)

    @torch.no_grad()
    def __call__(self, img, feats=True):
        """img: (H,W,3) float32 in [0,1] (BGR).  -> (embedding[1024], [5 x HWC device tensors])."""
        if Vgg16Encoder._owner.get((self.ctx._h.value, self.slot)) != id(self):
            self._load()                                          # another encoder used this slot in between
        dev = self.ctx.device
        x = torch.as_tensor(np.ascontiguousarray(img, dtype=np.float32)).to(dev)
        if x.dim() != 3 or x.shape[2] != 3:
            raise DisnError("Vgg16Encoder: image must be HxWx3, got %s" % (tuple(x.shape),))
        emb = torch.empty(1024, dtype=torch.float32, device=dev)
        outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in FEAT_SHAPES] if feats else []
        ptrs = (ctypes.c_void_p * 5)(*([t.data_ptr() for t in outs] if feats else [None] * 5))
        self.ctx._check(self.ctx.lib.disn_vgg_forward(self.ctx._h, self.slot, x.data_ptr(), x.shape[0], x.shape[1], ptrs,
                                                      emb.data_ptr(), _stream()), "disn_vgg_forward")
        self._keep = x                                            # read asynchronously by the kernels
        return emb, outs
