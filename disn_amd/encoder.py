"""Once-per-image VGG-16 encoder (NOT the hot path; PyTorch-ROCm plumbing).

Reference: models/model_normalization.py:65-78 (137->224 legacy bilinear
resize, slim vgg_16 with num_classes=1024, is_training=False,
spatial_squeeze=False; mirror models/CNN/vgg.py:182-218).  Returns the 1024-d
embedding and the five end points conv1_2..conv5_3 as NHWC device tensors in
the layout disn_sdf_set_image expects."""
import numpy as np
import torch
import torch.nn.functional as F

from .synth import VGG_CFG


class Vgg16Encoder:
    def __init__(self, ctx, weights):
        self.ctx = ctx
        self.w = {k: torch.as_tensor(np.asarray(v, np.float32)).to(ctx.device) for k, v in weights.items()}

    @torch.no_grad()
    def __call__(self, img):
        """img: (137,137,3) float32 in [0,1].  -> (embedding[1024], [5 x HWC tensors])."""
        dev = self.ctx.device
        x = torch.as_tensor(np.asarray(img, np.float32)).to(dev)
        h, w, c = x.shape
        if (h, w) != (224, 224):
            # TF1 legacy resize through the HIP kernel (channels padded to a multiple of 4)
            pad = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
            pad[:, :, :c] = x
            x = self.ctx.resize_bilinear(pad, 224, 224)[:, :, :c]
        x = x.permute(2, 0, 1)[None].contiguous()
        feats = []
        for bi, (n, _) in enumerate(VGG_CFG, 1):
            for li in range(1, n + 1):
                x = F.relu(F.conv2d(x, self.w["conv%d_%d_w" % (bi, li)], self.w["conv%d_%d_b" % (bi, li)], padding=1))
            feats.append(x[0].permute(1, 2, 0).contiguous())
            x = F.max_pool2d(x, 2, 2)
        x = F.relu(F.conv2d(x, self.w["fc6_w"], self.w["fc6_b"]))   # 7x7 VALID
        x = F.relu(F.conv2d(x, self.w["fc7_w"], self.w["fc7_b"]))   # dropout is the identity at inference
        x = F.conv2d(x, self.w["fc8_w"], self.w["fc8_b"])
        return x.reshape(-1).contiguous(), feats
