"""Camera estimator forward (once per image; NOT the hot path; PyTorch plumbing).

Reference: demo/demo.py:195-258 (cam_evl) -> cam_est/model_cam_old.py:47-109 ->
models/posenet_old.py:104-142 (get_cam_mat): a second VGG-16 embedding -> three FC
heads (scale 1024-64-32-1, ortho6d 1024-512-256-6, translation 1024-128-64-3 plus a
constant offset) -> Gram-Schmidt rotation (posenet_old.py:25-39) ->
pred_RT = [scale*R ; t] (4x3) -> pred_trans_mat = pred_RT @ K^T with the fixed
intrinsics of cam_est/model_cam_old.py:28.
Variable names follow the checkpoint: cameraprediction/{scale,ortho6d,translation}/fc{1,2,3}/{weights,biases}.
"""
import math

import numpy as np
import torch

K_INTRINSIC = np.array([[149.84375, 0., 68.5], [0., 149.84375, 68.5], [0., 0., 1.]], dtype=np.float32)
TRANSLATION_OFFSET = np.array([-0.00193892, 0.00169222, 1.3949631], dtype=np.float32)   # posenet_old.py:136
HEADS = {"scale": (1024, 64, 32, 1), "ortho6d": (1024, 512, 256, 6), "translation": (1024, 128, 64, 3)}


def random_weights(seed=0):
    """xavier-uniform FC weights, zero biases (utils/tf_util.py:328-364); translation/fc3 is a
    truncated normal with stddev 0.05 (posenet_old.py:127-132)."""
    rng = np.random.default_rng(seed + 2000)
    w = {}
    for head, dims in HEADS.items():
        for li in range(3):
            cin, cout = dims[li], dims[li + 1]
            name = "%s/fc%d" % (head, li + 1)
            if head == "translation" and li == 2:
                v = np.clip(rng.standard_normal((cin, cout)), -2, 2) * 0.05
            else:
                lim = math.sqrt(6.0 / (cin + cout))
                v = rng.uniform(-lim, lim, size=(cin, cout))
            w[name + "/weights"] = v.astype(np.float32)
            w[name + "/biases"] = np.zeros(cout, np.float32)
    return w


def _normalize(v):
    mag = torch.sqrt(torch.sum(v * v, dim=1, keepdim=True))
    return v / torch.clamp(mag, min=1e-8)


def rotation_from_ortho6d(p):
    """posenet_old.py:25-39; columns are x, y, z."""
    x = _normalize(p[:, 0:3])
    z = _normalize(torch.linalg.cross(x, p[:, 3:6]))
    y = torch.linalg.cross(z, x)
    return torch.stack((x, y, z), dim=2)


def pred_trans_mat(embedding, weights):
    """embedding: [B,1024] tensor (any device) -> pred_trans_mat [B,4,3]."""
    e = embedding.reshape(-1, 1024).float()
    w = {k: torch.as_tensor(v, dtype=torch.float32, device=e.device) for k, v in weights.items()}

    def head(name):
        h = e
        for li in (1, 2, 3):
            h = h @ w["%s/fc%d/weights" % (name, li)] + w["%s/fc%d/biases" % (name, li)]
            if li < 3:
                h = torch.relu(h)
        return h

    scale = head("scale").reshape(-1, 1, 1)
    rot = rotation_from_ortho6d(head("ortho6d"))
    trans = head("translation") + torch.as_tensor(TRANSLATION_OFFSET, device=e.device)
    eye = torch.eye(3, device=e.device).expand(e.shape[0], 3, 3)
    rot_s = torch.matmul(scale * eye, rot)                       # pred_scale @ R   (posenet_old.py:110,140)
    rt = torch.cat([rot_s, trans.reshape(-1, 1, 3)], dim=1)      # [B,4,3]
    kt = torch.as_tensor(K_INTRINSIC.T.copy(), device=e.device)
    return torch.matmul(rt, kt)                                  # cam_est/model_cam_old.py:102-103
