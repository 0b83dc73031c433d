"""Writes the small golden fixtures under tests/golden/ from the CPU oracle.
The reference holds no golden for this path; these freeze the
oracle at the state that reproduces the reference tool's known answers
(tests/test_mc_oracle.py) so later edits cannot drift silently."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import fields  # noqa: E402
from oracle import mc, net  # noqa: E402

g = fields.noise_field((13, 13, 13), seed=12)
bbox = np.array([-1, -0.5, -2, 1, 0.5, 2.0])
r = mc.marching_cubes(g, iso=0.05, bbox=tuple(bbox))
np.savez_compressed(os.path.join(HERE, "mc_noise12.npz"), grid=g, iso=np.float32(0.05), bbox=bbox,
                    verts=r["verts"], tris=r["tris"], keys=r["keys"], n_centre=r["n_centre"])
print("mc_noise12: v", len(r["verts"]), "t", r["n_tri"], "centres", r["n_centre"])

# network: tiny decoder case (64 points) with its float64 evaluation
maps, gf = net.synthetic_features(seed=5)
w = net.xavier_decoder_weights(seed=5, bias_scale=0.05)
rng = np.random.default_rng(5)
pts = rng.uniform(-1, 1, (64, 3)).astype(np.float32)
f32 = net.sdf_points(pts, pts, net.GT_TRANS_MAT, maps, gf, w)
import torch  # noqa: E402
f64 = net.sdf_points(pts, pts, net.GT_TRANS_MAT, maps, gf, w, dtype=torch.float64)
np.savez_compressed(os.path.join(HERE, "net_points64.npz"), pts=pts, pred_f32=f32, pred_f64=f64)
print("net_points64: max |f32-f64|", np.abs(f32 - f64).max())

# ref_narrowband_res64.npz: the reference's own shipped data file isosurface/model_normalized.obj.n.dist
# (narrow-band .dist, res 64, 69 149 cells) re-packed as arrays, with the mesh size the reference tool
# reports for it with `-n 1` ("#v 3800 #t 7604").  Data fixture, not source.
REF = "/root/reference/isosurface/model_normalized.obj.n.dist"
if os.path.exists(REF):
    import struct
    b = open(REF, "rb").read()
    n = struct.unpack_from("<i", b, 60)[0]
    rec = np.frombuffer(b, dtype=np.dtype([("i", "<i4"), ("j", "<i4"), ("k", "<i4"), ("d", "<f4")]), count=n, offset=64)
    np.savez_compressed(os.path.join(HERE, "ref_narrowband_res64.npz"),
                        header=np.array(struct.unpack_from("<3i", b, 0), np.int32), bbox=np.array(struct.unpack_from("<6d", b, 12)),
                        ijk=np.stack([rec["i"], rec["j"], rec["k"]], 1).astype(np.int16), d=rec["d"].copy(),
                        expect_vt=np.array([3800, 7604]))
