"""Distance-field timing at the reference's ground-truth configuration (res 256, box expansion 1.2, polygon-soup
pipeline with sigma 0.03; preprocessing/create_point_sdf_grid.py:331-347) on a synthetic ShapeNet-like soup:
a 20 k-triangle closed body with an internal cavity plus open sheets."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import meshes
from disn_amd import distance_field as dfh

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
v1, f1 = meshes.icosphere(0.45, 5)                      # 20 480 triangles
v2, f2 = meshes.icosphere(0.2, 3, (0.1, 0, 0))
v3, f3 = meshes.torus(0.55, 0.08, 96, 24)
sheet_v = np.array([[-0.7, -0.5, 0.6], [0.7, -0.5, 0.6], [0.7, 0.5, 0.6], [-0.7, 0.5, 0.6]], float)
sheet_f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
v = np.concatenate([v1, v2, v3, sheet_v])
f = np.concatenate([f1, f2[:, [0, 2, 1]] + len(v1), f3 + len(v1) + len(v2), sheet_f + len(v1) + len(v2) + len(v3)]).astype(np.int32)
bbox = dfh.auto_box(v, 1.2)
df = dfh.DistanceField()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    U = df.unsigned(v, f, res, bbox)
    torch.cuda.synchronize(); t1 = time.time()
    S = df.signed_polygon_soup(v, f, res, bbox, 0.03)
    torch.cuda.synchronize(); t2 = time.time()
print("res %d, %d input triangles: unsigned field %.3f s; polygon-soup signed field %.3f s (offset surface %d triangles kept, %d dropped); "
      "%.1f M grid points" % (res, len(f), t1 - t0, t2 - t1, df.last["offset_tris"], df.last["dropped"], (res + 1) ** 3 / 1e6))
print("min %.4f max %.4f, negative points %d" % (float(S.min()), float(S.max()), int((S < 0).sum())))
