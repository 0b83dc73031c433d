"""ctypes front-end of oracle/df_oracle.c plus a CPU restatement of the polygon-soup signed-distance pipeline
(Xu & Barbic, Graphics Interface 2014) that the reference's ground-truth tool runs for
`computeDistanceField <obj> res res res -s -m 1 -g sigma` (reference call site:
preprocessing/create_point_sdf_grid.py:200-210).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED against the tool (closed binary, not executed;
the reference holds no input mesh and no full output of it): what is restated is the published pipeline.
"""
import ctypes

import numpy as np

from . import mc


def _lib():
    L = mc.lib()
    if not getattr(L, "_df_ready", False):
        L.df_oracle_eval.restype = None
        L.df_oracle_eval.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L._df_ready = True
    return L


def pseudonormals(verts, faces):
    """[T][7][3]: unit face normal, angle-weighted vertex pseudonormals (a, b, c), edge pseudonormals (ab, ac, bc) --
    plain loops over dictionaries, on purpose not the vectorised code of the product."""
    verts = np.asarray(verts, np.float64)
    fn = []
    for f in faces:
        a, b, c = verts[f[0]], verts[f[1]], verts[f[2]]
        n = np.cross(b - a, c - a)
        fn.append(n / np.linalg.norm(n))
    vsum, esum = {}, {}
    for f, n in zip(faces, fn):
        for s in range(3):
            v0, v1, v2 = int(f[s]), int(f[(s + 1) % 3]), int(f[(s + 2) % 3])
            e1, e2 = verts[v1] - verts[v0], verts[v2] - verts[v0]
            cosang = float(np.dot(e1, e2) / (np.linalg.norm(e1) * np.linalg.norm(e2)))
            ang = np.arccos(min(1.0, max(-1.0, cosang)))
            vsum[v0] = vsum.get(v0, 0.0) + ang * n
            key = (min(v0, v1), max(v0, v1))
            esum[key] = esum.get(key, 0.0) + n
    out = np.zeros((len(faces), 7, 3), np.float64)
    for t, (f, n) in enumerate(zip(faces, fn)):
        a, b, c = int(f[0]), int(f[1]), int(f[2])
        out[t, 0] = n
        out[t, 1], out[t, 2], out[t, 3] = vsum[a], vsum[b], vsum[c]
        for slot, (u, v) in ((4, (a, b)), (5, (a, c)), (6, (b, c))):
            out[t, slot] = esum[(min(u, v), max(u, v))]
    return out


def distance_field(verts, faces, res, bbox, signed=False):
    """Brute-force float64 distance from every point of the (res+1)^3 grid (or res = (nx, ny, nz) points) to the mesh."""
    verts = np.asarray(verts, np.float64)
    faces = np.asarray(faces, np.int64)
    nx, ny, nz = (res + 1,) * 3 if np.isscalar(res) else res
    geom = np.ascontiguousarray(verts[faces].reshape(len(faces), 9))
    nrm = np.ascontiguousarray(pseudonormals(verts, faces)) if signed else None
    bb = np.asarray(bbox, np.float64)
    out = np.empty((nz, ny, nx), np.float32)
    _lib().df_oracle_eval(geom.ctypes.data, nrm.ctypes.data if signed else None, len(faces), nx, ny, nz, bb.ctypes.data,
                          out.ctypes.data)
    return out


def polygon_soup_field(verts, faces, res, bbox, sigma):
    """The `-m 1` pipeline: unsigned field -> offset surface at sigma (marching cubes) -> its exterior part (the grid
    points with U > sigma reachable from the box wall) -> signed field of that closed surface -> + sigma, so that the
    zero level comes back to the input surface."""
    from scipy import ndimage
    n = res + 1
    U = distance_field(verts, faces, res, bbox)
    m = mc.marching_cubes(U, iso=sigma, bbox=bbox)
    lab, _ = ndimage.label(U > np.float32(sigma))
    wall = np.zeros_like(lab, bool)
    wall[0], wall[-1], wall[:, 0], wall[:, -1], wall[:, :, 0], wall[:, :, -1] = (True,) * 6
    ext = np.isin(lab, np.unique(lab[wall & (lab > 0)])).reshape(-1)
    nc = m["n_centre"]
    keys = m["keys"][nc:]
    p, d = keys // 3, keys % 3
    stride = np.array([1, n, n * n])[d]
    Uf = U.reshape(-1)
    pos = np.where(Uf[p] > np.float32(sigma), p, p + stride)      # the endpoint on the positive side
    v_ext = np.concatenate([np.zeros(nc, bool), ext[pos]])
    keep = v_ext[m["tris"]].any(axis=1)
    kept = m["tris"][keep]
    V = m["verts"]
    a, b, c = V[kept[:, 0]], V[kept[:, 1]], V[kept[:, 2]]
    ok = (np.cross(b - a, c - a) ** 2).sum(1) > 0
    kept = kept[ok]
    return distance_field(V, kept, res, bbox, signed=True) + np.float32(sigma), dict(unsigned=U, offset_tris=int(keep.sum()),
                                                                                      dropped=int((~keep).sum()))
