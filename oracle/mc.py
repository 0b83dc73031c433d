"""ctypes front-end of oracle/mc_oracle.c (CPU restatement of the reference's
`computeMarchingCubes`, reference: demo/demo.py:355-369; spec SURVEY.md App. B).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libdisn_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdisn_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.mc_oracle_run.restype = ctypes.c_void_p
        L.mc_oracle_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_float]
        for n in ("n_centre", "n_edge", "n_tri", "n_bad"):
            f = getattr(L, "mc_oracle_" + n)
            f.restype = ctypes.c_int64
            f.argtypes = [ctypes.c_void_p]
        L.mc_oracle_copy.argtypes = [ctypes.c_void_p] * 4
        L.mc_oracle_copy.restype = None
        L.mc_oracle_copy_layers.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.mc_oracle_copy_layers.restype = None
        L.mc_oracle_free.argtypes = [ctypes.c_void_p]
        L.mc_oracle_free.restype = None
        L.mc_oracle_classify.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_float, ctypes.c_void_p]
        L.mc_oracle_classify.restype = None
        _LIB = L
    return _LIB


def marching_cubes(grid, iso=0.0, bbox=(-1, -1, -1, 1, 1, 1)):
    """grid[k][j][i] float32 (z, y, x; x fastest).  Returns dict with verts
    (V,3) f64, tris (T,3) i32 0-based, keys (V,) i64, n_centre, n_bad."""
    g = np.ascontiguousarray(grid, dtype=np.float32)
    nz, ny, nx = g.shape
    bb = np.asarray(bbox, dtype=np.float64)
    L = lib()
    r = L.mc_oracle_run(g.ctypes.data, nx, ny, nz, bb.ctypes.data, ctypes.c_float(iso))
    if not r:
        raise MemoryError("mc_oracle_run")
    try:
        nc, ne, nt = L.mc_oracle_n_centre(r), L.mc_oracle_n_edge(r), L.mc_oracle_n_tri(r)
        verts = np.empty((nc + ne, 3), np.float64)
        tris = np.empty((nt, 3), np.int32)
        keys = np.empty((nc + ne,), np.int64)
        L.mc_oracle_copy(r, verts.ctypes.data, tris.ctypes.data, keys.ctypes.data)
        layers = np.empty((nt,), np.int32)
        L.mc_oracle_copy_layers(r, layers.ctypes.data)
        return dict(verts=verts, tris=tris, keys=keys, tri_layer=layers, n_centre=nc, n_edge=ne, n_tri=nt,
                    n_bad=L.mc_oracle_n_bad(r))
    finally:
        L.mc_oracle_free(r)


def classify(grid, iso=0.0):
    g = np.ascontiguousarray(grid, dtype=np.float32)
    nz, ny, nx = g.shape
    out = np.empty(g.shape, np.int16)
    lib().mc_oracle_classify(g.ctypes.data, nx, ny, nz, ctypes.c_float(iso), out.ctypes.data)
    return out


def marching_cubes_slab(grid_full, lo, hi, own0, own1, iso=0.0, bbox=(-1, -1, -1, 1, 1, 1)):
    """CPU restatement of ONE z-slab of the sharded extraction, computed from
    the planes [lo,hi) only.  Returns the slab's owned counts and, given the index bases,
    its mesh with global indices -- the same contract as disn_mc_count / disn_mc_emit.
    Call .counts first, exchange them, then .emit(centre_base, edge_base)."""
    g = np.ascontiguousarray(grid_full[lo:hi], dtype=np.float32)
    nzg, ny, nx = grid_full.shape
    bb = np.asarray(bbox, np.float64)
    h = (bb[5] - bb[2]) / (nzg - 1)
    sub_bbox = (bb[0], bb[1], bb[2] + lo * h, bb[3], bb[4], bb[2] + (hi - 1) * h)
    r = marching_cubes(g, iso=iso, bbox=sub_bbox)
    sxy = nx * ny
    nc = r["n_centre"]
    ck, ek = r["keys"][:nc], r["keys"][nc:]
    ck_layer = (-(ck + 1)) // sxy + (((-(ck + 1)) % sxy) != 0)      # centre key = -k*sxy + j*nx + i - 1
    ck_layer = np.where(ck + 1 <= 0, ck_layer, 0)                    # k = 0 keys are >= -1
    ek_plane = (ek // 3) // sxy
    c_own = (ck_layer >= own0) & (ck_layer < own1)
    e_own = (ek_plane >= own0) & (ek_plane < own1)
    t_own = (r["tri_layer"] >= own0) & (r["tri_layer"] < own1)
    e_low = int((ek_plane < own0).sum())
    c_above = int((ck_layer >= own1).sum())                          # centres come in descending k

    class Slab:
        counts = (int(c_own.sum()), int(e_own.sum()), int(t_own.sum()))

        @staticmethod
        def emit(centre_base, edge_base):
            verts = np.concatenate([r["verts"][:nc][c_own], r["verts"][nc:][e_own]])
            # positions: recompute z from the global plane index so that they equal the single pass bit for bit
            keys = np.concatenate([ck[c_own] - lo * sxy, ek[e_own] + 3 * lo * sxy])
            t = r["tris"][t_own].astype(np.int64)
            is_c = t < nc
            glob = np.where(is_c, centre_base + (t - c_above), edge_base + (t - nc - e_low))
            # re-canonicalise: global numbering can change which corner is smallest
            m = np.argmin(glob, axis=1)
            rows = np.arange(len(glob))
            out = np.stack([glob[rows, m], glob[rows, (m + 1) % 3], glob[rows, (m + 2) % 3]], 1).astype(np.int32)
            return verts, out, keys
    return Slab
