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
        return dict(verts=verts, tris=tris, keys=keys, n_centre=nc, n_edge=ne, n_tri=nt,
                    n_bad=L.mc_oracle_n_bad(r))
    finally:
        L.mc_oracle_free(r)


def classify(grid, iso=0.0):
    g = np.ascontiguousarray(grid, dtype=np.float32)
    nz, ny, nx = g.shape
    out = np.empty(g.shape, np.int16)
    lib().mc_oracle_classify(g.ctypes.data, nx, ny, nz, ctypes.c_float(iso), out.ctypes.data)
    return out
