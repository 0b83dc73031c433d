"""Host-side mirror of the reference's inference surface over the C ABI.

Reference counterparts:
  Context.set_weights    <- Saver.restore                   (demo/demo.py:153-165)
  Context.set_image      <- the VGG end points fed per chunk (models/model_normalization.py:171-185)
  Context.sdf_grid       <- test_one_epoch's chunk loop      (demo/demo.py:292-335)
  Context.sdf_points     <- sess.run(pred_sdf, feed_dict)    (demo/demo.py:317-326)
  Context.marching_cubes <- create_obj / computeMarchingCubes (demo/demo.py:355-369)
torch is used only for device memory and streams.
"""
import ctypes

import numpy as np
import torch

from . import _lib

FEAT_SHAPES = [(224, 224, 64), (112, 112, 128), (56, 56, 256), (28, 28, 512), (14, 14, 512)]
WEIGHT_SHAPES = {"f1c1": (3, 64), "f1c2": (64, 256), "f1c3": (256, 512), "f2c2": (512, 256), "f2c5": (256, 1)}
F2C1_ROWS = {"g": 1536, "l": 1984}


class DisnError(RuntimeError):
    pass


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t, device):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        t = torch.as_tensor(np.asarray(t, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


class Context:
    """One DisnCtx on one GPU (one per thread/device; not thread-safe)."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise DisnError("disn_amd needs a GPU: torch.cuda.is_available() is False (no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        torch.cuda.set_device(self.device)
        h = ctypes.c_void_p()
        rc = self.lib.disn_create(self.device.index or 0, ctypes.byref(h))
        if rc != 0:
            raise DisnError("disn_create(%s) failed with %d" % (self.device, rc))
        self._h = h
        self._mc_counts = None
        self.auto_decoder = True       # decoder arithmetic chosen by measurement at the first evaluation after set_weights
        self._tuned = False
        self.decoder = "mixed"         # what the library runs right now (its own default until tuned / set)
        self.decoder_report = None     # last autotune: {"chosen", "errors", "tolerance", "margin"}
        self.weights_version = 0       # bumped by set_weights (MeshPipeline re-agrees on the decoder across ranks)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.disn_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise DisnError("%s failed (%d): %s" % (what, rc, self.lib.disn_last_error(self._h).decode()))

    # ---- decoder ---------------------------------------------------------
    def set_weights(self, w):
        """w: dict name -> float32 array, names g_f1c1_w ... l_f2c5_b (include/disn.h)."""
        st = _lib.DisnDecoderWeights()
        keep = []
        for s in ("g", "l"):
            for l in ("f1c1", "f1c2", "f1c3", "f2c1", "f2c2", "f2c5"):
                cin, cout = (F2C1_ROWS[s], 512) if l == "f2c1" else WEIGHT_SHAPES[l]
                for k, shape in (("w", (cin, cout)), ("b", (cout,))):
                    name = "%s_%s_%s" % (s, l, k)
                    if name not in w:
                        raise DisnError("set_weights: missing %s" % name)
                    a = np.ascontiguousarray(np.asarray(w[name], dtype=np.float32))
                    if a.shape != shape:
                        raise DisnError("set_weights: %s has shape %s, expected %s" % (name, a.shape, shape))
                    keep.append(a)
                    setattr(st, name, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        self._check(self.lib.disn_sdf_set_weights(self._h, ctypes.byref(st)), "disn_sdf_set_weights")
        self._tuned = False            # new weights: the arithmetic is measured again
        self.weights_version += 1

    DECODERS = {"mixed": 0, "bf16x3": 1, "mixed4": 2}
    DECODER_NAMES = {v: k for k, v in DECODERS.items()}
    TOLERANCE = 1e-4                   # BASELINE.json north_star: SDF within 1e-4 absolute
    MARGIN = 5.0                       # the chosen arithmetic keeps 5x to it on the sample points

    def set_decoder(self, name):
        """'auto' (default): at the first evaluation after set_weights the library measures, on 4096 points of the
        box being evaluated, the error of each arithmetic against the as-written fp32 graph and takes the cheapest
        one that keeps MARGIN x to TOLERANCE: 'mixed4' (fp16 + 2 fp8 residual products in all four MFMA layers),
        'mixed' (the same on three layers, bf16x3 on fold2/conv2) or 'bf16x3' (every layer).  A name pins one."""
        if name == "auto":
            self.auto_decoder, self._tuned = True, False
            return
        if name not in self.DECODERS:
            raise DisnError("set_decoder: unknown decoder %r (have auto, %s)" % (name, ", ".join(sorted(self.DECODERS))))
        self._check(self.lib.disn_sdf_set_decoder(self._h, self.DECODERS[name]), "disn_sdf_set_decoder")
        self.auto_decoder, self.decoder = False, name

    def selfcheck(self, trans_mat, sdf_params=(-1, -1, -1, 1, 1, 1), n=4096):
        """(max |decoder - as-written fp32 graph|, max |SDF|) over n sample points of the box (disn_sdf_selfcheck)."""
        tm = np.ascontiguousarray(np.asarray(trans_mat, dtype=np.float32).reshape(12))
        sp = np.ascontiguousarray(np.asarray(sdf_params, dtype=np.float64).reshape(6))
        err, mx = ctypes.c_float(), ctypes.c_float()
        self._check(self.lib.disn_sdf_selfcheck(self._h, tm.ctypes.data, sp.ctypes.data, n, ctypes.byref(err),
                                                ctypes.byref(mx), _stream()), "disn_sdf_selfcheck")
        return float(err.value), float(mx.value)

    def autotune(self, trans_mat, sdf_params=(-1, -1, -1, 1, 1, 1), n=4096, tolerance=None, margin=None):
        """Pick the decoder arithmetic by measurement (disn_sdf_autotune); raises if none keeps the margin."""
        tm = np.ascontiguousarray(np.asarray(trans_mat, dtype=np.float32).reshape(12))
        sp = np.ascontiguousarray(np.asarray(sdf_params, dtype=np.float64).reshape(6))
        tol = self.TOLERANCE if tolerance is None else tolerance
        mg = self.MARGIN if margin is None else margin
        chosen, errs = ctypes.c_int(-1), (ctypes.c_float * 3)()
        rc = self.lib.disn_sdf_autotune(self._h, tm.ctypes.data, sp.ctypes.data, n, tol, mg, ctypes.byref(chosen), errs,
                                        _stream())
        if chosen.value >= 0:
            self.decoder = self.DECODER_NAMES[chosen.value]
        self.decoder_report = {"chosen": self.decoder, "tolerance": tol, "margin": mg, "points": n,
                               "errors": {self.DECODER_NAMES[i]: (float(errs[i]) if errs[i] >= 0 else None) for i in range(3)}}
        self._check(rc, "disn_sdf_autotune")
        self._tuned = True
        return self.decoder_report

    def _auto(self, trans_mat, sdf_params):
        if self.auto_decoder and not self._tuned:
            self.autotune(trans_mat, sdf_params)

    def set_image(self, feat_maps, global_feat):
        """feat_maps: five NHWC/HWC float32 end points (conv1_2..conv5_3); global_feat: [1024]."""
        if len(feat_maps) != 5:
            raise DisnError("set_image: need the five VGG end points")
        dev = []
        for f, shape in zip(feat_maps, FEAT_SHAPES):
            t = _dev_f32(f, self.device)
            if t.dim() == 4:
                t = t[0]
            if tuple(t.shape) != shape:
                raise DisnError("set_image: feature map shape %s, expected %s" % (tuple(t.shape), shape))
            dev.append(t.contiguous())
        g = _dev_f32(global_feat, self.device).reshape(-1)
        if g.numel() != 1024:
            raise DisnError("set_image: global feature must have 1024 values")
        ptrs = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in dev])
        self._check(self.lib.disn_sdf_set_image(self._h, ptrs, g.data_ptr(), _stream()), "disn_sdf_set_image")
        self._img_keep = (dev, g)  # the kernels read them asynchronously

    def sdf_grid(self, trans_mat, res, z0=0, z1=None, sdf_params=(-1, -1, -1, 1, 1, 1), out=None):
        """SDF on planes z0..z1 of the (res+1)^3 grid -> float32 tensor [(z1-z0+1), res+1, res+1]."""
        z1 = res if z1 is None else z1
        tm = np.ascontiguousarray(np.asarray(trans_mat, dtype=np.float32).reshape(12))
        sp = np.ascontiguousarray(np.asarray(sdf_params, dtype=np.float64).reshape(6))
        self._auto(tm, sp)
        shape = (z1 - z0 + 1, res + 1, res + 1)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise DisnError("sdf_grid: bad output tensor")
        self._check(self.lib.disn_sdf_grid(self._h, tm.ctypes.data, sp.ctypes.data, res, z0, z1, out.data_ptr(),
                                           _stream()), "disn_sdf_grid")
        return out

    def sdf_points(self, trans_mat, pts, pts_rot=None):
        """Raw pred_sdf (not /10) at arbitrary points, like sess.run(pred_sdf)."""
        tm = np.ascontiguousarray(np.asarray(trans_mat, dtype=np.float32).reshape(12))
        p = _dev_f32(pts, self.device).reshape(-1, 3)
        pr = p if pts_rot is None else _dev_f32(pts_rot, self.device).reshape(-1, 3)
        if pr.shape != p.shape:
            raise DisnError("sdf_points: pts and pts_rot differ in shape")
        self._auto(tm, (-1, -1, -1, 1, 1, 1))
        out = torch.empty(p.shape[0], dtype=torch.float32, device=self.device)
        self._check(self.lib.disn_sdf_points(self._h, tm.ctypes.data, p.data_ptr(), pr.data_ptr(), p.shape[0],
                                             out.data_ptr(), _stream()), "disn_sdf_points")
        return out

    def project_points(self, trans_mat, pts):
        tm = np.ascontiguousarray(np.asarray(trans_mat, dtype=np.float32).reshape(12))
        p = _dev_f32(pts, self.device).reshape(-1, 3)
        out = torch.empty((p.shape[0], 2), dtype=torch.float32, device=self.device)
        self._check(self.lib.disn_project_points(self._h, tm.ctypes.data, p.data_ptr(), p.shape[0], out.data_ptr(),
                                                 _stream()), "disn_project_points")
        return out

    def resize_bilinear(self, feat, oh, ow):
        f = _dev_f32(feat, self.device)
        ih, iw, c = f.shape
        out = torch.empty((oh, ow, c), dtype=torch.float32, device=self.device)
        self._check(self.lib.disn_resize_bilinear(self._h, f.data_ptr(), ih, iw, c, out.data_ptr(), oh, ow, _stream()),
                    "disn_resize_bilinear")
        return out

    def sample_features(self, xy):
        x = _dev_f32(xy, self.device).reshape(-1, 2)
        out = torch.empty((x.shape[0], 1472), dtype=torch.float32, device=self.device)
        self._check(self.lib.disn_sample_features(self._h, x.data_ptr(), x.shape[0], out.data_ptr(), _stream()),
                    "disn_sample_features")
        return out

    # ---- marching cubes ---------------------------------------------------
    def mc_count(self, sdf, iso=0.0, z_offset=0, nz_global=None, own=None):
        """Phase 1 on a device float32 tensor [nz, ny, nx]; returns (n_centre, n_edge, n_tri)."""
        if not (sdf.is_cuda and sdf.dtype == torch.float32 and sdf.is_contiguous() and sdf.dim() == 3):
            raise DisnError("mc_count: sdf must be a contiguous CUDA float32 [nz,ny,nx] tensor")
        nz, ny, nx = sdf.shape
        own = (0, nz) if own is None else own
        g = _lib.DisnMcGrid(nx, ny, nz, z_offset, nz if nz_global is None else nz_global, own[0], own[1])
        c = (ctypes.c_uint64 * 3)()
        self._check(self.lib.disn_mc_count(self._h, sdf.data_ptr(), ctypes.byref(g), ctypes.c_float(iso), c, _stream()),
                    "disn_mc_count")
        self._mc_counts = (int(c[0]), int(c[1]), int(c[2]))
        self._mc_keep = sdf
        return self._mc_counts

    def mc_emit(self, bbox=(-1, -1, -1, 1, 1, 1), centre_base=0, edge_base=None, sort=True, want_keys=False):
        if self._mc_counts is None:
            raise DisnError("mc_emit: call mc_count first")
        nc, ne, nt = self._mc_counts
        edge_base = nc if edge_base is None else edge_base
        bb = np.ascontiguousarray(np.asarray(bbox, dtype=np.float64).reshape(6))
        verts = torch.empty((nc + ne, 3), dtype=torch.float64, device=self.device)
        tris = torch.empty((nt, 3), dtype=torch.int32, device=self.device)
        keys = torch.empty((nc + ne,), dtype=torch.int64, device=self.device) if want_keys else None
        self._check(self.lib.disn_mc_emit(self._h, bb.ctypes.data, centre_base, edge_base, verts.data_ptr(),
                                          keys.data_ptr() if want_keys else None, tris.data_ptr(), 1 if sort else 0,
                                          _stream()), "disn_mc_emit")
        return (verts, tris, keys) if want_keys else (verts, tris)

    def marching_cubes(self, sdf, iso=0.0, bbox=(-1, -1, -1, 1, 1, 1), want_keys=False):
        """Whole-grid isosurface: the reference's computeMarchingCubes on a device grid."""
        self.mc_count(sdf, iso)
        return self.mc_emit(bbox, want_keys=want_keys)

    def mc_debug_info(self):
        sdf = self._mc_keep
        out = torch.empty(sdf.shape, dtype=torch.int16, device=self.device)
        self._check(self.lib.disn_mc_debug_info(self._h, out.data_ptr(), _stream()), "disn_mc_debug_info")
        return out

    def sort_triangles(self, tris, max_index, static=False):
        """In-place lexicographic sort; static=True takes the allocation-free entry point (<= 2^20 triangles)."""
        if not (tris.is_cuda and tris.dtype == torch.int32 and tris.is_contiguous()):
            raise DisnError("sort_triangles: need a contiguous CUDA int32 [n,3] tensor")
        fn, name = ((self.lib.disn_sort_triangles_static, "disn_sort_triangles_static") if static else
                    (self.lib.disn_sort_triangles, "disn_sort_triangles"))
        self._check(fn(self._h, tris.data_ptr(), tris.shape[0], max_index, _stream()), name)
        return tris


# ---- host file formats -------------------------------------------------------
def write_dist(path, res, bbox, values):
    v = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
    if v.size != (res + 1) ** 3:
        raise DisnError("write_dist: need (res+1)^3 values")
    bb = np.ascontiguousarray(np.asarray(bbox, dtype=np.float64).reshape(6))
    rc = _lib.load().disn_write_dist(path.encode(), res, bb.ctypes.data, v.ctypes.data)
    if rc != 0:
        raise DisnError("disn_write_dist(%s) failed (%d)" % (path, rc))


def read_dist(path):
    lib = _lib.load()
    res = ctypes.c_int()
    bb = np.empty(6, np.float64)
    rc = lib.disn_read_dist_header(path.encode(), ctypes.byref(res), bb.ctypes.data)
    if rc != 0:
        raise DisnError("disn_read_dist_header(%s) failed (%d)" % (path, rc))
    n = res.value + 1
    v = np.empty(n * n * n, np.float32)
    rc = lib.disn_read_dist(path.encode(), v.ctypes.data, v.size)
    if rc != 0:
        raise DisnError("disn_read_dist(%s) failed (%d)" % (path, rc))
    return res.value, bb, v.reshape(n, n, n)


def read_narrowband_dist(path):
    """Narrow-band .n.dist (the reference tool's `-n` input) -> (res, bbox, dense grid [z,y,x], n_cells);
    cells the file does not list read +FLT_MAX outside the surface and -FLT_MAX inside it (side taken from the
    last listed cell passed on the way along +x from the box wall)."""
    lib = _lib.load()
    res, n = ctypes.c_int(), ctypes.c_int64()
    bb = np.empty(6, np.float64)
    rc = lib.disn_read_ndist_header(path.encode(), ctypes.byref(res), bb.ctypes.data, ctypes.byref(n))
    if rc != 0:
        raise DisnError("disn_read_ndist_header(%s) failed (%d)" % (path, rc))
    n1 = res.value + 1
    v = np.empty(n1 * n1 * n1, np.float32)
    rc = lib.disn_read_ndist(path.encode(), v.ctypes.data, v.size)
    if rc != 0:
        raise DisnError("disn_read_ndist(%s) failed (%d)" % (path, rc))
    return res.value, bb, v.reshape(n1, n1, n1), n.value


def write_narrowband_dist(path, res, bbox, values, band):
    """the cells with |value| < band as a narrow-band .n.dist (the distance-field tool's `-n -w band` output)"""
    v = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
    if v.size != (res + 1) ** 3:
        raise DisnError("write_narrowband_dist: need (res+1)^3 values")
    bb = np.ascontiguousarray(np.asarray(bbox, dtype=np.float64).reshape(6))
    rc = _lib.load().disn_write_ndist(path.encode(), res, bb.ctypes.data, v.ctypes.data, ctypes.c_float(band))
    if rc != 0:
        raise DisnError("disn_write_ndist(%s) failed (%d)" % (path, rc))


def write_obj(path, verts, tris):
    v = np.ascontiguousarray(np.asarray(verts, dtype=np.float64).reshape(-1, 3))
    t = np.ascontiguousarray(np.asarray(tris, dtype=np.int32).reshape(-1, 3))
    rc = _lib.load().disn_write_obj(path.encode(), v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0])
    if rc != 0:
        raise DisnError("disn_write_obj(%s) failed (%d)" % (path, rc))
