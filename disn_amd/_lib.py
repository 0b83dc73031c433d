"""ctypes binding of libdisn_hip.so (include/disn.h).  Fails loudly when the
library is missing or lacks a symbol -- there is no fallback path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdisn_hip.so")

c_void_p, c_int, c_int64, c_uint64, c_float, c_char_p = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, ctypes.c_char_p)


class DisnDecoderWeights(ctypes.Structure):
    _names = [s + "_" + l + "_" + k for s in ("g", "l")
              for l in ("f1c1", "f1c2", "f1c3", "f2c1", "f2c2", "f2c5") for k in ("w", "b")]
    _fields_ = [(n, ctypes.POINTER(ctypes.c_float)) for n in _names]


class DisnVggWeights(ctypes.Structure):
    _fields_ = [("conv_w", ctypes.POINTER(ctypes.c_float) * 13), ("conv_b", ctypes.POINTER(ctypes.c_float) * 13),
                ("fc_w", ctypes.POINTER(ctypes.c_float) * 3), ("fc_b", ctypes.POINTER(ctypes.c_float) * 3)]


class DisnMcGrid(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz", ctypes.c_int32),
                ("z_offset", ctypes.c_int32), ("nz_global", ctypes.c_int32),
                ("own0", ctypes.c_int32), ("own1", ctypes.c_int32)]


# name -> (restype, argtypes); mirrors include/disn.h one to one
SIGNATURES = {
    "disn_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "disn_destroy": (None, [c_void_p]),
    "disn_last_error": (c_char_p, [c_void_p]),
    "disn_version": (c_char_p, []),
    "disn_sdf_set_weights": (c_int, [c_void_p, ctypes.POINTER(DisnDecoderWeights)]),
    "disn_sdf_set_decoder": (c_int, [c_void_p, c_int]),
    "disn_sdf_selfcheck": (c_int, [c_void_p, c_void_p, c_void_p, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p]),
    "disn_sdf_autotune": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, ctypes.POINTER(c_int),
                                  ctypes.POINTER(c_float), c_void_p]),
    "disn_sdf_set_image": (c_int, [c_void_p, ctypes.POINTER(c_void_p), c_void_p, c_void_p]),
    "disn_sdf_grid": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "disn_sdf_points": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "disn_project_points": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "disn_resize_bilinear": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "disn_sample_features": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "disn_vgg_set_weights": (c_int, [c_void_p, c_int, ctypes.POINTER(DisnVggWeights)]),
    "disn_vgg_forward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, ctypes.POINTER(c_void_p), c_void_p, c_void_p]),
    "disn_mc_count": (c_int, [c_void_p, c_void_p, ctypes.POINTER(DisnMcGrid), c_float,
                              ctypes.POINTER(c_uint64), c_void_p]),
    "disn_mc_emit": (c_int, [c_void_p, c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "disn_mc_debug_info": (c_int, [c_void_p, c_void_p, c_void_p]),
    "disn_sort_triangles": (c_int, [c_void_p, c_void_p, c_uint64, c_uint64, c_void_p]),
    "disn_sort_triangles_static": (c_int, [c_void_p, c_void_p, c_uint64, c_uint64, c_void_p]),
    "disn_df_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "disn_df_flood": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    "disn_write_dist": (c_int, [c_char_p, c_int, c_void_p, c_void_p]),
    "disn_read_dist_header": (c_int, [c_char_p, ctypes.POINTER(c_int), c_void_p]),
    "disn_read_dist": (c_int, [c_char_p, c_void_p, ctypes.c_size_t]),
    "disn_read_ndist_header": (c_int, [c_char_p, ctypes.POINTER(c_int), c_void_p, ctypes.POINTER(c_int64)]),
    "disn_read_ndist": (c_int, [c_char_p, c_void_p, ctypes.c_size_t]),
    "disn_write_ndist": (c_int, [c_char_p, c_int, c_void_p, c_void_p, c_float]),
    "disn_crc32c": (ctypes.c_uint32, [ctypes.c_uint32, c_void_p, ctypes.c_size_t]),
    "disn_write_obj": (c_int, [c_char_p, c_void_p, c_uint64, c_void_p, c_uint64]),
}

_LIB = None


def load():
    """Load the HIP library; raises RuntimeError if it is absent (build it with
    `python -c 'import __graft_entry__ as g; g.build()'` or `make -C disn_amd/csrc`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("disn_amd: %s is missing -- the HIP extension is mandatory (no CPU fallback); "
                           "run `make -C disn_amd/csrc`" % LIB_PATH)
    # torch bundles its own HIP runtime; it must be the one already loaded when
    # libdisn_hip.so resolves libamdhip64 (two runtimes in one process do not share devices)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
