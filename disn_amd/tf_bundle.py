"""TensorFlow checkpoint ("tensor bundle" V2) reader without TensorFlow.

Replaces tf.train.get_checkpoint_state + Saver.restore of the reference
(demo/demo.py:153-165, 228-241) for inference: parses `<prefix>.index` (a
LevelDB-style SSTable whose values are BundleEntryProto messages) and slices the
tensors out of `<prefix>.data-00000-of-0000N`.  Only what the released DISN
checkpoints need: uncompressed table blocks, DT_FLOAT / DT_INT32 / DT_INT64
tensors, unsliced entries.

Every table block's trailer checksum and every tensor's BundleEntryProto.crc32c
(masked CRC-32C, as TensorFlow writes them) are verified: a truncated or
corrupted checkpoint is an error, never a silently different mesh.

PARITY UNPINNED: the released checkpoints are not in the reference tree
(README.md:39-50 fetches them), so this reader is exercised against bundles
written by tests/tf_bundle_writer.py, which follows the same published format
(tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table).
"""
import ctypes
import os
import re
import struct

import numpy as np

from . import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


class BundleError(RuntimeError):
    pass


def _varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise BundleError("bad varint")


def _block_entries(block):
    """Yield (key, value) of one table block (prefix-compressed keys + restart array)."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def crc32c(data, crc=0):
    """CRC-32C of a bytes-like object or a contiguous uint8 array (libdisn_hip.so's host helper)."""
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
    return int(_lib.load().disn_crc32c(crc, a.ctypes.data_as(ctypes.c_void_p), a.size))


def crc_mask(crc):
    """tensorflow/core/lib/hash/crc32c.h Mask(): rotate right by 15 and add a constant."""
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _read_block(blob, offset, size):
    if offset + size + 5 > len(blob):
        raise BundleError("block handle outside the file (truncated index?)")
    ctype = blob[offset + size]
    stored = struct.unpack_from("<I", blob, offset + size + 1)[0]
    if crc_mask(crc32c(blob[offset:offset + size + 1])) != stored:
        raise BundleError("table block at offset %d fails its checksum (corrupt index file)" % offset)
    if ctype != 0:
        raise BundleError("compressed table blocks (type %d) are not supported" % ctype)
    return blob[offset:offset + size]


def _parse_proto(buf):
    """Minimal protobuf wire parser -> {field: [values]} (varint -> int, len-delimited -> bytes, fixed32 -> int)."""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise BundleError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _shape(msg):
    dims = []
    for d in _parse_proto(msg).get(2, []):          # TensorShapeProto.dim
        size = _parse_proto(d).get(1, [0])[0]
        dims.append(size - (1 << 64) if size >= (1 << 63) else size)
    return tuple(dims)


def read_index(prefix):
    """-> (num_shards, {name: dict(dtype, shape, shard, offset, size)})"""
    path = prefix + ".index"
    try:
        blob = open(path, "rb").read()
    except OSError as e:
        raise BundleError("cannot read %s: %s" % (path, e))
    if len(blob) < 48 or struct.unpack_from("<Q", blob, len(blob) - 8)[0] != TABLE_MAGIC:
        raise BundleError("%s is not a tensor-bundle index (bad table magic)" % path)
    footer = blob[-48:]
    _, p = _varint(footer, 0)          # metaindex handle
    _, p = _varint(footer, p)
    ioff, p = _varint(footer, p)       # index handle
    isize, p = _varint(footer, p)
    entries, shards = {}, 1
    for _, handle in _block_entries(_read_block(blob, ioff, isize)):
        boff, q = _varint(handle, 0)
        bsize, q = _varint(handle, q)
        for key, val in _block_entries(_read_block(blob, boff, bsize)):
            if key == b"":
                hdr = _parse_proto(val)                     # BundleHeaderProto: 1 num_shards, 2 endianness
                shards = hdr.get(1, [1])[0]
                if hdr.get(2, [0])[0] != 0:
                    raise BundleError("big-endian bundles are not supported")
                continue
            m = _parse_proto(val)                           # BundleEntryProto
            if 7 in m:
                raise BundleError("sliced (partitioned) variable %r is not supported" % key.decode())
            dt = m.get(1, [0])[0]
            if dt not in DTYPES:
                raise BundleError("tensor %r has unsupported dtype enum %d" % (key.decode(), dt))
            entries[key.decode()] = dict(dtype=DTYPES[dt], shape=_shape(m[2][0]) if 2 in m else (),
                                         shard=m.get(3, [0])[0], offset=m.get(4, [0])[0], size=m.get(5, [0])[0],
                                         crc32c=m.get(6, [None])[0])
    return shards, entries


def read_bundle(prefix, names=None):
    """-> {name: ndarray}.  names: optional iterable or regex string to restrict what is loaded."""
    shards, entries = read_index(prefix)
    if isinstance(names, str):
        rx = re.compile(names)
        wanted = [n for n in entries if rx.search(n)]
    else:
        wanted = list(entries) if names is None else list(names)
    files, out = {}, {}
    for n in wanted:
        if n not in entries:
            raise BundleError("checkpoint %s has no tensor %r" % (prefix, n))
        e = entries[n]
        path = "%s.data-%05d-of-%05d" % (prefix, e["shard"], shards)
        if path not in files:
            try:
                files[path] = np.memmap(path, dtype=np.uint8, mode="r")
            except (OSError, ValueError) as err:
                raise BundleError("cannot map %s: %s" % (path, err))
        count = int(np.prod(e["shape"], dtype=np.int64))
        if count * np.dtype(e["dtype"]).itemsize != e["size"] or e["offset"] + e["size"] > files[path].shape[0]:
            raise BundleError("tensor %r: size/offset do not match its shape" % n)
        raw = np.array(files[path][e["offset"]:e["offset"] + e["size"]])      # one copy out of the map
        if e["crc32c"] is None:
            raise BundleError("tensor %r carries no checksum" % n)
        if crc_mask(crc32c(raw)) != e["crc32c"]:
            raise BundleError("tensor %r fails its checksum (corrupt or truncated data file %s)" % (n, path))
        out[n] = raw.view(e["dtype"]).reshape(e["shape"])
    return out


def latest_checkpoint(log_dir):
    """tf.train.get_checkpoint_state(log_dir).model_checkpoint_path (reference demo/demo.py:155-158)."""
    state = os.path.join(log_dir, "checkpoint")
    try:
        txt = open(state).read()
    except OSError:
        return None
    m = re.search(r'^model_checkpoint_path:\s*"([^"]+)"', txt, re.M)
    if not m:
        raise BundleError("%s has no model_checkpoint_path" % state)
    p = m.group(1)
    return p if os.path.isabs(p) else os.path.join(log_dir, p)


# ---- variable-name maps of the path -----------------------
_LAYER = {"f1c1": "fold1/conv1", "f1c2": "fold1/conv2", "f1c3": "fold1/conv3",
          "f2c1": "fold2/conv1", "f2c2": "fold2/conv2", "f2c5": "fold2/conv5"}
_SCOPE = {"g": "sdfprediction", "l": "sdfprediction_imgfeat"}
VGG_BLOCKS = [(1, 2), (2, 2), (3, 3), (4, 3), (5, 3)]


def decoder_weights(prefix):
    """The 24 arrays of include/disn.h's DisnDecoderWeights from an SDF_DISN checkpoint."""
    names = {}
    for s, scope in _SCOPE.items():
        for l, layer in _LAYER.items():
            names["%s_%s_w" % (s, l)] = "%s/%s/weights" % (scope, layer)
            names["%s_%s_b" % (s, l)] = "%s/%s/biases" % (scope, layer)
    t = read_bundle(prefix, names.values())
    out = {}
    for k, n in names.items():
        a = t[n]
        out[k] = a.reshape(a.shape[-2:]) if k.endswith("_w") else a.reshape(-1)   # [1,1,Cin,Cout] -> [Cin,Cout]
    return out


def vgg_weights(prefix, scope="vgg_16"):
    """slim vgg_16 variables -> the OIHW dict disn_amd.encoder.Vgg16Encoder takes."""
    names = {}
    for b, n in VGG_BLOCKS:
        for i in range(1, n + 1):
            names["conv%d_%d" % (b, i)] = "%s/conv%d/conv%d_%d" % (scope, b, b, i)
    for fc in ("fc6", "fc7", "fc8"):
        names[fc] = "%s/%s" % (scope, fc)
    t = read_bundle(prefix, [v + "/weights" for v in names.values()] + [v + "/biases" for v in names.values()])
    out = {}
    for k, n in names.items():
        out[k + "_w"] = np.ascontiguousarray(t[n + "/weights"].transpose(3, 2, 0, 1))   # HWIO -> OIHW
        out[k + "_b"] = t[n + "/biases"]
    return out


def camera_weights(prefix):
    """cameraprediction/{scale,ortho6d,translation}/fc{1,2,3}/{weights,biases} (models/posenet_old.py:105-134)."""
    keys = ["%s/fc%d/%s" % (h, i, p) for h in ("scale", "ortho6d", "translation") for i in (1, 2, 3)
            for p in ("weights", "biases")]
    t = read_bundle(prefix, ["cameraprediction/" + k for k in keys])
    return {k: t["cameraprediction/" + k] for k in keys}
