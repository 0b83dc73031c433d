"""Test helper: writes a TensorFlow tensor-bundle (V2 checkpoint) in the published format --
`<prefix>.index` = LevelDB-style table (uncompressed blocks, restart interval 16, masked CRC-32C
block trailers) of BundleEntryProto values (with the masked CRC-32C of the tensor bytes),
`<prefix>.data-00000-of-00001` = raw tensor bytes -- so that disn_amd/tf_bundle.py can be
exercised without TensorFlow.  Not a reference file; format per tensorflow/core/util/tensor_bundle."""
import struct

import numpy as np

DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}

_TAB = []


def crc32c_py(data, crc=0):
    """bytewise table CRC-32C (Castagnoli) in plain Python: the independent check of the library's helper."""
    if not _TAB:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82f63b78 if c & 1 else 0)
            _TAB.append(c)
    c = crc ^ 0xffffffff
    for b in bytes(data):
        c = (c >> 8) ^ _TAB[(c ^ b) & 255]
    return c ^ 0xffffffff


def _crc(data):
    from disn_amd import tf_bundle
    return tf_bundle.crc32c(data)         # fast (the VGG bundle is 550 MB); pinned by crc32c_py + the known answer in the tests


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _trailer(blk):
    return b"\x00" + struct.pack("<I", _mask(_crc(blk + b"\x00")))


def _vi(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    return _vi(num << 3 | wt) + payload


def _block(items, restart_interval=16):
    buf, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        buf += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", max(len(restarts), 1))
    return bytes(buf)


def write_bundle(prefix, tensors, block_entries=7):
    data, items = bytearray(), [(b"", _field(1, 0, _vi(1)) + _field(2, 0, _vi(0)))]
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        shape = b"".join(_field(2, 2, _vi(len(d)) + d) for d in (_field(1, 0, _vi(s)) for s in a.shape))
        entry = (_field(1, 0, _vi(DT[a.dtype])) + _field(2, 2, _vi(len(shape)) + shape) + _field(3, 0, _vi(0))
                 + _field(4, 0, _vi(len(data))) + _field(5, 0, _vi(a.nbytes)) + _field(6, 5, struct.pack("<I", _mask(_crc(a.tobytes())))))
        items.append((name.encode(), entry))
        data += a.tobytes()
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    out, index = bytearray(), []
    for s in range(0, len(items), block_entries):          # several data blocks, like a real table
        blk = _block(items[s:s + block_entries])
        index.append((items[min(s + block_entries, len(items)) - 1][0] + b"\x00", _vi(len(out)) + _vi(len(blk))))
        out += blk + _trailer(blk)
    meta = _block([])
    moff = len(out)
    out += meta + _trailer(meta)
    iblk = _block(index, restart_interval=1)
    ioff = len(out)
    out += iblk + _trailer(iblk)
    footer = _vi(moff) + _vi(len(meta)) + _vi(ioff) + _vi(len(iblk))
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    open(prefix + ".index", "wb").write(bytes(out))
