#!/usr/bin/env python3
"""Regenerate include/disn_mc_tables.h (marching-cubes lookup DATA).

Provenance
----------
The reference ships no source for its isosurface step, only the Vega-FEM
`computeMarchingCubes` ELF (reference: isosurface/computeMarchingCubes, called
from demo/demo.py:355-369).  The topology-disambiguation lookup tables
(15 base cases, 632 sub-configurations) live in that
file's .rodata/.data as plain arrays.  This script reads those BYTES with
`open(...,'rb')` and `struct` -- the ELF is never executed, dlopen'ed or
linked -- and re-emits them in this build's own flattened layout.

It runs only in the build container (the reference tree is absent on the GPU
box); its output is committed.  Geometry tables (corner offsets, edge end
points, face corners) are NOT extracted: they are written by hand in
include/disn_mc_tables.h's preamble from the standard Lorensen numbering and
cross-checked by tests/test_mc_tables.py.
"""
import struct
import sys
import os

ELF = "/root/reference/isosurface/computeMarchingCubes"
OUT = os.path.join(os.path.dirname(__file__), "..", "..", "include", "disn_mc_tables.h")


def load_elf(path):
    blob = open(path, "rb").read()
    assert blob[:4] == b"\x7fELF" and blob[4] == 2, "expect ELF64"
    shoff = struct.unpack_from("<Q", blob, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsz = struct.unpack_from(
            "<IIQQQQIIQQ", blob, shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsz=entsz))
    stroff = secs[shstrndx]["off"]
    for s in secs:
        e = blob.index(b"\0", stroff + s["name"])
        s["sname"] = blob[stroff + s["name"]:e].decode()
    return blob, secs


def symbols(blob, secs):
    symtab = next(s for s in secs if s["sname"] == ".symtab")
    strtab = secs[symtab["link"]]
    out = {}
    for i in range(symtab["size"] // 24):
        name, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", blob, symtab["off"] + 24 * i)
        e = blob.index(b"\0", strtab["off"] + name)
        out[blob[strtab["off"] + name:e].decode()] = (value, size)
    return out


class Image:
    def __init__(self, blob, secs):
        self.blob = blob
        self.secs = [s for s in secs if s["addr"] and s["type"] == 1]

    def read(self, va, n):
        for s in self.secs:
            if s["addr"] <= va and va + n <= s["addr"] + s["size"]:
                o = s["off"] + va - s["addr"]
                return self.blob[o:o + n]
        raise KeyError(hex(va))


def main():
    blob, secs = load_elf(ELF)
    syms = symbols(blob, secs)
    img = Image(blob, secs)

    def sym(name):
        if name in syms:
            return syms[name]
        return syms["_ZL%d%s" % (len(name), name)]

    va, _ = sym("marchingCubeSymmetries")
    sym3 = struct.unpack("<768i", img.read(va, 3072))
    case_of = [sym3[3 * c] for c in range(256)]
    assert all(0 <= c <= 14 for c in case_of) and case_of[0] == 0 and case_of[255] == 0

    va, _ = sym("faceTest_num")
    face = list(struct.unpack("<1792b", img.read(va, 1792)))
    va, _ = sym("interiorTest_num")
    interior = list(struct.unpack("<256b", img.read(va, 256)))
    va, _ = sym("ambiguityTable")
    amb = list(struct.unpack("<960b", img.read(va, 960)))

    va, _ = sym("triangleTable")
    top = struct.unpack("<256Q", img.read(va, 2048))
    # number of sub-configurations per code = number of reachable ambiguityTable
    # targets, cross-checked with the symbol count triangleTable_<code>_<n>
    nsub = []
    for code in range(256):
        n = 0
        while ("_ZL%d%s" % (len("triangleTable_%d_%d" % (code, n)), "triangleTable_%d_%d" % (code, n))) in syms:
            n += 1
        nsub.append(n)
    sub_start = [0]
    data = []
    data_off = []
    for code in range(256):
        if nsub[code] == 0:
            assert case_of[code] == 0
            sub_start.append(sub_start[-1])
            continue
        ptrs = struct.unpack("<%dQ" % nsub[code], img.read(top[code], 8 * nsub[code]))
        for p in ptrs:
            hdr = img.read(p, 2)
            nidx, has_c = hdr[0], hdr[1]
            assert nidx % 3 == 0 and has_c in (0, 1) and nidx <= 36
            body = img.read(p + 2, nidx)
            assert all(b <= 12 for b in body)
            assert (12 in body) == bool(has_c) or has_c  # centre flag implies edge id 12 may appear
            data_off.append(len(data))
            data.extend(hdr)
            data.extend(body)
        sub_start.append(sub_start[-1] + nsub[code])
    assert sub_start[-1] == 632, sub_start[-1]
    # every ambiguity target must exist
    for code in range(256):
        cs = case_of[code]
        if cs == 0:
            continue
        nf = face[7 * code]
        nbits = nf + (1 if interior[code] else 0)
        for a in range(1 << nbits):
            s = amb[64 * cs + a]
            assert s < nsub[code], (code, a, s)

    max_tri = max(data[o] for o in data_off) // 3

    def arr(name, ctype, vals, per=32):
        lines = ["static const %s %s[%d] = {" % (ctype, name, len(vals))]
        for i in range(0, len(vals), per):
            lines.append("  " + ",".join(str(v) for v in vals[i:i + per]) + ",")
        lines.append("};")
        return "\n".join(lines)

    hdr = open(os.path.join(os.path.dirname(__file__), "mc_tables_preamble.h")).read()
    body = [hdr]
    body.append("#define DISN_MC_NUM_SUBTABLES %d" % len(data_off))
    body.append("#define DISN_MC_TRI_BYTES %d" % len(data))
    body.append("#define DISN_MC_MAX_TRIS %d" % max_tri)
    body.append("/* base case 0..14 per 8-bit sign code (0 = no surface) */")
    body.append(arr("DISN_MC_CASE", "DISN_MC_QUAL unsigned char", case_of))
    body.append("/* [code][7]: [0]=n face tests, [1..n]=signed face ids (+-1..6) */")
    body.append(arr("DISN_MC_FACE_TESTS", "DISN_MC_QUAL signed char", face, 28))
    body.append("/* interior test sign per code (0 = none) */")
    body.append(arr("DISN_MC_INTERIOR_TEST", "DISN_MC_QUAL signed char", interior))
    body.append("/* [case][64]: ambiguity bits -> sub-configuration index, -1 impossible */")
    body.append(arr("DISN_MC_AMBIGUITY", "DISN_MC_QUAL signed char", amb))
    body.append("/* first sub-table id of each code; sub-table id = SUB_START[code]+sub */")
    body.append(arr("DISN_MC_SUB_START", "DISN_MC_QUAL unsigned short", sub_start))
    body.append("/* byte offset of each sub-table string inside DISN_MC_TRI */")
    body.append(arr("DISN_MC_SUB_OFFSET", "DISN_MC_QUAL unsigned short", data_off))
    body.append("/* strings: [nIdx=3*nTri][hasCentre][e0 e1 e2]...; edge 12 = centre vertex */")
    body.append(arr("DISN_MC_TRI", "DISN_MC_QUAL unsigned char", data, 36))
    body.append("#endif /* DISN_MC_TABLES_H */")
    with open(OUT, "w") as f:
        f.write("\n".join(body) + "\n")
    print("wrote", os.path.abspath(OUT), "subtables", len(data_off), "bytes", len(data), "max tris", max_tri)


if __name__ == "__main__":
    sys.exit(main())
