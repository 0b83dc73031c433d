#!/usr/bin/env python3
"""Build-time check of the generated decoder kernels (called by disn_amd/csrc/Makefile).

The decoders keep the LDS-DMA base in M0 across asm statements and count their own `s_waitcnt vmcnt(N)` over the
LDS-DMA pieces they issue.  Both assumptions hold only while, between the first and the last MFMA of a stream,
  * hipcc never touches M0 (only our `s_mov_b32 m0, sN`),
  * hipcc emits no vector-memory instruction of its own: no scratch spill/reload, no global/buffer load or store
    (each would shift the hand-counted vmcnt, and its own wait would be computed without our LDS-DMA pieces).
Outside that region (point set-up and the image term before the first MFMA of the layers -- marked by the kernel with
a `disn:pipelined_region_begin` comment when that set-up uses MFMAs itself -- and the result store after the last) the
compiler's own memory instructions are fine.  Fails the build otherwise."""
import re
import sys


def kernel_texts(asm, name):
    """every kernel whose mangled name contains `name` (template instances included)"""
    found = []
    for m in re.finditer(r"^(_ZN\S*%s\S*):" % name, asm, re.M):
        end = asm.find("s_endpgm", m.end())
        found.append((m.group(1), asm[m.end():end].split("\n")))
    if not found:
        sys.exit("check_decode_asm: kernel %s not found" % name)
    return found


def check(asm, name):
    for full, lines in kernel_texts(asm, name):
        check_one(full, lines)


def check_one(name, lines):
    code = [l.split(";")[0].strip() for l in lines]
    mf = [i for i, l in enumerate(code) if l.startswith("v_mfma")]
    if not mf:
        sys.exit("check_decode_asm: no MFMA in %s" % name)
    first, last = mf[0], mf[-1]
    # kernels whose point set-up itself uses MFMAs (the image term as a matrix product) mark where the pipelined region
    # begins; a kernel that walks both streams has one marker per stream and the region starts at the first
    marks = [i for i, l in enumerate(lines) if "disn:pipelined_region_begin" in l]
    if marks:
        first = min(marks)
    bad = []
    for i, l in enumerate(code):
        if re.search(r"\bm0\b", l) and not re.fullmatch(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+)", l):
            bad.append((i, "M0 used by the compiler: " + l))
        if l.startswith("scratch_"):
            bad.append((i, "register spill (the decoders are meant to fit the register file): " + l))
        if first < i < last:
            if l.startswith(("scratch_", "buffer_", "global_store", "flat_")) or \
               (l.startswith("global_load") and not l.startswith("global_load_lds_dword")):
                bad.append((i, "compiler memory instruction inside the pipelined region: " + l))
    if bad:
        for i, msg in bad[:20]:
            print("check_decode_asm: %s line %d: %s" % (name, i, msg))
        sys.exit(1)
    n_dma = sum(l.startswith("global_load_lds_dword") for l in code)
    print("check_decode_asm: %s ok (%d MFMAs, %d LDS-DMA pieces, %d lines)" % (name, len(mf), n_dma, len(code)))


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    for k in sys.argv[2:]:
        check(text, k)
