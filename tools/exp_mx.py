# NOTE: timing-only variants of the ROUND-2 kernels (patch anchors no longer match the source); kept as the record of how
# the round-2 numbers in DESIGN.md s3.1 were taken.  Current kernels: tools/seg_patch_mx4.py.
"""timing-only variants of sdf_decode_mx (experiment, not product): builds tools/dbg/libdisn_<name>.so with the
per-segment counters of tools/seg_patch_mx.py plus one thing removed, so that the per-stage cost of that thing can be
read off tools/seg_probe_mx.py (results of these variants are WRONG on purpose).  Run here: python tools/exp_mx.py
then on the GPU box: SEG_LIB=tools/dbg/libdisn_<name>.so python tools/seg_probe_mx.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "disn_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "dbg")


def variant(name, s):
    def rep(a, b, count=1):
        nonlocal s
        assert s.count(a) >= 1, (name, a)
        s = s.replace(a, b) if count == 0 else s.replace(a, b, count)
    if name in ("base", "base_global"):
        pass
    elif name == "nodma":     # steady-state refill pieces not issued
        rep("  constexpr int J = (U + 16) & 31;\n  pipe.template dma_sub<J>();", "  constexpr int J = (U + 16) & 31;")
    elif name == "noread":    # fragment reads of the ring not issued
        rep("  ring_put<U & 7>(rg, src[0]);\n", "  (void)src;\n")
    elif name == "nosync":    # no rendezvous in the middle of a stage
        rep("    __builtin_amdgcn_s_barrier();\n    long long tw2_", "    long long tw2_")
    elif name == "nocvt":     # operand conversions in the shadows dropped
        rep("          if constexpr (NEXT) mx_uops<7 * T + mx_q_start(M), MX_Q[M]>(nx, cs);\n", "")
        rep("        if constexpr (nb4 < 15) bf_uops<bf_q_start(P, M), BF_Q[P & 1][M]>(acc4[nn], cs5, yhi[nx], ylo[nx]);\n", "")
    elif name == "g_same":    # every lane gathers pixel row 0 (no misses, one line per instruction)
        i0 = s.find("float stream_pass_mx(")
        s = s[:i0] + s[i0:].replace("(size_t)cn.pix[c] * 512", "(size_t)0 * 512", 1)
    elif name == "g_hot":     # 32 distinct rows per instruction, always the same 32 (cache-resident)
        i0 = s.find("float stream_pass_mx(")
        s = s[:i0] + s[i0:].replace("(size_t)cn.pix[c] * 512", "(size_t)(lane & 31) * 512", 1)
    elif name.startswith("g_expr_"):   # gather row = expression of cn.pix[c] / lane, e.g. g_expr_mod64
        expr = {"mod64": "(size_t)(cn.pix[c] & 63) * 512", "mod4096": "(size_t)(cn.pix[c] & 4095) * 512",
                "rows8": "(size_t)((lane & 31) >> 2) * 512", "stride137": "(size_t)((lane & 31) * 137) * 512"}[name[7:]]
        i0 = s.find("float stream_pass_mx(")
        s = s[:i0] + s[i0:].replace("(size_t)cn.pix[c] * 512", expr, 1)
    elif name.startswith("mxq_"):   # another distribution of the split's ops over a group's six shadows
        rep("#define MX_CVT_Q 0, 0, 0, 0, 3, 4", "#define MX_CVT_Q " + ", ".join(name[4:]))
    elif name.startswith("bfq_"):   # ... of the bf16 split over a pair step's three shadows (even, odd pair)
        d = name[4:]
        rep("#define BF_CVT_Q {1, 2, 2}, {1, 1, 2}", "#define BF_CVT_Q {%s, %s, %s}, {%s, %s, %s}" % tuple(d))
    else:
        raise SystemExit("unknown variant " + name)
    return s


def main():
    names = sys.argv[1:] or ["base", "nodma", "noread", "nosync", "nocvt"]
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "sdf.hip")).read()
    os.environ["SEG_LOCAL"] = "1"
    env = {"s": src}
    exec(open(os.path.join(ROOT, "tools", "seg_patch_mx.py")).read(), env)
    patched = env["s"]
    procs = []
    for n in names:
        if n.endswith("_global"):
            os.environ["SEG_LOCAL"] = "0"
            env = {"s": src}
            exec(open(os.path.join(ROOT, "tools", "seg_patch_mx.py")).read(), env)
            patched_n = env["s"]
        else:
            patched_n = patched
        d = os.path.join("/tmp", "exp_mx", n)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "sdf.hip"), "w").write(variant(n, patched_n))
        cmd = ("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -ffp-contract=off "
               "-I%s -c %s/sdf.hip -o %s/sdf.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s/libdisn_%s.so "
               "%s/abi.o %s/hostio.o %s/mc.o %s/sdf.o" % (CSRC, d, d, OUT, n, CSRC, CSRC, CSRC, d))
        procs.append((n, subprocess.Popen(cmd, shell=True)))
    for n, p in procs:
        print(n, "rc", p.wait())


main()
