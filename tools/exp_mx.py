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
    if name == "base":
        pass
    elif name == "nodma":     # steady-state refill pieces not issued
        rep("  constexpr int J = (U + 16) & 31;\n  pipe.template dma_sub<J>();", "  constexpr int J = (U + 16) & 31;")
    elif name == "noread":    # fragment reads of the ring not issued
        rep("  ring_put<U & 7>(rg, src[0]);\n", "  (void)src;\n")
    elif name == "nosync":    # no rendezvous in the middle of a stage
        rep("    __builtin_amdgcn_s_barrier();\n    long long tw2_", "    long long tw2_")
    elif name == "nocvt":     # operand conversions in the shadows dropped
        rep("          if constexpr (NEXT && M == 5 && (T & 1) == 0) cvt_mx_a<(T >> 1)>(nx, cs);\n", "")
        rep("          if constexpr (NEXT && M == 5 && (T & 1) == 1) cvt_mx_b<(T >> 1)>(nx, cs);\n", "")
        rep("        if constexpr (nb4 < 15 && decltype(mm)::value == 2) cvt_part<P>(acc4[nn], cs5, yhi[nx], ylo[nx]);\n", "")
    else:
        raise SystemExit("unknown variant " + name)
    return s


def main():
    names = sys.argv[1:] or ["base", "nodma", "noread", "nosync", "nocvt"]
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "sdf.hip")).read()
    env = {"s": src}
    exec(open(os.path.join(ROOT, "tools", "seg_patch_mx.py")).read(), env)
    patched = env["s"]
    procs = []
    for n in names:
        d = os.path.join("/tmp", "exp_mx", n)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "sdf.hip"), "w").write(variant(n, patched))
        cmd = ("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -ffp-contract=off "
               "-I%s -c %s/sdf.hip -o %s/sdf.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s/libdisn_%s.so "
               "%s/abi.o %s/hostio.o %s/mc.o %s/sdf.o" % (CSRC, d, d, OUT, n, CSRC, CSRC, CSRC, d))
        procs.append((n, subprocess.Popen(cmd, shell=True)))
    for n, p in procs:
        print(n, "rc", p.wait())


main()
