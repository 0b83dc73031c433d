"""A/B builds of the decoder with another distribution of the operand-split ops over the MFMA shadows
(-DMX5_CVT_Q=... / -DMX_CVT_Q=... / -DBF_CVT_Q=...): python tools/exp_q.py mx5_333333 mx5_112266 ...
-> tools/dbg/libdisn_<name>.so; time them on the GPU box with SEG_LIB=tools/dbg/libdisn_<name>.so python tools/time_lib.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "disn_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "dbg")
os.makedirs(OUT, exist_ok=True)
procs = []
for n in sys.argv[1:]:
    kind, digits = n.split("_")
    macro = {"mx5": "MX5_CVT_Q", "mx": "MX_CVT_Q", "dma": "MX_DMA_GAPS"}[kind]
    d = os.path.join("/tmp", "exp_q", n)
    os.makedirs(d, exist_ok=True)
    objs = " ".join(os.path.join(CSRC, o) for o in ("abi.o", "hostio.o", "mc.o", "vgg.o"))
    cmd = ("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -ffp-contract=off "
           "'-D%s=%s' -c %s/sdf.hip -o %s/sdf.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o %s/libdisn_%s.so "
           "%s %s/sdf.o" % (macro, ",".join(digits), CSRC, d, OUT, n, objs, d))
    procs.append((n, subprocess.Popen(cmd, shell=True)))
for n, p in procs:
    print(n, "rc", p.wait())
