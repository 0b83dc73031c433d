"""Marching cubes alone on the analytic 3-primitive field at sdf_res 256/512/1024:
wall time of count + emit(sorted) against the algorithmic bytes 4 B/point + 24 B/vertex + 12 B/triangle."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import disn_amd._lib as _lib
if os.environ.get('SEG_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SEG_LIB'])
from disn_amd.api import Context


def field(n):
    l = torch.linspace(-1, 1, n, device="cuda", dtype=torch.float64)
    grid = torch.empty((n, n, n), dtype=torch.float32, device="cuda")
    y, x = torch.meshgrid(l, l, indexing="ij")
    for k0 in range(0, n, 64):
        z = l[k0:k0 + 64][:, None, None]
        s1 = torch.sqrt((x - 0.2) ** 2 + y ** 2 + z ** 2) - 0.45
        s2 = torch.sqrt((x + 0.3) ** 2 + (y - 0.1) ** 2 + (z + 0.2) ** 2) - 0.35
        q = torch.sqrt(x * x + z * z) - 0.6
        t = torch.sqrt(q * q + (y + 0.3) ** 2) - 0.12
        grid[k0:k0 + 64] = torch.minimum(torch.minimum(s1, s2), t).float()
    return grid


ctx = Context(0)
for res in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
    g = field(res + 1)
    for it in range(3):
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ctx.mc_count(g, 0.0)
        e1.record()
        v, f = ctx.mc_emit(sort=True)
        e2.record()
        torch.cuda.synchronize()
    P = (res + 1) ** 3
    b = 4 * P + 24 * len(v) + 12 * len(f)
    ms = e0.elapsed_time(e2)
    print("res %4d  %9d v %9d t  count %.3f ms  emit+sort %.3f ms  total %.3f ms  %.1f MB algorithmic -> %.0f GB/s (%.1f %% of 8 TB/s)"
          % (res, len(v), len(f), e0.elapsed_time(e1), e1.elapsed_time(e2), ms, b / 1e6, b / ms / 1e6, b / ms / 1e6 / 80.0), flush=True)
    del g, v, f
