"""Development check of the matrix-product image term (blend_mfma): dense grids through the mixed decoders against the
bf16x3 decoder (which gathers) and, on a small grid, against the float64 oracle.  GPU only."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from disn_amd import synth
from disn_amd.api import Context
from oracle import net

ctx = Context(0)
maps, gf = net.synthetic_features(seed=3)
w = synth.decoder_weights(seed=0)
ctx.set_weights(w)
ctx.set_image(maps, gf)
for res in (16, 24, 32, 64, 128, 256):
    grids = {}
    for dec in ("bf16x3", "mixed", "mixed4"):
        ctx.set_decoder(dec)
        grids[dec] = ctx.sdf_grid(synth.GT_TRANS_MAT, res).cpu().numpy()
    torch.cuda.synchronize()
    line = "sdf_res %3d:" % res
    for dec in ("mixed", "mixed4"):
        d = np.abs(grids[dec] - grids["bf16x3"])
        line += "  %s vs bf16x3 max %.3e nan %d" % (dec, d.max(), int(np.isnan(grids[dec]).sum()))
    if res <= 24:
        ref = net.sdf_grid(res, synth.GT_TRANS_MAT, maps, gf, w, dtype=torch.float64).reshape(res + 1, res + 1, res + 1)
        for dec in grids:
            line += "  %s vs oracle %.3e" % (dec, np.abs(grids[dec] - ref).max())
    print(line, flush=True)
for box in ((-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), (-0.3, -0.2, -0.4, 0.5, 0.9, 0.1)):
    g = {}
    for dec in ("bf16x3", "mixed4"):
        ctx.set_decoder(dec)
        g[dec] = ctx.sdf_grid(synth.GT_TRANS_MAT, 96, sdf_params=box).cpu().numpy()
    print("box", box, "mixed4 vs bf16x3 max %.3e" % np.abs(g["mixed4"] - g["bf16x3"]).max(), flush=True)
ctx.set_decoder("mixed4")
print("selfcheck mixed4:", ctx.selfcheck(synth.GT_TRANS_MAT), flush=True)
ctx.close()
