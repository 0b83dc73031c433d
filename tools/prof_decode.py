"""Short driver for profiling: N launches of the res-256 SDF grid (decoder only)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disn_amd import synth
from disn_amd.api import Context
from disn_amd.encoder import Vgg16Encoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = Context(0)
ctx.set_decoder(sys.argv[3] if len(sys.argv) > 3 else "mixed4")   # pinned: a profile holds launches of one arithmetic only
ctx.set_weights(synth.decoder_weights(0))
g, f = Vgg16Encoder(ctx, synth.vgg_weights(0))(synth.image(0))
ctx.set_image(f, g)
out = None
for _ in range(n):
    out = ctx.sdf_grid(synth.GT_TRANS_MAT, res, out=out)
torch.cuda.synchronize()
print("done", float(out.mean()))
