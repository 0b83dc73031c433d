"""time the res-256 SDF grid with the library given in SEG_LIB (default: the built one); for A/B runs of variants"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import disn_amd._lib as _lib
if os.environ.get("SEG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SEG_LIB"])
from disn_amd import synth
from disn_amd.api import Context
from disn_amd.encoder import Vgg16Encoder
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = Context(0)
ctx.set_weights(synth.decoder_weights(seed=0))
g, f = Vgg16Encoder(ctx, synth.vgg_weights(seed=0))(synth.image(seed=0))
ctx.set_image(f, g)
out = ctx.sdf_grid(synth.GT_TRANS_MAT, res)
torch.cuda.synchronize()
ts = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.sdf_grid(synth.GT_TRANS_MAT, res, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("%s res %d: %s ms, checksum %.6f" % (os.path.basename(_lib.LIB_PATH), res, " ".join("%.2f" % t for t in ts), float(out.double().sum())))
