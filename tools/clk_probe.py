"""effective shader clock during sdf_decode (needs the clk patch of the kernel; experiment only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disn_amd import synth
from disn_amd.api import Context
from disn_amd.encoder import Vgg16Encoder
ctx = Context(0)
ctx.set_weights(synth.decoder_weights(seed=0))
enc = Vgg16Encoder(ctx, synth.vgg_weights(seed=0))
g, f = enc(synth.image(seed=0))
ctx.set_image(f, g)
for it in range(4):
    sdf = ctx.sdf_grid(synth.GT_TRANS_MAT, 256)
    torch.cuda.synchronize()
    v = sdf.view(-1)[:4].view(torch.int64).cpu().tolist()
    print("iter", it, "cycles", v[0], "wall100MHz", v[1], "-> shader clock %.3f GHz, %.2f ms" % (v[0] / (v[1] * 10e-9) / 1e9, v[1] * 10e-6))
