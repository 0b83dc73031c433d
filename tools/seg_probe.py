"""per-segment shader cycles of sdf_decode (needs the seg patch of the kernel; experiment only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disn_amd import synth
from disn_amd.api import Context
from disn_amd.encoder import Vgg16Encoder
ctx = Context(0)
ctx.set_weights(synth.decoder_weights(seed=0))
ctx.set_decoder("bf16x3")
enc = Vgg16Encoder(ctx, synth.vgg_weights(seed=0))
g, f = enc(synth.image(seed=0))
ctx.set_image(f, g)
names = ["start local (x2)", "L2 2 stages+cvt", "L3 first+cvt", "L3 stage x15", "start global (x2)", "L4 x32", "L5 16 stages", "final"]
stages = [0, 2, 1, 15, 0, 32, 16, 0]
for it in range(2):
    sdf = ctx.sdf_grid(synth.GT_TRANS_MAT, 256)
    torch.cuda.synchronize()
    v = sdf.view(-1)[:18].view(torch.int64).cpu().tolist()
    wall = v[8]; v = v[:8]
    tiles = 518 * 2
    tot = sum(v)
    print("total cycles", tot, "wall %.2f ms -> %.3f GHz" % (wall * 1e-5, tot / (wall * 1e-8) / 1e9))
    for n, c, st in zip(names, v, stages):
        per = c / tiles
        print("%-24s %8.0f cyc/stream  %5.1f%%  %s" % (n, per, 100.0 * c / tot, ("%.0f cyc/stage (ideal 1536)" % (per / st)) if st else ""))
