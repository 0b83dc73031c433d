"""per-segment shader cycles of sdf_decode_mx<*, true> (needs tools/seg_patch_mx4.py applied; experiment only)"""
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
ctx.set_decoder("mixed4")
names = ["between passes (point, store)", "image term / bias", "fold1/conv1 + split", "fold1/conv2 (2 stages)", "fold1/conv3 first 2 stages",
         "chain steps 0..6 (42 stages)", "chain step 7 (4 stages)", "fold2/conv2 (16 stages)", "last layer"]
stages = [0, 0, 0, 2, 2, 42, 4, 16, 0]
for res in (int(a) for a in (sys.argv[1:] or ["256"])):
    sdf = ctx.sdf_grid(synth.GT_TRANS_MAT, res)
    torch.cuda.synchronize()
    v = sdf.view(-1)[:26].view(torch.int64).cpu().tolist()
    wall = v[12]; v = v[:9]
    tot = sum(v)
    ntiles = -(-((res + 1) ** 3) // 128)
    tiles = max(1, round(ntiles / 256))
    print("sdf_res %d: total cycles %d of one wave, wall %.2f ms -> %.3f GHz; ~%d tiles" % (res, tot, wall * 1e-5, tot / (wall * 1e-8) / 1e9, tiles))
    for n, c, st in zip(names, v, stages):
        per = c / tiles
        print("  %-32s %8.0f cyc/tile  %5.1f%%  %s" % (n, per, 100.0 * c / tot, ("%.0f cyc/stage (MFMA issue time 1024)" % (per / st)) if st else ""))
