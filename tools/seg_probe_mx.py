"""NOTE: goes with tools/seg_patch_mx.py (round-2 kernels); current kernels: tools/seg_probe_mx4.py."""
"""per-segment shader cycles of sdf_decode_mx (needs tools/seg_patch_mx.py applied; experiment only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import disn_amd._lib as _lib
if os.environ.get('SEG_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SEG_LIB'])
    print('variant', _lib.LIB_PATH)
from disn_amd import synth
from disn_amd.api import Context
from disn_amd.encoder import Vgg16Encoder
ctx = Context(0)
ctx.set_weights(synth.decoder_weights(seed=0))
enc = Vgg16Encoder(ctx, synth.vgg_weights(seed=0))
g, f = enc(synth.image(seed=0))
ctx.set_image(f, g)
names = ["start local", "f1c2 2 stages+split", "f1c3 first 2 st+split", "f1c3 14 stages", "start global", "f2c1 32 stages", "f2c2 16 stages", "final"]
stages = [0, 2, 2, 14, 0, 32, 16, 0]
ideal = [0, 1024, 1024, 1024, 0, 1024, 1536, 0]
for it in range(1):
    sdf = ctx.sdf_grid(synth.GT_TRANS_MAT, 256)
    torch.cuda.synchronize()
    v = sdf.view(-1)[:22].view(torch.int64).cpu().tolist()
    wall, tw, tb = v[8], v[9], v[10]; v = v[:8]
    print("mid-stage waits per tile: vmcnt %.0f cycles, barrier %.0f cycles (66 rendezvous per tile and stream)" % (tw / 518, tb / 518))
    tiles = 518
    tot = sum(v)
    print("total cycles", tot, "wall %.2f ms -> %.3f GHz" % (wall * 1e-5, tot / (wall * 1e-8) / 1e9))
    for n, c, st, idl in zip(names, v, stages, ideal):
        per = c / tiles
        print("%-24s %8.0f cyc/stream  %5.1f%%  %s" % (n, per, 100.0 * c / tot, ("%.0f cyc/stage (ideal %d)" % (per / st, idl)) if st else ""))
