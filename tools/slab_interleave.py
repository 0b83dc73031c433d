"""experiment: does alternating the two decoder streams more finely (z-slabs, each global launch + local launch) change
the time of a sdf_res-256 grid?  (the global-stream launch is power-limited, the local one is not)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disn_amd import synth
from disn_amd.api import Context
from disn_amd.encoder import Vgg16Encoder
res = 256
ctx = Context(0)
ctx.set_decoder("mixed4")
ctx.set_weights(synth.decoder_weights(seed=0))
g, f = Vgg16Encoder(ctx, synth.vgg_weights(seed=0))(synth.image(seed=0))
ctx.set_image(f, g)
out = torch.empty((res + 1, res + 1, res + 1), dtype=torch.float32, device=ctx.device)
ref = ctx.sdf_grid(synth.GT_TRANS_MAT, res).clone()
for rnd in range(2):
    for planes in (257, 128, 64, 32, 16, 8):
        bounds = list(range(0, res + 1, planes))
        if res + 1 - bounds[-1] < 4 and len(bounds) > 1:
            bounds.pop()
        bounds.append(res + 1)
        def run():
            for a, b in zip(bounds, bounds[1:]):
                ctx.sdf_grid(synth.GT_TRANS_MAT, res, z0=a, z1=b - 1, out=out[a:b])
        run(); torch.cuda.synchronize()
        assert torch.equal(out, ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            run()
        e1.record(); torch.cuda.synchronize()
        print("round %d: %3d planes per slab (%2d slabs): %.2f ms per grid" % (rnd, planes, len(bounds) - 1, e0.elapsed_time(e1) / 4), flush=True)
