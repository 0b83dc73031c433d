"""batch mode (sdf_res 128, synthetic views) with N lanes: wall per view; run under rocprofv3 --kernel-trace for the
per-kernel GPU time per view.  usage: batch_lanes.py [lanes] [views]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disn_amd import api, synth, create_sdf
from disn_amd.encoder import Vgg16Encoder
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 26
ctx = api.Context(0)
w, vw = synth.decoder_weights(0), synth.vgg_weights(0)
ctx.set_weights(w)
enc = Vgg16Encoder(ctx, vw)
extra = [create_sdf.second_lane(ctx, w, vw) for _ in range(lanes - 1)]
with tempfile.TemporaryDirectory() as td:
    bm = create_sdf.BatchMesher(ctx, enc, 128, 0.0, td, median_iso=True, extra_lanes=extra)
    views = list(create_sdf.synthetic_views(n + 2 * lanes))
    for v in views[:2 * lanes]:
        bm.submit(v)
    bm.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for v in views[2 * lanes:]:
        bm.submit(v)
    bm.drain(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bm.finish()
print("lanes %d: %.2f ms per view (%d views)" % (lanes, dt / n * 1e3, n))
