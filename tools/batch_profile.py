"""where a batch-mode view spends its time (sdf_res 128, synthetic views)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from disn_amd import api, synth, create_sdf
from disn_amd.encoder import Vgg16Encoder
from disn_amd.pipeline import MeshPipeline, median_iso
ctx = api.Context(0)
ctx.set_weights(synth.decoder_weights(0))
enc = Vgg16Encoder(ctx, synth.vgg_weights(0))
pipe = MeshPipeline(ctx)
res = 128
T = {k: 0.0 for k in ("encoder", "grid_for_median", "median", "grid+mc", "d2h", "write_obj")}
def tick():
    torch.cuda.synchronize(); return time.perf_counter()
views = list(create_sdf.synthetic_views(12))
for it, view in enumerate(views):
    t0 = tick(); g, f = enc(view["img"]); t1 = tick()
    params = tuple(float(x) for x in view["sdf_params"])
    pipe.run(f, g, view["trans_mat"], res, sdf_params=params, want_mesh=False); t2 = tick()
    iso = median_iso(pipe.last_sdf); t3 = tick()
    v, t = pipe.run(f, g, view["trans_mat"], res, iso, params); t4 = tick()
    vn, tn = v.cpu().numpy(), t.cpu().numpy(); t5 = tick()
    api.write_obj("/tmp/x.obj", vn, tn); t6 = tick()
    if it >= 2:
        for k, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            T[k] += d
n = len(views) - 2
print({k: "%.2f ms" % (v / n * 1e3) for k, v in T.items()}, "faces", len(tn), "obj MB %.1f" % (os.path.getsize("/tmp/x.obj") / 1e6))
