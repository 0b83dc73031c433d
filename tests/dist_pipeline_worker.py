"""Worker of tests/test_gpu_pipeline.py: every rank runs its z-slab on GPU 0, rank 0 also runs the
single-rank pipeline and records whether the stitched mesh is identical."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disn_amd import synth  # noqa: E402
from disn_amd.api import Context  # noqa: E402
from disn_amd.pipeline import MeshPipeline, median_iso  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("DISN_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
ctx = Context(0)
ctx.set_weights(synth.decoder_weights(seed=2))
rng = np.random.default_rng(2)
shapes = [(224, 224, 64), (112, 112, 128), (56, 56, 256), (28, 28, 512), (14, 14, 512)]
if rank == 0:
    feats = [torch.from_numpy(np.maximum(rng.standard_normal(s).astype(np.float32), 0)).cuda() for s in shapes]
    gfeat = torch.from_numpy(rng.standard_normal(1024).astype(np.float32)).cuda()
else:   # receive buffers only: the broadcast must deliver the features
    feats = [torch.zeros(s, dtype=torch.float32, device="cuda") for s in shapes]
    gfeat = torch.zeros(1024, dtype=torch.float32, device="cuda")
res, bbox = 30, (-1, -1, -1, 1, 1, 1)
iso_t = torch.zeros(1, device="cuda")
if rank == 0:
    single = MeshPipeline(ctx)
    single.run(feats, gfeat, synth.GT_TRANS_MAT, res, want_mesh=False)
    iso_t[0] = median_iso(single.last_sdf)
    v1, t1 = single.run(feats, gfeat, synth.GT_TRANS_MAT, res, float(iso_t.item()), bbox)
c = iso_t.cpu()
dist.broadcast(c, src=0)
iso = float(c.item())
pipe = MeshPipeline(ctx, rank, world, dist, halo=os.environ.get("DISN_HALO", "exchange"))
v, t = pipe.run(feats, gfeat, synth.GT_TRANS_MAT, res, iso, bbox)
torch.cuda.synchronize()
if rank == 0:
    np.savez(sys.argv[1], tris_equal=torch.equal(t, t1), verts_equal=torch.equal(v, v1), n_tris=t.shape[0],
             points=pipe.last_points, planes=pipe.last_sdf.shape[0], stitch=pipe.last_stitch, decoder=ctx.decoder)
dist.barrier()
dist.destroy_process_group()
ctx.close()
