"""Multi-rank path on CPU (gloo, world_size 2 and 3): slab ranges, the count all-gather,
global index bases and the rank-0 stitch, driven by the CPU slab restatement
(oracle/mc.py::marching_cubes_slab).  The stitched mesh must equal the single pass."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disn_amd.slabs import slab_bases, slab_ranges, stitch_slabs
    from oracle import mc
    g = fields.noise_field((19, 11, 12), seed=21)
    bbox = (-1, -1, -1, 1, 1, 1.5)
    lo, hi, own0, own1 = slab_ranges(g.shape[0], world)[rank]
    slab = mc.marching_cubes_slab(g, lo, hi, own0, own1, iso=0.03, bbox=bbox)
    mine = torch.tensor(slab.counts, dtype=torch.int64)
    allc = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    allc = [tuple(int(v) for v in c) for c in allc]
    cb, eb = slab_bases(allc)[rank]
    v, t, k = slab.emit(cb, eb)
    if rank != 0:
        for a in (v, t, k):
            dist.send(torch.from_numpy(np.ascontiguousarray(a)), dst=0)
    else:
        outs = [(v, t, k)]
        for r in range(1, world):
            nc, ne, nt = allc[r]
            bv = torch.empty((nc + ne, 3), dtype=torch.float64)
            bt = torch.empty((nt, 3), dtype=torch.int32)
            bk = torch.empty((nc + ne,), dtype=torch.int64)
            for b in (bv, bt, bk):
                dist.recv(b, src=r)
            outs.append((bv.numpy(), bt.numpy(), bk.numpy()))
        verts, tris, keys = stitch_slabs(None, allc, outs)
        ref = mc.marching_cubes(g, iso=0.03, bbox=bbox)
        ok = (np.array_equal(tris, ref["tris"]) and np.array_equal(keys, ref["keys"])
              and float(np.abs(verts - ref["verts"]).max()) <= 1e-12 and ref["n_centre"] > 0)
        open(tmp, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_exchange_and_stitch_gloo(world, tmp_path):
    port = 29500 + os.getpid() % 2000 + world
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_slab_ranges_cover_and_halo():
    from disn_amd.slabs import slab_ranges, split_layers
    for nz in (2, 5, 17, 257):
        for n in (1, 2, 3, 8):
            rs = slab_ranges(nz, n)
            own = [(lo + o0, lo + o1) for lo, hi, o0, o1 in rs]
            assert own[0][0] == 0 and own[-1][1] == nz
            assert all(a[1] == b[0] for a, b in zip(own, own[1:]))          # owned planes partition the grid
            for (lo, hi, o0, o1), (g0, g1) in zip(rs, own):
                assert lo <= max(0, g0 - 1) and hi >= min(nz, g1 + 2)      # one plane below, two above
            assert split_layers(nz, n)[-1] == nz - 1
