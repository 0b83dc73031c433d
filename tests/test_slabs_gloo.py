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


def _exchange_worker(rank, world, port, tmp):
    """pipeline._Comm.exchange in its non-blocking form (the one RCCL runs: one batch of isend / irecv into the final
    row ranges), here over gloo with host tensors."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disn_amd.pipeline import _Comm
    comm = _Comm(dist)
    comm.host = False                      # tensors are already on the host: take the batched path
    sizes = [(3 + r, 5 * r, 2 * r + 1) for r in range(world)]          # (centres, edges, triangles) per rank
    nc, ne, nt = sizes[rank]
    verts = torch.arange((nc + ne) * 3, dtype=torch.float64).reshape(-1, 3) + 1000.0 * rank
    tris = torch.arange(nt * 3, dtype=torch.int32).reshape(-1, 3) + 100 * rank
    if rank != 0:
        comm.exchange([(verts[:nc], 0), (verts[nc:], 0), (tris, 0)], [])
    else:
        c_total = sum(s[0] for s in sizes)
        gv = torch.full((c_total + sum(s[1] for s in sizes), 3), -1.0, dtype=torch.float64)
        gt = torch.full((sum(s[2] for s in sizes), 3), -1, dtype=torch.int32)
        recvs, c_off, e_off, t_off, want_v, want_t = [], 0, c_total, 0, gv.clone(), gt.clone()
        for r in range(world):
            rc, re_, rt = sizes[r]
            v_r = torch.arange((rc + re_) * 3, dtype=torch.float64).reshape(-1, 3) + 1000.0 * r
            t_r = torch.arange(rt * 3, dtype=torch.int32).reshape(-1, 3) + 100 * r
            want_v[c_off:c_off + rc], want_v[e_off:e_off + re_], want_t[t_off:t_off + rt] = v_r[:rc], v_r[rc:], t_r
            if r == 0:
                gv[c_off:c_off + rc], gv[e_off:e_off + re_], gt[t_off:t_off + rt] = verts[:rc], verts[rc:], tris
            else:
                recvs += [(gv[c_off:c_off + rc], r), (gv[e_off:e_off + re_], r), (gt[t_off:t_off + rt], r)]
            c_off, e_off, t_off = c_off + rc, e_off + re_, t_off + rt
        comm.exchange([], recvs)
        open(tmp, "w").write("ok" if torch.equal(gv, want_v) and torch.equal(gt, want_t) else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_batched_mesh_gather_lands_in_place(world, tmp_path):
    port = 31500 + os.getpid() % 2000 + world
    out = str(tmp_path / "result.txt")
    mp.spawn(_exchange_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


@pytest.mark.parametrize("nslab", [2, 3, 8])
def test_sorted_slab_lists_only_need_placing(nslab):
    """The multi-GPU stitch without the rank-0 sort (pipeline.triangle_segments): every slab sorts its own triangles
    (global indices); the centre-first runs in reverse slab order followed by the edge-first runs in slab order ARE
    the single-pass list.  CPU restatement on a noise field full of centre vertices."""
    from disn_amd.pipeline import triangle_segments
    from disn_amd.slabs import slab_bases, slab_ranges
    from oracle import mc
    g = fields.noise_field((26, 9, 10), seed=5)
    ref = mc.marching_cubes(g, iso=0.02)
    assert ref["n_centre"] > 10
    slabs = [mc.marching_cubes_slab(g, lo, hi, o0, o1, iso=0.02) for lo, hi, o0, o1 in slab_ranges(g.shape[0], nslab)]
    counts = [tuple(int(c) for c in s.counts) for s in slabs]
    c_total = sum(c[0] for c in counts)
    lists, ncf = [], []
    for s, (cb, eb) in zip(slabs, slab_bases(counts)):
        _, t, _ = s.emit(cb, eb)
        t = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))] if len(t) else t
        lists.append(t)
        ncf.append(int((t[:, 0] < c_total).sum()) if len(t) else 0)
    out = np.full((sum(c[2] for c in counts), 3), -1, np.int32)
    for t, f, (cfo, efo) in zip(lists, ncf, triangle_segments(counts, ncf)):
        out[cfo:cfo + f] = t[:f]
        out[efo:efo + len(t) - f] = t[f:]
    assert np.array_equal(out, ref["tris"])
    assert sum(ncf) > 10


def test_slab_plan_world_8_and_thin_slabs():
    """Rehearsal of the 8-rank configurations the driver launches (bench.py --gpus 8: sdf_res 512; demo at 1024) and of
    grids too small for them: with halo exchange every rank evaluates exactly its own planes and together they cover
    the grid once; when a rank would own fewer than two planes ALL ranks fall back to evaluating their halo too."""
    from disn_amd.slabs import slab_plan
    for res in (512, 1024, 64, 16):
        nz = res + 1
        plans = [slab_plan(nz, 8, r) for r in range(8)]
        cores = [(p[4], p[5]) for p in plans]
        assert cores[0][0] == 0 and cores[-1][1] == nz
        assert all(a[1] == b[0] for a, b in zip(cores, cores[1:]))            # evaluated exactly once
        for lo, hi, o0, o1, c0, c1 in plans:
            assert lo <= c0 < c1 <= hi and 0 <= o0 < o1 <= hi - lo
            assert hi - c1 <= 2 and c0 - lo <= 1                              # halo: one plane below, two above
    for res in (8, 12, 15):                                                   # some rank owns a single plane
        plans = [slab_plan(res + 1, 8, r) for r in range(8)]
        assert all((p[4], p[5]) == (p[0], p[1]) for p in plans)               # everyone recomputes its halo
    assert all(slab_plan(41, 8, r, halo="recompute")[4:] == slab_plan(41, 8, r, halo="recompute")[:2] for r in range(8))
    assert slab_plan(33, 1, 0) == (0, 33, 0, 33, 0, 33)


def _empty_exchange_worker(rank, world, port, tmp):
    """A slab whose mesh is empty (no surface in its planes) sends nothing; the others' pieces still land."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from disn_amd.pipeline import _Comm, triangle_segments
    comm = _Comm(dist)
    comm.host = False
    counts = [(0, 0, 0) if r in (1, world - 2) else (r + 1, 2 + r, 3 + r) for r in range(world)]   # two empty slabs
    ncf = [min(c[0], c[2]) for c in counts]
    seg = triangle_segments(counts, ncf)
    nt = counts[rank][2]
    tris = torch.full((nt, 3), rank, dtype=torch.int32)
    if rank != 0:
        comm.exchange([(tris[:ncf[rank]], 0), (tris[ncf[rank]:], 0)], [])
    else:
        gt = torch.full((sum(c[2] for c in counts), 3), -1, dtype=torch.int32)
        recvs = []
        for r in range(world):
            cfo, efo = seg[r]
            if r == 0:
                gt[cfo:cfo + ncf[0]] = tris[:ncf[0]]
                gt[efo:efo + nt - ncf[0]] = tris[ncf[0]:]
            else:
                recvs += [(gt[cfo:cfo + ncf[r]], r), (gt[efo:efo + counts[r][2] - ncf[r]], r)]
        comm.exchange([], recvs)
        want = [r for r in reversed(range(world)) for _ in range(ncf[r])] + \
               [r for r in range(world) for _ in range(counts[r][2] - ncf[r])]
        ok = gt[:, 0].tolist() == want
        open(tmp, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 8])
def test_exchange_with_empty_slabs(world, tmp_path):
    port = 33500 + os.getpid() % 2000 + world
    out = str(tmp_path / "result.txt")
    mp.spawn(_empty_exchange_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_seam_check():
    from disn_amd.pipeline import seams_in_order
    gt = torch.tensor([[0, 5, 6], [0, 7, 1], [2, 3, 4], [2, 3, 5], [9, 9, 9]], dtype=torch.int32)
    assert seams_in_order(gt, [(0, 2), (2, 2), (4, 1)])
    assert seams_in_order(gt, [(2, 2), (0, 2), (4, 0), (4, 1)])          # any listing order, empty pieces skipped
    assert seams_in_order(gt, [(0, 5)]) and seams_in_order(gt, [])
    bad = gt.clone()
    bad[2] = torch.tensor([0, 6, 9])                                       # first row of the second piece below the seam
    assert not seams_in_order(bad, [(0, 2), (2, 2), (4, 1)])
    dup = gt.clone()
    dup[2] = dup[1]                                                        # equal triangles across a seam never occur
    assert not seams_in_order(dup, [(0, 2), (2, 2), (4, 1)])
