"""z-slab partition of the (res+1)^3 grid across GPUs and the host stitch that
reproduces the single-GPU mesh exactly (ordering rules App. B.6).

The reference has no multi-GPU path (demo/demo.py:86 pins one GPU); this is the
sharding BASELINE.json's north_star asks for.  Slab g owns the voxel layers and
edge slots of planes [k0_g, k1_g); it evaluates one halo plane below and two
above so that (i) the slot ranks of the first plane of slab g+1 can be computed
locally and (ii) the rare underflow rule of App. B.9 sees every neighbouring
voxel.  No data-path collective is needed: the only exchange is three counts
per rank (all-gather), after which every rank emits final global indices.
"""
import numpy as np


def split_layers(nz, nslab):
    """Balanced split of the nz-1 voxel layers: returns [k0_0, k0_1, ..., nz-1]."""
    layers = nz - 1
    nslab = max(1, min(nslab, layers))
    return [(layers * g) // nslab for g in range(nslab + 1)]


def slab_ranges(nz, nslab):
    """[(lo, hi, own0, own1)]: slab g needs global planes [lo,hi); within that
    sub-grid it owns local planes [own0,own1)."""
    b = split_layers(nz, nslab)
    out = []
    for g in range(len(b) - 1):
        k0, k1 = b[g], b[g + 1]
        last = g == len(b) - 2
        own_hi = nz if last else k1           # the last slab also owns the top plane's x/y edges
        lo = max(0, k0 - 1)
        hi = min(nz, own_hi + 2)
        out.append((lo, hi, k0 - lo, own_hi - lo))
    return out


def slab_plan(nz, world, rank, halo="exchange"):
    """What rank `rank` of `world` does for an nz-plane grid: (lo, hi, own0, own1, core0, core1).
    [lo, hi) = global planes its marching cubes needs (slab_ranges), [core0, core1) = the planes it evaluates
    itself.  halo="exchange": only its own planes (the halo planes come from the neighbours' all-gather), which
    needs every rank to hold at least two planes of its own -- otherwise, and with halo="recompute", every rank
    evaluates its whole [lo, hi) range and no halo collective runs (the decision is the same on every rank)."""
    if world <= 1:
        return 0, nz, 0, nz, 0, nz
    lo, hi, own0, own1 = slab_ranges(nz, world)[rank]
    core0, core1 = lo, hi
    if halo == "exchange":
        b = split_layers(nz, world)
        if len(b) - 1 == world and min(b[g + 1] - b[g] for g in range(world)) >= 2:
            core0, core1 = b[rank], (nz if rank == world - 1 else b[rank + 1])
    return lo, hi, own0, own1, core0, core1


def slab_bases(counts):
    """counts[g] = (n_centre, n_edge, n_tri) per slab (in slab order).  Returns
    [(centre_base, edge_base)]: centre vertices come first globally in
    DESCENDING k, i.e. slabs in reverse order; edge vertices follow in slab order."""
    c_total = sum(c[0] for c in counts)
    bases = []
    e_run = 0
    for g, c in enumerate(counts):
        centre_base = sum(cc[0] for cc in counts[g + 1:])
        bases.append((centre_base, c_total + e_run))
        e_run += c[1]
    return bases


def stitch_slabs(ctx, counts, outs):
    """outs[g] = (verts, tris, keys) numpy arrays emitted with slab_bases() offsets
    (tris unsorted or sorted).  Returns the global (verts, tris, keys)."""
    import torch
    centres_v = [o[0][:c[0]] for o, c in zip(outs, counts)][::-1]
    centres_k = [o[2][:c[0]] for o, c in zip(outs, counts)][::-1]
    edges_v = [o[0][c[0]:] for o, c in zip(outs, counts)]
    edges_k = [o[2][c[0]:] for o, c in zip(outs, counts)]
    verts = np.concatenate(centres_v + edges_v)
    keys = np.concatenate(centres_k + edges_k)
    tris = np.concatenate([o[1] for o in outs]).astype(np.int32)
    if len(tris) and ctx is None:      # host-only stitch (CPU tests)
        tris = tris[np.lexsort((tris[:, 2], tris[:, 1], tris[:, 0]))]
    elif len(tris):
        t = torch.from_numpy(np.ascontiguousarray(tris)).to(ctx.device)
        ctx.sort_triangles(t, max(len(verts), 1))
        torch.cuda.synchronize()
        tris = t.cpu().numpy()
    return verts, tris, keys
