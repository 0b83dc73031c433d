"""Image features -> SDF grid -> mesh, on one GPU or z-slab sharded over the
ranks of a torch.distributed group (one process per GPU; backend "nccl" is RCCL).

Replaces the body of the reference's test_one_epoch + create_obj
(reference: demo/demo.py:281-369) without the host round trips, the .dist file
hop and the subprocess.  Multi-GPU exchange: one broadcast of
the image features, one all-gather of three counts per rank; then every rank
emits its slab with final global vertex indices, ORDERS ITS OWN TRIANGLES on its
own GPU, and sends the sorted pieces to rank 0, which only places them: the
global lexicographic order is the centre-vertex-first runs of the slabs in
reverse slab order followed by the edge-vertex-first runs in slab order
(triangle_segments).  Rank 0 checks the seams between the pieces and re-sorts
only if one is out of order.  There is no all-reduce anywhere.
"""
import numpy as np
import torch

from .slabs import slab_bases, slab_plan, slab_ranges


class _Comm:
    """torch.distributed with device tensors.  Backend "nccl" (RCCL) moves them directly; with
    "gloo" (CPU rehearsals and tests: several ranks on one GPU) they are staged through the host."""

    def __init__(self, dist):
        self.dist = dist
        self.host = dist is not None and dist.get_backend() == "gloo"

    def broadcast(self, t, src):
        if not self.host:
            return self.dist.broadcast(t, src=src)
        c = t.cpu()
        self.dist.broadcast(c, src=src)
        t.copy_(c)

    def all_gather(self, out_list, t):
        if not self.host:
            return self.dist.all_gather(out_list, t)
        outs = [torch.empty_like(t, device="cpu") for _ in out_list]
        self.dist.all_gather(outs, t.cpu())
        for o, c in zip(out_list, outs):
            o.copy_(c)

    def send(self, t, dst):
        self.dist.send(t.cpu() if self.host else t, dst=dst)

    def recv(self, t, src):
        if not self.host:
            return self.dist.recv(t, src=src)
        c = torch.empty_like(t, device="cpu")
        self.dist.recv(c, src=src)
        t.copy_(c)

    def exchange(self, sends, recvs):
        """sends = [(tensor, dst)], recvs = [(tensor, src)] with contiguous device tensors (empty ones skipped).
        RCCL: one batch of non-blocking point-to-point operations, so a rank's pieces from all peers travel at the
        same time over their own xGMI links and land in their final place; gloo (rehearsals): one after the other."""
        sends = [(t, d) for t, d in sends if t.numel()]
        recvs = [(t, r) for t, r in recvs if t.numel()]
        if self.host:
            for t, d in sends:
                self.send(t, d)
            for t, r in recvs:
                self.recv(t, r)
            return
        ops = [self.dist.P2POp(self.dist.isend, t, d) for t, d in sends]
        ops += [self.dist.P2POp(self.dist.irecv, t, r) for t, r in recvs]
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()


def triangle_segments(counts, n_centre_first):
    """Where the sorted triangle lists of the slabs go in the global list.  counts[g] = (n_centre, n_edge, n_tri),
    n_centre_first[g] = how many of slab g's triangles start with a centre vertex (they lead its sorted list,
    centre indices being smaller than every edge index).  Global order (App. B.6 with the index bases of
    slab_bases): centre vertices are numbered slab G-1 first, so the centre-first runs come in REVERSE slab order,
    then the edge-first runs in slab order.  Returns [(centre_first_offset, edge_first_offset)] per slab."""
    world = len(counts)
    off, cf_off = 0, [0] * world
    for g in reversed(range(world)):
        cf_off[g] = off
        off += n_centre_first[g]
    ef_off = [0] * world
    for g in range(world):
        ef_off[g] = off
        off += counts[g][2] - n_centre_first[g]
    return list(zip(cf_off, ef_off))


def seams_in_order(gt, pieces):
    """pieces = [(offset, length)] of sorted runs laid out in gt (device [T,3] int32), in global order: True iff the
    last triangle of every non-empty piece is lexicographically below the first triangle of the next one."""
    runs = sorted((o, n) for o, n in pieces if n > 0)
    if len(runs) < 2:
        return True
    idx = []
    for (o, n), (o2, _) in zip(runs, runs[1:]):
        idx += [o + n - 1, o2]
    rows = gt[torch.tensor(idx, device=gt.device)].cpu().tolist()
    return all(tuple(a) < tuple(b) for a, b in zip(rows[0::2], rows[1::2]))


class MeshPipeline:
    def __init__(self, ctx, rank=0, world=1, dist=None, halo="exchange"):
        """halo: how a slab gets its one halo plane below and two above --
        "exchange": every rank evaluates only its own planes and one all-gather of three boundary planes
        per rank (3 MB per rank at sdf_res 512) fills the halos, bit-identical because the same kernel
        produced them; "recompute": every rank evaluates its halo planes itself (no collective, +4.7 %
        points per GPU with 8 slabs at sdf_res 512)."""
        self.ctx, self.rank, self.world = ctx, rank, world
        self.halo = halo
        self.last_points = 0        # grid points this rank evaluated in the last run
        self.dist = _Comm(dist) if world > 1 else None   # torch.distributed when world > 1
        self.decode_events = []     # (start, end) torch events around the decoder launch
        self.image_events = []      # ... around the per-image precompute (resize, projected map)
        self.mc_events = []         # ... around marching cubes (count + emit + sort), single GPU
        self.time_decode = False
        self.last_sdf = None
        self.last_stitch = None     # multi-GPU, rank 0: "placed" (sorted pieces only placed) or "resorted" (a seam was out of order)
        self._agreed_weights = None

    @staticmethod
    def _mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _broadcast_features(self, feats, gfeat):
        if self.world == 1:
            return feats, gfeat
        for t in list(feats) + [gfeat]:
            self.dist.broadcast(t, src=0)
        return feats, gfeat

    def run(self, feats, gfeat, trans_mat, res, iso=0.0, sdf_params=(-1, -1, -1, 1, 1, 1), want_mesh=True, image_set=False):
        """feats: five device NHWC tensors (valid on rank 0, receive buffers elsewhere);
        returns (verts[V,3] f64, tris[T,3] i32) device tensors on rank 0, (None, None) elsewhere.
        image_set=True (single GPU): the caller has already done ctx.set_image(feats, gfeat) for this image."""
        ctx = self.ctx
        nz = res + 1
        if self.world > res:
            raise ValueError("z-slab sharding needs at least one voxel layer per rank: world=%d > sdf_res=%d"
                             % (self.world, res))
        feats, gfeat = self._broadcast_features(feats, gfeat)
        ev = (lambda: self._mark()) if self.time_decode else (lambda: None)
        i0 = ev()
        if not (image_set and self.world == 1):
            ctx.set_image(feats, gfeat)
        if self.world > 1 and self._agreed_weights != ctx.weights_version:
            # once per weight set (set_weights is called by every rank alike, so every rank gets here together):
            # each rank measures the same broadcast features with the same kernels, and rank 0's choice of decoder
            # arithmetic is taken by all, so that slab seams can never meet two arithmetics
            if ctx.auto_decoder and not ctx._tuned:
                ctx.autotune(trans_mat, sdf_params)
            pick = torch.tensor([ctx.DECODERS[ctx.decoder]], dtype=torch.int32, device=ctx.device)
            self.dist.broadcast(pick, src=0)
            name = ctx.DECODER_NAMES[int(pick.item())]
            if name != ctx.decoder:
                auto = ctx.auto_decoder
                ctx.set_decoder(name)
                ctx.auto_decoder, ctx._tuned = auto, True
            self._agreed_weights = ctx.weights_version
        i1 = ev()
        if i0 is not None:
            self.image_events.append((i0, i1))
        lo, hi, own0, own1, core0, core1 = slab_plan(nz, self.world, self.rank, self.halo)   # core: planes evaluated here
        if self.time_decode:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if (core0, core1) == (lo, hi):
            sdf = ctx.sdf_grid(trans_mat, res, z0=lo, z1=hi - 1, sdf_params=sdf_params)
        else:
            sdf = torch.empty((hi - lo, nz, nz), dtype=torch.float32, device=ctx.device)
            ctx.sdf_grid(trans_mat, res, z0=core0, z1=core1 - 1, sdf_params=sdf_params, out=sdf[core0 - lo:core1 - lo])
        if self.time_decode:
            e1.record()
            self.decode_events.append((e0, e1))
        self.last_points = (core1 - core0) * nz * nz
        if (core0, core1) != (lo, hi):
            # halo planes: the last own plane of the rank below, the first two of the rank above
            core = sdf[core0 - lo:core1 - lo]
            edge = torch.stack([core[0], core[1], core[-1]])
            allp = [torch.empty_like(edge) for _ in range(self.world)]
            self.dist.all_gather(allp, edge)
            if lo < core0:
                sdf[0].copy_(allp[self.rank - 1][2])
            for t in range(hi - core1):
                sdf[core1 - lo + t].copy_(allp[self.rank + 1][t])
        self.last_sdf = sdf
        if not want_mesh:
            return None, None
        m0 = ev()
        counts = ctx.mc_count(sdf, iso, z_offset=lo, nz_global=nz, own=(own0, own1))
        if self.world == 1:
            mesh = ctx.mc_emit(sdf_params, sort=True)
            m1 = ev()
            if m0 is not None:
                self.mc_events.append((m0, m1))
            return mesh
        # ---- multi-GPU: exchange counts, emit with global indices, ship to rank 0
        mine = torch.tensor(counts, dtype=torch.int64, device=ctx.device)
        allc = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allc, mine)
        allc = [tuple(int(v) for v in c.tolist()) for c in allc]
        bases = slab_bases(allc)
        cb, eb = bases[self.rank]
        c_total = sum(c[0] for c in allc)
        v_total = c_total + sum(c[1] for c in allc)
        t_total = sum(c[2] for c in allc)
        nc0, ne0, nt0 = allc[self.rank]
        # every rank orders its own triangles (global indices) on its own GPU ...
        verts, tris = ctx.mc_emit(sdf_params, centre_base=cb, edge_base=eb, sort=True)
        ncf = torch.zeros(1, dtype=torch.int64, device=ctx.device)
        if nt0:
            ncf[0] = (tris[:, 0] < c_total).sum()
        allf = [torch.empty_like(ncf) for _ in range(self.world)]
        self.dist.all_gather(allf, ncf)
        allf = [int(f.item()) for f in allf]
        seg = triangle_segments(allc, allf)
        f0 = allf[self.rank]
        if self.rank != 0:
            # ... and ships four pieces, each contiguous at the receiver: centre vertices, edge vertices,
            # centre-first triangles, edge-first triangles
            self.dist.exchange([(verts[:nc0], 0), (verts[nc0:], 0), (tris[:f0], 0), (tris[f0:], 0)], [])
            return None, None
        gv = torch.empty((v_total, 3), dtype=torch.float64, device=ctx.device)
        gt = torch.empty((t_total, 3), dtype=torch.int32, device=ctx.device)
        recvs, pieces = [], []
        for r in range(self.world):
            nc, ne, nt = allc[r]
            cbr, ebr = bases[r]
            cfo, efo = seg[r]
            pieces += [(cfo, allf[r]), (efo, nt - allf[r])]
            if r == 0:
                gv[cbr:cbr + nc] = verts[:nc]          # centre vertices: slabs in reverse order
                gv[ebr:ebr + ne] = verts[nc:]          # edge vertices: slab order
                gt[cfo:cfo + f0] = tris[:f0]
                gt[efo:efo + nt - f0] = tris[f0:]
            else:
                recvs += [(gv[cbr:cbr + nc], r), (gv[ebr:ebr + ne], r), (gt[cfo:cfo + allf[r]], r),
                          (gt[efo:efo + nt - allf[r]], r)]
        self.dist.exchange([], recvs)
        self.last_stitch = "placed"
        if not seams_in_order(gt, pieces):             # not expected (see triangle_segments); one global sort repairs it
            ctx.sort_triangles(gt, max(v_total, 1))
            self.last_stitch = "resorted"
        return gv, gt


def median_iso(sdf):
    """iso that guarantees a surface for random-init weights."""
    return float(torch.median(sdf.reshape(-1)).item())


def coarse_median_iso(ctx, trans_mat, sdf_params=(-1, -1, -1, 1, 1, 1), res=31):
    """Median of the SDF over a coarse (res+1)^3 grid spanning the WHOLE box (32^3 points = 256 tiles of 128: one
    tile per compute unit, one round of the decoder's persistent workgroups): the iso used with random-init
    weights, which have no zero level set.  It needs the image
    to be set (ctx.set_image) and is the same number on every rank, whatever slab the rank owns."""
    return median_iso(ctx.sdf_grid(trans_mat, res, sdf_params=sdf_params))


def mesh_by_slabs(ctx, trans_mat, res, iso, nslabs, sdf_params=(-1, -1, -1, 1, 1, 1)):
    """The N-rank z-slab algorithm of MeshPipeline.run executed slab after slab on ONE GPU (image already set with
    ctx.set_image): every slab evaluates its own planes plus its halo, counts, and -- once all counts are known --
    emits with its global vertex bases; the pieces are placed and the triangles sorted once, exactly as rank 0
    does.  Use: grids too large to mesh in one piece next to other work, and the single-GPU check that the
    sharding reproduces the one-pass mesh.  Returns (verts f64 [V,3], tris i32 [T,3]) on the device."""
    nz = res + 1
    ranges = slab_ranges(nz, nslabs)
    grids, counts = [], []
    for lo, hi, own0, own1 in ranges:
        sdf = ctx.sdf_grid(trans_mat, res, z0=lo, z1=hi - 1, sdf_params=sdf_params)
        grids.append(sdf)
        counts.append(tuple(ctx.mc_count(sdf, iso, z_offset=lo, nz_global=nz, own=(own0, own1))))
    bases = slab_bases(counts)
    c_total = sum(c[0] for c in counts)
    v_total = c_total + sum(c[1] for c in counts)
    t_total = sum(c[2] for c in counts)
    gv = torch.empty((v_total, 3), dtype=torch.float64, device=ctx.device)
    parts = []
    for (lo, hi, own0, own1), sdf, (nc, ne, nt), (cb, eb) in zip(ranges, grids, counts, bases):
        ctx.mc_count(sdf, iso, z_offset=lo, nz_global=nz, own=(own0, own1))
        pv, pt = ctx.mc_emit(sdf_params, centre_base=cb, edge_base=eb, sort=True)   # each slab sorts its own list
        gv[cb:cb + nc] = pv[:nc]
        gv[eb:eb + ne] = pv[nc:]
        parts.append((pt, int((pt[:, 0] < c_total).sum().item()) if nt else 0))
    seg = triangle_segments(counts, [f for _, f in parts])
    gt = torch.empty((t_total, 3), dtype=torch.int32, device=ctx.device)
    pieces = []
    for (pt, f), (cfo, efo) in zip(parts, seg):
        gt[cfo:cfo + f] = pt[:f]
        gt[efo:efo + len(pt) - f] = pt[f:]
        pieces += [(cfo, f), (efo, len(pt) - f)]
    del parts
    info = {"stitch": "placed"}
    if not seams_in_order(gt, pieces):
        ctx.sort_triangles(gt, max(v_total, 1))
        info["stitch"] = "resorted"
    mesh_by_slabs.last_info = info
    return gv, gt
