"""Distance field of a triangle mesh on a regular grid: the host side of the replacement for the reference's
ground-truth tool

    computeDistanceField <obj> res res res -s -e 1.2 -o out.dist -m 1 [-g sigma]

(closed Vega-FEM binary; call site preprocessing/create_point_sdf_grid.py:200-210, output read back by get_sdf
:29-51 and meshed by computeMarchingCubes -i 0.003 :248-252).  Flags and their meaning follow the tool's help text:
-s signed, -m 0 BASIC (closed manifold mesh, pseudonormal sign) / -m 1 POLYGONSOUP, -g sigma / -G sigma in grid
cells, -e expansion ratio of the automatic box, -c cube box, -b explicit box, -o output, -n -w/-W narrow-band output.

POLYGONSOUP is the published pipeline the help text names (Xu & Barbic, "Signed Distance Fields for Polygon Soup
Meshes", Graphics Interface 2014), built here from the library's own pieces, all on the GPU:
  1. unsigned distance field U of the soup                       disn_df_eval (normals = NULL)
  2. offset surface U = sigma                                    disn_mc_count / disn_mc_emit
  3. keep its exterior part: triangles whose vertices touch a grid point with U > sigma that is connected to the
     box wall through such points                                disn_df_flood
  4. signed distance to that closed manifold surface             disn_df_eval with pseudonormals
  5. + sigma, so that the zero level returns to the input surface: interior cavities are filled (negative), and
     around an open sheet, which has no inside, the result is the plain distance again (DISN meshes its fields at
     iso 0.003 for that reason, create_point_sdf_grid.py:335)
PARITY UNPINNED against the tool: it is not executed, the reference holds no input mesh and no dense output of it.
"""
import argparse
import ctypes
import sys

import numpy as np
import torch

from . import api


def read_obj(path):
    """`v x y z` / `f a b c ...` (a/b/c and negative indices accepted, polygons fanned): verts f64 [V,3], faces i32 [F,3]."""
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for t in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[t], idx[t + 1]])
    if not verts or not faces:
        raise ValueError("%s: no vertices or no faces" % path)
    return np.asarray(verts, np.float64), np.asarray(faces, np.int32)


def auto_box(verts, expand=1.5, cube=False):
    """The tool's "automatic box with cubic cells" for equal resolutions: the mesh's bounding box scaled about its
    centre by `expand` (-e; the tool's default is 1.5, DISN passes 1.2) and made a cube of the longest side (cubic cells
    need that when the three resolutions are equal, so -c changes nothing here)."""
    del cube
    lo, hi = verts.min(0), verts.max(0)
    c, half = 0.5 * (lo + hi), 0.5 * float((hi - lo).max()) * expand
    return tuple(c - half) + tuple(c + half)


def _morton(cent, lo, hi):
    q = np.clip(((cent - lo) / np.maximum(hi - lo, 1e-300) * 1023.0), 0, 1023).astype(np.uint64)

    def spread(x):
        x = (x | (x << 16)) & np.uint64(0x030000FF)
        x = (x | (x << 8)) & np.uint64(0x0300F00F)
        x = (x | (x << 4)) & np.uint64(0x030C30C3)
        x = (x | (x << 2)) & np.uint64(0x09249249)
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))


def pack_mesh(verts, faces, signed):
    """What disn_df_eval reads: geom [T,9] (corners a, b, c of the non-degenerate triangles, ordered along a Morton
    curve so that a chunk of consecutive triangles is a compact patch) and, if signed, normals [T,7,3]: unit face
    normal, angle-weighted pseudonormals of the vertices a, b, c, pseudonormals of the edges ab, ac, bc (sum of the
    unit normals of the faces around the edge) -- Baerentzen & Aanaes 2005."""
    verts = np.asarray(verts, np.float64)
    faces = np.asarray(faces, np.int64)
    if not np.isfinite(verts).all():
        raise ValueError("the mesh has non-finite vertex coordinates")
    if len(faces) and (faces.min() < 0 or faces.max() >= len(verts)):
        raise ValueError("face index outside the vertex list")
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    n = np.cross(b - a, c - a)
    nn = np.sqrt((n * n).sum(1))
    ok = nn > 0
    faces, a, b, c, n, nn = faces[ok], a[ok], b[ok], c[ok], n[ok], nn[ok]
    if len(faces) == 0:
        raise ValueError("the mesh has no triangle of non-zero area")
    order = np.argsort(_morton((a + b + c) / 3.0, verts.min(0), verts.max(0)), kind="stable")
    normals = None
    if signed:
        fn = n / nn[:, None]
        V = len(verts)
        vsum = np.zeros((V, 3))
        corners = (a, b, c)
        for s in range(3):
            e1, e2 = corners[(s + 1) % 3] - corners[s], corners[(s + 2) % 3] - corners[s]
            cosang = (e1 * e2).sum(1) / (np.sqrt((e1 * e1).sum(1)) * np.sqrt((e2 * e2).sum(1)))
            ang = np.arccos(np.clip(cosang, -1.0, 1.0))
            np.add.at(vsum, faces[:, s], ang[:, None] * fn)
        pairs = [(0, 1), (0, 2), (1, 2)]
        ekeys = [np.minimum(faces[:, u], faces[:, v]) * V + np.maximum(faces[:, u], faces[:, v]) for u, v in pairs]
        uniq, inv = np.unique(np.concatenate(ekeys), return_inverse=True)
        esum = np.zeros((len(uniq), 3))
        np.add.at(esum, inv, np.concatenate([fn, fn, fn]))
        T = len(faces)
        normals = np.empty((T, 7, 3))
        normals[:, 0] = fn
        for s in range(3):
            normals[:, 1 + s] = vsum[faces[:, s]]
            normals[:, 4 + s] = esum[inv[s * T:(s + 1) * T]]
        normals = np.ascontiguousarray(normals[order])
    geom = np.ascontiguousarray(np.concatenate([a, b, c], 1)[order])
    return geom, normals


class DistanceField:
    """Distance fields on one GPU through the C ABI (disn_df_eval, disn_df_flood, disn_mc_*)."""

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx or api.Context(device)
        self.last = {}

    def _eval(self, geom, normals, n, bbox):
        ctx = self.ctx
        g = torch.from_numpy(geom).to(ctx.device)
        nr = torch.from_numpy(normals).to(ctx.device) if normals is not None else None
        out = torch.empty((n[2], n[1], n[0]), dtype=torch.float32, device=ctx.device)
        bb = (ctypes.c_double * 6)(*[float(x) for x in bbox])
        ctx._check(ctx.lib.disn_df_eval(ctx._h, g.data_ptr(), nr.data_ptr() if nr is not None else None, len(geom),
                                        n[0], n[1], n[2], bb, out.data_ptr(), api._stream()), "disn_df_eval")
        return out

    @staticmethod
    def _points(res):
        return (res + 1,) * 3 if np.isscalar(res) else tuple(res)

    def unsigned(self, verts, faces, res, bbox):
        """-> device float32 [nz, ny, nx] (x fastest, the .dist order): distance to the nearest triangle."""
        geom, _ = pack_mesh(verts, faces, False)
        return self._eval(geom, None, self._points(res), bbox)

    def signed_basic(self, verts, faces, res, bbox):
        """`-s -m 0`: the mesh is taken to be a closed manifold; negative inside."""
        geom, normals = pack_mesh(verts, faces, True)
        return self._eval(geom, normals, self._points(res), bbox)

    def exterior(self, field, level):
        """uint8 [nz, ny, nx]: 1 where the point is connected to the box wall through points with field > level."""
        ctx = self.ctx
        nz, ny, nx = field.shape
        ext = torch.empty(field.shape, dtype=torch.uint8, device=ctx.device)
        ctx._check(ctx.lib.disn_df_flood(ctx._h, field.data_ptr(), ctypes.c_float(level), nx, ny, nz, ext.data_ptr(),
                                         api._stream()), "disn_df_flood")
        return ext

    def signed_polygon_soup(self, verts, faces, res, bbox, sigma):
        """`-s -m 1 -g sigma` (module docstring).  sigma > 0 in world units."""
        if not sigma > 0:
            raise ValueError("No valid sigma given.")
        ctx = self.ctx
        n = self._points(res)
        U = self.unsigned(verts, faces, res, bbox)
        ov, ot, keys = ctx.marching_cubes(U, iso=float(sigma), bbox=bbox, want_keys=True)
        if len(ot) == 0:
            raise ValueError("the offset surface at sigma = %g is empty (sigma outside the field's range)" % sigma)
        ext = self.exterior(U, float(sigma)).reshape(-1)
        nc = ctx._mc_counts[0]
        ek = keys[nc:]
        p, d = torch.div(ek, 3, rounding_mode="floor"), ek % 3
        stride = torch.tensor([1, n[0], n[0] * n[1]], device=ctx.device)[d]
        Uf = U.reshape(-1)
        pos = torch.where(Uf[p] > np.float32(sigma), p, p + stride)      # the endpoint on the positive side
        v_ext = torch.cat([torch.zeros(nc, dtype=torch.bool, device=ctx.device), ext[pos] != 0])
        keep = v_ext[ot.long()].any(dim=1)
        kept = ot[keep].cpu().numpy()
        self.last = dict(offset_tris=int(keep.sum().item()), dropped=int((~keep).sum().item()))
        if len(kept) == 0:
            raise ValueError("no part of the offset surface is reachable from the box wall: enlarge the box (-e)")
        used, inv = np.unique(kept.reshape(-1), return_inverse=True)     # the kept component's own vertex list
        geom, normals = pack_mesh(ov[torch.from_numpy(used).to(ctx.device)].cpu().numpy(), inv.reshape(-1, 3), True)
        out = self._eval(geom, normals, n, bbox)
        out += np.float32(sigma)
        return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="Computes signed or unsigned distance field for a given triangle mesh "
                                             "(flags of the reference's computeDistanceField).")
    ap.add_argument("obj")
    ap.add_argument("resolution", type=int, nargs=3, metavar=("RX", "RY", "RZ"))
    ap.add_argument("-s", dest="signed", action="store_true", help="compute signed distance field")
    ap.add_argument("-o", dest="out", default="out.dist")
    ap.add_argument("-m", dest="mode", type=int, default=0, help="0 BASIC, 1 POLYGONSOUP, 2 AUTO")
    ap.add_argument("-b", dest="box", default=None, help="xmin,ymin,zmin,xmax,ymax,zmax")
    ap.add_argument("-c", dest="cube", action="store_true")
    ap.add_argument("-e", dest="expand", type=float, default=1.5)
    ap.add_argument("-n", dest="narrow", action="store_true", help="write only the narrow band |d| < band (.n.dist layout)")
    ap.add_argument("-w", dest="band", type=float, default=None, help="band width")
    ap.add_argument("-W", dest="band_grid", type=float, default=None, help="band width in grid sizes")
    ap.add_argument("-g", dest="sigma", type=float, default=None)
    ap.add_argument("-G", dest="sigma_grid", type=float, default=None)
    ap.add_argument("--gpu", type=int, default=0)
    argv = list(sys.argv[1:] if argv is None else argv)
    for q in range(len(argv) - 1):           # "-b -1,-1,-1,1,1,1": the value starts with a minus sign
        if argv[q] == "-b":
            argv[q:q + 2] = ["-b=" + argv[q + 1]]
            break
    a = ap.parse_args(argv)
    if len(set(a.resolution)) != 1 or a.resolution[0] < 1:
        sys.exit("Invalid resolution.")
    if a.mode not in (0, 1, 2):
        sys.exit("Invalid sign field computation mode: %d." % a.mode)
    if not a.expand > 0:
        sys.exit("Invalid expansion ratio: %G." % a.expand)
    res = a.resolution[0]
    verts, faces = read_obj(a.obj)
    bbox = tuple(float(x) for x in a.box.split(",")) if a.box else auto_box(verts, a.expand, a.cube)
    if len(bbox) != 6 or not all(bbox[3 + q] > bbox[q] for q in range(3)):
        sys.exit("Invalid bounding box.")
    df = DistanceField(device=a.gpu)
    mode = a.mode
    if a.signed and mode == 2:
        mode = 0 if is_closed_manifold(faces) else 1
    if not a.signed:
        field = df.unsigned(verts, faces, res, bbox)
    elif mode == 0:
        if not is_closed_manifold(faces):
            sys.exit("Error: input mesh is not manifold or has boundary. Signed distance field is ill-defined.")
        field = df.signed_basic(verts, faces, res, bbox)
    else:
        sigma = a.sigma if a.sigma is not None else (a.sigma_grid * (bbox[3] - bbox[0]) / res if a.sigma_grid else None)
        if sigma is None or not sigma > 0:
            sys.exit("No valid sigma given.")
        field = df.signed_polygon_soup(verts, faces, res, bbox, sigma)
    if a.narrow:
        # the whole field is computed and the band cut out of it afterwards (the tool confines the computation to the band)
        band = a.band if a.band is not None else (a.band_grid * (bbox[3] - bbox[0]) / res if a.band_grid else None)
        if band is None or not band > 0:
            sys.exit("Error: No valid band width is provided.")
        api.write_narrowband_dist(a.out, res, bbox, field.cpu().numpy(), band)
    else:
        api.write_dist(a.out, res, bbox, field.cpu().numpy())
    print("Saving the distance field to %s" % a.out)


def is_closed_manifold(faces):
    """every undirected edge is shared by exactly two triangles, once in each direction"""
    f = np.asarray(faces, np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    V = int(f.max()) + 1
    fwd = e[:, 0] * V + e[:, 1]
    und = np.minimum(e[:, 0], e[:, 1]) * V + np.maximum(e[:, 0], e[:, 1])
    _, cnt = np.unique(und, return_counts=True)
    return bool((cnt == 2).all()) and len(np.unique(fwd)) == len(fwd)


if __name__ == "__main__":
    main()
