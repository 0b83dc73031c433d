"""The reference's isosurface tool as a command line, flag for flag:

    computeMarchingCubes <distancefield file name> <output objmesh file name> [-i isovalue] [-n narrowband]

(closed Vega-FEM binary; call sites demo/demo.py:365-369, test/create_sdf.py:314,
preprocessing/create_point_sdf_grid.py:248-252 -- all `os.system(command_str)`, so a maintainer only swaps the
executable: `python -m disn_amd.marching_cubes in.dist out.obj -i 0.003`).  The .dist (or, with `-n 1`, the narrow-band
.n.dist) is read on the host, meshed on the GPU through the C ABI (disn_mc_count / disn_mc_emit) and
written as OBJ text in the tool's format.  Unlike the tool's callers, a failure is an exit status != 0.
"""
import argparse
import sys
import time

import torch

from . import api


def main(argv=None):
    ap = argparse.ArgumentParser(prog="computeMarchingCubes", description=__doc__.split("\n\n")[0],
                                 usage="%(prog)s <distancefield file name> <output objmesh file name> [-i isovalue] [-n narrowband]")
    ap.add_argument("dist")
    ap.add_argument("obj")
    ap.add_argument("-i", dest="iso", type=float, default=0.0)
    ap.add_argument("-n", dest="narrowband", type=int, default=0)
    ap.add_argument("--gpu", type=int, default=0)
    a = ap.parse_args(argv)
    try:
        if a.narrowband:
            res, bbox, grid, _ = api.read_narrowband_dist(a.dist)
        else:
            res, bbox, grid = api.read_dist(a.dist)
    except api.DisnError as e:
        sys.exit("Error: could not load the distance field %s (%s)" % (a.dist, e))
    ctx = api.Context(a.gpu)
    t0 = time.time()
    verts, tris = ctx.marching_cubes(torch.from_numpy(grid).to(ctx.device), iso=a.iso, bbox=tuple(float(b) for b in bbox))
    torch.cuda.synchronize()
    print("Computing the isosurface using marching cubes. Iso-level: %f." % a.iso)
    print("Done. Time: %.6fs  #v %d #t %d (resolution %d)" % (time.time() - t0, len(verts), len(tris), res))
    api.write_obj(a.obj, verts.cpu().numpy(), tris.cpu().numpy())
    ctx.close()


if __name__ == "__main__":
    main()
