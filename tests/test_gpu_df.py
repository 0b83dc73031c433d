"""GPU distance field (disn_df_eval, disn_df_flood and the polygon-soup pipeline of disn_amd/distance_field.py, all
through the C ABI) against the CPU oracle (oracle/df_oracle.c: brute force, a different point-triangle formulation).
Bar: float64 arithmetic on both sides, float32 results: |difference| <= 2e-7 (1 + |d|) (one float32 rounding step),
signs equal wherever |d| > 1e-6.  Parity with the reference tool itself is UNPINNED (see tests/test_df_oracle.py);
the one piece of the tool's own output the reference holds, the narrow band isosurface/model_normalized.obj.n.dist,
is used as a consistency pin below."""
import numpy as np
import pytest
import torch

import fields
import meshes
from disn_amd.distance_field import DistanceField
from oracle import df, mc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dfield(ctx):
    return DistanceField(ctx)


def close(got, want):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    err = np.abs(got - want) / (1 + np.abs(want))
    assert err.max() <= 2e-7, err.max()
    big = np.abs(want) > 1e-6
    assert np.array_equal(np.sign(got[big]), np.sign(want[big]))
    return got


def test_unsigned_random_soup_noncubic_grid(dfield):
    """700 random triangles (6 chunks) on a 19 x 23 x 17 grid over a non-cubic box: bricks that overhang the grid,
    a last chunk that is not full, culling on scattered geometry"""
    v, f = meshes.random_soup(700, seed=1)
    bbox = (-1.1, -0.9, -1.3, 1.0, 1.2, 0.8)
    got = dfield._eval(*_pack(v, f, False), (19, 23, 17), bbox)
    close(got, df.distance_field(v, f, (19, 23, 17), bbox))


def _pack(v, f, signed):
    from disn_amd.distance_field import pack_mesh
    return pack_mesh(v, f, signed)


@pytest.mark.parametrize("name", ["icosphere", "torus", "cube"])
def test_signed_closed_meshes(dfield, name):
    v, f = {"icosphere": lambda: meshes.icosphere(0.6, 3), "torus": meshes.torus, "cube": lambda: meshes.cube(0.45, (0.05, 0, -0.1))}[name]()
    bbox = (-1, -1, -1, 1, 1, 1)
    got = close(dfield.signed_basic(v, f, 24, bbox), df.distance_field(v, f, 24, bbox, signed=True))
    assert (got < 0).any() and (got > 0).any()
    un = dfield.unsigned(v, f, 24, bbox).cpu().numpy()
    assert np.array_equal(np.abs(got), un)                       # same distances, sign aside


def test_result_does_not_depend_on_triangle_order(dfield):
    v, f = meshes.icosphere(0.5, 2)
    a = dfield.signed_basic(v, f, 16, (-1, -1, -1, 1, 1, 1))
    b = dfield.signed_basic(v, f[::-1].copy(), 16, (-1, -1, -1, 1, 1, 1))
    assert torch.equal(a.abs(), b.abs())
    assert (torch.sign(a) != torch.sign(b)).sum() == 0


def test_exterior_flood_matches_labelling(dfield):
    from scipy import ndimage
    g = fields.noise_field((21, 18, 25), seed=9)                 # noise: many enclosed pockets, twisted passages
    for level in (0.0, 0.15):
        ext = dfield.exterior(torch.from_numpy(g).cuda(), level).cpu().numpy().astype(bool)
        lab, _ = ndimage.label(g > np.float32(level))
        wall = np.zeros_like(lab, bool)
        wall[0], wall[-1], wall[:, 0], wall[:, -1], wall[:, :, 0], wall[:, :, -1] = (True,) * 6
        want = np.isin(lab, np.unique(lab[wall & (lab > 0)]))
        assert np.array_equal(ext, want) and want.any() and (~want & (g > level)).any()


def test_polygon_soup_pipeline_matches_oracle(dfield):
    """closed box with an internal cavity + an open sheet (tests/meshes.py): every stage on the GPU (unsigned field,
    marching cubes at sigma, exterior part, signed field of the offset surface, + sigma) against the oracle's pipeline"""
    v, f = meshes.soup_with_cavity_and_sheet()
    res, bbox, sigma = 24, (-1.2, -1.2, -1.2, 1.2, 1.2, 1.2), 0.12
    want, info = df.polygon_soup_field(v, f, res, bbox, sigma)
    got = dfield.signed_polygon_soup(v, f, res, bbox, sigma).cpu().numpy()
    assert dfield.last["dropped"] == info["dropped"] > 0 and dfield.last["offset_tris"] == info["offset_tris"]
    # the offset surface is built from float32 fields that may differ in the last bit: its vertices move by ~1e-7
    assert np.abs(got - want).max() <= 2e-6
    c = res // 2
    assert got[c, c, c] < -0.4                                   # cavity filled
    with pytest.raises(ValueError):
        dfield.signed_polygon_soup(v, f, res, bbox, 0.0)


def test_consistent_with_the_reference_tools_own_band(dfield, ctx, tmp_path):
    """The reference ships one output of its distance-field tool: the narrow band isosurface/model_normalized.obj.n.dist
    (res 64, 69 149 cells, values -0.9 h .. +5 h; re-packed in tests/golden).  A distance field is the distance to each
    of its own level sets, shifted: the level 0.006 (half a cell: the object has sheets thinner than a cell, which
    have no sign change at level 0 -- the reason DISN meshes at 0.003) meshed by our marching cubes is a closed surface
    whose signed distance field must give back (tool's value - 0.006) over the whole band: same sign everywhere off
    the surface, values within the mesher's interpolation error.  Measured: median 0.09 h, 99 % within 0.45 h,
    max 0.95 h (the CPU oracle gives the same figures)."""
    from disn_amd import api
    from test_mc_oracle import _narrowband_file
    p, z = _narrowband_file(tmp_path)
    res, bbox, grid, _ = api.read_narrowband_dist(p)
    bbox = tuple(float(b) for b in bbox)
    level = 0.006
    band = np.abs(grid) < 1e30
    v, f = ctx.marching_cubes(torch.from_numpy(grid).cuda(), iso=level, bbox=bbox)
    v, f = v.cpu().numpy(), f.cpu().numpy()
    assert fields.mesh_is_closed_manifold(f)
    got = dfield.signed_basic(v, f, res, bbox).cpu().numpy()
    h = (bbox[3] - bbox[0]) / res
    want = grid - np.float32(level)
    err = np.abs(got - want)[band] / h
    print("band cells %d: |ours - (tool - level)| median %.3f h, 99 %% %.3f h, max %.3f h"
          % (band.sum(), np.median(err), np.percentile(err, 99), err.max()))
    assert np.median(err) <= 0.15 and np.percentile(err, 99) <= 0.6 and err.max() <= 1.25
    off = band & (np.abs(want) > 0.5 * h)
    assert off.sum() > 40000 and np.array_equal(np.sign(got[off]), np.sign(want[off]))


def test_res128_mesh_of_35k_vertices(dfield, ctx):
    """a real-sized input: the analytic field's res-128 isosurface (35 690 v / 71 380 t) back to a res-128 signed field;
    where the analytic field is an exact distance (outside the union of the primitives) the two agree to the
    mesh's faceting error, and the culling keeps the work far below brute force (seconds on the GPU otherwise)"""
    import time
    g = fields.analytic_field(128)
    v, f = ctx.marching_cubes(torch.from_numpy(g).cuda(), iso=0.0)
    v, f = v.cpu().numpy(), f.cpu().numpy()
    torch.cuda.synchronize()
    t0 = time.time()
    got = dfield.signed_basic(v, f, 128, (-1, -1, -1, 1, 1, 1))
    torch.cuda.synchronize()
    dt = time.time() - t0
    got = got.cpu().numpy()
    h = 2.0 / 128
    out = g > 2 * h
    assert np.abs(got - g)[out].max() <= 0.05 * h
    assert np.array_equal(np.sign(got[np.abs(g) > 0.5 * h]), np.sign(g[np.abs(g) > 0.5 * h]))
    print("res 128, %d triangles: %.3f s incl. host packing" % (len(f), dt))
    assert dt < 20


def test_argument_errors(ctx):
    import ctypes
    from disn_amd import api
    bb = (ctypes.c_double * 6)(-1, -1, -1, 1, 1, 1)
    out = torch.empty(8, device="cuda")
    g = torch.zeros(9, dtype=torch.float64, device="cuda")
    assert ctx.lib.disn_df_eval(ctx._h, g.data_ptr(), None, 0, 2, 2, 2, bb, out.data_ptr(), None) != 0
    assert ctx.lib.disn_df_eval(ctx._h, g.data_ptr(), None, 1, 1, 2, 2, bb, out.data_ptr(), None) != 0
    assert ctx.lib.disn_df_eval(ctx._h, None, None, 1, 2, 2, 2, bb, out.data_ptr(), None) != 0
    bad = (ctypes.c_double * 6)(-1, -1, -1, 1, -1, 1)
    assert ctx.lib.disn_df_eval(ctx._h, g.data_ptr(), None, 1, 2, 2, 2, bad, out.data_ptr(), None) != 0
    assert b"box" in ctx.lib.disn_last_error(ctx._h)
    # non-finite corners that reach the library directly: no triangle wins, nothing is read through the missing index
    nan = torch.full((9,), float("nan"), dtype=torch.float64, device="cuda")
    nrm = torch.zeros(21, dtype=torch.float64, device="cuda")
    assert ctx.lib.disn_df_eval(ctx._h, nan.data_ptr(), nrm.data_ptr(), 1, 2, 2, 2, bb, out.data_ptr(), None) == 0
    assert torch.isinf(out).all()


def test_cli_surface_writes_the_tools_file_format(dfield, tmp_path):
    """`computeDistanceField obj res res res -s -e 1.2 -o out.dist -m 1 -g sigma` (create_point_sdf_grid.py:200-210)
    through disn_amd.distance_field.main: the .dist that get_sdf (:29-51) reads back -- negative first resolution,
    box, (res+1)^3 float32 x fastest -- holding the pipeline's field; -m 0 on a soup and a missing sigma stop with the
    tool's messages."""
    from disn_amd import api, distance_field as dfh
    v, f = meshes.soup_with_cavity_and_sheet()
    obj, out = str(tmp_path / "pc_norm.obj"), str(tmp_path / "0.dist")
    api.write_obj(obj, v, f)
    rv, rf = dfh.read_obj(obj)                                   # what the CLI will see (6 significant digits)
    dfh.main([obj, "20", "20", "20", "-s", "-e", "1.2", "-o", out, "-m", "1", "-g", "0.1"])
    res, bbox, grid = api.read_dist(out)
    import struct
    with open(out, "rb") as fh:
        assert struct.unpack("<3i", fh.read(12)) == (-20, 20, 20)
    assert res == 20 and grid.shape == (21, 21, 21)
    assert np.allclose(bbox, dfh.auto_box(rv, 1.2))
    want = dfield.signed_polygon_soup(rv, rf, 20, tuple(bbox), 0.1).cpu().numpy()
    assert np.array_equal(grid, want)
    with pytest.raises(SystemExit, match="manifold"):
        dfh.main([obj, "20", "20", "20", "-s", "-m", "0", "-o", out])
    with pytest.raises(SystemExit, match="sigma"):
        dfh.main([obj, "20", "20", "20", "-s", "-m", "1", "-o", out])
    dfh.main([obj, "20", "20", "20", "-o", out, "-b", "-1,-1,-1,1,1,1"])               # unsigned, explicit box
    _, bbox2, un = api.read_dist(out)
    assert np.array_equal(bbox2, [-1, -1, -1, 1, 1, 1]) and (un >= 0).all()
