"""Pins the CPU marching-cubes oracle (oracle/mc_oracle.c) to the known answers
the survey recorded from the reference tool, and checks mesh invariants."""
import numpy as np
import pytest

import fields
from oracle import mc


def test_sphere_res8_known_answer():
    # r=0.6 sphere, res 8 -> "#v 126 #t 248"
    r = mc.marching_cubes(fields.sphere_field(8))
    assert (len(r["verts"]), r["n_tri"]) == (126, 248)
    assert len(r["verts"]) - 3 * r["n_tri"] // 2 + r["n_tri"] == 2   # Euler characteristic


@pytest.mark.parametrize("res", [64, 128])
def test_analytic_field_known_answer(res):
    # SURVEY.md s6 / BASELINE.md: sizes printed by the reference tool
    r = mc.marching_cubes(fields.analytic_field(res))
    assert (len(r["verts"]), r["n_tri"]) == fields.KNOWN_ANALYTIC[res]
    assert r["n_bad"] == 0
    assert fields.mesh_is_closed_manifold(r["tris"])


def test_ordering_rules():
    # SURVEY.md App. B.6
    r = mc.marching_cubes(fields.noise_field((10, 9, 11), seed=3))
    assert r["n_centre"] > 0
    k = r["keys"]
    assert (np.diff(k) > 0).all()                      # ascending signed key, unique
    assert (k[: r["n_centre"]] < 0).all() and (k[r["n_centre"]:] >= 0).all()
    t = r["tris"]
    assert (t[:, 0] < t[:, 1]).all() and (t[:, 0] < t[:, 2]).all()   # smallest index first
    order = np.lexsort((t[:, 2], t[:, 1], t[:, 0]))
    assert np.array_equal(order, np.arange(len(t)))
    assert len(np.unique(t, axis=0)) == len(t)
    assert fields.mesh_is_closed_manifold(t)


def test_iso_nudge_and_bbox():
    g = fields.sphere_field(8)
    a = mc.marching_cubes(g, iso=0.05)
    b = mc.marching_cubes(g - np.float32(0.05), iso=0.0)
    assert np.array_equal(a["tris"], b["tris"])
    # exact-iso samples count as outside (B.1): a plane through grid points
    l = fields.lin(8)
    z, y, x = np.meshgrid(l, l, l, indexing="ij")
    p = mc.marching_cubes(x.astype(np.float32), iso=0.0)
    assert p["n_tri"] > 0 and np.all(p["verts"][:, 0] <= 0.0)
    # anisotropic bbox scales vertices linearly
    s = mc.marching_cubes(g, bbox=(-2, -1, -3, 2, 1, 3))
    u = mc.marching_cubes(g)
    assert np.allclose(s["verts"], u["verts"] * np.array([2.0, 1.0, 3.0]), atol=1e-12)


def test_golden_regression():
    """tests/golden/mc_noise12.npz: oracle output frozen at the commit that
    matched the reference's known answers (generator: tests/golden/make_golden.py)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mc_noise12.npz"))
    r = mc.marching_cubes(z["grid"], iso=float(z["iso"]), bbox=tuple(z["bbox"]))
    assert np.array_equal(r["tris"], z["tris"])
    assert np.array_equal(r["verts"], z["verts"])


def _narrowband_file(tmp_path):
    """Re-create the reference's shipped narrow-band file from the golden arrays (tests/golden/make_golden.py)."""
    import os
    import struct
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_narrowband_res64.npz"))
    p = str(tmp_path / "model_normalized.obj.n.dist")
    with open(p, "wb") as f:
        f.write(struct.pack("<3i", *z["header"]) + struct.pack("<6d", *z["bbox"]) + struct.pack("<i", len(z["d"])))
        rec = np.zeros(len(z["d"]), dtype=np.dtype([("i", "<i4"), ("j", "<i4"), ("k", "<i4"), ("d", "<f4")]))
        rec["i"], rec["j"], rec["k"], rec["d"] = z["ijk"][:, 0], z["ijk"][:, 1], z["ijk"][:, 2], z["d"]
        f.write(rec.tobytes())
    return p, z


def test_reference_narrowband_fixture_known_answer(tmp_path):
    """The reference's own data file meshes to exactly the size its tool reports with `-n 1`
    (#v 3800 #t 7604): a reference fixture pinning the oracle."""
    from disn_amd import api
    p, z = _narrowband_file(tmp_path)
    res, bbox, grid, n = api.read_narrowband_dist(p)
    assert res == 64 and n == 69149 and np.array_equal(bbox, z["bbox"])
    assert int((np.abs(grid) < 1e38).sum()) == 69149
    r = mc.marching_cubes(grid, iso=0.0, bbox=tuple(bbox))
    assert (len(r["verts"]), r["n_tri"]) == tuple(z["expect_vt"]) == (3800, 7604)
    assert r["n_bad"] == 0 and fields.mesh_is_closed_manifold(r["tris"])


def test_narrowband_missing_cells_take_the_sign_of_their_side(tmp_path):
    """A narrow band (|d| < 2.5 h) of two spheres written as .n.dist: the reader returns -FLT_MAX for the missing
    cells inside a sphere and +FLT_MAX outside, and the band grid meshes exactly like the full field."""
    import struct
    from disn_amd import api
    res = 24
    l = np.linspace(-1, 1, res + 1)
    z, y, x = np.meshgrid(l, l, l, indexing="ij")
    full = np.minimum(np.sqrt((x - 0.3) ** 2 + y ** 2 + z ** 2) - 0.45, np.sqrt((x + 0.45) ** 2 + y ** 2 + z ** 2) - 0.3)
    full = full.astype(np.float32)
    band = np.abs(full) < 2.5 * (2.0 / res)
    k, j, i = np.nonzero(band)
    p = str(tmp_path / "two_spheres.n.dist")
    with open(p, "wb") as f:
        f.write(struct.pack("<3i", -res, res, res) + struct.pack("<6d", -1, -1, -1, 1, 1, 1) + struct.pack("<i", len(i)))
        rec = np.zeros(len(i), dtype=np.dtype([("i", "<i4"), ("j", "<i4"), ("k", "<i4"), ("d", "<f4")]))
        rec["i"], rec["j"], rec["k"], rec["d"] = i, j, k, full[band]
        f.write(rec.tobytes())
    r, bbox, grid, n = api.read_narrowband_dist(p)
    assert r == res and n == len(i)
    assert np.array_equal(grid[band], full[band])
    assert bool((grid[~band & (full < 0)] == -np.finfo(np.float32).max).all()) and int((~band & (full < 0)).sum()) > 50
    assert bool((grid[~band & (full > 0)] == np.finfo(np.float32).max).all())
    a, b = mc.marching_cubes(grid, iso=0.0), mc.marching_cubes(full, iso=0.0)
    assert a["n_tri"] > 500 and np.array_equal(a["tris"], b["tris"]) and np.array_equal(a["verts"], b["verts"])


def test_narrowband_band_open_at_the_box_wall(tmp_path):
    """A surface that leaves the box: a sphere centred on the x = -1 wall, so grid lines through it START inside the
    surface, plus a slab through the whole box whose interior lines hold no band cell at all (resolved along y / z).
    Every missing cell must take the sign of its side and the band grid must mesh like the full field."""
    import struct
    from disn_amd import api
    res = 24
    l = np.linspace(-1, 1, res + 1)
    z, y, x = np.meshgrid(l, l, l, indexing="ij")
    sphere = np.sqrt((x + 1.0) ** 2 + y ** 2 + z ** 2) - 0.55          # cut open by the wall x = -1
    slab = np.abs(z - 0.6) - 0.22                                      # spans the box in x and y: interior rows list nothing
    full = np.minimum(sphere, slab).astype(np.float32)
    band = np.abs(full) < 1.6 * (2.0 / res)
    k, j, i = np.nonzero(band)
    p = str(tmp_path / "open.n.dist")
    with open(p, "wb") as f:
        f.write(struct.pack("<3i", -res, res, res) + struct.pack("<6d", -1, -1, -1, 1, 1, 1) + struct.pack("<i", len(i)))
        rec = np.zeros(len(i), dtype=np.dtype([("i", "<i4"), ("j", "<i4"), ("k", "<i4"), ("d", "<f4")]))
        rec["i"], rec["j"], rec["k"], rec["d"] = i, j, k, full[band]
        f.write(rec.tobytes())
    r, bbox, grid, n = api.read_narrowband_dist(p)
    assert r == res and n == len(i) and np.array_equal(grid[band], full[band])
    inside, outside = ~band & (full < 0), ~band & (full > 0)
    assert int(inside[:, :, 0].sum()) > 20                             # rows that begin inside the sphere at the wall
    rows_without_cells = ~band.any(axis=2)
    assert int((rows_without_cells & (full[:, :, 0] < 0)).sum()) > 20  # interior rows of the slab
    assert bool((grid[inside] == -np.finfo(np.float32).max).all())
    assert bool((grid[outside] == np.finfo(np.float32).max).all())
    a, b = mc.marching_cubes(grid, iso=0.0), mc.marching_cubes(full, iso=0.0)
    assert a["n_tri"] > 500 and np.array_equal(a["tris"], b["tris"]) and np.array_equal(a["verts"], b["verts"])
