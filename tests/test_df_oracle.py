"""CPU: the distance-field oracle (oracle/df_oracle.c, oracle/df.py) against closed forms, and the product's host-side
mesh preparation (disn_amd/distance_field.py: pseudonormals, box, OBJ reader) against the oracle's own.
Parity with the reference tool `computeDistanceField` is UNPINNED (closed binary, not executed; the reference holds no
input mesh and no dense output): what is pinned here is the definition -- exact distance, sign of a closed surface."""
import numpy as np
import pytest

import meshes
from disn_amd import distance_field as dfh
from oracle import df


def grid(res, bbox=(-1, -1, -1, 1, 1, 1)):
    ax = [np.linspace(bbox[a], bbox[3 + a], res + 1) for a in range(3)]
    z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    return x, y, z


def test_oracle_cube_closed_form():
    v, f = meshes.cube(0.5)
    s = df.distance_field(v, f, 16, (-1, -1, -1, 1, 1, 1), signed=True)
    x, y, z = grid(16)
    q = np.stack([np.abs(x), np.abs(y), np.abs(z)], -1) - 0.5
    want = np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0)
    assert np.abs(s - want).max() <= 1e-6
    assert np.array_equal(np.abs(s), df.distance_field(v, f, 16, (-1, -1, -1, 1, 1, 1)))


def test_oracle_sphere_within_faceting_error():
    v, f = meshes.icosphere(0.6, 3)
    s = df.distance_field(v, f, 12, (-1, -1, -1, 1, 1, 1), signed=True)
    x, y, z = grid(12)
    want = np.sqrt(x * x + y * y + z * z) - 0.6
    assert np.abs(s - want).max() <= 0.004          # sagitta of the level-3 icosphere's facets
    assert (np.sign(s) == np.sign(want))[np.abs(want) > 0.01].all()


def test_host_pseudonormals_match_oracle():
    for v, f in (meshes.icosphere(0.5, 1), meshes.torus(), meshes.cube(0.3, (0.1, 0, -0.2))):
        geom, normals = dfh.pack_mesh(v, f, True)
        on, og = df.pseudonormals(v, f), v[f].reshape(-1, 9)
        lut = {g.tobytes(): t for t, g in enumerate(og)}
        rows = [lut[g.tobytes()] for g in geom]
        assert sorted(rows) == list(range(len(f)))          # a permutation (Morton order), nothing dropped
        assert np.allclose(normals, on[rows], rtol=1e-12, atol=1e-14)


def test_pack_drops_degenerate_triangles_and_rejects_empty():
    v, f = meshes.cube()
    f2 = np.concatenate([f, [[0, 0, 1], [2, 2, 2]]]).astype(np.int32)
    geom, _ = dfh.pack_mesh(v, f2, False)
    assert len(geom) == 12
    with pytest.raises(ValueError):
        dfh.pack_mesh(v, np.array([[0, 0, 1]], np.int32), False)
    with pytest.raises(ValueError, match="finite"):
        dfh.pack_mesh(v * np.array([1, np.nan, 1]), f, False)
    with pytest.raises(ValueError, match="index"):
        dfh.pack_mesh(v, f + 1, False)


def test_manifold_check_box_and_obj_reader(tmp_path):
    v, f = meshes.icosphere(0.5, 1)
    assert dfh.is_closed_manifold(f)
    assert not dfh.is_closed_manifold(f[:-1])                     # a hole
    assert not dfh.is_closed_manifold(np.concatenate([f[:1][:, [0, 2, 1]], f[1:]]))   # one face flipped
    sv, sf = meshes.soup_with_cavity_and_sheet()
    assert not dfh.is_closed_manifold(sf)
    box = dfh.auto_box(v * [1.0, 0.5, 0.25] + [0.1, 0, 0], 1.2)
    assert np.allclose(box, (0.1 - 0.6, -0.6, -0.6, 0.1 + 0.6, 0.6, 0.6), atol=1e-12)
    p = tmp_path / "m.obj"
    with open(p, "w") as fh:
        fh.write("# c\n")
        for q in v:
            fh.write("v %.17g %.17g %.17g\n" % tuple(q))
        fh.write("vn 0 0 1\n")
        for a, b, c in f[:-2]:
            fh.write("f %d//1 %d//1 %d//1\n" % (a + 1, b + 1, c + 1))
        a, b, c = f[-2]
        fh.write("f %d %d %d\n" % (a - len(v), b - len(v), c - len(v)))          # negative indices
        a, b, c = f[-1]
        fh.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
    rv, rf = dfh.read_obj(str(p))
    assert np.array_equal(rv, v) and np.array_equal(rf, f)
    with open(p, "w") as fh:
        fh.write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")                 # a quad is fanned
    rv, rf = dfh.read_obj(str(p))
    assert rf.tolist() == [[0, 1, 2], [0, 2, 3]]


def test_oracle_polygon_soup_pipeline_fills_cavity_and_keeps_open_sheets():
    v, f = meshes.soup_with_cavity_and_sheet()
    res, bbox, sigma = 24, (-1.2, -1.2, -1.2, 1.2, 1.2, 1.2), 0.12
    s, info = df.polygon_soup_field(v, f, res, bbox, sigma)
    U = info["unsigned"]
    assert info["dropped"] > 0                                   # the cavity's offset surface is removed
    c = res // 2
    assert s[c, c, c] < -0.4                                     # the cavity is filled: deep inside the outer box
    x, y, z = grid(res, bbox)
    far = U > 2.5 * sigma
    outside = far & (np.maximum(np.maximum(np.abs(x), np.abs(y)), np.abs(z)) > 0.5)
    assert np.abs(s - U)[outside].max() <= 0.35 * sigma          # outside, away from the surface: the plain distance
    on_sheet = (np.abs(x - 0.75) < 0.05) & (np.abs(y) < 0.2) & (np.abs(z) < 0.2)
    # an open sheet has no inside: around it the result is the plain distance again (DISN meshes such fields at 0.003)
    assert on_sheet.any() and np.abs(s - U)[on_sheet].max() <= 1e-3


def test_narrowband_writer_round_trips_through_the_reader(tmp_path):
    """disn_write_ndist (the tool's `-n -w band` output) -> disn_read_ndist: listed cells come back bit for bit, the
    others as +-FLT_MAX on the right side, and the band meshes exactly like the dense field (CPU oracle)."""
    import struct
    import fields
    from disn_amd import api
    from oracle import mc
    res = 32
    g = fields.analytic_field(res)
    bbox = (-1, -1, -1, 1, 1, 1)
    h = 2.0 / res
    p = str(tmp_path / "a.n.dist")
    api.write_narrowband_dist(p, res, bbox, g, 3 * h)
    with open(p, "rb") as fh:
        raw = fh.read()
    assert struct.unpack("<3i", raw[:12]) == (-res, res, res) and struct.unpack("<6d", raw[12:60]) == tuple(float(b) for b in bbox)
    n = struct.unpack("<i", raw[60:64])[0]
    band = np.abs(g) < np.float32(3 * h)
    assert n == int(band.sum()) and len(raw) == 64 + 16 * n
    r, bb, back, cells = api.read_narrowband_dist(p)
    assert r == res and cells == n and np.array_equal(back[band], g[band])
    assert (np.abs(back[~band]) > 1e30).all() and np.array_equal(np.sign(back[~band]), np.sign(g[~band]))
    a, b = mc.marching_cubes(g, bbox=bbox), mc.marching_cubes(back, bbox=bbox)
    assert np.array_equal(a["tris"], b["tris"]) and np.array_equal(a["verts"], b["verts"])
    with pytest.raises(api.DisnError):
        api.write_narrowband_dist(p, res, bbox, g, 0.0)
