"""Marching cubes pinned WITHOUT the oracle: literal single-voxel expectations worked out from SURVEY.md App. B
(vertex formula B.4, centre vertex B.5, ordering and orientation B.6) and geometric facts about analytic surfaces
(distance of every vertex to the true surface, enclosed volume, Euler characteristic).  The oracle appears only in
the last test, where the point is that GPU and oracle apply the same rule to a voxel whose ambiguity lookup is
impossible (B.3)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def gpu_mesh(ctx, g, iso=0.0, bbox=(-1, -1, -1, 1, 1, 1)):
    t = torch.from_numpy(np.ascontiguousarray(g, dtype=np.float32)).cuda()
    v, f, k = ctx.marching_cubes(t, iso=iso, bbox=bbox, want_keys=True)
    torch.cuda.synchronize()
    return v.cpu().numpy(), f.cpu().numpy(), k.cpu().numpy()


def voxel(values):
    """values[v] for cube corners v0..v7 = (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) (1,0,1) (1,1,1) (0,1,1) -> grid[k][j][i]"""
    g = np.zeros((2, 2, 2), np.float32)
    for v, (i, j, k) in enumerate([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]):
        g[k, j, i] = values[v]
    return g


def normals_point_to_positive(v, f, inside_point):
    """for these small convex-ish patches: every triangle's normal points away from the negative corner"""
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(b - a, c - a)
    return bool((np.einsum("ij,ij->i", n, (a + b + c) / 3 - inside_point) > 0).all())


def test_single_voxel_plain_corner(ctx):
    """one negative corner v0 = -1 with neighbours +3 (x), +1 (y), +0.5 (z); box [-1,1]^3, h = 2.
    B.4: p[dir] = lo + (c_a * h) / (float)(c_a - c_b) -> x: -1 + (-2)/(-4), y: -1 + (-2)/(-2), z: -1 + (-2)/(-1.5);
    keys 3*point + dir = 0, 1, 2; one triangle, normal towards the positive corners."""
    v, f, k = gpu_mesh(ctx, voxel([-1, 3, 1, 1, 0.5, 1, 1, 1]))
    assert k.tolist() == [0, 1, 2]
    assert v.tolist() == [[-0.5, -1.0, -1.0], [-1.0, 0.0, -1.0], [-1.0, -1.0, -1.0 + (-1.0 * 2.0) / -1.5]]
    assert f.tolist() == [[0, 1, 2]]
    assert normals_point_to_positive(v, f, np.array([-1.0, -1.0, -1.0]))


def test_single_voxel_face_ambiguity_both_ways(ctx):
    """v0 and v2 negative: the bottom face (0,1,2,3) is ambiguous.  faceTest (B.3) compares A*C with B*D:
    (-1)(-1) > (1)(1) is false -> the two negative corners stay separate (two corner triangles);
    (-3)(-3) > (1)(1) is true  -> they are joined through the face (four triangles, a tunnel)."""
    v, f, k = gpu_mesh(ctx, voxel([-1, 1, -1, 1, 1, 1, 1, 1]))
    assert k.tolist() == [0, 1, 2, 4, 6, 11]                   # edges of v0: x,y,z; (1,0,0) y; (0,1,0) x; (1,1,0) z
    assert v.tolist() == [[0, -1, -1], [-1, 0, -1], [-1, -1, 0], [1, 0, -1], [0, 1, -1], [1, 1, 0]]
    assert f.tolist() == [[0, 1, 2], [3, 5, 4]]
    assert normals_point_to_positive(v, f[:1], np.array([-1.0, -1, -1])) and normals_point_to_positive(v, f[1:], np.array([1.0, 1, -1]))
    v, f, k = gpu_mesh(ctx, voxel([-3, 1, -3, 1, 1, 1, 1, 1]))
    assert k.tolist() == [0, 1, 2, 4, 6, 11]
    assert v.tolist() == [[0.5, -1, -1], [-1, 0.5, -1], [-1, -1, 0.5], [1, -0.5, -1], [-0.5, 1, -1], [1, 1, 0.5]]
    assert f.tolist() == [[0, 3, 5], [0, 5, 2], [1, 2, 5], [1, 5, 4]]
    # one connected, consistently oriented strip: every interior edge is used once in each direction
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    und = {}
    for a, b in e.tolist():
        und.setdefault((min(a, b), max(a, b)), []).append(a < b)
    assert all(len(d) <= 2 and (len(d) == 1 or d[0] != d[1]) for d in und.values())
    assert sum(len(d) == 2 for d in und.values()) == 3


def test_single_voxel_with_centre_vertex(ctx):
    """corners (v0..v7) = -2,-2,1,-2, 1,1,-1,2: eight crossed edges; the sub-configuration carries a centre vertex
    (B.5) = mean of ALL crossed-edge vertices, listed first (key -k*NX*NY + j*NX + i - 1 = -1), and a fan of eight
    triangles around it."""
    v, f, k = gpu_mesh(ctx, voxel([-2, -2, 1, -2, 1, 1, -1, 2]))
    t = -1.0 + (-2.0 * 2.0) / -3.0                          # -2 against +1: 1/3 from the lower corner ... = 0.33333333333333326
    edge = [[-1, -1, t], [1, t, -1], [1, -1, t], [t, 1, -1], [-1, 1, 0.0], [1, 1, 0.0], [1, 0.0, 1], [t, 1, 1]]
    assert k.tolist() == [-1, 2, 4, 5, 6, 8, 11, 16, 18]
    assert v[1:].tolist() == edge
    assert v[0].tolist() == (np.sum(np.array(edge), axis=0) / 8.0).tolist() == [0.33333333333333326, 0.29166666666666663, 0.08333333333333331]
    assert f.tolist() == [[0, 1, 3], [0, 2, 4], [0, 3, 7], [0, 4, 5], [0, 5, 1], [0, 6, 2], [0, 7, 8], [0, 8, 6]]
    ring = {}
    for a, b, c in f.tolist():                              # the fan closes: the rim is one 8-cycle
        ring[b] = c
    x, seen = 1, []
    for _ in range(8):
        seen.append(x)
        x = ring[x]
    assert x == 1 and sorted(seen) == list(range(1, 9))


def _sphere(x, y, z):
    return np.sqrt(x * x + y * y + z * z) - 0.6


def _torus(x, y, z):
    q = np.sqrt(x * x + z * z) - 0.6
    return np.sqrt(q * q + y * y) - 0.25


@pytest.mark.parametrize("res", [64, 128])
@pytest.mark.parametrize("name,fn,kmax,volume,euler", [("sphere", _sphere, 1 / 0.6, 4 / 3 * np.pi * 0.6 ** 3, 2),
                                                       ("torus", _torus, 1 / 0.25, 2 * np.pi ** 2 * 0.6 * 0.25 ** 2, 0)])
def test_analytic_surface_geometry(ctx, res, name, fn, kmax, volume, euler):
    """Exact distance fields.  Linear interpolation along an edge of length h misses the surface by at most
    kappa_max * h^2 / 8; the enclosed volume of the inscribed mesh converges at second order from below; the
    triangle orientation (B.6: normals toward positive values = outward) makes the signed volume positive."""
    from fields import mesh_is_closed_manifold
    l = np.linspace(-1, 1, res + 1)
    z, y, x = np.meshgrid(l, l, l, indexing="ij")
    v, f, k = gpu_mesh(ctx, fn(x, y, z))
    h = 2.0 / res
    resid = np.abs(fn(v[:, 0], v[:, 1], v[:, 2])).max()
    assert resid <= 1.02 * kmax * h * h / 8 + 2e-7           # + float32 rounding of the samples
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    vol = float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
    assert vol > 0
    assert -3.0 * h * h <= vol - volume < 0                   # measured: -1.50 h^2 (sphere), -2.30 h^2 (torus)
    assert mesh_is_closed_manifold(f)
    assert len(v) - len(f) // 2 == euler                      # V - E + F with E = 3F/2
    print("%s res %d: max |f(v)| = %.3f h^2, volume error = %.3f h^2" % (name, res, resid / h ** 2, (vol - volume) / h ** 2))


def test_volume_converges_at_second_order(ctx):
    errs = []
    for res in (32, 64, 128):
        l = np.linspace(-1, 1, res + 1)
        z, y, x = np.meshgrid(l, l, l, indexing="ij")
        v, f, _ = gpu_mesh(ctx, _sphere(x, y, z))
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        errs.append(4 / 3 * np.pi * 0.6 ** 3 - float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0))
    assert 3.6 < errs[0] / errs[1] < 4.4 and 3.6 < errs[1] / errs[2] < 4.4


# corner magnitudes found by a search over near-ties of the float32 face-test products (tools/mc_bad_voxel_search.py):
# code 165 (v0, v2, v5, v7 positive), face tests give an opposite-face pair "both joined" next to a pair "both
# separated", which no trilinear field can produce -> ambiguityTable[13][..] = -1
BAD = {"p0": 0.6127959489822388, "p2": 0.6401180028915405, "p5": 1.804636001586914, "p7": 3.020339012145996,
       "n1": 1.1097490787506104, "n3": 1.8573377132415771, "n4": 0.9965077638626099, "n6": 1.0409380197525024}


def test_impossible_ambiguity_voxel_emits_nothing(ctx):
    """B.3: `sub < 0` -> the reference prints an internal error and emits nothing for the voxel.  Alone in a grid the
    voxel yields an empty mesh (its 12 sign-changing edges own no vertex, because vertices are created by emitting
    voxels only); inside a larger field the neighbours emit as usual.  GPU and oracle agree on both."""
    from oracle import mc
    b = BAD
    g = voxel([b["p0"], -b["n1"], b["p2"], -b["n3"], -b["n4"], b["p5"], -b["n6"], b["p7"]])
    ref = mc.marching_cubes(g, iso=0.0)
    assert ref["n_bad"] == 1 and (ref["n_centre"], ref["n_edge"], ref["n_tri"]) == (0, 0, 0)
    v, f, k = gpu_mesh(ctx, g)
    assert len(v) == 0 and len(f) == 0
    rng = np.random.default_rng(5)
    big = rng.standard_normal((5, 6, 7)).astype(np.float32)
    big[0], big[-1], big[:, 0], big[:, -1], big[:, :, 0], big[:, :, -1] = 1.0, 1.0, 1.0, 1.0, 1.0, 1.0
    big[2:4, 2:4, 3:5] = g
    ref = mc.marching_cubes(big, iso=0.0)
    assert ref["n_bad"] == 1 and ref["n_tri"] > 50
    v, f, k = gpu_mesh(ctx, big)
    assert np.array_equal(f, ref["tris"]) and np.array_equal(k, ref["keys"]) and np.array_equal(v, ref["verts"])
