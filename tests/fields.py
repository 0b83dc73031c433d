"""Seeded synthetic SDF grids shared by the CPU and GPU tests."""
import numpy as np


def lin(res):
    return np.linspace(-1, 1, res + 1)


def analytic_field(res):
    """SURVEY.md s8d MC benchmark field: two spheres + a torus about the y axis.
    The reference tool's output sizes for it are known."""
    l = lin(res)
    z, y, x = np.meshgrid(l, l, l, indexing="ij")
    s1 = np.sqrt((x - 0.2) ** 2 + y ** 2 + z ** 2) - 0.45
    s2 = np.sqrt((x + 0.3) ** 2 + (y - 0.1) ** 2 + (z + 0.2) ** 2) - 0.35
    q = np.sqrt(x * x + z * z) - 0.6
    t = np.sqrt(q * q + (y + 0.3) ** 2) - 0.12
    return np.minimum(np.minimum(s1, s2), t).astype(np.float32)


KNOWN_ANALYTIC = {64: (8978, 17940), 128: (35690, 71380), 256: (143683, 287370)}  # (verts, tris) from the reference tool


def sphere_field(res, r=0.6):
    l = lin(res)
    z, y, x = np.meshgrid(l, l, l, indexing="ij")
    return (np.sqrt(x * x + y * y + z * z) - r).astype(np.float32)


def noise_field(shape, seed, pad=True):
    """White noise (many ambiguous cases and centre vertices).  pad=True keeps
    the outer layer positive so no centre vertex touches layer k=0
    (SURVEY.md App. B.8 -- the reference is undefined there)."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal(shape).astype(np.float32)
    if pad:
        g[0], g[-1] = 1.0, 1.0
        g[:, 0], g[:, -1] = 1.0, 1.0
        g[:, :, 0], g[:, :, -1] = 1.0, 1.0
    return g


def noisy_sphere(res, seed, amp=0.08):
    rng = np.random.default_rng(seed)
    return (sphere_field(res) + amp * rng.standard_normal((res + 1,) * 3)).astype(np.float32)


def mesh_is_closed_manifold(tris):
    """Every directed edge appears once and its reverse appears once."""
    t = np.asarray(tris, np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    key = e[:, 0] * (t.max() + 1) + e[:, 1]
    rkey = e[:, 1] * (t.max() + 1) + e[:, 0]
    u, c = np.unique(key, return_counts=True)
    return bool((c == 1).all() and np.array_equal(np.sort(u), np.unique(rkey)))
