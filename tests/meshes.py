"""Small closed meshes and polygon soups for the distance-field tests."""
import numpy as np


def cube(half=0.5, centre=(0, 0, 0)):
    v = np.array([[x, y, z] for z in (-half, half) for y in (-half, half) for x in (-half, half)], float) + np.asarray(centre, float)
    f = np.array([[0, 2, 1], [1, 2, 3], [4, 5, 6], [5, 7, 6], [0, 1, 4], [1, 5, 4], [2, 6, 3], [3, 6, 7], [0, 4, 2], [2, 4, 6],
                  [1, 3, 5], [3, 7, 5]], np.int32)
    return v, f


def icosphere(radius=0.6, levels=2, centre=(0, 0, 0)):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, float) / np.linalg.norm(p) for p in v]
    for _ in range(levels):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v) * radius + np.asarray(centre, float), np.asarray(f, np.int32)


def torus(R=0.55, r=0.2, nu=24, nv=12):
    u, w = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, np.arange(nv) * 2 * np.pi / nv, indexing="ij")
    v = np.stack([(R + r * np.cos(w)) * np.cos(u), r * np.sin(w), (R + r * np.cos(w)) * np.sin(u)], -1).reshape(-1, 3)
    f = []
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = i * nv + (j + 1) % nv, ((i + 1) % nu) * nv + (j + 1) % nv
            f += [(a, c, b), (b, c, d)]
    return v, np.asarray(f, np.int32)


def outward(v, f, inside_point):
    """orient every face away from inside_point (valid for the star-shaped pieces used here)"""
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    flip = (np.cross(b - a, c - a) * ((a + b + c) / 3 - np.asarray(inside_point, float))).sum(1) < 0
    f = f.copy()
    f[flip] = f[flip][:, [0, 2, 1]]
    return f


def random_soup(n, seed, spread=0.8, size=0.25):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 1, 3))
    v = (c + rng.normal(0, size, (n, 3, 3))).reshape(-1, 3)
    return v, np.arange(3 * n, dtype=np.int32).reshape(n, 3)


def soup_with_cavity_and_sheet():
    """closed box with a closed box inside it (a cavity the pipeline must fill) and an open square sheet beside it"""
    v1, f1 = cube(0.5)
    v2, f2 = cube(0.2)
    f2 = f2[:, [0, 2, 1]]                      # inner box faces the cavity
    v3 = np.array([[0.75, -0.3, -0.3], [0.75, 0.3, -0.3], [0.75, 0.3, 0.3], [0.75, -0.3, 0.3]], float)
    f3 = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return np.concatenate([v1, v2, v3]), np.concatenate([f1, f2 + 8, f3 + 16]).astype(np.int32)
