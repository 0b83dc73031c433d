"""Search for float32 corner values whose face tests give a pattern that no trilinear field has,
i.e. a voxel with ambiguityTable = -1.  Such patterns exist only for the fully alternating codes 90 / 165 (case 13):
with L_f = (product of the face's two positive corners) > (product of its two negative corners), opposite faces obey
L1*L3 = L2*L4 = L5*L6 in exact arithmetic, so "one opposite pair both true, another both false" is impossible -- and
needs all four products within rounding of a tie to happen in float32.  Prints the first hit (used by
tests/test_gpu_mc_geometry.py)."""
import numpy as np

f32 = np.float32
rng = np.random.default_rng(0)


def ulps(x, k):
    return (x.view(np.int32) + k).view(f32)


def gt(a, b, c, d):
    return (a * b).astype(f32) > (c * d).astype(f32)


for it in range(40):
    N = 1_000_000
    p0, p2, p5, n1, n3 = [rng.uniform(0.5, 2.0, N).astype(f32) for _ in range(5)]
    n4 = ulps((p0 * p5 / n1).astype(f32), rng.integers(-2, 3, N).astype(np.int32))
    n6 = ulps((p2 * p5 / n1).astype(f32), rng.integers(-2, 3, N).astype(np.int32))
    p7 = ulps((n3 * p5 / n1).astype(f32), rng.integers(-2, 3, N).astype(np.int32))
    # the face tests exactly as evaluated (A*C > B*D with A the face's first corner): faces 2, 3 and 6 start on a
    # negative corner, so their comparison is (negatives' product > positives' product) -- ties matter here
    L = [gt(p0, p5, n1, n4), ~gt(n1, n6, p2, p5), ~gt(n3, n6, p2, p7), gt(p0, p7, n3, n4), gt(p0, p2, n1, n3),
         ~gt(n4, n6, p5, p7)]
    pairs = [(L[0], L[2]), (L[1], L[3]), (L[4], L[5])]
    both = [a & b for a, b in pairs]
    none = [~a & ~b for a, b in pairs]
    bad = (both[0] | both[1] | both[2]) & (none[0] | none[1] | none[2])
    if bad.any():
        k = np.flatnonzero(bad)[0]
        print({n: float(x[k]) for n, x in zip("p0 p2 p5 p7 n1 n3 n4 n6".split(), (p0, p2, p5, p7, n1, n3, n4, n6))})
        break
