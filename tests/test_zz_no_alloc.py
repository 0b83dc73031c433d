"""disn_sort_triangles_static -- the allocation-free entry point (include/disn.h): same result as
disn_sort_triangles, scratch in a static device array, no hipMalloc.  Through the C ABI (ctypes), bit-exact bar
(integer work).  Two calls in a row, the second on a smaller input: the static array keeps the first call's
contents and the second must not be disturbed by them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tris(n, max_index, seed):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, max_index, size=(n, 3), dtype=np.int64).astype(np.int32)
    t[: n // 8] = t[n // 8: 2 * (n // 8)][: n // 8]          # duplicates: ties on every key
    return t


def _both(ctx, t, max_index):
    a = torch.from_numpy(t.copy()).to(ctx.device)
    b = torch.from_numpy(t.copy()).to(ctx.device)
    ctx.sort_triangles(a, max_index)                  # existing entry point (ctx-owned hipMalloc scratch)
    ctx.sort_triangles(b, max_index, static=True)     # allocation-free entry point
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_static_sort_equals_allocating_sort_twice(ctx):
    for n, max_index, seed in ((300_000, 151_234, 1), (4_097, 70_000, 2), (1, 5, 3), (0, 5, 4)):
        t = _tris(n, max_index, seed)
        a, b = _both(ctx, t, max_index)
        assert np.array_equal(a, b), (n, max_index)
        if n:
            want = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
            assert np.array_equal(b, want)


def test_static_sort_full_size_and_limit(ctx):
    from disn_amd.api import DisnError
    n = 1 << 20
    t = _tris(n, 2_000_000_000, 5)                    # 31-bit indices: every radix pass runs
    a, b = _both(ctx, t, 2_000_000_000)
    assert np.array_equal(a, b)
    big = torch.zeros((n + 1, 3), dtype=torch.int32, device=ctx.device)
    with pytest.raises(DisnError, match="static scratch limit"):
        ctx.sort_triangles(big, 10, static=True)
    # a rejected call leaves the entry point usable
    t = _tris(1000, 999, 6)
    a, b = _both(ctx, t, 999)
    assert np.array_equal(a, b)


def test_sort_with_long_runs_of_equal_first_index(ctx):
    """The public sort entry points take arbitrary triangle lists: a fan of 300 000 triangles around one smallest
    vertex, an all-equal first column, and runs just below / above the look-ahead bound of the run-ordering kernel
    (the long-run inputs take the gated full radix passes instead) -- both entry points against np.lexsort."""
    rng = np.random.default_rng(11)
    cases = []
    fan = rng.integers(1, 50_000, size=(300_000, 3)).astype(np.int32)
    fan[:, 0] = 0
    cases.append((fan, 50_000))
    allsame = rng.integers(0, 1 << 20, size=(70_001, 3)).astype(np.int32)
    allsame[:, 0] = 777
    cases.append((allsame, 1 << 20))
    for run in (63, 64, 65, 66, 130):                 # runs around TIES_MAX_LOOK = 64
        t = rng.integers(0, 3000, size=(20_000, 3)).astype(np.int32)
        t[5000:5000 + run, 0] = 1234
        cases.append((t, 3000))
    mixed = _tris(100_000, 900, 12)                   # ~111 triangles per first index on average
    cases.append((mixed, 900))
    for t, max_index in cases:
        a, b = _both(ctx, t, max_index)
        want = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
        assert np.array_equal(a, want), (len(t), max_index)
        assert np.array_equal(b, want), (len(t), max_index)
    # a short-run input right after a long-run one: the flag of the previous call must not leak
    t = _tris(10_000, 70_000, 13)
    a, b = _both(ctx, t, 70_000)
    want = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
    assert np.array_equal(a, want) and np.array_equal(b, want)
