"""GPU marching cubes (through the C ABI) against the CPU oracle.
Bar (BASELINE.json north_star): case indices and triangle topology bit-exact,
vertices within 1e-5 (they are in fact required to be bit-equal here, both
sides evaluate the same float64 expressions)."""
import numpy as np
import pytest
import torch

import fields
from oracle import mc

pytestmark = pytest.mark.gpu


def run_gpu(ctx, g, iso=0.0, bbox=(-1, -1, -1, 1, 1, 1)):
    t = torch.from_numpy(np.ascontiguousarray(g)).cuda()
    v, f, k = ctx.marching_cubes(t, iso=iso, bbox=bbox, want_keys=True)
    torch.cuda.synchronize()
    return v.cpu().numpy(), f.cpu().numpy(), k.cpu().numpy(), ctx._mc_counts


def check(ctx, g, iso=0.0, bbox=(-1, -1, -1, 1, 1, 1)):
    ref = mc.marching_cubes(g, iso=iso, bbox=bbox)
    v, f, k, counts = run_gpu(ctx, g, iso, bbox)
    assert counts == (ref["n_centre"], ref["n_edge"], ref["n_tri"])
    assert np.array_equal(f, ref["tris"])
    assert np.array_equal(k, ref["keys"])
    same = bool(np.array_equal(v, ref["verts"], equal_nan=True))   # inf where c_a == c_b underflows (B.9)
    fin = np.isfinite(ref["verts"])
    err = float(np.abs(v[fin] - ref["verts"][fin]).max(initial=0.0))
    assert err <= 1e-5, err
    assert same
    return ref


def test_case_indices_bit_exact(ctx):
    g = fields.noise_field((17, 15, 19), seed=7)
    t = torch.from_numpy(g).cuda()
    ctx.mc_count(t, 0.1)
    info = ctx.mc_debug_info().cpu().numpy().astype(np.uint16)
    sub = (info & 0x3ff).astype(np.int32)
    sub[sub == 0x3ff] = -1
    assert np.array_equal(sub, mc.classify(g, 0.1).astype(np.int32))


@pytest.mark.parametrize("seed", range(4))
def test_noise_fields(ctx, seed):
    ref = check(ctx, fields.noise_field((9 + seed, 9, 9 + 2 * seed), seed=seed), iso=0.05)
    assert ref["n_centre"] > 0


@pytest.mark.parametrize("seed", range(3))
def test_noisy_spheres_noncubic_bbox(ctx, seed):
    check(ctx, fields.noisy_sphere(16, seed), bbox=(-1, -0.5, -2, 1, 0.5, 2))


def test_golden_fixture(ctx):
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mc_noise12.npz"))
    v, f, k, _ = run_gpu(ctx, z["grid"], float(z["iso"]), tuple(z["bbox"]))
    assert np.array_equal(f, z["tris"]) and np.array_equal(v, z["verts"]) and np.array_equal(k, z["keys"])


def test_unpadded_noise_touching_every_border(ctx):
    # centre vertices in layer k=0 are where the reference tool is undefined
    # (SURVEY.md App. B.8); ours and the oracle share the defined-away rule
    check(ctx, fields.noise_field((8, 8, 8), seed=11, pad=False))


def test_edge_cases(ctx):
    check(ctx, np.ones((2, 2, 2), np.float32))                      # smallest grid, no surface
    g = np.ones((2, 2, 2), np.float32); g[0, 0, 0] = -1
    check(ctx, g)                                                   # one triangle
    l = fields.lin(8)
    z, y, x = np.meshgrid(l, l, l, indexing="ij")
    check(ctx, x.astype(np.float32))                                # samples exactly on the iso value (B.1)
    tiny = np.full((4, 4, 4), 1e-30, np.float32); tiny[1:3, 1:3, 1:3] = -1e-30
    check(ctx, tiny)                                                # products underflow to zero (B.9)
    tiny2 = np.full((5, 5, 5), 1e-30, np.float32); tiny2[2, 2, 2] = -1.0
    check(ctx, tiny2)
    check(ctx, fields.sphere_field(8), iso=0.25)
    with pytest.raises(Exception):
        ctx.mc_count(torch.zeros((1, 4, 4), device="cuda"))


@pytest.mark.parametrize("res", [64, 128])
def test_analytic_field_matches_oracle_and_reference_counts(ctx, res):
    ref = check(ctx, fields.analytic_field(res))
    assert (len(ref["verts"]), ref["n_tri"]) == fields.KNOWN_ANALYTIC[res]


def test_res256_known_answer_and_invariants(ctx):
    # full-size config: size-independent properties + the reference tool's counts
    g = fields.analytic_field(256)
    v, f, k, counts = run_gpu(ctx, g)
    assert (len(v), len(f)) == fields.KNOWN_ANALYTIC[256]
    assert (np.diff(k) > 0).all()
    assert (f[:, 0] < f[:, 1]).all() and (f[:, 0] < f[:, 2]).all()
    order = np.lexsort((f[:, 2], f[:, 1], f[:, 0]))
    assert np.array_equal(order, np.arange(len(f)))
    assert fields.mesh_is_closed_manifold(f)
    assert len(v) - 3 * len(f) // 2 + len(f) == -2                # the primitives merge into one genus-2 surface


@pytest.mark.parametrize("nslab", [2, 3])
def test_slabs_reproduce_whole_grid(ctx, nslab):
    """z-slab partition with halo planes: stitched slabs == single pass."""
    from disn_amd.slabs import slab_ranges, stitch_slabs
    g = fields.noise_field((21, 12, 13), seed=5)
    nz = g.shape[0]
    ref = mc.marching_cubes(g, iso=0.02, bbox=(-1, -1, -1, 1, 1, 2))
    parts = []
    full = torch.from_numpy(g).cuda()
    ranges = slab_ranges(nz, nslab)
    counts = []
    for (lo, hi, own0, own1) in ranges:
        sub = full[lo:hi].contiguous()
        counts.append(ctx.mc_count(sub, 0.02, z_offset=lo, nz_global=nz, own=(own0, own1)))
        parts.append((sub, lo, own0, own1))
    from disn_amd.slabs import slab_bases
    bases = slab_bases(counts)
    outs = []
    for (sub, lo, own0, own1), (cb, eb) in zip(parts, bases):
        ctx.mc_count(sub, 0.02, z_offset=lo, nz_global=nz, own=(own0, own1))
        v, f, k = ctx.mc_emit((-1, -1, -1, 1, 1, 2), centre_base=cb, edge_base=eb, sort=False, want_keys=True)
        outs.append((v.cpu().numpy(), f.cpu().numpy(), k.cpu().numpy()))
    verts, tris, keys = stitch_slabs(ctx, counts, outs)
    assert np.array_equal(tris, ref["tris"])
    assert np.array_equal(keys, ref["keys"])
    assert np.array_equal(verts, ref["verts"])


def test_reference_narrowband_fixture(ctx, tmp_path):
    """GPU on the reference's own shipped grid (narrow-band file): equals the oracle, and both give the
    reference tool's reported size 3800 v / 7604 t."""
    from disn_amd import api
    from test_mc_oracle import _narrowband_file
    p, z = _narrowband_file(tmp_path)
    res, bbox, grid, _ = api.read_narrowband_dist(p)
    ref = check(ctx, grid, 0.0, tuple(bbox))
    assert (len(ref["verts"]), ref["n_tri"]) == (3800, 7604)


def test_contexts_are_reentrant_across_threads():
    """the reference calls the tool from 4 pool threads at once (demo/demo.py:291,338); the
    replacement promises one independent DisnCtx per thread.  Four threads, each with its own context and
    stream, mesh different fields concurrently; every result must equal the oracle's."""
    import threading
    from disn_amd.api import Context
    jobs = [(fields.noise_field((20 + 3 * t, 18, 22), seed=40 + t), 0.05 * t) for t in range(4)]
    refs = [mc.marching_cubes(g, iso=iso) for g, iso in jobs]
    out, errs = [None] * 4, []

    def work(t):
        try:
            c = Context(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(5):
                    g = torch.from_numpy(jobs[t][0]).cuda()
                    v, f = c.marching_cubes(g, iso=jobs[t][1])
                s.synchronize()
                out[t] = (v.cpu().numpy(), f.cpu().numpy())
            c.close()
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for (v, f), ref in zip(out, refs):
        assert np.array_equal(f, ref["tris"]) and np.array_equal(v, ref["verts"], equal_nan=True)


@pytest.mark.parametrize("shape", [(2, 2, 2), (2, 3, 5), (3, 2, 2), (5, 4, 3), (4, 4, 4), (9, 2, 7), (3, 3, 1031),
                                   (2, 1029, 3), (6, 5, 4), (33, 2, 2), (2, 2, 260)])
def test_tiny_and_skewed_grids(ctx, shape):
    """rows shorter than a thread's run of 4 points, rows longer than a tile, tiles that end mid-row:
    the boundary path of every kernel (shape is (nz, ny, nx))"""
    g = fields.noise_field(shape, seed=sum(shape))
    check(ctx, g, iso=0.0)
    check(ctx, g, iso=0.2, bbox=(-1, -2, -3, 2, 1, 0.5))


def test_classification_matches_oracle_per_point(ctx):
    """The walking classifier's per-point record (disn_mc_debug_info: sub-table id of the point's voxel and its
    three edge-slot flags) against the oracle on noise (every voxel mixed), an analytic field, slabs with halo
    and rows that need padding to the arrays' row pitch (nx = 21, 6, 30, 49, 97: none a multiple of 4)."""
    cases = [(fields.noise_field((9, 13, 21), seed=3), 0.01, {}), (fields.noise_field((37, 5, 6), seed=4), -0.02, {}),
             (fields.analytic_field(48), 0.0, {}),
             (fields.noise_field((23, 17, 30), seed=5), 0.0, dict(z_offset=4, nz_global=40, own=(1, 12))),
             (fields.analytic_field(96)[10:41], 0.0, dict(z_offset=10, nz_global=97, own=(1, 29)))]
    for g, iso, kw in cases:
        g = np.ascontiguousarray(g)
        counts = ctx.mc_count(torch.from_numpy(g).cuda(), iso, **kw)
        assert counts[2] > 0
        info = ctx.mc_debug_info().cpu().numpy().astype(np.uint16)
        sub = (info & 0x3ff).astype(np.int32)
        sub[sub == 0x3ff] = -1
        assert np.array_equal(sub, mc.classify(g, iso).astype(np.int32))
        ref = mc.marching_cubes(g, iso=iso)
        want = np.zeros(g.size, np.uint16)
        ek = ref["keys"][ref["n_centre"]:]
        np.bitwise_or.at(want, ek // 3, (1 << (ek % 3)).astype(np.uint16))
        assert np.array_equal((info >> 10).reshape(-1), want)


def test_tool_command_line(tmp_path):
    """`computeMarchingCubes <dist> <obj> [-i iso] [-n narrowband]` (demo/demo.py:365-369) as
    `python -m disn_amd.marching_cubes`: the dense .dist of the analytic field and the reference's own narrow-band file
    through files only; mesh sizes = the tool's, OBJ text = the oracle's mesh in the tool's format."""
    from disn_amd import api, marching_cubes as cli
    from test_mc_oracle import _narrowband_file
    g = fields.analytic_field(64)
    dist, obj = str(tmp_path / "result.dist"), str(tmp_path / "result.obj")
    api.write_dist(dist, 64, (-1, -1, -1, 1, 1, 1), g)
    cli.main([dist, obj, "-i", "0.0"])
    lines = open(obj).read().splitlines()
    assert lines[0].startswith("#") and "Number of vertices: 8978" in lines[1] and "Number of faces: 17940" in lines[2]
    ref = mc.marching_cubes(g)
    want = str(tmp_path / "want.obj")
    api.write_obj(want, ref["verts"], ref["tris"])
    assert open(obj, "rb").read() == open(want, "rb").read()
    cli.main([dist, obj, "-i", "0.05"])
    assert "Number of vertices: %d" % len(mc.marching_cubes(g, iso=0.05)["verts"]) in open(obj).read(200)
    p, _ = _narrowband_file(tmp_path)
    cli.main([p, obj, "-n", "1"])
    head = open(obj).read(200)
    assert "Number of vertices: 3800" in head and "Number of faces: 7604" in head
    with pytest.raises(SystemExit):
        cli.main([str(tmp_path / "missing.dist"), obj])
