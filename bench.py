#!/usr/bin/env python3
"""bench.py -- SDF query points/sec and ms per mesh (BASELINE.json metric).

A step = one mesh: per-image precompute (137x137 resize + projected feature
map), dense SDF evaluation of the (res+1)^3 grid, marching cubes with sorted
output.  Inputs (VGG end points of a synthetic view, random-init decoder
weights) are resident in HBM before the timed region; the VGG encoder itself
is the once-per-image plumbing north_star excludes from the hot path.

N=1: sdf_res=256 (BASELINE.json configs[1]).  N>1: ONE grid z-slab-sharded over
the ranks with per-GPU points held at ~257^3 (res = round(256*N^(1/3)); N=8 is
configs[2], sdf_res=512), RCCL broadcast of features + all-gather of counts,
mesh assembled on rank 0 -> "weak" scaling.  Halo planes are exchanged (one all-gather of three boundary
planes per rank), not recomputed, so every GPU evaluates exactly its own planes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_FLOP_PER_POINT = 3671808        # SURVEY.md s8d (global term hoisted)
EXECUTED_FLOP_PER_POINT = 2 * 2 * (64 * 256 + 256 * 512 + 512 * 512 + 512 * 256)   # MFMA layers actually run (image term pre-projected per pixel)
# MFMA pipe time per point in bf16-rate flops (2 streams x 2 flop/MAC): the bf16x3 decoder runs 3 bf16 products per
# layer; the mixed decoder 1 f16 product + 2 fp8 products at twice the rate (= 2 units) on fold1/conv2, fold1/conv3
# and fold2/conv1, and the bf16 triple on fold2/conv2
_L123, _L4 = 64 * 256 + 256 * 512 + 512 * 512, 512 * 256
MFMA_FLOP_PER_POINT = {"bf16x3": 2 * 2 * 3 * (_L123 + _L4), "mixed": 2 * 2 * (2 * _L123 + 3 * _L4),
                       "mixed4": 2 * 2 * 2 * (_L123 + _L4)}
DTYPE = {"bf16x3": "f32 (bf16x3 split MFMA, f32 accumulate)",
         "mixed": "f32 (f16 + 2 block-scaled fp8 residual MFMAs on three layers, bf16x3 on fold2/conv2; f32 accumulate)",
         "mixed4": "f32 (f16 + 2 block-scaled fp8 residual MFMAs on all four MFMA layers; f32 accumulate)"}
PEAK_BF16_TFLOPS = 2500.0           # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def cpu_baseline(res, trans_mat, feats_np, gfeat_np, weights, iso, budget_s=12.0):
    """The reference pipeline's stand-in on the host cores: oracle/net.py (torch-CPU fp32
    restatement of the TF graph) on a bounded sample of the grid + oracle/mc_oracle.c on the
    full grid.  kind="port": TensorFlow and the closed MC tool cannot run here."""
    from oracle import mc, net
    pts = net.grid_points(res)
    P = len(pts)
    n0 = min(32768, P)
    t0 = time.time()
    net.sdf_points(pts[:n0], pts[:n0], trans_mat, feats_np, gfeat_np, weights)
    dt0 = time.time() - t0
    n1 = int(max(1, min(P, max(n0, (budget_s - dt0) / max(dt0, 1e-3) * n0))))
    sel = np.linspace(0, P - 1, n1).astype(np.int64)
    t0 = time.time()
    net.sdf_points(pts[sel], pts[sel], trans_mat, feats_np, gfeat_np, weights)
    dt1 = time.time() - t0
    net_pps = n1 / dt1
    return net_pps, n1, dt1, mc


def batch_res128(ctx, enc, weights, vgg_w, n=26, lanes=3):
    """BASELINE.json configs[3]: n views (two per category) through create_sdf.BatchMesher at sdf_res 128 -- encoder,
    coarse median iso, decode, marching cubes, D2H and OBJ hand-off to the writer threads; `lanes` views in flight on
    their own contexts / HIP streams; wall time per view from the first submit to the last mesh on the host (files
    are written by the pool, off the critical path)."""
    import tempfile
    from disn_amd import create_sdf
    extra = [create_sdf.second_lane(ctx, weights, vgg_w) for _ in range(lanes - 1)]
    with tempfile.TemporaryDirectory() as td:
        bm = create_sdf.BatchMesher(ctx, enc, 128, 0.0, td, median_iso=True, extra_lanes=extra)
        views = list(create_sdf.synthetic_views(n + 2 * lanes))
        for v in views[:2 * lanes]:                  # warm-up (allocations, decoder choice of every lane)
            bm.submit(v)
        bm.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for v in views[2 * lanes:]:
            bm.submit(v)
        bm.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stats = bm.finish()
    for lane in extra:
        lane[0].close()
    return {"views": n, "sdf_res": 128, "lanes": lanes, "ms_per_view": dt / n * 1e3, "views_per_s": n / dt,
            "faces_per_view": float(np.mean([s[2] for s in stats[2 * lanes:]])),
            "note": "wall per view incl. VGG encoder, coarse-grid median iso, decode, MC, D2H; %d views in flight on their own "
                    "contexts and HIP streams; OBJ text written by 4 pool threads" % lanes}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves as fresh children
    (one process per GPU under torch.distributed.run), BEFORE this process touches the GPU -- a process that has
    initialised HIP must never be replaced or forked into ranks.  The children print; rank 0's JSON line is
    relayed on our stdout and any child failure is our exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if proc.returncode != 0 or len(lines) != 1:
        print("bench.py: %d-rank launch failed (exit %d, %d result lines)" % (n, proc.returncode, len(lines)),
              file=sys.stderr)
        raise SystemExit(proc.returncode or 1)
    print(lines[0], flush=True)
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=0, help="sdf_res (default 256 at N=1, round(256*N^(1/3)) otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the other_kernels.batch_res128 measurement "
                    "(keeps a profile of this command to launches of the benchmark grid only)")
    ap.add_argument("--decoder", choices=("auto", "mixed4", "mixed", "bf16x3"), default="auto",
                    help="decoder arithmetic; auto (default) = disn_sdf_autotune: the cheapest one whose error, measured on the "
                         "loaded weights and image against the as-written fp32 graph, keeps 5x margin to the 1e-4 tolerance")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])          # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    comm = None
    local_rank = local_rank % max(torch.cuda.device_count(), 1)   # identity on a node with one GPU per rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        import torch.distributed as dist
        backend = os.environ.get("DISN_DIST_BACKEND", "nccl")   # "gloo" only to rehearse N ranks on one GPU
        comm = backend
        if backend == "nccl":
            # RCCL over xGMI.  If the communicator cannot be built or its first collective fails (every rank sees the
            # same failure: IPC mode, several ranks on one GPU), fall back to gloo with host staging rather than
            # report nothing -- the line then says so in config.comm.
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("all_reduce probe returned %r" % probe.item())
                comm = "rccl"
            except Exception as e:   # noqa: BLE001
                print("bench.py rank %d: RCCL unavailable (%s: %s); falling back to gloo with host staging"
                      % (rank, type(e).__name__, str(e).splitlines()[0][:200]), file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:   # noqa: BLE001
                    pass
                port = int(os.environ.get("MASTER_PORT", "29500")) + 1
                dist.init_process_group("gloo", init_method="tcp://%s:%d" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), port),
                                        rank=rank, world_size=world)
                comm = "gloo (fallback: RCCL unavailable)"
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from disn_amd import synth
    from disn_amd.api import Context
    from disn_amd.encoder import Vgg16Encoder
    from disn_amd.pipeline import MeshPipeline, coarse_median_iso, median_iso

    res = args.res or (256 if world == 1 else int(round(256 * world ** (1.0 / 3.0))))
    ctx = Context(local_rank)
    ctx.set_decoder(args.decoder)
    weights = synth.decoder_weights(seed=0)
    ctx.set_weights(weights)
    vgg_w = synth.vgg_weights(seed=0)
    enc = Vgg16Encoder(ctx, vgg_w)
    gfeat, feats = enc(synth.image(seed=0))          # once per image, outside the timed region
    if world > 1 and rank != 0:                      # non-root ranks only hold receive buffers
        feats = [torch.zeros_like(f) for f in feats]
        gfeat = torch.zeros_like(gfeat)
    pipe = MeshPipeline(ctx, rank, world, dist)
    tm = synth.GT_TRANS_MAT

    # iso: median of the predicted grid so that random-init weights give a surface
    pipe.run(feats, gfeat, tm, res, want_mesh=False)   # (with --decoder auto this first evaluation also picks the arithmetic)
    decoder = ctx.decoder
    if world == 1:
        iso = median_iso(pipe.last_sdf)
    else:
        # a rank holds only its slab: the median of a coarse grid over the WHOLE box, which every rank
        # computes from the (broadcast, bit-identical) features with the same kernel, then agreed on
        iso_t = torch.tensor([coarse_median_iso(ctx, tm, res=64)], device=ctx.device)
        pipe.dist.broadcast(iso_t, src=0)
        iso = float(iso_t.item())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        verts, tris = pipe.run(feats, gfeat, tm, res, iso)
    pipe.time_decode = True
    pipe.decode_events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        verts, tris = pipe.run(feats, gfeat, tm, res, iso)
    barrier()
    dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())

    points = (res + 1) ** 3
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b in pipe.decode_events]))
    local_pts = pipe.last_points
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        alg_tflops = local_pts * ALG_FLOP_PER_POINT / (dec_ms * 1e-3) / 1e12
        mfma_tflops = local_pts * MFMA_FLOP_PER_POINT[decoder] / (dec_ms * 1e-3) / 1e12
        out = {
            "metric": "SDF query points/sec (node), ms per mesh incl. marching cubes",
            "value": points / (dt / args.steps), "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[decoder] + ("; plan chosen by disn_sdf_autotune: measured max |error| %.2e on %d points, tolerance "
                                       "%g / margin %g" % (ctx.decoder_report["errors"][decoder], ctx.decoder_report["points"],
                                                           ctx.decoder_report["tolerance"], ctx.decoder_report["margin"])
                                       if args.decoder == "auto" else "; pinned by --decoder"),
            "data": "synthetic",
            "config": {"workload": "sdf_res=%d two-stream, full sorted mesh output in HBM, %d z-slab(s); device->host copy and "
                                   "OBJ text excluded (other_kernels.mesh_output)" % (res, world),
                       "sdf_res": res, "decoder": decoder, "decoder_selection": args.decoder, "points": points, "mesh_vertices": int(verts.shape[0]),
                       "mesh_triangles": int(tris.shape[0]), "iso": iso,
                       "per_gpu_points_evaluated": local_pts, "halo": pipe.halo if world > 1 else None, "comm": comm,
                       "stitch": pipe.last_stitch},
            "roofline": {"bound": "mfma", "kernel": "sdf_decode" if decoder == "bf16x3" else "sdf_decode_mx", "achieved": alg_tflops, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": alg_tflops / PEAK_BF16_TFLOPS, "traffic": None,
                         "kernel_ms": dec_ms, "launches_per_pass": 1 if decoder == "bf16x3" else 2,
                         "alg_flop_per_point": ALG_FLOP_PER_POINT, "executed_flop_per_point": EXECUTED_FLOP_PER_POINT,
                         "executed_mfma_tflops_bf16_rate": mfma_tflops, "mfma_issue_frac": mfma_tflops / PEAK_BF16_TFLOPS},
        }
        if world == 1 and pipe.mc_events:
            # the HBM-bound parts: marching cubes against its algorithmic bytes
            # 4 B/grid point + 24 B/vertex + 12 B/triangle, and the per-image precompute
            mc_ms = float(np.mean([a.elapsed_time(b) for a, b in pipe.mc_events]))
            img_ms = float(np.mean([a.elapsed_time(b) for a, b in pipe.image_events]))
            mc_bytes = 4 * points + 24 * int(verts.shape[0]) + 12 * int(tris.shape[0])
            out["other_kernels"] = {
                "marching_cubes": {"bound": "hbm", "ms": mc_ms, "algorithmic_bytes": mc_bytes,
                                   "achieved": mc_bytes / (mc_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": mc_bytes / (mc_ms * 1e-3) / 1e9 / 8000.0,
                                   "note": "count + emit + triangle sort (radix on the first index + run ordering), incl. the host sync that returns the counts"},
                "per_image_precompute": {"ms": img_ms, "note": "5 x resize to 137x137 + fp32 MFMA GEMM 18769x1472x512 + bias fold"},
            }
        if world == 1:
            # what follows the timed region when the caller wants result.obj (reference demo/demo.py:355-369)
            import tempfile
            from disn_amd import api
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            vh, th = verts.cpu().numpy(), tris.cpu().numpy()
            t2 = time.perf_counter()
            with tempfile.TemporaryDirectory() as td:
                api.write_obj(os.path.join(td, "result.obj"), vh, th)
                t3 = time.perf_counter()
                obj_bytes = os.path.getsize(os.path.join(td, "result.obj"))
            out.setdefault("other_kernels", {})["mesh_output"] = {
                "d2h_ms": (t2 - t1) * 1e3, "obj_write_ms": (t3 - t2) * 1e3, "obj_bytes": obj_bytes,
                "note": "NOT in ms_per_step: copy of the sorted mesh to pageable host memory, then disn_write_obj "
                        "(multi-threaded text formatting) into a temporary directory"}
        # HBM-side traffic of the dominant kernel, measured separately with rocprofv3 PMC passes
        # (tools/pmc_decode.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md) and committed under profiles/
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if res == tr.get("sdf_res"):
                out["roofline"]["traffic"] = tr["sdf_decode_bytes_per_launch"]
                out["roofline"]["traffic_note"] = tr["note"]
        except (OSError, ValueError, KeyError):
            pass
        if world == 1 and not args.no_batch:
            out.setdefault("other_kernels", {})["batch_res128"] = batch_res128(ctx, enc, weights, vgg_w)
        if world == 1 and not args.no_cpu_baseline:
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            feats_np = [f.cpu().numpy() for f in feats]
            net_pps, n1, dt1, mc = cpu_baseline(res, tm, feats_np, gfeat.cpu().numpy(), weights, iso)
            sdf_np = pipe.last_sdf.cpu().numpy()
            t1 = time.time()
            ref = mc.marching_cubes(sdf_np, iso=iso)
            mc_s = time.time() - t1
            cpu_s = points / net_pps + mc_s
            out["cpu_baseline"] = {
                "value": points / cpu_s, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "oracle/net.py on %d of %d grid points (%.1f s, scaled linearly) + oracle/mc_oracle.c on the "
                          "full grid (%.2f s)" % (n1, points, dt1, mc_s),
                "ms_per_mesh": cpu_s * 1e3,
                "mesh_matches_gpu": bool(np.array_equal(ref["tris"], tris.cpu().numpy())),
            }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
