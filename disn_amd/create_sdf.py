"""Batch driver: the `test/create_sdf.py --create_obj` surface of the reference
(reference: test/create_sdf.py:25-62 flags, :87-95 output layout, :224-289 per-view
loop, :305-317 create_obj) on the HIP path.

    python -m disn_amd.create_sdf --img_feat_twostream --sdf_res 128 --synthetic 16 --log_dir /tmp/exp
    torchrun --nproc-per-node 8 -m disn_amd.create_sdf ... --synthetic 1024      (BASELINE.json configs[3])

Per view: image -> VGG end points (once), [--cam_est: estimated camera], the
(sdf_res+1)^3 grid spanning that object's own `sdf_params` box
(test/create_sdf.py:246-249), marching cubes, and
`<log_dir>/test_objs/[camest_]<res+1>_<iso>/<cat>/<cat>_<obj>_<view>.obj`.
The .dist file the reference writes and deletes again (:311-315) is skipped
unless --store_dist.  OBJ text is written by a small thread pool (the reference
uses 4 threads around its subprocess, :232) while the GPU goes on.

Views come from one of
  --synthetic N     seeded synthetic views (seed = index, category = index % 13)
  --views_dir DIR   one .npz per view: img_arr (H,W,3|4 uint8, BGR(A) as cv2 / the h5 hold it),
                    trans_mat (4,3), sdf_params (6), optional cat_id / obj_nm / view_id
  --h5              the reference's own h5 layout (data/data_sdf_h5_queue.py:137,153-186);
                    needs h5py, which this image does not have -> fails loudly.
Multi-GPU = plain replicas (batched config): rank r takes views
r, r+world, ...; there is no data-path collective.
"""
import argparse
import glob
import os
import queue
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

# reference preprocessing/info.json:3-17 (category name -> ShapeNet synset), in all_cats order (:18-32)
CATS = {
    "airplane": "02691156", "bench": "02828884", "cabinet": "02933112", "car": "02958343", "chair": "03001627",
    "display": "03211117", "lamp": "03636649", "speaker": "03691459", "rifle": "04090263", "sofa": "04256520",
    "table": "04379243", "phone": "04401088", "watercraft": "04530566",
}
ALL_CATS = list(CATS)


def result_obj_path(log_dir, res, iso, cam_est):
    """reference test/create_sdf.py:87-93"""
    tag = "%d_%s" % (res + 1, str(iso))
    return os.path.join(log_dir, "test_objs", ("camest_" + tag) if cam_est else tag)


def obj_file_name(root, cat_id, obj_nm, view_id):
    """reference test/create_sdf.py:305-311"""
    if not isinstance(view_id, str):
        view_id = "%02d" % view_id
    return os.path.join(root, cat_id, "%s_%s_%s.obj" % (cat_id, obj_nm, view_id))


def prepare_image(img_arr):
    """h5 `img_arr` -> network input: first three channels / 255 (reference data/data_sdf_h5_queue.py:166-186)."""
    a = np.asarray(img_arr)
    if a.ndim != 3 or a.shape[2] < 3:
        raise ValueError("img_arr must be HxWx3 or HxWx4, got %s" % (a.shape,))
    a = a[:, :, :3]
    if a.dtype != np.float32 or a.max() > 1.0:
        a = np.clip(a.astype(np.float32), 0, 255) / np.float32(255.0)
    return np.ascontiguousarray(a, dtype=np.float32)


def synthetic_views(n, first=0):
    """image seed = index, category = index % 13; every third object gets a non-unit box."""
    from . import synth
    for idx in range(first, first + n):
        rng = np.random.default_rng(1000 + idx)
        if idx % 3 == 2:
            half = 0.6 + 0.35 * rng.random(3)
            centre = (rng.random(3) - 0.5) * 0.1
            params = np.concatenate([centre - half, centre + half])
        else:
            params = np.array([-1, -1, -1, 1, 1, 1], dtype=np.float64)
        yield {"img": synth.image(idx), "trans_mat": np.asarray(synth.GT_TRANS_MAT, dtype=np.float32).reshape(4, 3),
               "sdf_params": params.astype(np.float64), "cat_id": CATS[ALL_CATS[idx % 13]],
               "obj_nm": "synthetic%06d" % idx, "view_id": idx % 24}


def npz_views(directory):
    files = sorted(glob.glob(os.path.join(directory, "*.npz")))
    if not files:
        raise SystemExit("create_sdf: no .npz views in %s" % directory)
    for f in files:
        with np.load(f, allow_pickle=False) as z:
            stem = os.path.splitext(os.path.basename(f))[0]
            yield {"img": prepare_image(z["img_arr"]), "trans_mat": z["trans_mat"].astype(np.float32).reshape(4, 3),
                   "sdf_params": z["sdf_params"].astype(np.float64).reshape(6),
                   "cat_id": str(z["cat_id"]) if "cat_id" in z else "00000000",
                   "obj_nm": str(z["obj_nm"]) if "obj_nm" in z else stem,
                   "view_id": int(z["view_id"]) if "view_id" in z else 0}


def h5_views(rendered_dir, sdf_dir, listinfo, cam_est):
    """The reference's data layout: <rendered_dir>/<cat>/<obj>/<view:02d>.h5 with img_arr + trans_mat,
    <sdf_dir>/<cat>/<obj>/ori_sample.h5 with sdf_params (reference data/data_sdf_h5_queue.py:96-118,137,153-186)."""
    try:
        import h5py
    except ImportError:
        raise SystemExit("create_sdf: --h5 needs h5py, which is not installed here; convert the views to .npz "
                         "(keys img_arr, trans_mat, sdf_params) and use --views_dir")
    for cat_id, obj, view in listinfo:
        with h5py.File(os.path.join(sdf_dir, cat_id, obj, "ori_sample.h5"), "r") as f:
            params = f["sdf_params"][:].astype(np.float64).reshape(6)
        with h5py.File(os.path.join(rendered_dir, cat_id, obj, "%02d.h5" % view), "r") as f:
            img = prepare_image(f["img_arr"][:])
            tm = f["trans_mat"][:].astype(np.float32).reshape(4, 3)
        yield {"img": img, "trans_mat": tm, "sdf_params": params, "cat_id": cat_id, "obj_nm": obj, "view_id": view}


def read_list(lst_dir, cat_ids, view_num, seed=0):
    """reference test/create_sdf.py:119-127 (<cat>_test.lst, view_num random renders of 24 per object)"""
    import random
    rnd = random.Random(seed)
    out = []
    for cat_id in cat_ids:
        with open(os.path.join(lst_dir, cat_id + "_test.lst")) as f:
            for line in f.read().splitlines():
                for render in rnd.sample(range(24), view_num):
                    out.append((cat_id, line.strip(), render))
    return out


class _Lane:
    """One context + encoder (+ camera network) on its own HIP stream, fed by its own thread."""

    def __init__(self, ctx, encoder, cam):
        from .pipeline import MeshPipeline
        self.ctx, self.enc, self.cam = ctx, encoder, cam
        self.pipe = MeshPipeline(ctx)
        self.stream = None
        self.thread = None
        self.queue = None


class BatchMesher:
    """One GPU's share of the job: views in, OBJ files out.

    With `extra_lanes` (more (ctx, encoder, cam) triples on the same GPU, each with its own weights and per-image
    buffers) views are dealt round-robin to one worker thread per lane, each on its own HIP stream: while one view's
    decoder occupies the CUs, the other lane's encoder, coarse-iso launch, marching cubes and device->host copies
    run in its shadow and its next decode is already queued -- the GPU never waits for the host.  This replaces the
    reference's overlap of MC subprocesses with the next batch (reference demo/demo.py:291,338)."""

    def __init__(self, ctx, encoder, res, iso, out_root, cam=None, writers=4, store_dist=False, median_iso=False,
                 extra_lanes=()):
        self.ctx, self.enc, self.res, self.iso, self.root = ctx, encoder, res, iso, out_root
        self.cam = cam                    # (camera Vgg16Encoder, camera weights) or None
        self.lanes = [_Lane(ctx, encoder, cam)] + [_Lane(*l) for l in extra_lanes]
        self.pipe = self.lanes[0].pipe
        self.pool = ThreadPoolExecutor(max_workers=writers)
        self.store_dist, self.median_iso = store_dist, median_iso
        self.pending = []
        self.stats = []                   # (obj file, vertices, faces, gpu ms), in submission order after finish()
        self._n = 0
        self._lock = threading.Lock()
        self._error = None
        if len(self.lanes) > 1:
            for lane in self.lanes:
                lane.stream = torch.cuda.Stream(device=lane.ctx.device)
                lane.queue = queue.Queue(maxsize=2)
                lane.thread = threading.Thread(target=self._work, args=(lane,), daemon=True)
                lane.thread.start()

    def _write(self, path, verts, tris, sdf, params):
        from . import api
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if sdf is not None:
            api.write_dist(path[:-4] + ".dist", self.res, params, sdf)
        api.write_obj(path, verts, tris)
        return path

    def _work(self, lane):
        torch.cuda.set_device(lane.ctx.device)
        with torch.cuda.stream(lane.stream):
            while True:
                item = lane.queue.get()
                if item is None:
                    return
                try:
                    if self._error is None:
                        self._mesh(lane, *item)
                except BaseException as e:     # surfaced by submit() / finish(): nothing fails silently
                    self._error = e

    def _mesh(self, lane, seq, view, path):
        from .pipeline import coarse_median_iso
        t0 = time.perf_counter()
        gfeat, feats = lane.enc(view["img"])
        trans_mat = view["trans_mat"]
        if lane.cam is not None:
            from . import camera
            emb, _ = lane.cam[0](view["img"], feats=False)
            trans_mat = camera.pred_trans_mat(emb, lane.cam[1])[0].cpu().numpy()
        params = tuple(float(v) for v in view["sdf_params"])
        iso = self.iso
        lane.ctx.set_image(feats, gfeat)
        if self.median_iso:
            # one coarse launch over the view's box (36 K points), not a second full grid
            iso = coarse_median_iso(lane.ctx, trans_mat, params)
        verts, tris = lane.pipe.run(feats, gfeat, trans_mat, self.res, iso, params, image_set=True)
        v, t = verts.cpu().numpy(), tris.cpu().numpy()        # synchronises this lane's stream only
        sdf = lane.pipe.last_sdf.cpu().numpy() if self.store_dist else None
        ms = (time.perf_counter() - t0) * 1e3
        fut = self.pool.submit(self._write, path, v, t, sdf, params)
        with self._lock:
            self.pending.append(fut)
            self.stats.append((seq, path, len(v), len(t), ms))

    def submit(self, view):
        if self._error is not None:
            raise self._error
        path = obj_file_name(self.root, view["cat_id"], view["obj_nm"], view["view_id"])
        seq, self._n = self._n, self._n + 1
        if len(self.lanes) == 1:
            self._mesh(self.lanes[0], seq, view, path)
        else:
            self.lanes[seq % len(self.lanes)].queue.put((seq, view, path))   # blocks when that lane is two views behind
        return path

    def drain(self):
        """Wait until every submitted view has been meshed (the OBJ writers may still be running)."""
        if len(self.lanes) > 1:
            for lane in self.lanes:
                lane.queue.put(None)
            for lane in self.lanes:
                lane.thread.join()
            for lane in self.lanes:       # ready for more work
                lane.thread = threading.Thread(target=self._work, args=(lane,), daemon=True)
                lane.thread.start()
        if self._error is not None:
            raise self._error

    def finish(self):
        if len(self.lanes) > 1:
            for lane in self.lanes:
                lane.queue.put(None)
            for lane in self.lanes:
                lane.thread.join()
        if self._error is not None:
            raise self._error
        for f in self.pending:
            f.result()                    # re-raises a writer's exception: nothing fails silently
        self.pool.shutdown()
        return [st[1:] for st in sorted(self.stats)]


def second_lane(ctx, decoder_weights, vgg_weights, cam_weights=None):
    """Another (ctx, encoder, cam) on ctx's GPU with the same weights, for BatchMesher(extra_lanes=[...])."""
    from . import api
    from .encoder import Vgg16Encoder
    c2 = api.Context(ctx.device.index or 0)
    c2.set_weights(decoder_weights)
    if not ctx.auto_decoder:
        c2.set_decoder(ctx.decoder)
    cam = None
    if cam_weights is not None:
        cam = (Vgg16Encoder(c2, cam_weights[0], slot=1), cam_weights[1])
    return c2, Vgg16Encoder(c2, vgg_weights), cam


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpu", type=str, default="0")
    ap.add_argument("--sdf_res", type=int, default=64)
    ap.add_argument("--iso", type=float, default=0.0)
    ap.add_argument("--log_dir", default="checkpoint/exp_200")
    ap.add_argument("--cam_log_dir", default="cam_est/checkpoint/cam_DISN")
    ap.add_argument("--img_feat_twostream", action="store_true")
    ap.add_argument("--cam_est", action="store_true")
    ap.add_argument("--create_obj", action="store_true", help="accepted for compatibility: meshes are always created")
    ap.add_argument("--category", default="all")
    ap.add_argument("--view_num", type=int, default=24)
    ap.add_argument("--test_lst_dir", default="./data/filelists")
    ap.add_argument("--synthetic", type=int, default=0, metavar="N")
    ap.add_argument("--views_dir", default=None)
    ap.add_argument("--h5", nargs=2, metavar=("RENDERED_DIR", "SDF_DIR"), default=None)
    ap.add_argument("--store_dist", action="store_true")
    ap.add_argument("--median_iso", action="store_true",
                    help="iso = median of a coarse grid over each view's box (implied by --random_init)")
    ap.add_argument("--random_init", action="store_true",
                    help="run on seeded random-init weights instead of checkpoints (no checkpoint is otherwise an error)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=3,
                    help="views in flight per GPU, each with its own context and HIP stream (1 = strictly serial)")
    FLAGS = ap.parse_args(argv)
    if not FLAGS.img_feat_twostream:
        raise SystemExit("create_sdf: only the two-stream decoder (--img_feat_twostream) is built (SURVEY.md s2 row 3)")

    from . import api, synth, tf_bundle
    from .encoder import Vgg16Encoder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = int(os.environ.get("LOCAL_RANK", FLAGS.gpu.split(",")[0]))
    torch.cuda.set_device(dev)
    log = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)
    ctx = api.Context(dev)
    random_init = FLAGS.random_init
    if not random_init:
        ckpt = tf_bundle.latest_checkpoint(FLAGS.log_dir)
        if ckpt is None:                      # the reference swallows this (test/create_sdf.py:188-196); we do not
            raise SystemExit("create_sdf: no checkpoint under --log_dir %s (give --random_init to run on random "
                             "weights)" % FLAGS.log_dir)
        log("Model loaded in file: %s" % ckpt)
        dec_w, vgg_w = tf_bundle.decoder_weights(ckpt), tf_bundle.vgg_weights(ckpt)
    else:
        log("create_sdf: --random_init -> seeded random-init weights (seed %d)" % FLAGS.seed)
        dec_w, vgg_w = synth.decoder_weights(FLAGS.seed), synth.vgg_weights(FLAGS.seed)
    ctx.set_weights(dec_w)
    enc = Vgg16Encoder(ctx, vgg_w)
    cam = cam_w = None
    if FLAGS.cam_est:
        from . import camera
        if random_init:
            log("create_sdf: --random_init -> seeded random-init camera network")
            cam_w = (synth.vgg_weights(FLAGS.seed + 1), camera.random_weights(FLAGS.seed))
        else:
            cam_ckpt = tf_bundle.latest_checkpoint(FLAGS.cam_log_dir)
            if cam_ckpt is None:
                raise SystemExit("create_sdf: no checkpoint under --cam_log_dir %s (give --random_init to run on "
                                 "random weights)" % FLAGS.cam_log_dir)
            cam_w = (tf_bundle.vgg_weights(cam_ckpt), tf_bundle.camera_weights(cam_ckpt))
        cam = (Vgg16Encoder(ctx, cam_w[0], slot=1), cam_w[1])

    if FLAGS.synthetic:
        views = synthetic_views(FLAGS.synthetic)
    elif FLAGS.views_dir:
        views = npz_views(FLAGS.views_dir)
    elif FLAGS.h5:
        cat_ids = list(CATS.values()) if FLAGS.category == "all" else [CATS[FLAGS.category]]
        views = h5_views(FLAGS.h5[0], FLAGS.h5[1], read_list(FLAGS.test_lst_dir, cat_ids, FLAGS.view_num, FLAGS.seed),
                         FLAGS.cam_est)
    else:
        raise SystemExit("create_sdf: give --synthetic N, --views_dir DIR or --h5 RENDERED_DIR SDF_DIR")

    root = result_obj_path(FLAGS.log_dir, FLAGS.sdf_res, FLAGS.iso, FLAGS.cam_est)
    log("RESULT_OBJ_PATH: ", root)
    extra = [second_lane(ctx, dec_w, vgg_w, cam_w) for _ in range(max(1, FLAGS.lanes) - 1)]
    mesher = BatchMesher(ctx, enc, FLAGS.sdf_res, FLAGS.iso, root, cam=cam, store_dist=FLAGS.store_dist,
                         median_iso=FLAGS.median_iso or random_init, extra_lanes=extra)
    t0 = time.perf_counter()
    n = 0
    for idx, view in enumerate(views):
        if idx % world != rank:
            continue
        path = mesher.submit(view)
        n += 1
        log("%d, submit create_obj %s, %s, %s -> %s" % (idx, view["cat_id"], view["obj_nm"], view["view_id"], path))
    stats = mesher.finish()
    dt = time.perf_counter() - t0
    pts = n * (FLAGS.sdf_res + 1) ** 3
    print("create_sdf[rank %d/%d]: %d views, %.2f s, %.1f ms/view, %.1f M points/s, %d faces"
          % (rank, world, n, dt, dt / max(n, 1) * 1e3, pts / dt / 1e6, sum(s[2] for s in stats)), flush=True)
    for lane in extra:
        lane[0].close()
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
