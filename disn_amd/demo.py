"""`demo.py` surface of the reference (reference: demo/demo.py:26-65,123-193,261-369)
on the HIP path:  python -m disn_amd.demo --sdf_res 256 --img_feat_twostream

Kept flags: --sdf_res --img_feat_twostream --cam_est --log_dir --cam_log_dir --iso
--img_h --img_w --gpu (+ --image, --out, --ngpu is `torchrun --nproc-per-node`).
Checkpoints are read without TensorFlow (disn_amd/tf_bundle.py).  A --log_dir (or,
with --cam_est, a --cam_log_dir) without a checkpoint is an ERROR (the reference
swallows the failed restore and meshes random weights, demo/demo.py:157-165).
--random_init is the explicit opt-in to seeded random-init decoder / encoder /
camera weights (the released checkpoints are a separate download,
README.md:39-50); iso then defaults to the median of a coarse whole-box grid so
that a surface exists.  With a restored checkpoint --iso defaults to 0.0 like the
reference (demo/demo.py:50).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch


DEMO_IMAGE = "03001627_17e916fc863540ee3def89b32cef8e45_20.png"     # reference demo/demo.py:262


def default_image():
    """The image the reference's demo meshes when none is given (demo/demo.py:262): ./demo/<name> relative to the
    working directory like the reference, else this repository's data copy of it; None if neither exists."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join("demo", DEMO_IMAGE), os.path.join(here, "tests", "golden", "demo_chair_20.png")):
        if os.path.exists(p):
            return p
    return None


def read_image(path):
    """(137,137,3) float32 in [0,1], BGR channel order like cv2.imread (reference demo/demo.py:263-264)."""
    from PIL import Image
    img = np.asarray(Image.open(path))
    if img.ndim == 2:
        img = np.stack([img] * 3, -1)
    img = img[:, :, :3][:, :, ::-1]          # RGB(A) -> BGR
    if img.shape[:2] != (137, 137):
        raise SystemExit("demo: image must be 137x137 (got %s)" % (img.shape[:2],))
    return np.ascontiguousarray(img).astype(np.uint8).astype(np.float32) / np.float32(255.0)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--gpu", type=str, default="0")
    ap.add_argument("--sdf_res", type=int, default=64)
    ap.add_argument("--img_feat_twostream", action="store_true")
    ap.add_argument("--cam_est", action="store_true")
    ap.add_argument("--log_dir", default="checkpoint/SDF_DISN")
    ap.add_argument("--cam_log_dir", default="cam_est/checkpoint/cam_DISN")
    ap.add_argument("--iso", type=float, default=None,
                    help="iso value (default 0.0 as in the reference; with --random_init: median of a coarse grid)")
    ap.add_argument("--random_init", action="store_true",
                    help="run on seeded random-init weights instead of checkpoints (no checkpoint is otherwise an error)")
    ap.add_argument("--img_h", type=int, default=137)
    ap.add_argument("--img_w", type=int, default=137)
    ap.add_argument("--image", default=None,
                    help="137x137 PNG (default: the reference's demo chair, demo/demo.py:262, looked up as "
                         "./demo/%s or the copy under tests/golden/; a seeded synthetic view if neither exists)" % DEMO_IMAGE)
    ap.add_argument("--out", default="demo/result.obj")
    ap.add_argument("--seed", type=int, default=0)
    FLAGS = ap.parse_args(argv)
    if not FLAGS.img_feat_twostream:
        raise SystemExit("demo: only the two-stream decoder (--img_feat_twostream) is built (SURVEY.md s2 row 3)")
    if (FLAGS.img_h, FLAGS.img_w) != (137, 137):
        raise SystemExit("demo: the feature pyramid is sampled at 137x137 (reference default)")

    from . import api, synth
    from .encoder import Vgg16Encoder
    from .pipeline import MeshPipeline, coarse_median_iso

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = int(os.environ.get("LOCAL_RANK", FLAGS.gpu.split(",")[0]))
    dist = None
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    log = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)

    ctx = api.Context(dev)
    from . import tf_bundle
    if FLAGS.random_init:
        log("demo: --random_init -> seeded random-init decoder/encoder weights (seed %d)" % FLAGS.seed)
        ctx.set_weights(synth.decoder_weights(FLAGS.seed))
        enc = Vgg16Encoder(ctx, synth.vgg_weights(FLAGS.seed))
    else:
        ckpt = tf_bundle.latest_checkpoint(FLAGS.log_dir)      # reference demo/demo.py:153-165 (but failing loudly)
        if ckpt is None:
            raise SystemExit("demo: no checkpoint under --log_dir %s (give --random_init to run on random weights)"
                             % FLAGS.log_dir)
        log("Model loading from %s" % ckpt)
        ctx.set_weights(tf_bundle.decoder_weights(ckpt))
        enc = Vgg16Encoder(ctx, tf_bundle.vgg_weights(ckpt))
    img_path = FLAGS.image or default_image()
    if img_path:
        log("demo: image %s" % img_path)
        img = read_image(img_path)
    else:
        log("demo: no --image and the demo chair is not around: seeded synthetic view (seed %d)" % FLAGS.seed)
        img = synth.image(FLAGS.seed)
    if FLAGS.cam_est:
        # reference demo/demo.py:195-258: a second VGG-16 (own weights) + the pose heads
        from . import camera
        log("here we use our cam est network to estimate cam parameters:")
        if FLAGS.random_init:
            log("demo: --random_init -> seeded random-init camera network")
            cam_vgg, cam_w = synth.vgg_weights(FLAGS.seed + 1), camera.random_weights(FLAGS.seed)
        else:
            cam_ckpt = tf_bundle.latest_checkpoint(FLAGS.cam_log_dir)
            if cam_ckpt is None:
                raise SystemExit("demo: no checkpoint under --cam_log_dir %s (give --random_init to run on random "
                                 "weights)" % FLAGS.cam_log_dir)
            log("Model loading from %s" % cam_ckpt)
            cam_vgg, cam_w = tf_bundle.vgg_weights(cam_ckpt), tf_bundle.camera_weights(cam_ckpt)
        cam_emb, _ = Vgg16Encoder(ctx, cam_vgg, slot=1)(img, feats=False)
        tm_t = camera.pred_trans_mat(cam_emb, cam_w)[0].contiguous()
        if world > 1:                                  # one matrix for all ranks: slab seams need identical projections
            from .pipeline import _Comm
            _Comm(dist).broadcast(tm_t, src=0)
        trans_mat = tm_t.cpu().numpy()
        log("pred_trans_mat_val", trans_mat)
    else:
        log("here we use gt cam parameters")
        trans_mat = synth.GT_TRANS_MAT                 # reference demo/demo.py:272-276
    sdf_params = (-1, -1, -1, 1, 1, 1)                 # reference demo/demo.py:278
    t0 = time.time()
    gfeat, feats = enc(img)
    if world > 1 and rank != 0:
        feats, gfeat = [torch.zeros_like(f) for f in feats], torch.zeros_like(gfeat)
    pipe = MeshPipeline(ctx, rank, world, dist)
    iso = FLAGS.iso
    if iso is None and not FLAGS.random_init:
        iso = 0.0                                      # reference demo/demo.py:50
    if iso is None:
        # every rank holds the same features after this broadcast, so every rank computes the same number
        feats, gfeat = pipe._broadcast_features(feats, gfeat)
        ctx.set_image(feats, gfeat)
        iso_t = torch.tensor([coarse_median_iso(ctx, trans_mat, sdf_params)], dtype=torch.float32, device=ctx.device)
        if world > 1:                                  # rank 0's value for everyone: counts and vertex bases must agree
            pipe.dist.broadcast(iso_t, src=0)
        iso = float(iso_t.item())
        log("demo: --random_init without --iso: using the median of a coarse whole-box grid, %.9g "
            "(random weights have no zero level set)" % iso)
    torch.cuda.synchronize()
    t1 = time.time()
    verts, tris = pipe.run(feats, gfeat, trans_mat, FLAGS.sdf_res, iso, sdf_params)
    torch.cuda.synchronize()
    t2 = time.time()
    if rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(FLAGS.out)), exist_ok=True)
        api.write_obj(FLAGS.out, verts.cpu().numpy(), tris.cpu().numpy())
        log("demo: encoder %.1f ms, grid+mesh %.1f ms (%d^3 points, %d GPU(s)), %d vertices / %d faces -> %s"
            % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, FLAGS.sdf_res + 1, world, verts.shape[0], tris.shape[0], FLAGS.out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
