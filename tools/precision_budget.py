#!/usr/bin/env python3
"""Per-layer precision budget of the decoder's MFMA layers (SURVEY.md s7 "measure
per-layer error contributions first").  Host-side experiment: every candidate operand scheme is emulated
on the CPU (operands rounded exactly as the hardware formats round, products accumulated in float64) and
compared with the float64 evaluation of the same network.  Tolerance: 1e-4 on the SDF after /10
(BASELINE.json north_star).

Schemes (w = weights, x = post-ReLU activations; hi/lo = leading / residual part in the named format):
  bf16x3      w_hi x_hi + w_hi x_lo + w_lo x_hi, all bf16                  (round-1 kernel, 3 MFMA units)
  f16x3       the same in fp16                                             (3 units)
  f16x1       w_hi x_hi only                                               (1 unit)
  f16x2w / f16x2x   plus only the weight residual / only the activation residual   (2 units)
  f16+f8b     as f16+fp8c but the activation-side operands in bf8 e5m2 (fp16's range, no scaling or clamping needed)
  f16+fp8c    fp16 main product; both residual products in fp8 e4m3 at K=64 (2x rate): 1 + 2*0.5 = 2 units
  f16+fp6c    residual products in fp6 e2m3, block scales on the weights (4x rate): 1.5 units
  f16+fp4w    residual products with fp4 e2m1 weights x fp8 activations
  bf16+fp8c   bf16 main + fp8 residuals (the variant DESIGN.md r1 rejected), for the record
Writes profiles/precision_budget.json.  Uses oracle/net.py for the reference evaluation (this is a
measurement tool, not product code)."""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import net  # noqa: E402

torch.set_num_threads(8)
F64 = torch.float64


def q_bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(F64)


def q_f16(t):
    return t.to(torch.float32).to(torch.float16).to(F64)


def q_e5m2(t):
    return t.clamp(-57344.0, 57344.0).to(torch.float32).to(torch.float8_e5m2).to(F64)


def q_e4m3(t):
    return t.clamp(-448.0, 448.0).to(torch.float32).to(torch.float8_e4m3fn).to(F64)


def _q_small(t, mant_bits, emax, vmax):
    """generic tiny float: 1 sign, exponent bias 1, `mant_bits` mantissa bits, subnormals, saturating, RNE"""
    a = t.abs().clamp(max=vmax)
    e = torch.floor(torch.log2(a.clamp(min=1e-300))).clamp(min=0, max=emax)
    step = torch.pow(torch.tensor(2.0, dtype=F64), e) / (1 << mant_bits)
    q = torch.round(a / step) * step          # torch.round is half-to-even
    return torch.sign(t) * q.clamp(max=vmax)


def q_e2m3(t):
    return _q_small(t, 3, 2, 7.5)


def q_e2m1(t):
    return _q_small(t, 1, 2, 6.0)


def pow2_scale(maxabs, fmt_max):
    """largest power of two s with maxabs * s <= fmt_max"""
    m = float(maxabs)
    if m == 0.0:
        return 1.0
    return 2.0 ** math.floor(math.log2(fmt_max / m))


def q_blocked(w, q, fmt_max, block=32):
    """W[K][N] quantised with one power-of-two scale per (32 k-rows, output column): the MX block scale"""
    K, N = w.shape
    wb = w.reshape(K // block, block, N)
    m = wb.abs().amax(dim=1, keepdim=True).clamp(min=1e-300)
    s = torch.pow(torch.tensor(2.0, dtype=F64), torch.floor(torch.log2(fmt_max / m)))
    return (q(wb * s) / s).reshape(K, N)


def layer(x, w, scheme):
    """x[N][K] (post-ReLU, float64 holding float32 values), w[K][Nout] -> x @ w under `scheme`"""
    if scheme == "f64":
        return x @ w
    if scheme.startswith("bf16x") or scheme.startswith("f16x"):
        q = q_bf16 if scheme.startswith("bf16") else q_f16
        xh, wh = q(x), q(w)
        xl, wl = q(x - xh), q(w - wh)
        kind = scheme.split("x", 1)[1]
        y = xh @ wh
        if kind in ("3", "2x"):
            y = y + xl @ wh
        if kind in ("3", "2w"):
            y = y + xh @ wl
        return y
    main, corr = scheme.split("+")
    q = q_bf16 if main == "bf16" else q_f16
    xh, wh = q(x), q(w)
    xl, wl = x - xh, w - wh
    res_bits = 9 if main == "bf16" else 12         # |xl| <= 2^-res_bits * |x| (RNE, normal range)
    y = xh @ wh
    if corr == "fp8c":          # uniform power-of-two scales per layer; activations converted as they are
        sw = pow2_scale(wl.abs().max(), 448.0)
        sw2 = pow2_scale(w.abs().max(), 448.0)
        sx = 2.0 ** res_bits
        y = y + q_e4m3(x) @ (q_e4m3(wl * sw) / sw)
        y = y + (q_e4m3(xl * sx) / sx) @ (q_e4m3(w * sw2) / sw2)
    elif corr == "f8b":         # weights fp8 e4m3 (uniform scale), activations bf8 e5m2 unscaled (fp16's range: no clamp)
        sw = pow2_scale(wl.abs().max(), 448.0)
        sw2 = pow2_scale(w.abs().max(), 448.0)
        sx = 2.0 ** (res_bits - 1)
        y = y + q_e5m2(x) @ (q_e4m3(wl * sw) / sw)
        y = y + (q_e5m2(xl * sx) / sx) @ (q_e4m3(w * sw2) / sw2)
    elif corr == "fp8cb":       # block-scaled weights
        sx = 2.0 ** res_bits
        y = y + q_e4m3(x) @ q_blocked(wl, q_e4m3, 448.0)
        y = y + (q_e4m3(xl * sx) / sx) @ q_blocked(w, q_e4m3, 448.0)
    elif corr in ("fp6c", "fp6w", "fp4w", "fp4c"):
        qw, wmax = (q_e2m3, 7.5) if corr.startswith("fp6") else (q_e2m1, 6.0)
        if corr.endswith("c"):   # activations in the small format too, fixed scale: x/4 (range 30), xl * 2^res_bits / 4
            qa = qw
            sa, sxl = 0.25, (2.0 ** res_bits) * 0.25
        else:                    # activations in fp8
            qa = q_e4m3
            sa, sxl = 1.0, 2.0 ** res_bits
        y = y + (qa(x * sa) / sa) @ q_blocked(wl, qw, wmax)
        y = y + (qa(xl * sxl) / sxl) @ q_blocked(w, qw, wmax)
    else:
        raise ValueError(scheme)
    return y


LAYERS = ("f1c2", "f1c3", "f2c1", "f2c2")      # 64->256, 256->512, 512->512 (point part), 512->256


def decoder(pts, gfeat, img_feat, w, schemes):
    """both streams; schemes: dict layer -> scheme (missing = f64).  Everything outside the four MFMA layers
    is evaluated in float64 on float32 inputs (the kernel does those parts in exact fp32)."""
    t = {k: torch.from_numpy(np.asarray(v, np.float32)).to(F64) for k, v in w.items()}
    p = torch.from_numpy(pts).to(F64)
    out = 0
    for s, extra in (("g", torch.from_numpy(gfeat).to(F64).reshape(1, -1)), ("l", torch.from_numpy(img_feat).to(F64))):
        f32 = lambda v: v.to(torch.float32).to(F64)   # activations exist as fp32 accumulator values
        h = f32(torch.relu(p @ t[s + "_f1c1_w"] + t[s + "_f1c1_b"]))
        h = f32(torch.relu(layer(h, t[s + "_f1c2_w"], schemes.get("f1c2", "f64")) + t[s + "_f1c2_b"]))
        h = f32(torch.relu(layer(h, t[s + "_f1c3_w"], schemes.get("f1c3", "f64")) + t[s + "_f1c3_b"]))
        w4 = t[s + "_f2c1_w"]
        const = extra @ w4[512:] + t[s + "_f2c1_b"]                 # hoisted global term / Pmap blend: exact fp32 in the kernel
        h = f32(torch.relu(layer(h, w4[:512], schemes.get("f2c1", "f64")) + const))
        h = f32(torch.relu(layer(h, t[s + "_f2c2_w"], schemes.get("f2c2", "f64")) + t[s + "_f2c2_b"]))
        out = out + (h @ t[s + "_f2c5_w"] + t[s + "_f2c5_b"]).reshape(-1)
    return out.numpy()


def weight_sets():
    sets = {}
    sets["xavier_seed0 (bench config: zero biases)"] = net.xavier_decoder_weights(0)
    sets["xavier_seed1 + N(0,1) biases"] = net.xavier_decoder_weights(1, bias_scale=1.0)
    w = net.xavier_decoder_weights(2, bias_scale=0.3)
    rng = np.random.default_rng(2)
    for s in "gl":
        for l in ("f1c3", "f2c1"):           # row norms x4 on a quarter of the input features
            k = w["%s_%s_w" % (s, l)].shape[0]
            f = np.where(rng.random(k) < 0.25, 4.0, 1.0).astype(np.float32)
            w["%s_%s_w" % (s, l)] = w["%s_%s_w" % (s, l)] * f[:, None]
        w[s + "_f2c5_w"] = w[s + "_f2c5_w"] * 4
    sets["row norms x4 on f1c3/f2c1, biases 0.3, |raw| scaled to 10"] = w
    w = net.xavier_decoder_weights(3, bias_scale=0.5)
    rng = np.random.default_rng(3)
    for k in list(w):
        if k.endswith("_w") and "f1c1" not in k:      # heavy-tailed (log-normal, sigma 1) element magnitudes
            w[k] = (w[k] * rng.lognormal(0.0, 1.0, size=w[k].shape) * 0.6).astype(np.float32)
    for s in "gl":
        w[s + "_f2c5_w"] = w[s + "_f2c5_w"] * 2
    sets["log-normal element scales (sigma 1), biases 0.5, |raw| scaled to 10"] = w
    return sets


def main():
    npts = 12288
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, size=(npts, 3)).astype(np.float32)
    maps, gfeat = net.synthetic_features(seed=3)
    xy = net.get_img_points(pts, net.GT_TRANS_MAT)
    img_feat = net.sample_point_features(maps, xy)
    schemes = ["bf16x3", "f16x3", "f16x1", "f16x2w", "f16x2x", "bf16+fp8c", "f16+fp8c", "f16+f8b", "f16+fp8cb", "f16+fp6w",
               "f16+fp6c", "f16+fp4w", "f16+fp4c"]
    units = {"bf16x3": 3, "f16x3": 3, "f16x1": 1, "f16x2w": 2, "f16x2x": 2, "bf16+fp8c": 2, "f16+fp8c": 2, "f16+f8b": 2, "f16+fp8cb": 2,
             "f16+fp6w": 2, "f16+fp6c": 1.5, "f16+fp4w": 2, "f16+fp4c": 1.5}
    report = {"tolerance_after_div10": 1e-4, "points": npts,
              "note": "max |SDF - float64 SDF| after /10; 'all' = scheme on all four MFMA layers of both streams, "
                      "per-layer columns = scheme on that layer only (others float64). units = MFMA time per "
                      "k-step relative to one 16-bit product (fp8 K=64 counts 0.5, fp6/fp4 0.25; mixed fp8 x fp6/fp4 "
                      "runs at the fp8 rate). Products accumulated in float64 (the MFMA's fp32 accumulation adds "
                      "~1e-7 relative, common to all schemes).",
              "mfma_units": units, "weight_sets": {}}
    plans = {   # layer -> scheme; MFMA time relative to bf16x3 everywhere (pairs: f1c2 32, f1c3 256, f2c1 512, f2c2 256)
        "P_today  bf16x3 everywhere": {l: "bf16x3" for l in LAYERS},
        "P_mixed  f16+fp8c on f1c2/f1c3/f2c1, bf16x3 on f2c2": {"f1c2": "f16+fp8c", "f1c3": "f16+fp8c", "f2c1": "f16+fp8c", "f2c2": "bf16x3"},
        "P_mixed16 f16+fp8c on f1c2/f1c3/f2c1, f16x3 on f2c2": {"f1c2": "f16+fp8c", "f1c3": "f16+fp8c", "f2c1": "f16+fp8c", "f2c2": "f16x3"},
        "P_mixedb f16+f8b  on f1c2/f1c3/f2c1, bf16x3 on f2c2": {"f1c2": "f16+f8b", "f1c3": "f16+f8b", "f2c1": "f16+f8b", "f2c2": "bf16x3"},
        "P_mixedh f16+f8b  on f1c2/f1c3/f2c1, f16x3 on f2c2": {"f1c2": "f16+f8b", "f1c3": "f16+f8b", "f2c1": "f16+f8b", "f2c2": "f16x3"},
        "P_allb   f16+f8b everywhere": {l: "f16+f8b" for l in LAYERS},
    }
    pairs = {"f1c2": 32, "f1c3": 256, "f2c1": 512, "f2c2": 256}
    report["plans"] = {k: {"layers": v, "mfma_time_vs_today": sum(pairs[l] * units[v[l]] for l in LAYERS) / (3.0 * 1056)}
                       for k, v in plans.items()}
    for name, w in weight_sets().items():
        ref = decoder(pts, gfeat, img_feat, w, {})
        if "scaled to 10" in name:          # normalise the output range to |raw| <= 10 (trained SDFs reach +-0.5 after /10)
            f = np.float32(10.0 / np.abs(ref).max())
            for s in "gl":
                w[s + "_f2c5_w"] = w[s + "_f2c5_w"] * f
                w[s + "_f2c5_b"] = w[s + "_f2c5_b"] * f
            ref = decoder(pts, gfeat, img_feat, w, {})
        prow = {}
        for pname, plan in plans.items():
            e = float(np.abs(decoder(pts, gfeat, img_feat, w, plan) - ref).max() / 10)
            prow[pname] = {"err": e, "margin": 1e-4 / e}
            print("%-62s %-58s %.2e (margin %6.1fx)" % (name[:62], pname, e, 1e-4 / e), flush=True)
        rows = {}
        for sc in schemes:
            row = {"all": float(np.abs(decoder(pts, gfeat, img_feat, w, {l: sc for l in LAYERS}) - ref).max() / 10)}
            for l in LAYERS:
                row[l] = float(np.abs(decoder(pts, gfeat, img_feat, w, {l: sc}) - ref).max() / 10)
            row["margin_all"] = 1e-4 / max(row["all"], 1e-300)
            rows[sc] = row
            print("%-62s %-10s all %.2e (margin %6.1fx)  " % (name[:62], sc, row["all"], row["margin_all"]) +
                  " ".join("%s %.1e" % (l, row[l]) for l in LAYERS), flush=True)
        report["weight_sets"][name] = {"max_abs_raw_output": float(np.abs(ref).max()), "schemes": rows, "plans": prow}
    worst = {sc: min(ws["schemes"][sc]["margin_all"] for ws in report["weight_sets"].values()) for sc in schemes}
    report["worst_margin_over_weight_sets"] = worst
    report["passing_with_5x_margin"] = sorted([sc for sc in schemes if worst[sc] >= 5.0], key=lambda s: units[s])
    out = os.path.join(ROOT, "profiles", "precision_budget.json")
    report["worst_plan_margin"] = {p: min(ws["plans"][p]["margin"] for ws in report["weight_sets"].values()) for p in plans}
    json.dump(report, open(out, "w"), indent=1)
    print("plan margins:", {k: round(v, 1) for k, v in report["worst_plan_margin"].items()})
    print("worst margins:", {k: round(v, 1) for k, v in worst.items()})
    print("wrote", out)


if __name__ == "__main__":
    main()
