"""CPU model of where the HIP path rounds to 16 bits.  TEST INFRASTRUCTURE ONLY.

Restates the product's pipeline (cognitive_aim_depth_estimation_amd/model.py) on top of the fp32 oracle with one
rounding to the 16-bit operand type at every buffer the kernels store in 16 bits, each switchable on its own:

    patch    im2col columns and patch-embedding weight
    ln       LayerNorm outputs (GEMM A operands)
    wqkv/wo/w1/w2  packed weights
    qkv      fused projection output (Q carries the softmax scale)
    p        softmax probabilities fed to P V
    ctx      attention output
    gelu     fc1 + GELU output
    fh       focal tokens + position table
    fy       focal projection Y = X~ (Wk^T Wq) (and its folded weight fw)

Run as a script it prints, for bf16 and fp16, the depth error of the whole pipeline and of single stages against the
fp32 oracle: which buffers carry the bf16 error budget.  Nothing in the product imports this file.

    python -m tests.precision_model [--layers 12 --size 518 --batch 1]
"""
from __future__ import annotations

import argparse
import math

import torch
import torch.nn.functional as F

from oracle import cogaim_oracle as O
from oracle.weights import reference_state_spec, synth_state_dict

STAGES = ("patch", "ln", "wqkv", "wo", "w1", "w2", "qkv", "p", "ctx", "gelu", "fh", "fy", "fw")
LOG2E = 1.4426950408889634


class Rounder:
    def __init__(self, spec):
        """spec: {stage: torch dtype or None}"""
        self.spec = spec

    def __call__(self, stage, x):
        dt = self.spec.get(stage)
        return x if dt is None else x.to(dt).to(torch.float32)


def forward(sd, images, exif, rnd: Rounder, focus_strength=1.5, n_heads=12, eps=1e-6):
    cfg = O.dinov2_config_from_sd(sd)
    D, p = cfg["hidden"], cfg["patch"]
    pre = "backbone."
    B, _, H, W = images.shape
    w = sd[pre + "embeddings.patch_embeddings.projection.weight"]
    cols = F.unfold(images, p, stride=p).transpose(1, 2)                 # [B, N, 3 p p]
    x = rnd("patch", cols) @ rnd("patch", w.reshape(D, -1)).t() + sd[pre + "embeddings.patch_embeddings.projection.bias"]
    x = torch.cat((sd[pre + "embeddings.cls_token"].expand(B, -1, -1), x), 1) + O.dinov2_pos_embed(sd, pre, H // p, W // p)
    dh = D // n_heads
    T = x.shape[1]
    for i in range(cfg["layers"]):
        lp = f"{pre}encoder.layer.{i}."
        a = lp + "attention.attention."
        h = rnd("ln", F.layer_norm(x, (D,), sd[lp + "norm1.weight"], sd[lp + "norm1.bias"], eps))
        qs = dh ** -0.5 * LOG2E
        q = rnd("qkv", (h @ rnd("wqkv", sd[a + "query.weight"]).t() + sd[a + "query.bias"]) * qs)
        k = rnd("qkv", h @ rnd("wqkv", sd[a + "key.weight"]).t() + sd[a + "key.bias"])
        v = rnd("qkv", h @ rnd("wqkv", sd[a + "value.weight"]).t() + sd[a + "value.bias"])
        q, k, v = (t.view(B, T, n_heads, dh).transpose(1, 2) for t in (q, k, v))
        s = q @ k.transpose(2, 3)
        pr = torch.exp2(s - s.amax(-1, keepdim=True))
        prr = rnd("p", pr)
        ctx = rnd("ctx", (prr @ v) / pr.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, T, D)
        x = (ctx @ rnd("wo", sd[lp + "attention.output.dense.weight"]).t() + sd[lp + "attention.output.dense.bias"]) \
            * sd[lp + "layer_scale1.lambda1"] + x
        h = rnd("ln", F.layer_norm(x, (D,), sd[lp + "norm2.weight"], sd[lp + "norm2.bias"], eps))
        g = rnd("gelu", F.gelu(h @ rnd("w1", sd[lp + "mlp.fc1.weight"]).t() + sd[lp + "mlp.fc1.bias"]))
        x = (g @ rnd("w2", sd[lp + "mlp.fc2.weight"]).t() + sd[lp + "mlp.fc2.bias"]) * sd[lp + "layer_scale2.lambda1"] + x
    tokens = F.layer_norm(x, (D,), sd[pre + "layernorm.weight"], sd[pre + "layernorm.bias"], eps)
    cls, patches = tokens[:, 0], tokens[:, 1:]
    mc = O.model_config_from_sd(sd)
    feats = [O.ambient_stream(sd, cls)]
    N = patches.shape[1]
    pe, cb = O.pe2d_table(N, D), O.center_bias(N)
    cur, fl, a_vec = patches, [], None
    for i in range(mc["num_iterations"]):
        fp = f"focal_stream.focal_streams.{i}."
        xt = cur + pe
        h = rnd("fh", xt)
        wq, wk = sd[fp + "query_proj.weight"], sd[fp + "key_proj.weight"]
        ss = LOG2E / math.sqrt(D // 8)
        y = rnd("fy", h @ rnd("fw", (wk.t() @ wq) * ss).t() + (sd[fp + "query_proj.bias"] @ wk) * ss)
        S = y @ h.transpose(1, 2)
        P = torch.softmax(S / LOG2E, -1)
        pa = P.mean(1) + cb
        pa = pa / (pa.sum(-1, keepdim=True) + 1e-8)
        a_vec = pa.clamp_min(1e-8)
        a_vec = a_vec / (a_vec.sum(-1, keepdim=True) + 1e-8)
        u = torch.einsum("bn,bnc->bc", a_vec, P)
        wt = torch.einsum("bc,bcd->bd", u, xt)
        wv = wt @ sd[fp + "value_proj.weight"].t() + sd[fp + "value_proj.bias"] * u.sum(-1, keepdim=True)
        fl.append(O._lin(sd, fp + "projection.3", F.relu(O._lin(sd, fp + "projection.0", wv))))
        if i < mc["num_iterations"] - 1:
            cur = cur * (1 + focus_strength * a_vec.unsqueeze(-1))
    hf = F.relu(O._lin(sd, "focal_stream.fusion.0", torch.cat(fl, 1)))
    feats.append(O._lin(sd, "focal_stream.fusion.2", hf))
    if mc["use_exif"] and exif is not None:
        feats.append(O.exif_prior(sd, exif))
    z = torch.cat(feats, 1)
    if z.size(1) < 192:
        z = torch.cat([z, torch.zeros(z.size(0), 192 - z.size(1))], 1)
    depth, conf, _ = O.fusion_and_heads(sd, z)
    return depth, conf, a_vec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--grid", type=int, default=4)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    spec = reference_state_spec(layers=args.layers, native_grid=args.grid)
    sd = synth_state_dict(spec, seed=args.seed)
    torch.manual_seed(5)
    x = torch.randn(args.batch, 3, args.size, args.size)
    g = torch.Generator().manual_seed(99)
    B = args.batch
    exif = {"camera_idx": torch.randint(0, 71, (B,), generator=g), "focal_length": 10 + 190 * torch.rand(B, generator=g),
            "aperture": 1 + 21 * torch.rand(B, generator=g), "iso": 50 + 6350 * torch.rand(B, generator=g)}
    d0, c0, a0 = O.forward(sd, x, exif, return_attention=True)

    def err(spec_):
        d, c, a = forward(sd, x, exif, Rounder(spec_))
        return (float(((d - d0).abs() / d0.abs()).max()), float(((c - c0).abs() / c0.abs()).max()),
                float(((a - a0).abs() / a0.abs()).max()))

    print("restatement vs oracle, no rounding:", err({}))
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        print(f"{name}: all stages", err({s: dt for s in STAGES}))
        for s in STAGES:
            print(f"   only {s:6s}", "%.2e %.2e %.2e" % err({s: dt}))
        for s in STAGES:
            other = torch.float16 if dt == torch.bfloat16 else torch.bfloat16
            print(f"   all but {s:6s} (that one {'fp16' if other == torch.float16 else 'bf16'})",
                  "%.2e %.2e %.2e" % err({**{t: dt for t in STAGES}, s: other}))


if __name__ == "__main__":
    main()
