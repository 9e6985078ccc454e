"""CPU oracle for the Cognitive-Aim depth forward pass.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 restatement of the reference algorithm
(/root/reference/src/model.py plus the third-party DINOv2 arithmetic in
transformers/models/dinov2/modeling_dinov2.py, cited below as ``hf:``).  It is
the checker for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
never imports it and has no CPU fallback.

Pinning: ``oracle/make_goldens.py`` path-loads the real reference module in the
build container and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.  The reference ships no
tests or golden vectors of its own, and the DINOv2 arithmetic lives in the
un-vendored ``transformers`` dependency (requirements.txt:12, ``>=4.20.0``;
5.15.0 in the build container), so the goldens made from the reference run
locally are the only pin there is.

All functions are stateless and take a flat ``state_dict`` (``{key: tensor}``)
using the reference's parameter names.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

INSTRUCTIONS = ("center", "left", "right", "top", "bottom",
                "top-left", "top-right", "bottom-left", "bottom-right")


def _lin(sd: SD, key: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


# --------------------------------------------------------------------------
# DINOv2 backbone (third-party arithmetic; call sites model.py:973,1095,1129,1173)
# --------------------------------------------------------------------------
def dinov2_config_from_sd(sd: SD, prefix: str = "backbone.") -> dict:
    """Recover the architecture from tensor shapes (hf:43-55,126-139)."""
    w = sd[prefix + "embeddings.patch_embeddings.projection.weight"]
    hidden, _, patch, _ = w.shape
    n_layers = 0
    while f"{prefix}encoder.layer.{n_layers}.norm1.weight" in sd:
        n_layers += 1
    n_pos = sd[prefix + "embeddings.position_embeddings"].shape[1] - 1
    return {"hidden": hidden, "patch": patch, "layers": n_layers,
            "native_grid": int(round(math.sqrt(n_pos)))}


def dinov2_pos_embed(sd: SD, prefix: str, gh: int, gw: int) -> Tensor:
    """hf:57-95: position table for a gh x gw patch grid, [1, 1+gh*gw, D].

    Native table returned untouched when the grid matches and is square;
    otherwise bicubic interpolation in fp32, align_corners=False, class
    position kept."""
    pos = sd[prefix + "embeddings.position_embeddings"]
    n_pos = pos.shape[1] - 1
    if gh * gw == n_pos and gh == gw:
        return pos
    d = pos.shape[-1]
    s = int(n_pos ** 0.5)
    grid = pos[:, 1:].reshape(1, s, s, d).permute(0, 3, 1, 2)
    grid = F.interpolate(grid.to(torch.float32), size=(gh, gw), mode="bicubic",
                         align_corners=False).to(pos.dtype)
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, d)
    return torch.cat((pos[:, :1], grid), dim=1)


def dinov2_embeddings(sd: SD, prefix: str, images: Tensor) -> Tensor:
    """hf:97-116 + hf:141-149: conv patch-embed, CLS concat, + pos-embed."""
    w = sd[prefix + "embeddings.patch_embeddings.projection.weight"]
    b = sd[prefix + "embeddings.patch_embeddings.projection.bias"]
    p = w.shape[-1]
    B, _, H, W = images.shape
    x = F.conv2d(images, w, b, stride=p).flatten(2).transpose(1, 2)
    cls = sd[prefix + "embeddings.cls_token"].expand(B, -1, -1)
    x = torch.cat((cls, x), dim=1)
    return x + dinov2_pos_embed(sd, prefix, H // p, W // p)


def dinov2_layer(sd: SD, lp: str, x: Tensor, n_heads: int, eps: float) -> Tensor:
    """hf:361-380 (block), hf:203-234 (QKV/heads), hf:153-178 (softmax attn),
    hf:249-252 (out dense), hf:277-278 (LayerScale), hf:293-297 (MLP, erf GELU)."""
    B, T, D = x.shape
    dh = D // n_heads
    h = F.layer_norm(x, (D,), sd[lp + "norm1.weight"], sd[lp + "norm1.bias"], eps)
    q = _lin(sd, lp + "attention.attention.query", h).view(B, T, n_heads, dh).transpose(1, 2)
    k = _lin(sd, lp + "attention.attention.key", h).view(B, T, n_heads, dh).transpose(1, 2)
    v = _lin(sd, lp + "attention.attention.value", h).view(B, T, n_heads, dh).transpose(1, 2)
    att = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * dh ** -0.5, dim=-1)
    ctx = torch.matmul(att, v).transpose(1, 2).reshape(B, T, D)
    x = _lin(sd, lp + "attention.output.dense", ctx) * sd[lp + "layer_scale1.lambda1"] + x
    h = F.layer_norm(x, (D,), sd[lp + "norm2.weight"], sd[lp + "norm2.bias"], eps)
    h = _lin(sd, lp + "mlp.fc2", F.gelu(_lin(sd, lp + "mlp.fc1", h)))
    return h * sd[lp + "layer_scale2.lambda1"] + x


def dinov2_forward(sd: SD, images: Tensor, prefix: str = "backbone.",
                   n_heads: int = 12, eps: float = 1e-6) -> Tensor:
    """hf:451-477: embeddings -> layers -> final LayerNorm.  Returns
    last_hidden_state [B, 1+N, D]."""
    cfg = dinov2_config_from_sd(sd, prefix)
    x = dinov2_embeddings(sd, prefix, images)
    for i in range(cfg["layers"]):
        x = dinov2_layer(sd, f"{prefix}encoder.layer.{i}.", x, n_heads, eps)
    D = x.shape[-1]
    return F.layer_norm(x, (D,), sd[prefix + "layernorm.weight"],
                        sd[prefix + "layernorm.bias"], eps)


# --------------------------------------------------------------------------
# Cognitive modules
# --------------------------------------------------------------------------
def ambient_stream(sd: SD, cls: Tensor, prefix: str = "ambient_stream.") -> Tensor:
    """model.py:32-53: Linear(D,256) ReLU (Dropout) Linear(256,128) ReLU Linear(128,64)."""
    h = F.relu(_lin(sd, prefix + "mlp.0", cls))
    h = F.relu(_lin(sd, prefix + "mlp.3", h))
    return _lin(sd, prefix + "mlp.5", h)


def pe2d_table(num_patches: int, dim: int) -> Tensor:
    """model.py:140-177, vectorised.  Square N: first half of the channels
    encodes the row, second half the column, sin on even / cos on odd slots,
    frequency exp(-2k ln(1e4)/(dim/2)).  Non-square N: plain 1-D table."""
    pe = torch.zeros(num_patches, dim)
    g = int(num_patches ** 0.5)
    if g * g == num_patches:
        if dim >= 4:
            half = dim // 2
            div = torch.exp(torch.arange(0, half, 2, dtype=torch.float)
                            * -(math.log(10000.0) / half))
            idx = torch.arange(num_patches)
            row = (idx // g).to(torch.float).unsqueeze(1)
            col = (idx % g).to(torch.float).unsqueeze(1)
            if len(div) > 0:
                pe[:, 0:half:2] = torch.sin(row * div)
                pe[:, 1:half:2] = torch.cos(row * div)
                pe[:, half::2] = torch.sin(col * div)
                pe[:, half + 1::2] = torch.cos(col * div)
    else:
        pos = torch.arange(0, num_patches, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float)
                        * -(math.log(10000.0) / dim))
        if len(div) > 0:
            pe[:, 0::2] = torch.sin(pos * div)
            if dim > 1:
                pe[:, 1::2] = torch.cos(pos * div)
    return pe


def center_bias(num_patches: int, strength: float = 0.3) -> Tensor:
    """model.py:208-231: Gaussian bump at (G//2, G//2), sigma G/6, x0.3."""
    g = int(num_patches ** 0.5)
    if g * g != num_patches:
        c = num_patches // 2
        d = torch.abs(torch.arange(num_patches, dtype=torch.float) - c)
        sigma = num_patches / 12
        return torch.exp(-d ** 2 / (2 * sigma ** 2)) * strength
    c = g // 2
    y, x = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
    d = torch.sqrt((x - c).float() ** 2 + (y - c).float() ** 2)
    sigma = g / 6
    return torch.exp(-d ** 2 / (2 * sigma ** 2)).flatten() * strength


def focal_stream(sd: SD, prefix: str, x: Tensor, num_heads: int = 8,
                 return_parts: bool = False):
    """model.py:128-313 with curiosity modulation off (it is disabled by every
    shipped config and cancels in the renormalisation at model.py:282 anyway).

    Returns (features [B,64], attention [B,N])."""
    B, N, D = x.shape
    xt = x + pe2d_table(N, D).unsqueeze(0)                       # :184
    q = _lin(sd, prefix + "query_proj", xt)                      # :192
    k = _lin(sd, prefix + "key_proj", xt)
    v = _lin(sd, prefix + "value_proj", xt)
    scale = math.sqrt(D // num_heads)                            # :69
    p = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) / scale, dim=-1)  # :197-200
    o = torch.matmul(p, v)                                       # :204
    cb = center_bias(N)
    colmean = p.mean(dim=1)                                      # :234
    pa = colmean + cb.unsqueeze(0)                               # :239
    branch = 0                                                   # which summary ends up in use (0 = column mean)
    if pa.var() < 1e-6:                                          # :242
        pa = torch.diagonal(p, dim1=-2, dim2=-1) + cb.unsqueeze(0)
        branch = 1
    if pa.var() < 1e-6:                                          # :247
        pa = p.max(dim=-1)[0] + cb.unsqueeze(0)
        branch = 2
    if pa.var() < 1e-6:                                          # :252
        norms = torch.norm(xt, dim=-1)
        pa = norms + torch.randn_like(norms) * 0.1 * norms.std()
        branch = 3
    pa = pa / (pa.sum(dim=-1, keepdim=True) + 1e-8)              # :261
    a = torch.clamp(pa, min=1e-8)                                # :281
    a = a / (a.sum(dim=-1, keepdim=True) + 1e-8)                 # :282
    w = torch.sum(o * a.unsqueeze(-1), dim=1)                    # :308
    f = _lin(sd, prefix + "projection.3", F.relu(_lin(sd, prefix + "projection.0", w)))  # :311
    if return_parts:
        return f, a, {"colmean": colmean, "weighted": w, "branch": branch}
    return f, a


def iterative_focal_stream(sd: SD, x: Tensor, num_iterations: int, focus_strength: float,
                           prefix: str = "focal_stream.") -> Tuple[Tensor, Tensor]:
    """model.py:391-455: I FocalStreams with separate weights; between
    iterations tokens are rescaled by (1 + fs * a); features concatenated and
    fused by Linear(64 I,128) ReLU Linear(128,64)."""
    feats, a, cur = [], None, x
    for i in range(num_iterations):
        f, a = focal_stream(sd, f"{prefix}focal_streams.{i}.", cur)
        feats.append(f)
        if i < num_iterations - 1:
            cur = cur * (1 + focus_strength * a.unsqueeze(-1))   # :426
    h = F.relu(_lin(sd, prefix + "fusion.0", torch.cat(feats, dim=1)))
    return _lin(sd, prefix + "fusion.2", h), a


def exif_prior(sd: SD, exif: Dict[str, Tensor], prefix: str = "exif_prior.") -> Tensor:
    """model.py:482-519."""
    def sq(t):
        return t.squeeze(1) if t.dim() > 1 else t
    cam = F.embedding(sq(exif["camera_idx"]), sd[prefix + "camera_embedding.weight"])
    cont = torch.stack([sq(exif["focal_length"]), sq(exif["aperture"]),
                        torch.log(sq(exif["iso"]) + 1)], dim=1)
    e = _lin(sd, prefix + "exif_encoder.2", F.relu(_lin(sd, prefix + "exif_encoder.0", cont)))
    h = F.relu(_lin(sd, prefix + "fusion.0", torch.cat([cam, e], dim=1)))
    return _lin(sd, prefix + "fusion.3", h)


def curiosity_reward(sd: SD, cls: Tensor, eps: Tensor, noise: Tensor, prefix: str = "curiosity_module.") -> Tensor:
    """model.py:586-688 as called from the forward paths (features = CLS token, exif_data=None, loss_type
    "robust", uncertainty_weight 0.1), with the two random draws (:609 ``eps`` [B,192], :744 ``noise`` [B,768],
    already scaled by 0.01) passed in.  Returns the reward that :680 writes to the exploration history (before
    the final clamp of :685)."""
    def seq(name, x, last_act=None):
        i0, i1 = (("0", "3") if name in ("encoder_mean", "encoder_logvar", "decoder") else ("0", "2"))
        return _lin(sd, f"{prefix}{name}.{i1}", F.relu(_lin(sd, f"{prefix}{name}.{i0}", x)))
    mu, logvar = seq("encoder_mean", cls), seq("encoder_logvar", cls)
    z = mu + eps * torch.exp(0.5 * logvar)                                   # :608-610
    rec = seq("decoder", z)
    diff = rec - cls[:, :rec.size(1)]
    re = torch.sqrt(torch.sum(diff ** 2, dim=1) + 1e-8)                      # :621-623
    re = (re / (1.0 + re)).clamp(min=0.0)
    kl = (-0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1)).clamp(min=0.0)   # :635, :642
    unc = F.softplus(seq("uncertainty_head", cls)).squeeze(-1).clamp(0.0, 10.0)            # :638, :643
    basic = re + 0.1 * kl + 0.1 * unc                                        # :646
    lb = torch.sigmoid(seq("local_curiosity", cls)).squeeze(-1)              # :738
    ln = torch.sigmoid(seq("local_curiosity", cls + noise)).squeeze(-1)      # :745-748
    local = (lb + (lb - ln).abs() * 0.2).clamp(0.0, 1.0)                     # :751-758
    w = torch.softmax(sd[prefix + "curiosity_weights"], dim=0)               # :666
    return w[0] * 0.5 + w[1] * local + w[2] * basic                          # :669-671, geometric = 0.5 (:698-700)


def fusion_and_heads(sd: SD, feats: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """model.py:908-912 (fusion), :925-928 (depth: Linear+Softplus),
    :936-941 (confidence: Linear, ReLU, Linear(1,1), Sigmoid)."""
    fused = F.relu(_lin(sd, "fusion.0", feats))
    depth = F.softplus(_lin(sd, "decision_head.0", fused))
    conf = torch.sigmoid(_lin(sd, "confidence_head.2",
                              F.relu(_lin(sd, "confidence_head.0", fused))))
    return depth, conf, fused


# --------------------------------------------------------------------------
# Guided attention (model.py:1242-1426)
# --------------------------------------------------------------------------
def guidance_mask(instruction: str, g: int) -> Tensor:
    """model.py:1262-1379: G x G mask for a named instruction, flattened.
    Unknown names give the all-ones mask."""
    name = instruction.lower()
    mask = torch.ones(g, g)
    if name == "center":
        fy, fx, r, hi, mid = g // 2, g // 2, max(1, g // 4), 3.0, 1.5
    else:
        pos = {"left": (g // 2, g // 4), "right": (g // 2, g * 3 // 4),
               "top": (g // 4, g // 2), "bottom": (g * 3 // 4, g // 2),
               "top-left": (g // 4, g // 4), "topleft": (g // 4, g // 4),
               "top-right": (g // 4, g * 3 // 4), "topright": (g // 4, g * 3 // 4),
               "bottom-left": (g * 3 // 4, g // 4), "bottomleft": (g * 3 // 4, g // 4),
               "bottom-right": (g * 3 // 4, g * 3 // 4), "bottomright": (g * 3 // 4, g * 3 // 4)}
        if name not in pos:
            return mask.flatten()
        (fy, fx), r, hi, mid = pos[name], max(1, g // 6), 5.0, 2.0
    for y in range(g):
        for x in range(g):
            d = math.sqrt((y - fy) ** 2 + (x - fx) ** 2)
            if d <= r:
                mask[y, x] = hi
            elif d <= r * 2:
                mask[y, x] = mid
    return mask.flatten()


def resolve_guidance(guidance: Union[str, Tensor], n: int) -> Tensor:
    """model.py:1262-1398: string -> mask; tensor of another square length ->
    bilinear resize (align_corners=False) to the patch grid."""
    if isinstance(guidance, str):
        guidance = guidance_mask(guidance, int(math.sqrt(n)))
    if guidance.size(0) != n:
        gs, ts = int(math.sqrt(guidance.size(0))), int(math.sqrt(n))
        g2 = F.interpolate(guidance.view(1, 1, gs, gs), size=(ts, ts),
                           mode="bilinear", align_corners=False)
        guidance = g2.squeeze().flatten()
    return guidance


def guided_attention(base_attention: Tensor, guidance: Tensor) -> Tensor:
    """model.py:1401-1409: softmax((0.7 g + 0.3 a) / 0.05)."""
    mixed = 0.7 * guidance.unsqueeze(0).expand(base_attention.size(0), -1) \
        + (1 - 0.7) * base_attention
    return torch.softmax(mixed / 0.05, dim=-1)


# --------------------------------------------------------------------------
# Whole-model forwards (de-duplicated; every repeat in the reference's
# schedule is bit-identical, model.py:1008/1021/1112)
# --------------------------------------------------------------------------
def model_config_from_sd(sd: SD) -> dict:
    n_iter = 0
    while f"focal_stream.focal_streams.{n_iter}.query_proj.weight" in sd:
        n_iter += 1
    single = "focal_stream.query_proj.weight" in sd            # a bare FocalStream (model.py:869-870)
    return {"use_ambient": "ambient_stream.mlp.0.weight" in sd, "single_focal": single,
            "use_focal": n_iter > 0 or single, "num_iterations": n_iter,
            "use_exif": "exif_prior.camera_embedding.weight" in sd}


def forward(sd: SD, images: Tensor, exif: Optional[Dict[str, Tensor]] = None,
            return_attention: bool = False, focus_strength: float = 1.5,
            n_heads: int = 12):
    """model.py:1064-1155 via get_features_aligned (:960-1056): backbone,
    ambient(CLS), iterative focal(patches), EXIF prior, concat in that order,
    zero-pad to 192, fusion, heads."""
    cfg = model_config_from_sd(sd)
    lhs = dinov2_forward(sd, images, n_heads=n_heads)
    cls, patches = lhs[:, 0], lhs[:, 1:]
    feats, attn = [], None
    if cfg["use_ambient"]:
        feats.append(ambient_stream(sd, cls))
    if cfg["single_focal"]:
        f, attn = focal_stream(sd, "focal_stream.", patches)
        feats.append(f)
    elif cfg["use_focal"]:
        f, attn = iterative_focal_stream(sd, patches, cfg["num_iterations"], focus_strength)
        feats.append(f)
    if cfg["use_exif"] and exif is not None:
        feats.append(exif_prior(sd, exif))
    z = torch.cat(feats, dim=1)
    if z.size(1) < 192:                                          # :1035-1039
        z = torch.cat([z, torch.zeros(z.size(0), 192 - z.size(1))], dim=1)
    depth, conf, _ = fusion_and_heads(sd, z)
    if return_attention:
        return depth, conf, attn
    return depth, conf


def forward_with_guidance(sd: SD, images: Tensor, exif: Optional[Dict[str, Tensor]],
                          guidance: Union[str, Tensor, None],
                          temp_projection: Optional[Tuple[Tensor, Tensor]] = None,
                          focus_strength: float = 1.5, n_heads: int = 12):
    """model.py:1157-1240.  Returns (depth, conf, guided_attention).

    With the iterative focal stream the reference projects the guided token
    sum through a freshly random-initialised Linear(768,64) on every call
    (:1417-1422), so its guided depth is not reproducible; the projection is
    an explicit argument here.  With ``temp_projection=None`` depth and conf
    are returned as None and only the attention (which is deterministic) is
    meaningful."""
    cfg = model_config_from_sd(sd)
    lhs = dinov2_forward(sd, images, n_heads=n_heads)
    cls, patches = lhs[:, 0], lhs[:, 1:]
    if cfg["single_focal"]:
        base_f, base_a = focal_stream(sd, "focal_stream.", patches)
    else:
        base_f, base_a = iterative_focal_stream(sd, patches, cfg["num_iterations"], focus_strength)
    if guidance is None:
        focal_f, attn = base_f, base_a
    else:
        g = resolve_guidance(guidance, patches.size(1))
        attn = guided_attention(base_a, g)
        weighted = torch.sum(patches * attn.unsqueeze(-1), dim=1)   # :1412-1414
        if cfg["single_focal"]:                                     # :1417-1418: the stream's own projection
            focal_f = _lin(sd, "focal_stream.projection.3", F.relu(_lin(sd, "focal_stream.projection.0", weighted)))
        elif temp_projection is None:
            return None, None, attn
        else:
            focal_f = F.linear(weighted, temp_projection[0], temp_projection[1])
    feats = []
    if cfg["use_ambient"]:
        feats.append(ambient_stream(sd, cls))
    feats.append(focal_f)
    if cfg["use_exif"] and exif is not None:
        feats.append(exif_prior(sd, exif))
    depth, conf, _ = fusion_and_heads(sd, torch.cat(feats, dim=1))  # no zero-pad here (:1223-1226)
    return depth, conf, attn


# --------------------------------------------------------------------------
# Faithful-schedule replay, used only to time the reference's as-executed
# CPU cost (2 backbone + 3 iterative focal passes per fresh forward, and the
# per-call Python position-table loop of model.py:149-166).
# --------------------------------------------------------------------------
def _pe2d_table_loop(num_patches: int, dim: int) -> Tensor:
    pe = torch.zeros(num_patches, dim)
    g = int(num_patches ** 0.5)
    half = dim // 2
    for i in range(num_patches):
        row, col = i // g, i % g
        div = torch.exp(torch.arange(0, half, 2, dtype=torch.float) * -(math.log(10000.0) / half))
        pe[i, 0:half:2] = torch.sin(row * div)
        pe[i, 1:half:2] = torch.cos(row * div)
        div = torch.exp(torch.arange(0, half, 2, dtype=torch.float) * -(math.log(10000.0) / half))
        pe[i, half::2] = torch.sin(col * div)
        pe[i, half + 1::2] = torch.cos(col * div)
    return pe


def forward_faithful_schedule(sd: SD, images: Tensor, exif, focus_strength: float = 1.5, n_heads: int = 12):
    """Replays the call structure of a fresh reference ``forward`` (attribute
    ``_last_attention_weights`` unset, model.py:960-1113): backbone x2,
    iterative focal x3, ambient x2, with the Python position loop rebuilt in
    every FocalStream call.  Returns (depth, conf)."""
    cfg = model_config_from_sd(sd)
    n_it = cfg["num_iterations"]

    def focal(patches):
        for _ in range(n_it):
            _pe2d_table_loop(patches.size(1), patches.size(2))
        return iterative_focal_stream(sd, patches, n_it, focus_strength)

    lhs = dinov2_forward(sd, images, n_heads=n_heads)    # :973
    cls, patches = lhs[:, 0], lhs[:, 1:]
    ambient_stream(sd, cls)                              # :983 (dead)
    focal(patches)                                       # :1008 (dead)
    feats = [ambient_stream(sd, cls), focal(patches)[0]]  # :1018-1021
    if cfg["use_exif"] and exif is not None:
        feats.append(exif_prior(sd, exif))
    z = torch.cat(feats, dim=1)
    if z.size(1) < 192:
        z = torch.cat([z, torch.zeros(z.size(0), 192 - z.size(1))], dim=1)
    depth, conf, _ = fusion_and_heads(sd, z)
    lhs2 = dinov2_forward(sd, images, n_heads=n_heads)   # :1095
    focal(lhs2[:, 1:])                                   # :1112
    return depth, conf
