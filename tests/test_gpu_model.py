"""End-to-end parity of the HIP path (through the nn.Module API, which calls the
C ABI) against the CPU oracle and the golden vectors made from the reference."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cogaim_oracle as O  # noqa: E402
from tests.util import default_exif, load_golden, rel_err, synth_model_and_sd  # noqa: E402

torch.set_grad_enabled(False)

# north-star bar: <= 1e-3 relative on fp32 depth / confidence.  It is met with fp16 MFMA
# operands (the shipped default: 11 significand bits, fp32 accumulation).  bf16 operands
# (8 bits) run at the same MFMA rate but land at 1-2e-3 -- the reference's own bf16
# autocast is 1.8e-3 off its fp32 result -- so bf16 is held to a looser, documented bound.
DEPTH_TOL = 1e-3
TOL = {"fp16": dict(depth=1e-3, attn=2e-2), "bf16": dict(depth=5e-3, attn=0.2)}
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


def report(**kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(kw) + "\n")


def _golden_exif(z, key, device):
    return {k: torch.from_numpy(z[f"{key}:exif:{k}"]).to(device)
            for k in ("camera_idx", "focal_length", "aperture", "iso")}


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("size", [56, 70, 224])
def test_golden_forward(dtype, size):
    """Reference outputs (tests/golden/g7_full_model.npz) reproduced on the GPU."""
    z = load_golden("g7_full_model")
    model, _ = synth_model_and_sd(int(z["layers"]), int(z["native_grid"]), seed=int(z["seed"]), dtype=dtype)
    key = f"full:{size}"
    x = torch.from_numpy(z[key + ":x"]).cuda()
    exif = _golden_exif(z, key, "cuda")
    d, c, a = model(x, exif, return_attention=True)
    e = dict(depth=rel_err(d, torch.from_numpy(z[key + ":depth"])), conf=rel_err(c, torch.from_numpy(z[key + ":conf"])),
             attn=rel_err(a, torch.from_numpy(z[key + ":attn"])))
    report(test="golden_forward", dtype=dtype, size=size, **e)
    assert e["depth"] <= TOL[dtype]["depth"] and e["conf"] <= TOL[dtype]["depth"], e
    assert e["attn"] <= TOL[dtype]["attn"], e
    # [B,1]-shaped EXIF and the no-EXIF zero-pad branch
    d2, c2 = model(x, {k: v.unsqueeze(1) for k, v in exif.items()})
    assert torch.equal(d2, d) and torch.equal(c2, c)
    d3, c3 = model(x, None)
    assert rel_err(d3, torch.from_numpy(z[key + ":depth_noexif"])) <= TOL[dtype]["depth"]
    assert rel_err(c3, torch.from_numpy(z[key + ":conf_noexif"])) <= TOL[dtype]["depth"]
    assert d.shape == (2, 1) and c.shape == (2, 1) and a.shape == (2, (size // 14) ** 2)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_golden_guided(dtype):
    """9-instruction sweep: guided attention heat-maps against the reference's."""
    z = load_golden("g7_full_model")
    model, _ = synth_model_and_sd(int(z["layers"]), int(z["native_grid"]), seed=int(z["seed"]), dtype=dtype)
    for size in (56, 224):
        key = f"full:{size}"
        x = torch.from_numpy(z[key + ":x"]).cuda()
        exif = _golden_exif(z, key, "cuda")
        worst = 0.0
        for name in O.INSTRUCTIONS:
            d, c, ga = model.forward_with_guidance(x, exif, name, return_attention=True)
            ref = torch.from_numpy(z[f"{key}:guided:{name}"])
            err = float((ga.cpu() - ref).abs().max() / ref.max())
            worst = max(worst, err)
            assert torch.equal(model.get_attention_weights(), ga)
        report(test="golden_guided", dtype=dtype, size=size, attn_max_err_rel_peak=worst)
        assert worst <= 1e-2, worst
    # guided depth with the reference's per-call random projection pinned
    key = "full:56"
    x = torch.from_numpy(z[key + ":x"]).cuda()
    proj = (torch.from_numpy(z[key + ":guided_proj_w"]), torch.from_numpy(z[key + ":guided_proj_b"]))
    d, c = model.forward_with_guidance(x, _golden_exif(z, key, "cuda"), "center", temp_projection=proj)
    e = dict(depth=rel_err(d, torch.from_numpy(z[key + ":guided_depth"])), conf=rel_err(c, torch.from_numpy(z[key + ":guided_conf"])))
    report(test="golden_guided_depth", dtype=dtype, **e)
    assert e["depth"] <= TOL[dtype]["depth"] and e["conf"] <= TOL[dtype]["depth"], e


def test_golden_variants():
    z = load_golden("g7_full_model")
    model, _ = synth_model_and_sd(int(z["layers"]), int(z["native_grid"]), cams=0, seed=int(z["seed"]), dtype="fp16")
    x = torch.from_numpy(z["noexif:56:x"]).cuda()
    d, c, a = model(x, None, return_attention=True)
    assert rel_err(d, torch.from_numpy(z["noexif:56:depth"])) <= DEPTH_TOL
    assert rel_err(c, torch.from_numpy(z["noexif:56:conf"])) <= DEPTH_TOL
    # guided call without EXIF: the reference falls back to forward() but keeps the guided map cached
    d2, c2 = model.forward_with_guidance(x, None, "left")
    assert torch.equal(d2, d) and torch.equal(c2, c)
    assert not torch.equal(model.get_attention_weights(), a)
    model, _ = synth_model_and_sd(int(z["layers"]), int(z["native_grid"]), iters=2, seed=int(z["seed"]),
                                  dtype="fp16", focus_strength=0.4)
    x = torch.from_numpy(z["it2:56:x"]).cuda()
    d, c, a = model(x, _golden_exif(z, "it2:56", "cuda"), return_attention=True)
    assert rel_err(d, torch.from_numpy(z["it2:56:depth"])) <= DEPTH_TOL
    assert rel_err(c, torch.from_numpy(z["it2:56:conf"])) <= DEPTH_TOL
    assert rel_err(a, torch.from_numpy(z["it2:56:attn"])) <= 2e-2


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_full_size_vs_oracle(dtype):
    """12-layer ViT-B/14 at 518x518 (BASELINE config 2 geometry, B=2) against the CPU oracle."""
    model, sd = synth_model_and_sd(12, 37, seed=11, dtype=dtype)
    torch.manual_seed(5)
    x = torch.randn(2, 3, 518, 518)
    exif = default_exif(2)
    d_ref, c_ref, a_ref = O.forward(sd, x, exif, return_attention=True)
    d, c, a = model(x.cuda(), {k: v.cuda() for k, v in exif.items()}, return_attention=True)
    e = dict(depth=rel_err(d, d_ref), conf=rel_err(c, c_ref), attn=rel_err(a, a_ref))
    report(test="full_size", dtype=dtype, depth_ref=d_ref.flatten().tolist(), depth_gpu=d.flatten().tolist(), **e)
    assert e["depth"] <= TOL[dtype]["depth"] and e["conf"] <= TOL[dtype]["depth"], e
    _, _, g_ref = O.forward_with_guidance(sd, x, exif, "bottom-right")
    _, _, g = model.forward_with_guidance(x.cuda(), {k: v.cuda() for k, v in exif.items()}, "bottom-right",
                                          return_attention=True)
    gerr = float((g.cpu() - g_ref).abs().max() / g_ref.max())
    report(test="full_size_guided", dtype=dtype, attn_max_err_rel_peak=gerr)
    assert gerr <= 1e-2


def test_focal_stream_vs_oracle():
    """Iterative focal stream alone (fp32 tokens in, features + attention out), N = 1369 and a ragged N."""
    model, sd = synth_model_and_sd(1, 4, iters=3, seed=3, dtype="fp16")
    model._pack()
    for B, g in ((2, 37), (3, 5)):
        N, D = g * g, 768
        ws = model._workspace(B, g, g)
        torch.manual_seed(g)
        tokens = torch.randn(B, N + 1, D)
        ws["tokens"].copy_(tokens.view(B * (N + 1), D))
        ws["cat"].zero_()
        model._focal(B, ws)
        model._focal_fused_into(ws, B, 64)
        torch.cuda.synchronize()
        f_ref, a_ref = O.iterative_focal_stream(sd, tokens[:, 1:], 3, 1.5)
        ea, ef = rel_err(ws["attn"], a_ref), float((ws["cat"][:, 64:128].cpu() - f_ref).abs().max() / f_ref.abs().max())
        report(test="focal_stream", N=N, attn=ea, feat=ef)
        assert ea <= 2e-2 and ef <= 5e-3, (ea, ef)
        assert int(ws["flags"].sum()) == 0


def test_batch_independence_and_determinism():
    """Image-sharding property: each image's result does not depend on its batch mates."""
    model, _ = synth_model_and_sd(2, 4, seed=7, dtype="fp16")
    x = torch.randn(5, 3, 224, 224, device="cuda")
    exif = default_exif(5, "cuda")
    d, c = model(x, exif)
    d_again, c_again = model(x, exif)
    assert torch.equal(d, d_again) and torch.equal(c, c_again)
    for sl in (slice(0, 2), slice(2, 5)):
        ds, cs = model(x[sl], {k: v[sl] for k, v in exif.items()})
        assert torch.equal(ds, d[sl]) and torch.equal(cs, c[sl])


@pytest.mark.parametrize("size,B", [((266, 266), 1), ((224, 448), 3), ((98, 70), 33)])
def test_odd_shapes_vs_oracle(size, B):
    """Odd grids (19x19), non-square images (non-square patch count -> 1-D position table and centre
    bias, bicubic position-embedding resampling to a rectangular grid) and a ragged batch."""
    model, sd = synth_model_and_sd(2, 4, seed=21, dtype="fp16")
    torch.manual_seed(B)
    x = torch.randn(B, 3, *size)
    exif = default_exif(B)
    d_ref, c_ref, a_ref = O.forward(sd, x, exif, return_attention=True)
    d, c, a = model(x.cuda(), {k: v.cuda() for k, v in exif.items()}, return_attention=True)
    e = dict(depth=rel_err(d, d_ref), conf=rel_err(c, c_ref), attn=rel_err(a, a_ref))
    report(test="odd_shapes", size=list(size), B=B, **e)
    assert e["depth"] <= DEPTH_TOL and e["conf"] <= DEPTH_TOL and e["attn"] <= 3e-2, e


def test_high_resolution_eight_iterations():
    """BASELINE config 5 geometry: 1036x1036 input (74x74 = 5476 patches), 8 focal iterations, fp16
    operands; 2-layer backbone so that the CPU oracle finishes in seconds."""
    model, sd = synth_model_and_sd(2, 37, iters=8, seed=5, dtype="fp16")
    torch.manual_seed(8)
    x = torch.randn(1, 3, 1036, 1036)
    exif = default_exif(1)
    d_ref, c_ref, a_ref = O.forward(sd, x, exif, return_attention=True)
    d, c, a = model(x.cuda(), {k: v.cuda() for k, v in exif.items()}, return_attention=True)
    e = dict(depth=rel_err(d, d_ref), conf=rel_err(c, c_ref), attn=rel_err(a, a_ref))
    report(test="config5_1036_I8", **e)
    assert a.shape == (1, 5476)
    assert e["depth"] <= DEPTH_TOL and e["conf"] <= DEPTH_TOL and e["attn"] <= 3e-2, e
    model.synchronize()


def test_graph_replay_equals_eager_launches():
    """The captured HIP graph replays the same kernels on the same buffers: results are bit-identical to the
    eager launch sequence, for new inputs on every call, for both forwards, and after new weights are loaded."""
    model, sd = synth_model_and_sd(2, 4, seed=7, dtype="fp16")
    eager, _ = synth_model_and_sd(2, 4, seed=7, dtype="fp16")
    eager.use_graph = False
    assert model.use_graph
    proj = (torch.randn(64, 768), torch.randn(64))
    for call_no in range(4):                                   # 0: eager, 1: capture + replay, 2..: replay
        x = torch.randn(3, 3, 224, 224, device="cuda")
        exif = default_exif(3, "cuda")
        d, c, a = model(x, exif, return_attention=True)
        d0, c0, a0 = eager(x, exif, return_attention=True)
        assert torch.equal(d, d0) and torch.equal(c, c0) and torch.equal(a, a0), call_no
        g = model.forward_with_guidance(x, exif, "top-left", True, temp_projection=proj)
        g0 = eager.forward_with_guidance(x, exif, "top-left", True, temp_projection=proj)
        assert all(torch.equal(u, v) for u, v in zip(g, g0)), call_no
        f, f0 = model.get_features(x, None), eager.get_features(x, None)
        assert torch.equal(f, f0), call_no
    ws = next(iter(model._ws.values()))
    assert sum(isinstance(v, torch.cuda.CUDAGraph) for v in ws["graphs"].values()) == 3
    # new weights: packed copies move, the graphs must be rebuilt, not replayed on stale pointers
    sd2 = {k: (v * 1.01 if v.is_floating_point() else v) for k, v in sd.items()}
    model.load_state_dict(sd2)
    eager.load_state_dict(sd2)
    for _ in range(3):
        x = torch.randn(3, 3, 224, 224, device="cuda")
        exif = default_exif(3, "cuda")
        assert all(torch.equal(u, v) for u, v in zip(model(x, exif), eager(x, exif)))
    model.synchronize()


def test_empty_batch_and_single_image():
    model, sd = synth_model_and_sd(2, 4, seed=7, dtype="fp16")
    x0 = torch.zeros(0, 3, 56, 56, device="cuda")
    d, c, a = model(x0, None, return_attention=True)
    assert d.shape == (0, 1) and c.shape == (0, 1) and a.shape == (0, 16)
    g = model.forward_with_guidance(x0, None, "center")
    assert g[0].shape == (0, 1)
    x1 = torch.randn(1, 3, 56, 56, device="cuda")
    d1, c1 = model(x1, default_exif(1, "cuda"))
    d_ref, c_ref, _ = O.forward(sd, x1.cpu(), default_exif(1), return_attention=True)
    assert rel_err(d1, d_ref) <= DEPTH_TOL and rel_err(c1, c_ref) <= DEPTH_TOL
