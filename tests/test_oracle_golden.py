"""The oracle against the golden vectors produced by the real reference
(oracle/make_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cogaim_oracle as O
from oracle.weights import reference_state_spec, synth_state_dict

torch.set_grad_enabled(False)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def sd_from(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def close(a, b, rtol=1e-5, atol=1e-6):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


def test_tiny_backbone(golden_dir):
    z = load(golden_dir, "g1_tiny_backbone")
    sd = sd_from(z, "w:")
    sd = {"backbone." + k: v for k, v in sd.items()}
    for size in (56, 70, 98):   # native grid, bicubic up-sampled 5x5 and 7x7
        y = O.dinov2_forward(sd, torch.from_numpy(z[f"x{size}"]), n_heads=4)
        close(y, z[f"y{size}"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_focal_stream(golden_dir, tag):
    z = load(golden_dir, "g2_focal_stream")
    sd = sd_from(z, f"{tag}:w:")
    x = torch.from_numpy(z[f"{tag}:x"])
    f, a = O.focal_stream(sd, "", x)
    close(f, z[f"{tag}:f"], rtol=1e-4, atol=1e-6)
    close(a, z[f"{tag}:a"], rtol=1e-4, atol=1e-8)
    close(O.pe2d_table(x.shape[1], x.shape[2]), z[f"{tag}:pe"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_iterative_focal(golden_dir, tag):
    z = load(golden_dir, "g3_iterative_focal")
    sd = sd_from(z, f"{tag}:w:")
    f, a = O.iterative_focal_stream(sd, torch.from_numpy(z[f"{tag}:x"]),
                                    int(z[f"{tag}:iters"]), float(z[f"{tag}:fs"]), prefix="")
    close(f, z[f"{tag}:f"], rtol=1e-4, atol=1e-6)
    close(a, z[f"{tag}:a"], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("g", [4, 5, 16, 37])
def test_guidance(golden_dir, g):
    z = load(golden_dir, "g5_guidance")
    base = torch.from_numpy(z[f"g{g}:base"])
    names = [k.split(":", 1)[1] for k in z.files
             if k.startswith(f"g{g}:") and k.split(":", 1)[1] not in
             ("base", "tensor196", "tensor196_in")]
    assert len(names) == 15
    for name in names:
        att = O.guided_attention(base, O.resolve_guidance(name, g * g))
        close(att, z[f"g{g}:{name}"], rtol=1e-4, atol=1e-9)
    att = O.guided_attention(base, O.resolve_guidance(torch.from_numpy(z[f"g{g}:tensor196_in"]), g * g))
    close(att, z[f"g{g}:tensor196"], rtol=1e-4, atol=1e-9)


def test_pe_table_full_size(golden_dir):
    z = load(golden_dir, "g6_pe_table")
    for n in (256, 1369):
        pe = O.pe2d_table(n, 768)
        close(pe[torch.from_numpy(z[f"n{n}:rows"])], z[f"n{n}:pe_rows"], rtol=0, atol=1e-6)
        close(pe.double().sum(0), z[f"n{n}:colsum"], rtol=0, atol=1e-4)
        close(pe.double().sum(1), z[f"n{n}:rowsum"], rtol=0, atol=1e-4)
        close(O._pe2d_table_loop(n, 768)[:40], pe[:40], rtol=0, atol=1e-6)


def _full_sd(z, iters=3, cams=71):
    spec = reference_state_spec(layers=int(z["layers"]), native_grid=int(z["native_grid"]),
                                num_iterations=iters, num_cameras=cams)
    return synth_state_dict(spec, seed=int(z["seed"]))


def _exif(z, key):
    return {k: torch.from_numpy(z[f"{key}:exif:{k}"])
            for k in ("camera_idx", "focal_length", "aperture", "iso")}


@pytest.mark.parametrize("size", [56, 70, 224])
def test_full_model_forward(golden_dir, size):
    z = load(golden_dir, "g7_full_model")
    sd = _full_sd(z)
    key = f"full:{size}"
    x = torch.from_numpy(z[key + ":x"])
    exif = _exif(z, key)
    d, c, a = O.forward(sd, x, exif, return_attention=True)
    close(d, z[key + ":depth"], rtol=1e-4)
    close(c, z[key + ":conf"], rtol=1e-4)
    close(a, z[key + ":attn"], rtol=1e-3, atol=1e-8)
    # the goldens keep the confidence ReLU alive (conf != sigmoid(bias))
    assert not np.allclose(z[key + ":conf"], z[key + ":conf"][0])
    exif2 = {k: v.unsqueeze(1) for k, v in exif.items()}
    d2, c2 = O.forward(sd, x, exif2)
    close(d2, d, rtol=0, atol=0)
    d3, c3 = O.forward(sd, x, None)
    close(d3, z[key + ":depth_noexif"], rtol=1e-4)
    close(c3, z[key + ":conf_noexif"], rtol=1e-4)


@pytest.mark.parametrize("size", [56, 70, 224])
def test_full_model_guided(golden_dir, size):
    z = load(golden_dir, "g7_full_model")
    sd = _full_sd(z)
    key = f"full:{size}"
    x = torch.from_numpy(z[key + ":x"])
    for name in O.INSTRUCTIONS:
        _, _, ga = O.forward_with_guidance(sd, x, _exif(z, key), name)
        close(ga, z[f"{key}:guided:{name}"], rtol=2e-3, atol=1e-9)
    if size == 56:
        proj = (torch.from_numpy(z[key + ":guided_proj_w"]), torch.from_numpy(z[key + ":guided_proj_b"]))
        d, c, _ = O.forward_with_guidance(sd, x, _exif(z, key), "center", temp_projection=proj)
        close(d, z[key + ":guided_depth"], rtol=1e-4)
        close(c, z[key + ":guided_conf"], rtol=1e-4)


def test_full_model_variants(golden_dir):
    z = load(golden_dir, "g7_full_model")
    x = torch.from_numpy(z["noexif:56:x"])
    d, c, a = O.forward(_full_sd(z, cams=0), x, None, return_attention=True)
    close(d, z["noexif:56:depth"], rtol=1e-4)
    close(c, z["noexif:56:conf"], rtol=1e-4)
    close(a, z["noexif:56:attn"], rtol=1e-3, atol=1e-8)
    x = torch.from_numpy(z["it2:56:x"])
    d, c, a = O.forward(_full_sd(z, iters=2), x, _exif(z, "it2:56"), return_attention=True,
                        focus_strength=0.4)
    close(d, z["it2:56:depth"], rtol=1e-4)
    close(c, z["it2:56:conf"], rtol=1e-4)
    close(a, z["it2:56:attn"], rtol=1e-3, atol=1e-8)


def test_curiosity_reward_and_ring_buffer_vs_reference():
    """oracle.curiosity_reward + the ring-buffer rule (src/model.py:760-773) reproduce the exploration history the
    real reference left after its first forward (tests/golden/g8_curiosity_state.npz, call 'a': two curiosity
    evaluations of 3 images, pointer starting at 997)."""
    import numpy as np
    import torch
    from oracle import cogaim_oracle as O
    from oracle.weights import reference_state_spec, synth_state_dict
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_curiosity_state.npz"))
    sd = synth_state_dict(reference_state_spec(layers=int(z["layers"]), native_grid=int(z["native_grid"])), seed=int(z["seed"]))
    x = torch.from_numpy(z["x56"])
    with torch.no_grad():
        cls = O.dinov2_forward(sd, x)[:, 0]
        torch.manual_seed(int(z["torch_seed"]))
        hist, ptr_ = np.zeros(1000, np.float32), int(z["pointer0"])
        for _ in range(2):
            eps = torch.randn(3, 192)
            noise = torch.randn(3, 768) * 0.01
            for r in O.curiosity_reward(sd, cls, eps, noise).tolist():
                hist[ptr_ % 1000] = r
                ptr_ = (ptr_ + 1) % 1000
    assert ptr_ == int(z["a:pointer"])
    assert np.allclose(hist, z["a:history"], rtol=1e-5, atol=1e-7), float(np.abs(hist - z["a:history"]).max())


def test_single_patch_degenerate_attention_vs_reference():
    """14 x 14 inputs (one patch): the reference takes all three degenerate-attention fallbacks
    (src/model.py:242-257, tests/golden/g9_single_patch.npz records the noise draws) and still returns the
    attention vector [1.0]; the oracle, which restates those branches, agrees."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_single_patch.npz"))
    sd = synth_state_dict(reference_state_spec(layers=int(z["layers"]), native_grid=int(z["native_grid"])), seed=int(z["seed"]))
    x = torch.from_numpy(z["x"])
    exif = {k: torch.from_numpy(z["exif:" + k]) for k in ("camera_idx", "focal_length", "aperture", "iso")}
    d, c, a = O.forward(sd, x, exif, return_attention=True)
    assert int(z["fallback_noise_draws"]) >= 3
    assert np.allclose(d.numpy(), z["depth"], rtol=1e-4) and np.allclose(c.numpy(), z["conf"], rtol=1e-4)
    assert np.array_equal(a.numpy(), z["attn"]) and np.array_equal(z["attn"], np.ones((3, 1), np.float32))


def test_bare_focal_stream_variant_vs_reference():
    """'focal_stream' in cognitive_modules (one FocalStream, its own projection on the guided path):
    tests/golden/g10_bare_focal_stream.npz from the real reference."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_bare_focal_stream.npz"))
    mods = ("ambient_stream", "focal_stream", "exif_prior_database")
    sd = synth_state_dict(reference_state_spec(layers=int(z["layers"]), native_grid=int(z["native_grid"]), modules=mods),
                          seed=int(z["seed"]))
    x = torch.from_numpy(z["x"])
    exif = {k: torch.from_numpy(z["exif:" + k]) for k in ("camera_idx", "focal_length", "aperture", "iso")}
    d, c, a = O.forward(sd, x, exif, return_attention=True)
    gd, gc, ga = O.forward_with_guidance(sd, x, exif, "bottom-left")
    for got, key in ((d, "depth"), (c, "conf"), (a, "attn"), (gd, "guided_depth"), (gc, "guided_conf"), (ga, "guided_attn")):
        assert np.allclose(got.numpy(), z[key], rtol=1e-4, atol=1e-7), key


def test_large_backbone_vs_reference():
    """backbone_size='large' (reference src/model.py:807-809: 1024 columns, 16 heads) against
    tests/golden/g11_large_backbone.npz from the real reference (two layers, 4 x 4 native grid)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_large_backbone.npz"))
    hidden, heads = int(z["hidden"]), int(z["heads"])
    sd = synth_state_dict(reference_state_spec(hidden=hidden, layers=int(z["layers"]), native_grid=int(z["native_grid"]),
                                               num_iterations=3, num_cameras=71), seed=int(z["seed"]))
    exif = {k: torch.from_numpy(z["exif:" + k]) for k in ("camera_idx", "focal_length", "aperture", "iso")}
    for size in (56, 98):
        x = torch.from_numpy(z[f"x{size}"])
        d, c, a = O.forward(sd, x, exif, return_attention=True, n_heads=heads)
        for got, key in ((d, "depth"), (c, "conf"), (a, "attn")):
            assert np.allclose(got.numpy(), z[f"{key}{size}"], rtol=1e-4, atol=1e-7), (key, size)
    proj = (torch.from_numpy(z["guided_proj_w"]), torch.from_numpy(z["guided_proj_b"]))
    gd, gc, ga = O.forward_with_guidance(sd, x, exif, "top-left", temp_projection=proj, n_heads=heads)
    for got, key in ((gd, "guided_depth"), (gc, "guided_conf"), (ga, "guided_attn")):
        assert np.allclose(got.numpy(), z[key], rtol=2e-4, atol=1e-7), key
    # the 768-wide curiosity module raises on 1024 columns before drawing anything; the reference catches it
    assert int(z["curiosity_failures56"]) == 2 and bool(z["history_untouched"])


@pytest.mark.parametrize("tag", ["rows_equal4", "rows_differ4", "rows_differ3", "rows_equal2"])
def test_degenerate_attention_fallbacks_vs_reference(tag):
    """Inputs on which the real FocalStream takes its first (diagonal) or second (row maximum) fallback
    (reference src/model.py:242-249) on grids of 2, 3 and 2 x 2 patches: tests/golden/g12_degenerate_attention.npz."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_degenerate_attention.npz"))
    sd = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + ":w:")}
    f, a = O.focal_stream(sd, "", torch.from_numpy(z[tag + ":x"]))
    assert np.allclose(a.numpy(), z[tag + ":a"], rtol=1e-4, atol=1e-7)
    assert np.allclose(f.numpy(), z[tag + ":f"], rtol=1e-4, atol=1e-6)
    assert int(z[tag + ":branch"]) == (1 if "differ" in tag else 2)
