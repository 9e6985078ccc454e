"""Generate tests/golden/*.npz by running the REAL reference.  Build-container only.

Path-loads /root/reference/src/model.py (``import src.model`` would drag in
torchvision through src/__init__.py:4), replaces the hub fetch
``Dinov2Model.from_pretrained('facebook/dinov2-base')`` (src/model.py:814) by
a local random-init ``Dinov2Model(Dinov2Config(...))``, runs the reference on
seeded inputs and stores inputs + outputs (+ weights when they are small).
Nothing from the reference is copied: the fixtures are data only.

    python oracle/make_goldens.py            # rewrites tests/golden/
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import os
import sys

import numpy as np
import torch

os.environ.setdefault("HF_HUB_OFFLINE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.weights import reference_state_spec, synth_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/src/model.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_model", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        yield buf


def npd(sd, prefix="w:"):
    return {prefix + k: v.detach().numpy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def g1_tiny_backbone(ref):
    from transformers import Dinov2Config, Dinov2Model
    torch.manual_seed(0)
    cfg = Dinov2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                       image_size=56, patch_size=14)
    model = Dinov2Model(cfg).eval()
    # HF init (std 0.02) gives near-uniform attention; widen q/k so softmax is exercised.
    with torch.no_grad():
        for layer in model.encoder.layer:
            layer.attention.attention.query.weight.mul_(8.0)
            layer.attention.attention.key.weight.mul_(8.0)
            layer.layer_scale1.lambda1.add_(0.1 * torch.randn(64))
            layer.norm1.bias.add_(0.1 * torch.randn(64))
    arrays = npd(model.state_dict())
    for size in (56, 70, 98):
        x = torch.randn(2, 3, size, size)
        with torch.no_grad():
            y = model(x, output_hidden_states=True).last_hidden_state
        arrays[f"x{size}"] = x.numpy()
        arrays[f"y{size}"] = y.numpy()
    save("g1_tiny_backbone", **arrays)


def g2_focal(ref):
    arrays = {}
    torch.manual_seed(1)
    for tag, dim, n in (("a", 64, 16), ("b", 64, 25), ("c", 64, 20), ("d", 128, 49)):
        fs = ref.FocalStream(dim, curiosity_guided=False).eval()
        x = torch.randn(2, n, dim)
        captured = {}
        hook = fs.query_proj.register_forward_pre_hook(
            lambda mod, inp: captured.__setitem__("xt", inp[0].detach().clone()))
        with torch.no_grad(), quiet():
            f, a = fs(x)
        hook.remove()
        arrays.update(npd(fs.state_dict(), f"{tag}:w:"))
        arrays[f"{tag}:x"] = x.numpy()
        arrays[f"{tag}:f"] = f.numpy()
        arrays[f"{tag}:a"] = a.numpy()
        arrays[f"{tag}:pe"] = (captured["xt"] - x)[0].numpy()
    save("g2_focal_stream", **arrays)


def g3_iterative(ref):
    arrays = {}
    torch.manual_seed(2)
    for tag, iters, fstr, n in (("a", 2, 0.1, 16), ("b", 3, 1.5, 25), ("c", 8, 1.5, 16)):
        it = ref.IterativeFocalStream(64, num_iterations=iters, curiosity_guided=False,
                                      focus_strength=fstr).eval()
        x = torch.randn(2, n, 64)
        with torch.no_grad(), quiet():
            f, a = it(x)
        arrays.update(npd(it.state_dict(), f"{tag}:w:"))
        arrays[f"{tag}:x"] = x.numpy()
        arrays[f"{tag}:f"] = f.numpy()
        arrays[f"{tag}:a"] = a.numpy()
        arrays[f"{tag}:iters"] = iters
        arrays[f"{tag}:fs"] = fstr
    save("g3_iterative_focal", **arrays)


def g5_guidance(ref):
    """Masks and guided softmax straight from CognitiveAimModel._guided_focal_stream
    driven with a stand-in ``self`` (only focal_stream / feature_dim are touched)."""
    arrays = {}
    torch.manual_seed(3)
    names = ["center", "left", "right", "top", "bottom", "top-left", "top-right",
             "bottom-left", "bottom-right", "topleft", "TopRight", "bottomleft",
             "BOTTOMRIGHT", "Center", "somewhere-else"]

    class Stub:
        feature_dim = 4

    for g in (4, 5, 16, 37):
        n = g * g
        base = torch.rand(2, n)
        base = base / base.sum(-1, keepdim=True)
        stub = Stub()
        stub.focal_stream = lambda p, c, _b=base: (torch.zeros(2, 64), _b)
        patches = torch.randn(2, n, 4)
        arrays[f"g{g}:base"] = base.numpy()
        for name in names:
            with torch.no_grad(), quiet():
                _, att = ref.CognitiveAimModel._guided_focal_stream(stub, patches, None, name)
            arrays[f"g{g}:{name}"] = att.numpy()
        gt = torch.rand(196)
        with torch.no_grad(), quiet():
            _, att = ref.CognitiveAimModel._guided_focal_stream(stub, patches, None, gt)
        arrays[f"g{g}:tensor196_in"] = gt.numpy()
        arrays[f"g{g}:tensor196"] = att.numpy()
    save("g5_guidance", **arrays)


def g6_pe_table(ref):
    """Sub-sampled rows of the full-size position table (N=1369, D=768)."""
    fs = ref.FocalStream(768, curiosity_guided=False).eval()
    arrays = {}
    for n in (256, 1369):
        x = torch.zeros(1, n, 768)
        captured = {}
        hook = fs.query_proj.register_forward_pre_hook(
            lambda mod, inp: captured.__setitem__("xt", inp[0].detach().clone()))
        with torch.no_grad(), quiet():
            fs(x)
        hook.remove()
        pe = captured["xt"][0]
        rows = sorted({0, 1, int(n ** 0.5) - 1, int(n ** 0.5), n // 2, n - 1})
        arrays[f"n{n}:rows"] = np.array(rows)
        arrays[f"n{n}:pe_rows"] = pe[rows].numpy()
        arrays[f"n{n}:colsum"] = pe.double().sum(0).numpy()
        arrays[f"n{n}:rowsum"] = pe.double().sum(1).numpy()
    save("g6_pe_table", **arrays)


def g7_full_model(ref):
    """Full-width model, 2-layer backbone, native 4x4 grid, synthetic weights
    regenerated from (seed, key) by oracle/weights.py."""
    from transformers import Dinov2Config, Dinov2Model
    layers, grid = 2, 4
    ref.Dinov2Model.from_pretrained = classmethod(
        lambda cls, name, *a, **k: Dinov2Model(Dinov2Config(image_size=14 * grid,
                                                            num_hidden_layers=layers)))
    arrays = {"layers": layers, "native_grid": grid, "seed": 7}
    for tag, cam, iters in (("full", {"num_cameras": 71}, 3), ("noexif", None, 3),
                            ("it2", {"num_cameras": 71}, 2)):
        cfg = {"cognitive_modules": ["ambient_stream", "iterative_focal_stream",
                                     "exif_prior_database"],
               "focal_config": {"num_iterations": iters,
                                "focus_strength": 1.5 if iters == 3 else 0.4}}
        with quiet():
            model = ref.create_model(cfg, cam).eval()
        spec = reference_state_spec(layers=layers, native_grid=grid, num_iterations=iters,
                                    num_cameras=71 if cam else 0)
        ref_sd = model.state_dict()
        assert list(ref_sd.keys()) == list(spec.keys()), \
            (set(ref_sd) ^ set(spec))
        for k, (shp, dt) in spec.items():
            assert tuple(ref_sd[k].shape) == tuple(shp) and ref_sd[k].dtype == dt, k
        sd = synth_state_dict(spec, seed=7)
        model.load_state_dict(sd, strict=True)
        for size in ((56, 70, 224) if tag == "full" else (56,)):
            torch.manual_seed(100 + size)
            x = torch.randn(2, 3, size, size)
            exif = {"camera_idx": torch.tensor([0, 5]),
                    "focal_length": torch.tensor([50.0, 24.0]),
                    "aperture": torch.tensor([2.8, 8.0]),
                    "iso": torch.tensor([100.0, 1600.0])}
            key = f"{tag}:{size}"
            arrays[key + ":x"] = x.numpy()
            for k, v in exif.items():
                arrays[f"{key}:exif:{k}"] = v.numpy()
            if hasattr(model, "_last_attention_weights"):
                delattr(model, "_last_attention_weights")
            with torch.no_grad(), quiet() as log:
                d, c, a = model(x, exif if cam else None, return_attention=True)
            assert "exception" not in log.getvalue(), log.getvalue()
            arrays[key + ":depth"], arrays[key + ":conf"], arrays[key + ":attn"] = \
                d.numpy(), c.numpy(), a.numpy()
            if tag != "full":
                continue
            # [B,1]-shaped EXIF must give the same answer (src/model.py:492-508)
            exif2 = {k: v.unsqueeze(1) for k, v in exif.items()}
            with torch.no_grad(), quiet():
                d2, c2 = model(x, exif2)
            assert torch.equal(d2, d) and torch.equal(c2, c)
            # exif omitted -> zero-pad branch (src/model.py:1035-1039)
            with torch.no_grad(), quiet():
                d3, c3 = model(x, None)
            arrays[key + ":depth_noexif"], arrays[key + ":conf_noexif"] = d3.numpy(), c3.numpy()
            # guided path: attention for every instruction; depth for 'center'
            # with the per-call random projection captured (src/model.py:1417-1422)
            for name in ("center", "left", "right", "top", "bottom", "top-left",
                         "top-right", "bottom-left", "bottom-right"):
                created = []
                real_linear = ref.nn.Linear

                class Spy(real_linear):
                    def __init__(self, *a, **k):
                        super().__init__(*a, **k)
                        created.append(self)

                ref.nn.Linear = Spy
                try:
                    with torch.no_grad(), quiet() as log:
                        gd, gc, ga = model.forward_with_guidance(x, exif, name,
                                                                 return_attention=True)
                finally:
                    ref.nn.Linear = real_linear
                assert "failed" not in log.getvalue(), log.getvalue()
                arrays[f"{key}:guided:{name}"] = ga.numpy()
                if name == "center" and size == 56:
                    assert len(created) == 1
                    arrays[key + ":guided_proj_w"] = created[0].weight.detach().numpy()
                    arrays[key + ":guided_proj_b"] = created[0].bias.detach().numpy()
                    arrays[key + ":guided_depth"] = gd.numpy()
                    arrays[key + ":guided_conf"] = gc.numpy()
    save("g7_full_model", **arrays)


def g8_curiosity_state(ref):
    """State the curiosity module leaves behind (src/model.py:760-773, :775-793) and where the guided path's
    per-call Linear falls in torch's CPU random stream (:609, :744, :1417-1422), over a fixed call sequence
    under one seed.  Same model and weights as g7 'full'; the ring pointer starts at 997 so that it wraps."""
    from transformers import Dinov2Config, Dinov2Model
    layers, grid = 2, 4
    ref.Dinov2Model.from_pretrained = classmethod(
        lambda cls, name, *a, **k: Dinov2Model(Dinov2Config(image_size=14 * grid, num_hidden_layers=layers)))
    cfg = {"cognitive_modules": ["ambient_stream", "iterative_focal_stream", "exif_prior_database"],
           "focal_config": {"num_iterations": 3, "focus_strength": 1.5}}
    with quiet():
        model = ref.create_model(cfg, {"num_cameras": 71}).eval()
    spec = reference_state_spec(layers=layers, native_grid=grid, num_iterations=3, num_cameras=71)
    model.load_state_dict(synth_state_dict(spec, seed=7), strict=True)
    model.curiosity_module.history_pointer.fill_(997)
    g = torch.Generator().manual_seed(31)
    x56, x70 = torch.randn(3, 3, 56, 56, generator=g), torch.randn(2, 3, 70, 70, generator=g)
    exif = {"camera_idx": torch.tensor([0, 5, 70]), "focal_length": torch.tensor([50.0, 24.0, 135.0]),
            "aperture": torch.tensor([2.8, 8.0, 1.8]), "iso": torch.tensor([100.0, 1600.0, 400.0])}
    arrays = {"layers": layers, "native_grid": grid, "seed": 7, "torch_seed": 4242, "pointer0": 997,
              "x56": x56.numpy(), "x70": x70.numpy()}
    for k, v in exif.items():
        arrays["exif:" + k] = v.numpy()
    created = []
    real_linear = ref.nn.Linear

    class Spy(real_linear):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            created.append(self)

    def snap(tag, out):
        arrays[tag + ":history"] = model.curiosity_module.exploration_history.numpy().copy()
        arrays[tag + ":pointer"] = int(model.curiosity_module.history_pointer)
        st = model.get_exploration_stats()
        arrays[tag + ":stats"] = np.array([st["mean"], st["std"], st["max"], st.get("min", 0.0), st["samples"]])
        arrays[tag + ":depth"], arrays[tag + ":conf"] = out[0].numpy(), out[1].numpy()

    torch.manual_seed(4242)
    ref.nn.Linear = Spy
    try:
        with torch.no_grad(), quiet() as log:
            snap("a", model(x56, exif))                                   # fresh: 2 curiosity calls
            snap("b", model(x56, exif, return_attention=True))            # attribute set: 1 + 1 calls
            snap("c", model.forward_with_guidance(x56, exif, "left"))     # 1 call, then the Linear
            snap("d", model(x70, None))                                   # 1 call
            snap("e", model.forward_with_guidance(x56, exif, "top-right"))
    finally:
        ref.nn.Linear = real_linear
    assert "failed" not in log.getvalue(), log.getvalue()
    assert len(created) == 2
    for tag, lin in zip("ce", created):
        arrays[tag + ":proj_w"], arrays[tag + ":proj_b"] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
    save("g8_curiosity_state", **arrays)


def g9_single_patch(ref):
    """14 x 14 inputs: ONE patch, the only grid on which the degenerate-attention fallbacks of src/model.py:242-257
    fire naturally (column mean + centre bias is the constant 1.3 for every image, so its whole-batch variance is
    0 for B >= 2; all three fallbacks then run, the last one drawing noise).  With one patch every branch
    normalises to the same attention vector [1.0], so outputs do not depend on the branch taken."""
    from transformers import Dinov2Config, Dinov2Model
    layers, grid = 2, 4
    ref.Dinov2Model.from_pretrained = classmethod(
        lambda cls, name, *a, **k: Dinov2Model(Dinov2Config(image_size=14 * grid, num_hidden_layers=layers)))
    cfg = {"cognitive_modules": ["ambient_stream", "iterative_focal_stream", "exif_prior_database"],
           "focal_config": {"num_iterations": 3, "focus_strength": 1.5}}
    with quiet():
        model = ref.create_model(cfg, {"num_cameras": 71}).eval()
    spec = reference_state_spec(layers=layers, native_grid=grid, num_iterations=3, num_cameras=71)
    model.load_state_dict(synth_state_dict(spec, seed=7), strict=True)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 3, 14, 14, generator=g)
    exif = {"camera_idx": torch.tensor([0, 5, 70]), "focal_length": torch.tensor([50.0, 24.0, 135.0]),
            "aperture": torch.tensor([2.8, 8.0, 1.8]), "iso": torch.tensor([100.0, 1600.0, 400.0])}
    arrays = {"layers": layers, "native_grid": grid, "seed": 7, "x": x.numpy()}
    for k, v in exif.items():
        arrays["exif:" + k] = v.numpy()
    drawn, real = [], torch.randn_like

    def spy(t, *a_, **k_):
        drawn.append(tuple(t.shape))
        return real(t, *a_, **k_)
    torch.randn_like = spy
    try:
        with torch.no_grad(), quiet():
            d, c, a = model(x, exif, return_attention=True)
            _, _, ga = model.forward_with_guidance(x, exif, "center", return_attention=True)
    finally:
        torch.randn_like = real
    assert drawn.count((3, 1)) >= 3          # the third fallback (:252-257) drew its noise in every FocalStream call
    arrays["fallback_noise_draws"] = drawn.count((3, 1))
    arrays["depth"], arrays["conf"], arrays["attn"], arrays["guided"] = d.numpy(), c.numpy(), a.numpy(), ga.numpy()
    save("g9_single_patch", **arrays)


def g10_bare_focal_stream(ref):
    """cognitive_modules with 'focal_stream' instead of 'iterative_focal_stream' (src/model.py:869-873): one
    FocalStream, curiosity modulator present (constructor default), its own projection on the guided path
    (:1417-1418), so guided depth is deterministic in this variant."""
    from transformers import Dinov2Config, Dinov2Model
    layers, grid = 2, 4
    ref.Dinov2Model.from_pretrained = classmethod(
        lambda cls, name, *a, **k: Dinov2Model(Dinov2Config(image_size=14 * grid, num_hidden_layers=layers)))
    mods = ["ambient_stream", "focal_stream", "exif_prior_database"]
    with quiet():
        model = ref.create_model({"cognitive_modules": mods}, {"num_cameras": 71}).eval()
    spec = reference_state_spec(layers=layers, native_grid=grid, num_cameras=71, modules=tuple(mods))
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(spec.keys()), (set(ref_sd) ^ set(spec))
    for k, (shp, dt) in spec.items():
        assert tuple(ref_sd[k].shape) == tuple(shp) and ref_sd[k].dtype == dt, k
    model.load_state_dict(synth_state_dict(spec, seed=7), strict=True)
    g = torch.Generator().manual_seed(55)
    x = torch.randn(2, 3, 70, 70, generator=g)
    exif = {"camera_idx": torch.tensor([3, 44]), "focal_length": torch.tensor([35.0, 200.0]),
            "aperture": torch.tensor([4.0, 11.0]), "iso": torch.tensor([200.0, 3200.0])}
    arrays = {"layers": layers, "native_grid": grid, "seed": 7, "x": x.numpy()}
    for k, v in exif.items():
        arrays["exif:" + k] = v.numpy()
    with torch.no_grad(), quiet() as log:
        d, c, a = model(x, exif, return_attention=True)
        gd, gc, ga = model.forward_with_guidance(x, exif, "bottom-left", return_attention=True)
    assert "failed" not in log.getvalue(), log.getvalue()
    for k, v in (("depth", d), ("conf", c), ("attn", a), ("guided_depth", gd), ("guided_conf", gc), ("guided_attn", ga)):
        arrays[k] = v.numpy()
    save("g10_bare_focal_stream", **arrays)


def g11_large_backbone(ref):
    """backbone_size='large' (src/model.py:807-809): 1024-wide tokens, 16 heads, every stream fed 1024 columns, the
    guided path's per-call Linear(1024, 64) (:1421).  Two layers here, like the other full-width fixtures."""
    from transformers import Dinov2Config, Dinov2Model
    layers, grid = 2, 4
    asked = []

    def local(cls, name, *a, **k):
        asked.append(name)
        return Dinov2Model(Dinov2Config(hidden_size=1024, num_attention_heads=16, image_size=14 * grid,
                                        num_hidden_layers=layers))
    ref.Dinov2Model.from_pretrained = classmethod(local)
    cfg = {"backbone_size": "large",
           "cognitive_modules": ["ambient_stream", "iterative_focal_stream", "exif_prior_database"],
           "focal_config": {"num_iterations": 3, "focus_strength": 1.5}}
    with quiet():
        model = ref.create_model(cfg, {"num_cameras": 71}).eval()
    assert asked == ["facebook/dinov2-large"] and model.feature_dim == 1024
    spec = reference_state_spec(hidden=1024, layers=layers, native_grid=grid, num_iterations=3, num_cameras=71)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(spec.keys()), (set(ref_sd) ^ set(spec))
    for k, (shp, dt) in spec.items():
        assert tuple(ref_sd[k].shape) == tuple(shp) and ref_sd[k].dtype == dt, k
    model.load_state_dict(synth_state_dict(spec, seed=11), strict=True)
    g = torch.Generator().manual_seed(111)
    exif = {"camera_idx": torch.tensor([7, 60]), "focal_length": torch.tensor([28.0, 85.0]),
            "aperture": torch.tensor([5.6, 1.4]), "iso": torch.tensor([800.0, 100.0])}
    arrays = {"layers": layers, "native_grid": grid, "seed": 11, "hidden": 1024, "heads": 16}
    for k, v in exif.items():
        arrays["exif:" + k] = v.numpy()
    created = []
    real_linear = ref.nn.Linear

    class Spy(real_linear):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            created.append(self)

    for size in (56, 98):
        x = torch.randn(2, 3, size, size, generator=g)
        arrays[f"x{size}"] = x.numpy()
        with torch.no_grad(), quiet() as log:
            d, c, a = model(x, exif, return_attention=True)
        arrays[f"curiosity_failures{size}"] = log.getvalue().count("calculation failed")
        arrays[f"depth{size}"], arrays[f"conf{size}"], arrays[f"attn{size}"] = d.numpy(), c.numpy(), a.numpy()
    arrays["history_untouched"] = bool((model.curiosity_module.exploration_history == 0).all()
                                       and int(model.curiosity_module.history_pointer) == 0)
    arrays["torch_seed"] = 999
    torch.manual_seed(999)
    ref.nn.Linear = Spy
    try:
        with torch.no_grad(), quiet() as log:
            gd, gc, ga = model.forward_with_guidance(x, exif, "top-left", return_attention=True)
    finally:
        ref.nn.Linear = real_linear
    # the curiosity module is 768 wide whatever the backbone (src/model.py:946-950): fed the 1024-wide CLS token
    # it raises, the reference catches that and carries on with its default score (which reaches no output)
    lines = [ln for ln in log.getvalue().splitlines() if "failed" in ln]
    assert all(ln.startswith("Preliminary curiosity calculation failed") for ln in lines), lines
    arrays["guided_log_failed_lines"] = len(lines)
    assert len(created) == 1 and tuple(created[0].weight.shape) == (64, 1024)
    arrays["guided_proj_w"], arrays["guided_proj_b"] = created[0].weight.detach().numpy(), created[0].bias.detach().numpy()
    arrays["guided_depth"], arrays["guided_conf"], arrays["guided_attn"] = gd.numpy(), gc.numpy(), ga.numpy()
    save("g11_large_backbone", **arrays)


def g12_degenerate_attention(ref):
    """Inputs on which the REAL FocalStream takes its degenerate-attention fallbacks (src/model.py:242-249) on grids of
    more than one patch: the tokens are chosen so that (column mean of the attention + centre bias) is the same number
    for every patch of every image, i.e. its whole-batch variance is 0.  Only grids of 2, 3 and 2 x 2 patches admit
    that (the constant must reach the bias at the centre, 0.3).  Case 'rows_equal': every attention row is the same,
    so the diagonal (first fallback) is degenerate too and the row maximum (second fallback) decides; 'rows_differ':
    rows differ, the diagonal decides.  The third fallback (:252-257) draws noise and is not pinned."""
    arrays = {}
    torch.manual_seed(12)
    taken = []
    real_var = torch.Tensor.var

    def spy_var(self, *a, **k):
        v = real_var(self, *a, **k)
        taken.append(float(v))
        return v

    for tag, n, differ in (("rows_equal4", 4, False), ("rows_differ4", 4, True), ("rows_differ3", 3, True),
                           ("rows_equal2", 2, False)):
        fs = ref.FocalStream(64, curiosity_guided=False).eval()
        B = 2
        captured = {}
        hook = fs.query_proj.register_forward_pre_hook(
            lambda mod, inp: captured.__setitem__("xt", inp[0].detach().clone()))
        with torch.no_grad(), quiet():
            fs(torch.zeros(1, n, 64))
        pe = captured["xt"][0].double()                                   # the position table the stream adds
        # centre bias exactly as the stream builds it: run it on uniform attention and subtract
        with torch.no_grad():
            fs.query_proj.weight.zero_(); fs.query_proj.bias.zero_()
            fs.key_proj.weight.zero_(); fs.key_proj.bias.zero_()
            with quiet():
                _, a_uniform = fs(torch.zeros(1, n, 64))
        # with zero scores P = 1/n; a = renorm(1/n + bias) -> bias up to the normalisation: recover it from ratios
        g = int(n ** 0.5)
        if g * g == n:
            yy, xx = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
            d = torch.sqrt((xx - g // 2).float() ** 2 + (yy - g // 2).float() ** 2)
            bias = (torch.exp(-d ** 2 / (2 * (g / 6) ** 2)).flatten() * 0.3).double()
        else:
            d = torch.abs(torch.arange(n, dtype=torch.float) - n // 2)
            bias = (torch.exp(-d ** 2 / (2 * (n / 12) ** 2)) * 0.3).double()
        chk = (1.0 / n + bias) / (1.0 / n + bias).sum()
        assert torch.allclose(a_uniform[0].double(), chk, atol=1e-6), (a_uniform, chk)
        C = (1.0 + bias.sum()) / n
        t = C - bias                                                      # the column means that make the sum constant
        assert (t > 0).all(), t
        s_gain = 4.0
        with torch.no_grad():
            fs.query_proj.bias[0] = s_gain                                # q_n = [s, xt_n[1], 0, ...]
            fs.query_proj.weight[1, 1] = 1.0
            fs.key_proj.weight[0, 0] = 1.0                                # k_c = [xt_c[0], xt_c[2], 0, ...]
            fs.key_proj.weight[1, 2] = 1.0
        xt = torch.randn(B, n, 64).double() * 0.5
        if not differ:
            xt[:, :, 1] = 0.0
        scale = (64 // 8) ** 0.5
        f = torch.log(t).repeat(B, 1)                                     # per-column offsets, solved per image
        for _ in range(200):
            S = (s_gain * f.unsqueeze(1) * scale / s_gain + xt[:, :, 1].unsqueeze(2) * xt[:, :, 2].unsqueeze(1)) / scale
            colmean = torch.softmax(S, dim=-1).mean(dim=1)
            f = f + torch.log(t.unsqueeze(0) / colmean)
        assert float((colmean - t).abs().max()) < 1e-12
        xt[:, :, 0] = f * scale / s_gain
        x = (xt - pe.unsqueeze(0)).float()
        taken.clear()
        torch.Tensor.var = spy_var
        try:
            with torch.no_grad(), quiet():
                feat, att = fs(x)
        finally:
            torch.Tensor.var = real_var
        small = [v < 1e-6 for v in taken]
        branch = 0
        for v in small:
            if not v:
                break
            branch += 1
        assert branch in (1, 2), (tag, taken)                             # 1: diagonal decided, 2: row maximum decided
        assert branch == (1 if differ else 2), (tag, taken)
        arrays.update(npd(fs.state_dict(), f"{tag}:w:"))
        arrays[f"{tag}:x"], arrays[f"{tag}:f"], arrays[f"{tag}:a"] = x.numpy(), feat.numpy(), att.numpy()
        arrays[f"{tag}:branch"] = branch
        arrays[f"{tag}:variances"] = np.array(taken[:3] + [0.0] * (3 - len(taken[:3])))
    save("g12_degenerate_attention", **arrays)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    g1_tiny_backbone(ref)
    g2_focal(ref)
    g3_iterative(ref)
    g5_guidance(ref)
    g6_pe_table(ref)
    g7_full_model(ref)
    g8_curiosity_state(ref)
    g9_single_patch(ref)
    g10_bare_focal_stream(ref)
    g11_large_backbone(ref)
    g12_degenerate_attention(ref)


if __name__ == "__main__":
    main()
