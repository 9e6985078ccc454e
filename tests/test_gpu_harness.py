"""End to end through the demo harness on the GPU (reference demo.py:26-600 surface): image files ->
device preprocessing -> batched forward -> per-image results, against the CPU oracle fed with the host-side
preprocessing of the same files."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from cognitive_aim_depth_estimation_amd import inference as inf  # noqa: E402
from oracle import cogaim_oracle as O  # noqa: E402
from oracle.weights import reference_state_spec, synth_state_dict  # noqa: E402
from tests.util import rel_err  # noqa: E402

PIL = pytest.importorskip("PIL.Image")
torch.set_grad_enabled(False)

LAYERS, GRID, SIZE = 2, 4, 56


@pytest.fixture()
def workdir(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)                      # rendered PNGs go to ./demo_results
    rng = np.random.default_rng(3)
    paths = []
    for i, (h, w, ext) in enumerate([(120, 160, "png"), (64, 48, "png"), (200, 200, "jpg"), (56, 56, "png"), (90, 31, "png")]):
        p = tmp_path / f"img{i}.{ext}"
        PIL.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), "RGB").save(p)
        paths.append(str(p))
    cfg = {"cognitive_modules": ["ambient_stream", "iterative_focal_stream", "exif_prior_database"],
           "dataset": {"image_size": SIZE}, "compute_dtype": "fp16",
           "backbone_config": {"num_hidden_layers": LAYERS, "image_size": 14 * GRID}}
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    sd = synth_state_dict(reference_state_spec(layers=LAYERS, native_grid=GRID, num_iterations=3, num_cameras=71), seed=11)
    ckpt = tmp_path / "ckpt.pth"
    torch.save({"model_state_dict": sd, "epoch": 3}, ckpt)
    return paths, str(cfg_path), str(ckpt), sd


def _oracle(sd, paths):
    x = torch.stack([inf.preprocess_image(PIL.open(p).convert("RGB"), (SIZE, SIZE)) for p in paths])
    exif = inf.exif_rows([None] * len(paths), {"unknown": 0})
    return O.forward(sd, x, exif, return_attention=True)


def test_predict_batch_matches_oracle_and_tolerates_bad_files(workdir):
    paths, cfg, ckpt, sd = workdir
    h = inf.CognitiveAimInference(cfg, ckpt, device="cuda", batch_size=2, render=False)     # 5 images -> batches 2,2,1
    res = h.predict_batch(paths + ["missing.png"])
    assert len(res) == 6 and res[5][0] is None and "error" in res[5][2]
    d_ref, c_ref, a_ref = _oracle(sd, paths)
    d = torch.tensor([[r[0]] for r in res[:5]])
    c = torch.tensor([[r[1]] for r in res[:5]])
    assert rel_err(d, d_ref) <= 1e-3 and rel_err(c, c_ref) <= 1e-3
    meta = res[0][2]
    assert meta["original_size"] == (160, 120) and meta["processed_size"] == (SIZE, SIZE)
    assert meta["exif_source"] == "default" and meta["instruction"] is None
    # single-image entry point, same contract as the reference's predict()
    d0, c0, m0 = h.predict(paths[0])
    assert abs(d0 - res[0][0]) <= 1e-6 * abs(d0) and m0["image_path"] == paths[0]
    with pytest.raises(ValueError):
        h.predict("missing.png")


def test_guided_batch_and_rendering(workdir):
    paths, cfg, ckpt, sd = workdir
    h = inf.CognitiveAimInference(cfg, ckpt, device="cuda", batch_size=4, render=True)
    res = h.predict_batch(paths[:3], ["center", None, "top-left"])
    assert all(r[0] is not None and np.isfinite(r[0]) and 0.0 < r[1] < 1.0 for r in res)
    assert [r[2]["instruction"] for r in res] == ["center", None, "top-left"]
    # the unguided image of a mixed batch equals its plain prediction
    d_ref, c_ref, _ = _oracle(sd, paths[1:2])
    assert rel_err(torch.tensor([[res[1][0]]]), d_ref) <= 1e-3
    try:
        import matplotlib  # noqa: F401
    except Exception:
        return
    out = sorted(os.listdir("demo_results"))
    assert out == ["img0_center_prediction.png", "img1_prediction.png", "img2_top-left_prediction.png"]
    assert all(os.path.getsize(os.path.join("demo_results", f)) > 10_000 for f in out)


def test_instruction_sweep_through_the_harness(workdir, monkeypatch):
    """BASELINE config 3 through predict_batch: one file asked for with the nine instructions is decoded once and
    goes through ONE guided forward (one backbone + focal pass); the heat maps of the sweep equal those of nine
    single calls bit for bit, and files with real EXIF feed their own values to the model."""
    from PIL import Image
    from PIL.TiffImagePlugin import IFDRational
    paths, cfg, ckpt, sd = workdir
    h = inf.CognitiveAimInference(cfg, ckpt, device="cuda", batch_size=8, render=False)
    calls = []
    real = h.model.forward_with_guidance

    def spy(images, exif_data=None, attention_guidance=None, *a, **k):
        calls.append((tuple(images.shape), attention_guidance))
        return real(images, exif_data, attention_guidance, *a, **k)
    monkeypatch.setattr(h.model, "forward_with_guidance", spy)
    names = list(O.INSTRUCTIONS)
    res = h.predict_batch([paths[0]] * 9 + [paths[1]], names + [None])
    assert len(calls) == 1 and calls[0][0][0] == 1 and list(calls[0][1]) == names
    assert [r[2]["instruction"] for r in res] == names + [None]
    assert all(np.isfinite(r[0]) and 0 < r[1] < 1 for r in res)
    x = h.preprocess(PIL.open(paths[0]))[None]
    exif = {k: v.cuda() for k, v in inf.exif_rows([None], h.camera_to_id).items()}
    _, _, sweep = real(x, exif, names, return_attention=True)
    for k, name in enumerate(names):
        _, _, one = real(x, exif, name, return_attention=True)
        assert torch.equal(one, sweep[k]), name
    # EXIF read from the file reaches the forward: same prediction as the oracle with those values
    ex = Image.Exif()
    ex[0x010F], ex[0x0110] = "Canon", "EOS R5"
    ifd = ex.get_ifd(0x8769)
    ifd[0x920A], ifd[0x829D], ifd[0x8827] = IFDRational(85, 1), IFDRational(18, 10), 800
    p = os.path.join(os.path.dirname(paths[0]), "with_exif.jpg")
    PIL.open(paths[2]).save(p, exif=ex, quality=95)
    h.camera_to_id["EOS R5"] = 12
    d, c, meta = h.predict(p)
    assert meta["exif_source"] == "real" and meta["exif_data"]["FocalLength"] == 85.0
    xo = inf.preprocess_image(PIL.open(p).convert("RGB"), (SIZE, SIZE))[None]
    rows = inf.exif_rows([inf.extract_exif(p)], h.camera_to_id)
    assert rows["camera_idx"].tolist() == [12]
    d_ref, c_ref = O.forward(sd, xo, rows)
    assert rel_err(torch.tensor([[d]]), d_ref) <= 1e-3 and rel_err(torch.tensor([[c]]), c_ref) <= 1e-3
    d_def, _, _ = h.predict(paths[2])
    assert abs(d_def - d) > 1e-6                               # the EXIF values matter


def test_mixed_image_modes_and_a_multi_block_image(workdir, tmp_path):
    """Files that are not plain small RGB: grayscale, RGBA, palette, CMYK (all converted like the reference's
    Image.open(...).convert('RGB'), demo.py:312) and an image too large for Pillow's single-block storage, where the
    zero-copy hand-over is not available and the packed copy is used.  Results against the oracle on the host path."""
    _, cfg, ckpt, sd = workdir
    rng = np.random.default_rng(21)
    files = []
    for name, mode, shape in (("gray.png", "L", (80, 100)), ("rgba.png", "RGBA", (70, 90, 4)), ("cmyk.jpg", "CMYK", (60, 60, 4)),
                              ("plain.png", "RGB", (50, 75, 3))):
        p = str(tmp_path / name)
        PIL.fromarray(rng.integers(0, 256, size=shape, dtype=np.uint8), mode).save(p)
        files.append(p)
    pal = str(tmp_path / "pal.png")
    PIL.fromarray(rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8), "RGB").convert("P").save(pal)
    big = str(tmp_path / "big.png")                                        # 2400 x 2400 x 4 bytes = 23 MB: several blocks
    tile = rng.integers(0, 256, size=(40, 40, 3), dtype=np.uint8)
    PIL.fromarray(np.tile(tile, (60, 60, 1)), "RGB").save(big, compress_level=1)
    files += [pal, big]
    arr, _, size = inf.CognitiveAimInference._decode(None, big)
    assert size == (2400, 2400) and arr.shape[:2] == (2400, 2400)
    h = inf.CognitiveAimInference(cfg, ckpt, device="cuda", batch_size=4, render=False)
    res = h.predict_batch(files)
    assert all(r[0] is not None for r in res), [r[2] for r in res if r[0] is None]
    d_ref, c_ref, _ = _oracle(sd, files)
    d = torch.tensor([[r[0]] for r in res])
    c = torch.tensor([[r[1]] for r in res])
    assert rel_err(d, d_ref) <= 1e-3 and rel_err(c, c_ref) <= 1e-3, (rel_err(d, d_ref), rel_err(c, c_ref))


@pytest.mark.parametrize("render", [False, True])
def test_many_chunks_reuse_the_input_slots(workdir, tmp_path, monkeypatch, render):
    """Eleven distinct files in chunks of two: six chunks over four input slots, so every slot is handed back and
    refilled while later forwards are still queued.  Each file's result must be the one a single-image prediction
    gives (a slot overwritten before its forward had read it would show up as another file's numbers), and with
    rendering on, the pixels handed to the PNG writer must be that file's own preprocessed pixels."""
    _, cfg, ckpt, sd = workdir
    rng = np.random.default_rng(5)
    files = []
    for i in range(11):
        p = str(tmp_path / f"many{i:02d}.png")
        PIL.fromarray(rng.integers(0, 256, size=(60 + 7 * i, 90 - 3 * i, 3), dtype=np.uint8), "RGB").save(p)
        files.append(p)
    h = inf.CognitiveAimInference(cfg, ckpt, device="cuda", batch_size=2, render=render)
    drawn = {}

    def record(image_path, image_tensor, depth, conf, grid, instruction):
        drawn[image_path] = image_tensor.clone()
    monkeypatch.setattr(h, "_save_prediction_image", record)
    res = h.predict_batch(files)
    assert all(r[0] is not None for r in res)
    d_ref, c_ref, _ = _oracle(sd, files)
    d = torch.tensor([[r[0]] for r in res])
    c = torch.tensor([[r[1]] for r in res])
    assert rel_err(d, d_ref) <= 1e-3 and rel_err(c, c_ref) <= 1e-3, (rel_err(d, d_ref), rel_err(c, c_ref))
    assert len(set(round(r[0], 6) for r in res)) == len(files)          # distinct inputs, distinct outputs
    single = inf.CognitiveAimInference(cfg, ckpt, device="cuda", batch_size=1, render=False)
    for p, r in zip(files, res):
        d1, c1, _ = single.predict(p)
        assert abs(d1 - r[0]) <= 2e-5 * abs(d1) and abs(c1 - r[1]) <= 2e-5, p
    if render:
        assert sorted(drawn) == sorted(files)
        for p in files:
            want = inf.preprocess_image(PIL.open(p).convert("RGB"), (SIZE, SIZE))
            assert torch.equal(drawn[p].cpu(), want), p


def test_consumer_failure_ends_the_producer_thread(workdir, monkeypatch):
    """A forward that raises must not leave the upload thread parked on the hand-over queue (it would keep the
    input slots, the pinned staging buffers and both thread pools alive)."""
    import threading
    paths, cfg, ckpt, sd = workdir
    h = inf.CognitiveAimInference(cfg, ckpt, device="cuda", batch_size=1, render=False)

    def boom(*a, **k):
        raise RuntimeError("forward failed")
    monkeypatch.setattr(h.model, "forward_async", boom)
    with pytest.raises(RuntimeError, match="forward failed"):
        h.predict_batch(paths * 3)
    assert not [t for t in threading.enumerate() if t.name == "cogaim-producer" and t.is_alive()]


def test_cli_end_to_end_with_a_nested_reference_style_config(tmp_path, monkeypatch, capsys):
    """``demo.py --image F --instruction center`` and ``--image_dir D`` through inference.main (reference
    demo.py:603-693) with a YAML laid out like the reference's own (everything under ``model:``, image_size 224):
    the nested values are ignored exactly as the reference ignores them, so the model is dinov2-base with three
    focal iterations; a checkpoint path that does not exist keeps the seeded initial weights (demo.py:148-150).
    The printed depth must be what predict() returns for the same file under the same torch seed (the guided path
    draws its Linear(768, 64) per call, src/model.py:1417-1422)."""
    import re
    monkeypatch.chdir(tmp_path)
    cfg = {"experiment": {"name": "full_cognition_enhanced", "version": "v1.0"},
           "model": {"backbone": "dinov2", "backbone_size": "base",
                     "cognitive_modules": ["ambient_stream", "iterative_focal_stream", "exif_prior_database"],
                     "curiosity_guided_attention": {"enable": True, "attention_heads": 8},
                     "enable_focal_stream": False,
                     "exif_config": {"num_cameras": 71, "exif_embedding_dim": 64},
                     "focal_config": {"num_iterations": 2, "focus_strength": 0.1}},
           "dataset": {"path": "dataset", "image_size": 224, "augmentation": True},
           "training": {"batch_size": 8, "epochs": 20}}
    cfg_path = tmp_path / "nested.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    rng = np.random.default_rng(9)
    img_dir = tmp_path / "imgs"
    img_dir.mkdir()
    files = []
    for i in range(3):
        p = img_dir / f"photo{i}.jpg"
        PIL.fromarray(rng.integers(0, 256, size=(300 + 20 * i, 400, 3), dtype=np.uint8), "RGB").save(p, quality=92)
        files.append(str(p))
    argv = ["--config", str(cfg_path), "--checkpoint", str(tmp_path / "absent.pth")]
    torch.manual_seed(123)
    inf.main(argv + ["--image", files[0], "--instruction", "center"])
    out = capsys.readouterr().out
    assert "Continuing with randomly initialized weights" in out and "Inference completed!" in out
    depth_cli = float(re.search(r"Depth value: ([0-9.eE+-]+)", out).group(1))
    conf_cli = float(re.search(r"Confidence: ([0-9.eE+-]+)", out).group(1))
    assert "Guidance instruction: center" in out
    assert os.path.exists(os.path.join("demo_results", "photo0_center_prediction.png")) or "skipped" in out
    engine = inf.CognitiveAimInference(str(cfg_path), str(tmp_path / "absent.pth"), "cuda", 32, render=False)
    m = engine.model
    assert (m.num_layers, m.feature_dim, m.num_iterations, m.focus_strength) == (12, 768, 3, 1.5)   # nested values unread
    assert engine.image_size == (224, 224)
    torch.manual_seed(123)
    d, c, meta = engine.predict(files[0], "center")
    assert abs(d - depth_cli) <= 5e-5 * max(1.0, abs(d)) + 5e-5 and abs(c - conf_cli) <= 1e-4
    assert meta["processed_size"] == (224, 224) and meta["instruction"] == "center"
    # directory mode: every file through one batched forward
    capsys.readouterr()
    inf.main(argv + ["--image_dir", str(img_dir), "--no_render"])
    out = capsys.readouterr().out
    assert "Found 3 images" in out and "Successfully processed: 3" in out and "Failed: 0" in out
    rows = engine.predict_batch(files)
    avg = float(np.mean([r[0] for r in rows]))
    assert abs(float(re.search(r"Average depth: ([0-9.]+)", out).group(1)) - avg) <= 1e-3 * avg + 1e-4
    # the reference's argument checks
    with pytest.raises(ValueError):
        inf.main(argv)
    with pytest.raises(ValueError):
        inf.main(argv + ["--image", files[0], "--image_dir", str(img_dir)])
