"""Device preprocessing (cogaim_preprocess_rgb8 through the C ABI) against the host path of the reference
(Pillow bilinear resize -> / 255 -> ImageNet normalise, demo.py:162-166): bit-exact, integer resampling and a
table of the host's own float results."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cognitive_aim_depth_estimation_amd.inference import DevicePreprocessor, preprocess_image  # noqa: E402
from oracle import preprocess_oracle as PO  # noqa: E402

PIL = pytest.importorskip("PIL.Image")

CASES = [((480, 640), (224, 224)),        # ordinary shrink
         ((3024, 4032), (518, 518)),      # 12-megapixel photo, the headline input size
         ((37, 53), (28, 28)),            # odd small sizes
         ((20, 31), (56, 42)),            # enlargement, non-square target
         ((518, 518), (518, 518)),        # identity
         ((1, 9), (14, 14)),              # one-row source
         ((700, 13), (14, 28))]           # strong shrink on one axis, enlargement on the other


@pytest.mark.parametrize("src,dst", CASES)
def test_device_preprocess_is_bit_exact(src, dst):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    rng = np.random.default_rng(src[0] * 7919 + src[1])
    arr = rng.integers(0, 256, size=(src[0], src[1], 3), dtype=np.uint8)
    image = PIL.fromarray(arr, "RGB")
    want = preprocess_image(image, dst)                       # host path (Pillow + torch fp32)
    pre = DevicePreprocessor(dst, "cuda")
    got = pre(image)
    torch.cuda.synchronize()
    assert got.shape == want.shape and got.dtype == torch.float32
    assert torch.equal(got.cpu(), want)
    if src[0] * src[1] <= 4096:                               # the loop oracle, where it finishes in seconds
        assert np.array_equal(PO.preprocess(arr, dst[0], dst[1]), got.cpu().numpy())


def test_device_preprocess_into_batch_slot_and_cache():
    rng = np.random.default_rng(5)
    pre = DevicePreprocessor((56, 56), "cuda")
    batch = torch.zeros(3, 3, 56, 56, device="cuda")
    images = [PIL.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), "RGB")
              for h, w in ((90, 120), (90, 120), (64, 200))]
    for i, im in enumerate(images):
        pre(im, out=batch[i])
    torch.cuda.synchronize()
    for i, im in enumerate(images):
        assert torch.equal(batch[i].cpu(), preprocess_image(im, (56, 56)))
    assert len(pre._axis) == 4          # (120->56), (90->56), (200->56), (64->56): tables are cached per axis length
    with pytest.raises(ValueError):
        pre(images[0], out=torch.zeros(3, 56, 57, device="cuda"))


def test_grayscale_and_rgba_inputs_follow_the_host_conversion():
    rng = np.random.default_rng(6)
    pre = DevicePreprocessor((28, 28), "cuda")
    gray = PIL.fromarray(rng.integers(0, 256, size=(40, 50), dtype=np.uint8), "L")
    rgba = PIL.fromarray(rng.integers(0, 256, size=(40, 50, 4), dtype=np.uint8), "RGBA")
    for im in (gray, rgba):
        assert torch.equal(pre(im).cpu(), preprocess_image(im, (28, 28)))


def test_batch_of_rgbx_views_equals_packed_rgb(tmp_path):
    """The batch loop hands the kernel Pillow's own RGBX memory (4 bytes a pixel, zero-copy view from
    CognitiveAimInference._decode) where it can and packed RGB copies where it cannot: same pixels, same bits."""
    from cognitive_aim_depth_estimation_amd.inference import CognitiveAimInference
    rng = np.random.default_rng(8)
    pre = DevicePreprocessor((56, 70), "cuda")
    sizes = ((90, 120), (33, 47), (518, 518), (64, 200))
    rgb = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    rgbx = [np.concatenate([a, rng.integers(0, 256, size=a.shape[:2] + (1,), dtype=np.uint8)], axis=2) for a in rgb]
    mixed = [rgbx[0], rgb[1], rgbx[2], rgbx[3]]
    a = pre.batch(rgb, torch.zeros(4, 3, 56, 70, device="cuda")).clone()
    b = pre.batch(mixed, torch.zeros(4, 3, 56, 70, device="cuda")).clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    for i, arr in enumerate(rgb):
        assert torch.equal(a[i].cpu(), preprocess_image(PIL.fromarray(arr, "RGB"), (56, 70)))
    # _decode: a view of the decoder's memory for an RGB file, a packed copy for a converted palette file; same pixels
    p_rgb, p_pal = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    PIL.fromarray(rgb[0], "RGB").save(p_rgb)
    PIL.fromarray(rgb[0], "RGB").convert("P").save(p_pal)
    arr, exif, size = CognitiveAimInference._decode(None, p_rgb)
    assert size == (120, 90) and exif is None and np.array_equal(arr[:, :, :3], rgb[0])
    import importlib.util
    if importlib.util.find_spec("pyarrow") is not None and hasattr(PIL.Image, "__arrow_c_array__"):
        assert arr.shape == (90, 120, 4) and not arr.flags.owndata
    arr2, _, _ = CognitiveAimInference._decode(None, p_pal)
    assert np.array_equal(arr2[:, :, :3], np.asarray(PIL.open(p_pal).convert("RGB")))


def test_batched_launch_equals_per_image_launches():
    """cogaim_preprocess_rgb8_batch (two launches for the whole chunk, records in device memory) against the
    per-image entry point: bit-identical, sizes and pixel strides mixed."""
    rng = np.random.default_rng(9)
    pre = DevicePreprocessor((70, 56), "cuda")
    arrays = [rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
              for h, w, c in ((90, 120, 3), (33, 47, 4), (518, 518, 4), (64, 200, 3), (7, 9, 3))]
    batch = pre.batch(arrays, torch.zeros(5, 3, 70, 56, device="cuda")).clone()
    torch.cuda.synchronize()
    for i, a in enumerate(arrays):
        single = pre(np.ascontiguousarray(a[:, :, :3]))
        assert torch.equal(batch[i], single), i
