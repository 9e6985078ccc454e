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
