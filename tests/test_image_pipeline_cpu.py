"""oracle/image.py (numpy restatement of Pillow's bicubic resample + the HF processor arithmetic) pinned against the
installed Pillow / transformers themselves, on the exact call sequence of the reference's dataset (train_halva.py:735-751,
vila/mm_utils.py:150-193).  Bit-exact: uint8 after the resize, float32 after rescale + normalize."""
import numpy as np
import pytest
from PIL import Image

from oracle import image as OI

CLIP_MEAN, CLIP_STD = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
SHAPES = [(480, 640), (640, 480), (336, 336), (100, 37), (37, 100), (700, 1333), (336, 500), (51, 51)]


def _img(h, w, seed):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (h // 7 + 2, w // 7 + 2, 3)).astype(np.uint8)          # blocky content + noise: real gradients
    big = np.kron(base, np.ones((7, 7, 1), dtype=np.uint8))[:h, :w]
    return np.clip(big.astype(np.int32) + rng.randint(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w", SHAPES)
@pytest.mark.parametrize("out", [(336, 336), (384, 384), (224, 301)])
def test_bicubic_resize_bit_exact_vs_pillow(h, w, out):
    img = _img(h, w, h * 1000 + w)
    want = np.asarray(Image.fromarray(img).resize((out[1], out[0]), resample=Image.BICUBIC))
    got = OI.resize_bicubic_u8(img, out[1], out[0])
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("h,w", SHAPES)
def test_clip_pad_pipeline_bit_exact_vs_transformers(h, w):
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    img = _img(h, w, 7 * h + w)
    pil = Image.fromarray(img)
    bg = tuple(int(x * 255) for x in proc.image_mean)
    side = max(pil.size)
    canvas = Image.new(pil.mode, (side, side), bg)                       # the reference's expand2square
    canvas.paste(pil, ((side - pil.size[0]) // 2, (side - pil.size[1]) // 2))
    want = proc.preprocess(canvas, return_tensors="np")["pixel_values"][0]
    got = OI.clip_preprocess(img, 336, proc.image_mean, proc.image_std, pad=True)
    assert got.dtype == np.float32 and got.shape == (3, 336, 336)
    np.testing.assert_array_equal(got, want)
    lut = OI.normalize_lut(proc.image_mean, proc.image_std)
    sq = OI.resize_bicubic_u8(OI.expand2square(img, bg), 336, 336)
    np.testing.assert_array_equal(np.stack([lut[c][sq[..., c]] for c in range(3)]), want)
    # without padding: shortest-edge resize + centre crop
    want2 = proc.preprocess(pil, return_tensors="np")["pixel_values"][0]
    np.testing.assert_array_equal(OI.clip_preprocess(img, 336, proc.image_mean, proc.image_std, pad=False), want2)


@pytest.mark.parametrize("h,w", SHAPES[:5])
def test_siglip_resize_pipeline_bit_exact_vs_transformers(h, w):
    from transformers import SiglipImageProcessor
    proc = SiglipImageProcessor(size={"height": 384, "width": 384})
    img = _img(h, w, 3 * h + w)
    pil = Image.fromarray(img).resize((384, 384))                        # vila/mm_utils.py:168 (PIL default filter: BICUBIC)
    want = proc.preprocess(pil, return_tensors="np")["pixel_values"][0]
    got = OI.siglip_preprocess(img, 384, proc.image_mean, proc.image_std)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("n_in,n_out", [(640, 336), (480, 336), (336, 336), (100, 336), (37, 384), (1333, 336), (2000, 384), (336, 301)])
def test_product_weight_tables_equal_oracle(n_in, n_out):
    """The vectorised host tables of halva_amd/image_pipeline.py are the oracle's (Pillow's) loop, integer for integer."""
    from halva_amd.image_pipeline import coeff_tables, normalize_lut
    ks, bd, cf = coeff_tables(n_in, n_out)
    ks2, bd2, cf2 = OI.precompute_coeffs(n_in, n_out)
    assert ks == ks2
    np.testing.assert_array_equal(bd, bd2)
    np.testing.assert_array_equal(cf, cf2)
    np.testing.assert_array_equal(normalize_lut(CLIP_MEAN, CLIP_STD), OI.normalize_lut(CLIP_MEAN, CLIP_STD))
