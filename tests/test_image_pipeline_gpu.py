"""GPU image preprocessing against the packages the reference calls (Pillow + transformers processors), on the
reference's own call sequence: bit-exact float32, and bf16 = the same values rounded."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from test_image_pipeline_cpu import SHAPES, _img  # noqa: E402


def _reference_pad(proc, img):
    pil = Image.fromarray(img)
    bg = tuple(int(x * 255) for x in proc.image_mean)
    side = max(pil.size)
    canvas = Image.new(pil.mode, (side, side), bg)
    canvas.paste(pil, ((side - pil.size[0]) // 2, (side - pil.size[1]) // 2))
    return proc.preprocess(canvas, return_tensors="np")["pixel_values"][0]


@pytest.mark.parametrize("mode", ["pad", "crop"])
def test_clip_batch_bit_exact(mode):
    from transformers import CLIPImageProcessor
    from halva_amd.image_pipeline import GpuImagePipeline
    proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    imgs = [_img(h, w, 11 * h + w) for h, w in SHAPES]
    pipe = GpuImagePipeline.from_processor(proc, "pad" if mode == "pad" else "square", out_dtype=torch.float32)
    assert pipe.mode == mode and pipe.size == 336
    got = pipe(imgs).cpu().numpy()
    for i, im in enumerate(imgs):
        want = _reference_pad(proc, im) if mode == "pad" else proc.preprocess(Image.fromarray(im), return_tensors="np")["pixel_values"][0]
        np.testing.assert_array_equal(got[i], want, err_msg=str(im.shape))
    bf = GpuImagePipeline.from_processor(proc, "pad" if mode == "pad" else "square")(imgs[:3])
    assert bf.dtype == torch.bfloat16 and torch.equal(bf.cpu(), torch.from_numpy(got[:3]).bfloat16())


def test_siglip_resize_bit_exact():
    from transformers import SiglipImageProcessor
    from halva_amd.image_pipeline import GpuImagePipeline
    proc = SiglipImageProcessor(size={"height": 384, "width": 384})
    imgs = [_img(h, w, 5 * h + w) for h, w in SHAPES[:6]] + [_img(384, 384, 1)]
    pipe = GpuImagePipeline.from_processor(proc, "resize", out_dtype=torch.float32)
    got = pipe(imgs).cpu().numpy()
    for i, im in enumerate(imgs):
        pil = Image.fromarray(im).resize((384, 384))
        want = proc.preprocess(pil, return_tensors="np")["pixel_values"][0]
        np.testing.assert_array_equal(got[i], want, err_msg=str(im.shape))
