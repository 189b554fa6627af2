"""reference vila/mm_utils.py:150-235 - the pieces HallDataset uses."""
import os

import torch
from PIL import Image

from llava.mm_utils import tokenizer_image_token  # noqa: F401


def _expand2square(img, fill):
    w, h = img.size
    if w == h:
        return img
    side = max(w, h)
    canvas = Image.new(img.mode, (side, side), fill)
    canvas.paste(img, ((side - w) // 2, (side - h) // 2))
    return canvas


def process_image(image_file, data_args, image_folder):
    """mm_utils.py:150-193.  'resize': PIL-resize to the processor's square size first; 'pad': expand to a square filled
    with the processor mean; then the tower's own preprocessing (SigLIP: resize + rescale + normalise)."""
    processor = data_args.image_processor
    if isinstance(image_file, str):
        path = os.path.join(image_folder, image_file) if image_folder is not None else image_file
        image = Image.open(path).convert("RGB")
    else:
        image = image_file
    if data_args.image_aspect_ratio == "resize":
        size = processor.crop_size if hasattr(processor, "crop_size") and processor.crop_size else processor.size
        image = image.resize((size["height"], size["width"]))
    if data_args.image_aspect_ratio == "pad":
        image = _expand2square(image, tuple(int(x * 255) for x in processor.image_mean))
    return processor.preprocess(image, return_tensors="pt")["pixel_values"][0]


def process_images(images, image_processor, model_cfg):
    model_cfg.image_processor = image_processor
    out = [process_image(im, model_cfg, None) for im in images]
    if all(x.shape == out[0].shape for x in out):
        out = torch.stack(out, dim=0)
    return out


def is_gemma_tokenizer(tokenizer):
    return "gemma" in tokenizer.__class__.__name__.lower()
