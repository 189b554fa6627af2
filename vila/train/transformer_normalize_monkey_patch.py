"""reference vila/train/transformer_normalize_monkey_patch.py: `normalize` that accepts single-channel images by
replicating the channel (train_halva_vila.py patches transformers.image_processing_utils.normalize with it)."""
from collections.abc import Iterable

import numpy as np


def patched_normalize(image, mean, std, data_format=None, input_data_format=None):
    from transformers.image_transforms import (get_channel_dimension_axis, infer_channel_dimension_format,
                                               to_channel_dimension_format)
    from transformers.image_utils import ChannelDimension
    if not isinstance(image, np.ndarray):
        raise ValueError("image must be a numpy array")
    input_data_format = infer_channel_dimension_format(image)
    axis = get_channel_dimension_axis(image)
    n = image.shape[axis]
    if isinstance(mean, Iterable):
        mean = list(mean)
        if len(mean) != n:
            if n != 1:
                raise ValueError(f"mean must have {n} elements if it is an iterable, got {len(mean)}")
            n = 3
            image = np.concatenate([image] * 3, axis=axis)
    else:
        mean = [mean] * n
    if isinstance(std, Iterable):
        std = list(std)
        if len(std) != n:
            raise ValueError(f"std must have {n} elements if it is an iterable, got {len(std)}")
    else:
        std = [std] * n
    mean, std = np.array(mean, dtype=image.dtype), np.array(std, dtype=image.dtype)
    image = (image - mean) / std if input_data_format == ChannelDimension.LAST else ((image.T - mean) / std).T
    return to_channel_dimension_format(image, data_format) if data_format is not None else image


def patch_normalize_preprocess():
    import transformers
    transformers.image_transforms.normalize = patched_normalize
