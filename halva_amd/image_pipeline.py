"""GPU image preprocessing for HallDataset batches (product code).

Stands in for the per-sample CPU work of reference llava/train/train_halva.py:735-751 (expand2square on a mean-coloured
canvas + CLIPImageProcessor.preprocess: bicubic shortest-edge resize, centre crop, rescale, normalise) and of the VILA twin
(vila/mm_utils.py:150-193: `image.resize((S, S))` + SiglipImageProcessor), once the image is DECODED (JPEG/PNG decoding
stays on the CPU workers).  The host computes Pillow's resample weights (Resample.c precompute_coeffs /
normalize_coeffs_8bpc: double precision, normalised, 22-bit fixed point; cached per geometry) and one descriptor per image;
`halva_image_preprocess` runs the two resample passes, the crop and the normalisation for the whole batch in two launches.
Results are bit-exact with Pillow + transformers (tests/test_image_pipeline_gpu.py).  No CPU fallback.
"""
import functools
import math

import numpy as np
import torch

from .hip import BF16, F32, call, ptr, stream_ptr

PRECISION_BITS = 22

DESC_DTYPE = np.dtype([("src_off", np.int64), ("tmp_off", np.int64), ("src_h", np.int32), ("src_w", np.int32),
                       ("pad_x", np.int32), ("pad_y", np.int32), ("bg", np.int32, (3,)), ("out_h", np.int32), ("out_w", np.int32),
                       ("row0", np.int32), ("tmp_rows", np.int32), ("kh_off", np.int32), ("ksize_h", np.int32), ("bh_off", np.int32),
                       ("kv_off", np.int32), ("ksize_v", np.int32), ("bv_off", np.int32), ("crop_y", np.int32), ("crop_x", np.int32),
                       ("need_h", np.int32), ("need_v", np.int32)], align=True)
assert DESC_DTYPE.itemsize == 104, DESC_DTYPE.itemsize      # must match HalvaImageDesc in include/halva_hip.h


def _bicubic(x, a=-0.5):
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


@functools.lru_cache(maxsize=256)
def coeff_tables(in_size, out_size):
    """(ksize, bounds int32 [out, 2], coef int32 [out, ksize]) of Pillow's bicubic resample from in_size to out_size."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    k = np.arange(ksize, dtype=np.int64)[None, :]
    w = _bicubic((k + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(k < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):                       # left-to-right accumulation, exactly like the C loop
        ww = ww + w[:, j]
    w = np.where((ww != 0.0)[:, None], w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(w < 0, np.trunc(-0.5 + w * (1 << PRECISION_BITS)), np.trunc(0.5 + w * (1 << PRECISION_BITS))).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return ksize, bounds, np.ascontiguousarray(fixed)


def normalize_lut(mean, std, rescale_factor=1 / 255):
    v = (np.arange(256, dtype=np.float64) * rescale_factor).astype(np.float32)
    return np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)


def _resize_output_size(h, w, shortest_edge):
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = shortest_edge, int(shortest_edge * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


class GpuImagePipeline:
    """mode 'pad'    : expand2square(mean colour) -> resize(shortest_edge=size) -> centre crop   (hallava_7b.sh: --image_aspect_ratio pad)
       mode 'crop'   : resize(shortest_edge=size) -> centre crop                                 (the processor's default behaviour)
       mode 'resize' : resize to (size, size)                                                    (halva_vila_13b.sh: --image_aspect_ratio resize)"""

    def __init__(self, mode, size, image_mean, image_std, rescale_factor=1 / 255, out_dtype=torch.bfloat16, device="cuda"):
        if mode not in ("pad", "crop", "resize"):
            raise ValueError("unknown image pipeline mode %r" % mode)
        self.mode, self.size, self.device, self.out_dtype = mode, int(size), torch.device(device), out_dtype
        self.bg = tuple(int(x * 255) for x in image_mean)
        self.lut = torch.from_numpy(normalize_lut(image_mean, image_std, rescale_factor)).to(self.device)

    @classmethod
    def from_processor(cls, processor, image_aspect_ratio, **kw):
        size = processor.size
        get = (lambda k: size.get(k)) if isinstance(size, dict) else (lambda k: getattr(size, k, None))
        s = get("shortest_edge") or get("height")
        mode = "pad" if image_aspect_ratio == "pad" else ("resize" if image_aspect_ratio == "resize" or not get("shortest_edge") else "crop")
        return cls(mode, s, processor.image_mean, processor.image_std, getattr(processor, "rescale_factor", 1 / 255), **kw)

    def plan(self, shapes):
        """Descriptors + weight tables for images of the given (h, w) shapes (host only, integer work)."""
        S = self.size
        descs = np.zeros(len(shapes), dtype=DESC_DTYPE)
        coefs, bounds, table_at = [], [], {}
        c_off = b_off = 0

        def table(n_in, n_out):
            nonlocal c_off, b_off
            key = (n_in, n_out)
            if key not in table_at:
                ks, bd, cf = coeff_tables(n_in, n_out)
                table_at[key] = (c_off, ks, b_off)
                coefs.append(cf.reshape(-1))
                bounds.append(bd.reshape(-1))
                c_off += cf.size
                b_off += bd.size
            return table_at[key]

        src_off = tmp_off = 0
        max_tmp = 1
        for i, (h, w) in enumerate(shapes):
            d = descs[i]
            d["src_off"], d["src_h"], d["src_w"] = src_off, h, w
            src_off += h * w * 3
            if self.mode == "pad":
                side = max(h, w)
                ch, cw = side, side
                d["pad_x"], d["pad_y"] = (side - w) // 2, (side - h) // 2
            else:
                ch, cw = h, w
            d["bg"] = self.bg
            oh, ow = (S, S) if self.mode == "resize" else _resize_output_size(ch, cw, S)
            if oh < S or ow < S:
                raise NotImplementedError("centre crop larger than the resized image is not on the HALVA path")
            d["out_h"], d["out_w"] = oh, ow
            d["need_h"], d["need_v"] = int(ow != cw), int(oh != ch)
            row0, rows = 0, ch
            if d["need_v"]:
                off, ks, boff = table(ch, oh)
                _, bd, _ = coeff_tables(ch, oh)
                row0, rows = int(bd[0, 0]), int(bd[-1, 0] + bd[-1, 1] - bd[0, 0])
                assert row0 == 0          # whole-image box: the first output row always reads from row 0 (bounds stay absolute)
                d["kv_off"], d["ksize_v"], d["bv_off"] = off, ks, boff
            if d["need_h"]:
                off, ks, boff = table(cw, ow)
                d["kh_off"], d["ksize_h"], d["bh_off"] = off, ks, boff
            d["row0"], d["tmp_rows"] = row0, rows
            d["tmp_off"] = tmp_off
            tmp_off += rows * ow * 3
            max_tmp = max(max_tmp, rows * ow)
            d["crop_y"], d["crop_x"] = (oh - S) // 2, (ow - S) // 2
        coef = np.concatenate(coefs) if coefs else np.zeros(1, np.int32)
        bnd = np.concatenate(bounds) if bounds else np.zeros(2, np.int32)
        return descs, coef.astype(np.int32), bnd.astype(np.int32), src_off, tmp_off, max_tmp

    def __call__(self, images):
        """images: list of [H, W, 3] uint8 arrays / tensors (decoded RGB).  Returns [n, 3, size, size] on the device."""
        arrs = [np.ascontiguousarray(im.numpy() if isinstance(im, torch.Tensor) else np.asarray(im)) for im in images]
        for a in arrs:
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise TypeError("GpuImagePipeline takes decoded [H, W, 3] uint8 RGB images, got %s %s" % (a.dtype, a.shape))
        descs, coef, bnd, n_src, n_tmp, max_tmp = self.plan([a.shape[:2] for a in arrs])
        pack = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).to(self.device, non_blocking=True)
        dev = self.device
        d_desc = torch.from_numpy(descs.view(np.uint8).copy()).to(dev, non_blocking=True)
        d_coef = torch.from_numpy(coef).to(dev, non_blocking=True)
        d_bnd = torch.from_numpy(bnd).to(dev, non_blocking=True)
        tmp = torch.empty(max(n_tmp, 1), dtype=torch.uint8, device=dev)
        S = self.size
        out = torch.empty(len(arrs), 3, S, S, dtype=self.out_dtype, device=dev)
        call("halva_image_preprocess", ptr(pack), ptr(d_desc), ptr(d_coef), ptr(d_bnd), ptr(self.lut), ptr(tmp), ptr(out), len(arrs),
             int(max_tmp), S, S, BF16 if self.out_dtype == torch.bfloat16 else F32, stream_ptr())
        return out
