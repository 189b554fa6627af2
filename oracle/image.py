"""Oracle (test infrastructure): the image side of HallDataset.__getitem__ restated in numpy.

Reference call sites: llava/train/train_halva.py:735-751 (`expand2square` with the processor mean as background, then
`processor.preprocess(image)['pixel_values'][0]`) and vila/mm_utils.py:150-193 (`image.resize((S, S))` for aspect ratio
'resize', then the tower's processor).  The arithmetic itself lives in two third-party packages absent from /root/reference:
  * Pillow (12.2.0 here) - `Image.resize(..., BICUBIC)`: src/libImaging/Resample.c (precompute_coeffs, normalize_coeffs_8bpc,
    ImagingResampleHorizontal_8bpc / Vertical_8bpc): separable convolution, double-precision weights normalised per output
    pixel, converted to 22-bit fixed point, accumulated in int32 with a +0.5 bias, clipped to uint8 after EACH pass;
  * transformers (5.15 here) - CLIPImageProcessor / SiglipImageProcessor (PIL backend): resize (shortest edge or fixed size),
    center crop, rescale in float64 -> float32, (x - mean) / std in float32.
Parity is pinned by tests/test_image_pipeline_cpu.py against those packages themselves (bit-exact uint8 and float32).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, in0=0.0, in1=None, support=2.0, filt=bicubic):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (ksize, bounds int32 [out, 2], kk int32 [out, ksize])."""
    in1 = float(in_size) if in1 is None else in1
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            w = filt((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    ki = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64))
    return ksize, bounds, ki.astype(np.int32)          # .astype(int) truncates toward zero like the C cast


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img, out_w, out_h):
    """PIL Image.resize((out_w, out_h), BICUBIC) on an [H, W, C] uint8 array (ImagingResample, box = whole image)."""
    h, w, _ = img.shape
    need_h, need_v = out_w != w, out_h != h
    cur = img
    ks_v, bv, kv = precompute_coeffs(h, out_h) if need_v else (0, None, None)
    if need_h:
        ks_h, bh, kh = precompute_coeffs(w, out_w)
        # only the rows the vertical pass will read are produced (ybox_first .. ybox_last)
        r0, r1 = (int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])) if need_v else (0, h)
        src = cur[r0:r1].astype(np.int64)
        out = np.empty((r1 - r0, out_w, img.shape[2]), dtype=np.uint8)
        for xx in range(out_w):
            xmin, n = bh[xx]
            acc = (src[:, xmin:xmin + n] * kh[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xx] = _clip8(acc)
        cur = out
        if need_v:
            bv = bv.copy()
            bv[:, 0] -= r0
    if need_v:
        src = cur.astype(np.int64)
        out = np.empty((out_h, cur.shape[1], img.shape[2]), dtype=np.uint8)
        for yy in range(out_h):
            ymin, n = bv[yy]
            acc = (src[ymin:ymin + n] * kv[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur if (need_h or need_v) else img.copy()


def expand2square(img, background):
    """train_halva.py:737-748 on an [H, W, 3] uint8 array; background = tuple(int(x * 255) for x in image_mean)."""
    h, w, c = img.shape
    if w == h:
        return img
    side = max(w, h)
    out = np.empty((side, side, c), dtype=np.uint8)
    out[:] = np.asarray(background, dtype=np.uint8)
    if w > h:
        out[(w - h) // 2:(w - h) // 2 + h] = img
    else:
        out[:, (h - w) // 2:(h - w) // 2 + w] = img
    return out


def resize_output_size(h, w, shortest_edge):
    """transformers.image_transforms.get_resize_output_image_size(size=int, default_to_square=False)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = shortest_edge, int(shortest_edge * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)          # (out_h, out_w)


def center_crop(img, ch, cw):
    """transformers center_crop for crops not larger than the image (the HALVA case) on [H, W, C]."""
    h, w, _ = img.shape
    top, left = (h - ch) // 2, (w - cw) // 2
    if top < 0 or left < 0:
        raise NotImplementedError("crop larger than the image (zero padding) is not on the HALVA path")
    return img[top:top + ch, left:left + cw]


def rescale_normalize(img_u8_hwc, mean, std, scale=1 / 255):
    x = (img_u8_hwc.astype(np.float64) * scale).astype(np.float32).transpose(2, 0, 1)
    mean = np.asarray(mean, dtype=np.float32)[:, None, None]
    std = np.asarray(std, dtype=np.float32)[:, None, None]
    return (x - mean) / std


def normalize_lut(mean, std, scale=1 / 255):
    """[3, 256] float32: the whole rescale+normalize per channel as a table of the 256 possible inputs."""
    v = (np.arange(256, dtype=np.float64) * scale).astype(np.float32)
    return np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)


def clip_preprocess(img_u8_hwc, size, mean, std, pad=True):
    """LLaVA path: [expand2square] -> resize(shortest_edge=size, BICUBIC) -> center_crop(size) -> rescale -> normalize."""
    if pad:
        img_u8_hwc = expand2square(img_u8_hwc, tuple(int(x * 255) for x in mean))
    oh, ow = resize_output_size(img_u8_hwc.shape[0], img_u8_hwc.shape[1], size)
    r = resize_bicubic_u8(img_u8_hwc, ow, oh)
    return rescale_normalize(center_crop(r, size, size), mean, std)


def siglip_preprocess(img_u8_hwc, size, mean, std):
    """VILA path, aspect ratio 'resize': image.resize((size, size)) [BICUBIC] -> processor resize (identity) -> rescale -> normalize."""
    r = resize_bicubic_u8(img_u8_hwc, size, size)
    return rescale_normalize(r, mean, std)
