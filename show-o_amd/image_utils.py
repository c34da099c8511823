"""Image pre / post-processing on the device: the host-side torchvision / PIL / numpy steps around the
tokenizer in the reference, with the same names and results:

  image_transform(image, resolution=256, normalize=True)   training/utils.py:178-185 (Resize(bicubic) -> CenterCrop -> ToTensor
                                                           -> Normalize(0.5, 0.5)); returns the fp32 [C,R,R] tensor ON THE GPU
  images_to_uint8(images)                                  inference_t2i.py:157-159 (clamp((x+1)/2,0,1)*255 -> uint8 NHWC)
  inpainting_token_mask(mask, resolution)                  inference_t2i.py:100-108 (bicubic down-sample by 16, >= 0.5 -> bool)

The resize is PIL's antialiased bicubic (what torchvision's Resize calls for PIL inputs): its coefficient tables depend only on
the two sizes, are computed here in double exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do, and are cached;
the integer resampling itself runs in csrc/image_ops.hip and is byte-exact with PIL (tests/test_image_gpu.py).
"""
import functools
import math

import numpy as np
import torch

from . import _lib

_PRECISION_BITS = 22  # Pillow Resample.c: 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=256)
def _coeff_table(in_size, out_size):
    """(bounds int32 [out,2], kk int32 [out,ksize]) of one axis; identity table when the size does not change"""
    if in_size == out_size:
        b = np.stack([np.arange(out_size), np.ones(out_size)], 1).astype(np.int32)
        return b, np.full((out_size, 1), 1 << _PRECISION_BITS, np.int32)
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * inv) for x in range(n)]
        total = 0.0
        for v in w:
            total += v
        for x, v in enumerate(w):
            if total != 0.0:
                v = v / total
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, n)
    return bounds, kk


@functools.lru_cache(maxsize=64)
def _device_tables(in_size, out_size, device):
    b, kk = _coeff_table(in_size, out_size)
    return torch.from_numpy(b).to(device), torch.from_numpy(kk).to(device), kk.shape[1]


def _as_u8_hwc(image, device):
    if isinstance(image, torch.Tensor):
        t = image
    else:  # PIL.Image or array-like: the decoded bytes are host data, one H2D copy
        t = torch.from_numpy(np.array(image))
    if t.dtype != torch.uint8:
        raise TypeError("image_transform expects 8-bit image data (a PIL image or a uint8 HWC tensor)")
    if t.dim() == 2:
        t = t[:, :, None]
    if t.dim() != 3 or t.shape[2] not in (1, 3):
        raise ValueError(f"image must be [H,W], [H,W,1] or [H,W,3] uint8, got {tuple(t.shape)}")
    return t.to(device).contiguous()


def image_transform(image, resolution=256, normalize=True, device="cuda", return_bytes=False):
    if not torch.cuda.is_available():
        raise RuntimeError("show-o_amd runs image_transform on the GPU (no CPU path exists)")
    img = _as_u8_hwc(image, device)
    H, W, C = img.shape
    # torchvision Resize(int): shorter side -> resolution, longer side int(resolution * long / short); CenterCrop rounds half-even
    if W <= H:
        ow, oh = resolution, int(resolution * H / W)
    else:
        oh, ow = resolution, int(resolution * W / H)
    ct, cl = int(round((oh - resolution) / 2.0)), int(round((ow - resolution) / 2.0))
    if oh < resolution or ow < resolution:
        raise ValueError("image_transform: crop larger than the resized image")
    bh, kh, ksh = _device_tables(W, ow, img.device)
    bv, kv, ksv = _device_tables(H, oh, img.device)
    tmp = torch.empty((H, ow, C), dtype=torch.uint8, device=img.device)
    out = torch.empty((C, resolution, resolution), dtype=torch.float32, device=img.device)
    u8 = torch.empty((resolution, resolution, C), dtype=torch.uint8, device=img.device) if return_bytes else None
    _lib.call("showo_image_resize_crop_normalize", _lib.ptr(img), H, W, C, oh, ow, _lib.ptr(bh), _lib.ptr(kh), ksh, _lib.ptr(bv),
              _lib.ptr(kv), ksv, ct, cl, resolution, resolution, int(bool(normalize)), _lib.ptr(tmp), _lib.ptr(out), _lib.ptr(u8),
              _lib.stream())
    return (out, u8) if return_bytes else out


def images_to_uint8(images):
    """fp32 [B,C,H,W] in [-1,1] (MAGVITv2.decode_code output) -> uint8 [B,H,W,C] on the device"""
    if not images.is_cuda:
        raise RuntimeError("show-o_amd converts images on the GPU (no CPU path exists)")
    x = images.detach().float().contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=torch.uint8, device=x.device)
    _lib.call("showo_images_to_uint8", _lib.ptr(x), _lib.ptr(out), B, C, H, W, _lib.stream())
    return out


def inpainting_token_mask(inpainting_mask, resolution, batch_size=1, return_values=False):
    """inpainting_mask: fp32 [1,R,R] in [0,1] (image_transform(mask, normalize=False)) -> bool [batch_size, (R/16)^2]: True where
    the token is to be regenerated (reference inference_t2i.py:100-108)"""
    if not inpainting_mask.is_cuda:
        raise RuntimeError("show-o_amd down-samples the mask on the GPU (no CPU path exists)")
    m = inpainting_mask.detach().float().reshape(resolution, resolution).contiguous()
    s = resolution // 16
    out = torch.empty((s * s,), dtype=torch.uint8, device=m.device)
    val = torch.empty((s * s,), dtype=torch.float32, device=m.device) if return_values else None
    _lib.call("showo_mask_downsample_threshold", _lib.ptr(m), resolution, s, _lib.ptr(out), _lib.ptr(val), _lib.stream())
    mask = out.to(torch.bool)[None].repeat(batch_size, 1)
    return (mask, val) if return_values else mask
