"""Device-side letterbox (first slice): one uint8 HWC frame -> its slot of the fp32 NCHW network input, on the GPU.

Host work is reduced to the tiny coefficient tables; the frame is uploaded as it was decoded (h0 x w0 x c bytes instead of a
letterboxed float tensor) and ``yh_letterbox_fwd`` (csrc/preprocess.hip) does resize + border + x/256 + HWC->CHW in two launches.

``resample_tables`` restates Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (src/libImaging/Resample.c, the library this
package's host loader resizes with: Image.BILINEAR, box = whole image) operation for operation in float64, so the device image is
bit-identical to ``utils.datasets.letterbox`` (tests/test_preprocess.py pins the restatement against Pillow itself on the CPU and
the kernel against the host loader on the GPU).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import hiplib
from .hiplib import LetterboxDesc

PRECISION_BITS = 32 - 8 - 2      # Pillow: 8-bit samples, 22 fractional coefficient bits
PAD_VALUE = 114


def resample_tables(in_size, out_size):
    """(bounds int32 [out][2], coefficients int32 [out][ksize], ksize) of Pillow's bilinear filter for one axis."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                      # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.empty(xmax, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            wv = 1.0 - v if v < 1.0 else 0.0
            w[x] = wv
            ww += wv
        if ww != 0.0:
            for x in range(xmax):
                w[x] /= ww
        for x in range(xmax):                         # normalize_coeffs_8bpc
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
    return bounds, kk, ksize


def identity_tables(size):
    """Pillow skips a pass whose size does not change; the same tables with one unit coefficient make the pass an exact copy."""
    bounds = np.stack([np.arange(size, dtype=np.int32), np.ones(size, dtype=np.int32)], 1)
    return bounds, np.full((size, 1), 1 << PRECISION_BITS, dtype=np.int32), 1


def letterbox_geometry(h0, w0, new_shape, auto=True, scaleup=True):
    """Sizes and offsets of utils.datasets.letterbox (reference datasets.py:611-646): (new_h, new_w, out_h, out_w, top, left, ratio, (dw, dh))."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / h0, new_shape[1] / w0)
    if not scaleup:
        r = min(r, 1.0)
    new_w, new_h = int(round(w0 * r)), int(round(h0 * r))
    dw, dh = new_shape[1] - new_w, new_shape[0] - new_h
    if auto:
        dw, dh = dw % 64, dh % 64
    dw, dh = dw / 2, dh / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_h, new_w, new_h + top + bottom, new_w + left + right, top, left, (r, r), (dw, dh)


_table_cache = {}


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    hit = _table_cache.get(key)
    if hit is None:
        b, k, ksize = identity_tables(in_size) if in_size == out_size else resample_tables(in_size, out_size)
        hit = (torch.from_numpy(b).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), ksize)
        if len(_table_cache) > 64:
            _table_cache.clear()
        _table_cache[key] = hit
    return hit


def letterbox_to_device(frame, new_shape, device, out=None, auto=True, scaleup=True, maxabsscaler=False, swap_rb=False):
    """uint8 HWC frame (numpy array or tensor; RGB as this package's loaders deliver it, or BGR with ``swap_rb``) -> fp32 tensor
    ``(c, out_h, out_w)`` on ``device`` holding the letterboxed, scaled image; ``out`` may be a slot of a preallocated batch.

    Returns ``(tensor, ratio, (dw, dh))`` like ``utils.datasets.letterbox`` returns ``(image, ratio, pad)``."""
    lib = hiplib.load()
    if isinstance(frame, np.ndarray):
        frame = np.ascontiguousarray(frame)
        if any(st < 0 for st in frame.strides):     # a flipped size-1 axis counts as contiguous for numpy but not for torch
            frame = frame.copy()
    src = torch.as_tensor(frame) if isinstance(frame, np.ndarray) else frame.contiguous()
    if src.dtype != torch.uint8 or src.dim() != 3:
        raise ValueError('expected a uint8 HWC frame')
    h0, w0, c = src.shape
    new_h, new_w, out_h, out_w, top, left, ratio, pad = letterbox_geometry(h0, w0, new_shape, auto=auto, scaleup=scaleup)
    device = torch.device(device)
    with hiplib.on_device(torch.empty(0, device=device)):
        src = src.to(device, non_blocking=True)
        hb, hk, hks = _device_tables(w0, new_w, device)
        vb, vk, vks = _device_tables(h0, new_h, device)
        tmp = torch.empty((h0, new_w, c), dtype=torch.uint8, device=device)
        if out is None:
            out = torch.empty((c, out_h, out_w), dtype=torch.float32, device=device)
        elif tuple(out.shape) != (c, out_h, out_w) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError('out must be a contiguous fp32 (%d, %d, %d) tensor' % (c, out_h, out_w))
        P = hiplib.ptr
        d = LetterboxDesc(src=P(src), tmp=P(tmp), dst=P(out), hbounds=P(hb), hk=P(hk), vbounds=P(vb), vk=P(vk), h0=h0, w0=w0, c=c,
                          src_pitch=w0 * c, hksize=hks, vksize=vks, new_h=new_h, new_w=new_w, out_h=out_h, out_w=out_w, top=top,
                          left=left, pad_value=PAD_VALUE, swap_rb=1 if swap_rb else 0,
                          scale=(2.0 / 256.0) if maxabsscaler else (1.0 / 256.0), shift=-1.0 if maxabsscaler else 0.0)
        hiplib.check(lib.yh_letterbox_fwd(C.byref(d), hiplib.stream_ptr()), 'yh_letterbox_fwd')
    return out, ratio, pad


def resample_reference(img, out_hw):
    """Integer restatement of the two passes on the host (numpy), used by the CPU tests to pin ``resample_tables`` to Pillow."""
    h0, w0, c = img.shape
    new_h, new_w = out_hw
    hb, hk, _ = identity_tables(w0) if w0 == new_w else resample_tables(w0, new_w)
    vb, vk, _ = identity_tables(h0) if h0 == new_h else resample_tables(h0, new_h)
    src = img.astype(np.int64)
    tmp = np.empty((h0, new_w, c), dtype=np.int64)
    for x in range(new_w):
        x0, n = hb[x]
        acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * hk[x, :n].astype(np.int64)[None, :, None]).sum(1)
        tmp[:, x, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.empty((new_h, new_w, c), dtype=np.uint8)
    for y in range(new_h):
        y0, n = vb[y]
        acc = (1 << (PRECISION_BITS - 1)) + (tmp[y0:y0 + n] * vk[y, :n].astype(np.int64)[:, None, None]).sum(0)
        out[y] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out
