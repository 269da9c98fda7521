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
import os

import numpy as np
import torch

from . import hiplib, imgtables
from .hiplib import LetterboxDesc

PRECISION_BITS = 32 - 8 - 2      # Pillow: 8-bit samples, 22 fractional coefficient bits
PAD_VALUE = 114

# Which library's uint8 arithmetic the device pipeline evaluates: 'pillow' (this package's host loader, bit-identical to it) or
# 'cv2' (the reference's loader: OpenCV restated, engine/imgtables.py; there is no host twin of it in this package - OpenCV is not
# installed - so this arithmetic exists on the GPU only).  Per call through ``arith=``, per process through YOLO_IMAGE_ARITH.
ARITHMETICS = ('pillow', 'cv2')


def default_arith():
    a = os.environ.get('YOLO_IMAGE_ARITH', 'pillow')
    if a not in ARITHMETICS:
        raise ValueError('YOLO_IMAGE_ARITH must be one of %s, got %r' % (ARITHMETICS, a))
    return a


def resample_tables(in_size, out_size, filt='bilinear'):
    """(bounds int32 [out][2], coefficients int32 [out][ksize], ksize) of one of Pillow's filters for one axis: 'bilinear' (triangle,
    support 1) or 'box' (support 0.5: what ``utils.datasets`` shrinks evaluation images with, Image.BOX)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    box = filt == 'box'
    support = (0.5 if box else 1.0) * filterscale
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
            if box:                                   # box_filter: 1 on (-0.5, 0.5]
                wv = 1.0 if -0.5 < v <= 0.5 else 0.0
            else:                                     # bilinear_filter
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


def _device_tables(in_size, out_size, device, filt='bilinear'):
    key = (in_size, out_size, str(device), filt)
    hit = _table_cache.get(key)
    if hit is None:
        b, k, ksize = identity_tables(in_size) if in_size == out_size else resample_tables(in_size, out_size, filt)
        hit = (torch.from_numpy(b).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), ksize)
        if len(_table_cache) > 64:
            _table_cache.clear()
        _table_cache[key] = hit
    return hit


_cv2_cache = {}


def _device_cv2_tables(in_size, out_size, horizontal, device):
    key = (in_size, out_size, horizontal, str(device))
    hit = _cv2_cache.get(key)
    if hit is None:
        idx, coef = imgtables.cv2_linear_tables(in_size, out_size, horizontal)
        hit = (torch.from_numpy(idx).to(device), torch.from_numpy(coef).to(device))
        if len(_cv2_cache) > 64:
            _cv2_cache.clear()
        _cv2_cache[key] = hit
    return hit


def letterbox_to_device(frame, new_shape, device, out=None, auto=True, scaleup=True, maxabsscaler=False, swap_rb=False, arith=None):
    """uint8 HWC frame (numpy array or tensor; RGB as this package's loaders deliver it, or BGR with ``swap_rb``) -> fp32 tensor
    ``(c, out_h, out_w)`` on ``device`` holding the letterboxed, scaled image; ``out`` may be a slot of a preallocated batch.
    ``arith``: 'pillow' (default; the host loader's resize) or 'cv2' (the reference's cv2.resize INTER_LINEAR, one fused launch).

    Returns ``(tensor, ratio, (dw, dh))`` like ``utils.datasets.letterbox`` returns ``(image, ratio, pad)``."""
    lib = hiplib.load()
    arith = arith or default_arith()
    if arith not in ARITHMETICS:
        raise ValueError('arith must be one of %s' % (ARITHMETICS,))
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
        if arith == 'cv2':
            (hb, hk), (vb, vk), hks, vks = _device_cv2_tables(w0, new_w, True, device), _device_cv2_tables(h0, new_h, False, device), 2, 2
            tmp = src                                  # unused by the fused kernel
        else:
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
                          scale=(2.0 / 256.0) if maxabsscaler else (1.0 / 256.0), shift=-1.0 if maxabsscaler else 0.0,
                          arith=imgtables.ARITH_CV2_LINEAR if arith == 'cv2' else imgtables.ARITH_PILLOW, out_u8=0)
        hiplib.check(lib.yh_letterbox_fwd(C.byref(d), hiplib.stream_ptr()), 'yh_letterbox_fwd')
    return out, ratio, pad


def _cv2_axis_tables(in_size, out_size, horizontal, code):
    """(bounds int32 [out][2], coefficients as int32 [out][ksize], ksize) for one axis of a cv2 resize of the given arithmetic."""
    if code == imgtables.ARITH_CV2_AREA and in_size != out_size:
        b, k, ksize = imgtables.cv2_area_tables(in_size, out_size)
        return b, k.view(np.int32), ksize
    if code == imgtables.ARITH_CV2_AREA:            # an axis that keeps its size inside an area resize: one unit weight
        b = np.stack([np.arange(out_size, dtype=np.int32), np.ones(out_size, dtype=np.int32)], 1)
        return b, np.ones((out_size, 1), dtype=np.float32).view(np.int32), 1
    b, k = imgtables.cv2_linear_tables(in_size, out_size, horizontal)
    return b, k, 2


def resize_to_device(frame, out_hw, device, code=imgtables.ARITH_CV2_LINEAR, lib=None):
    """cv2.resize of one uint8 HWC frame on the GPU -> uint8 HWC tensor (``code``: an ``imgtables.ARITH_CV2_*`` value; INTER_AREA with
    integer factors needs no tables).  The reference calls it in load_image (datasets.py:519-526)."""
    lib = lib or hiplib.load()
    src = torch.as_tensor(np.ascontiguousarray(frame)) if isinstance(frame, np.ndarray) else frame.contiguous()
    if src.dtype != torch.uint8 or src.dim() != 3:
        raise ValueError('expected a uint8 HWC frame')
    h0, w0, c = src.shape
    new_h, new_w = out_hw
    device = torch.device(device)
    with hiplib.on_device(torch.empty(0, device=device)):
        src = src.to(device, non_blocking=True)
        out = torch.empty((new_h, new_w, c), dtype=torch.uint8, device=device)
        P = hiplib.ptr
        d = LetterboxDesc(src=P(src), tmp=P(src), dst=P(out), h0=h0, w0=w0, c=c, src_pitch=w0 * c, new_h=new_h, new_w=new_w, out_h=new_h,
                          out_w=new_w, top=0, left=0, pad_value=PAD_VALUE, swap_rb=0, scale=1.0, shift=0.0, arith=code, out_u8=1)
        keep = []
        if code == imgtables.ARITH_CV2_AREA_FAST:
            fx, fy = imgtables.cv2_area_is_fast((w0, h0), (new_w, new_h))
            d.hksize, d.vksize = fx, fy
        else:
            hb, hk, d.hksize = _cv2_axis_tables(w0, new_w, True, code)
            vb, vk, d.vksize = _cv2_axis_tables(h0, new_h, False, code)
            keep = [torch.from_numpy(np.ascontiguousarray(t)).to(device) for t in (hb, hk, vb, vk)]
            d.hbounds, d.hk, d.vbounds, d.vk = (P(t) for t in keep)
        hiplib.check(lib.yh_letterbox_fwd(C.byref(d), hiplib.stream_ptr()), 'yh_letterbox_fwd')
    return out


def resample_reference(img, out_hw, filt='bilinear'):
    """Integer restatement of the two passes on the host (numpy), used by the CPU tests to pin ``resample_tables`` to Pillow."""
    h0, w0, c = img.shape
    new_h, new_w = out_hw
    hb, hk, _ = identity_tables(w0) if w0 == new_w else resample_tables(w0, new_w, filt)
    vb, vk, _ = identity_tables(h0) if h0 == new_h else resample_tables(h0, new_h, filt)
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


# ------------------------------------------------------------------------------------- evaluation / rect items (first slice)
def render_letterbox_items(batch, device, out=None, maxabsscaler=False, lib=None):
    """A ``utils.datasets.LetterboxBatch`` (or a list of ``LetterboxItem``) -> the network input ``(n, c, H, W)`` fp32 on ``device``:
    what ``imgs.to(device).float() / 256.0`` (test.py:95-96) makes of the host loader's items.  Per item ONE ``yh_letterbox_fwd``:
    load_image's resize (datasets.py:519-526) + letterbox's border (:637-643) + scaling + HWC->CHW; one upload per batch.

    Pillow arithmetic: Image.BOX when shrinking, as ``utils.datasets.load_image`` does for evaluation; cv2 arithmetic: INTER_AREA
    (general or integer-factor form), or INTER_LINEAR where the reference's load_image would use it."""
    from utils.datasets import LetterboxBatch
    lib = lib or hiplib.load()
    if not isinstance(batch, LetterboxBatch):
        batch = LetterboxBatch(batch)
    items = batch.items
    n, c = len(items), items[0].channels
    H, W = items[0].out_hw
    if any(tuple(it.out_hw) != (H, W) for it in items):
        raise ValueError('the items of a batch must share their rectangle')
    device = torch.device(device)
    if out is None:
        out = torch.empty((n, c, H, W), dtype=torch.float32, device=device)
    assert out.is_contiguous() and tuple(out.shape) == (n, c, H, W) and out.dtype == torch.float32
    dev_buf = batch.blob.to(device, non_blocking=True)
    base, P = dev_buf.data_ptr(), hiplib.ptr
    keep = []
    with hiplib.on_device(dev_buf):
        for i, it in enumerate(items):
            h0, w0 = it.frame
            h, w = it.resized_hw
            d = LetterboxDesc(src=base + int(batch.offsets[i]), tmp=base, dst=out[i].data_ptr(), h0=h0, w0=w0, c=c, src_pitch=w0 * c,
                              new_h=h, new_w=w, out_h=H, out_w=W, top=it.top, left=it.left, pad_value=PAD_VALUE, swap_rb=0,
                              scale=(2.0 / 256.0) if maxabsscaler else (1.0 / 256.0), shift=-1.0 if maxabsscaler else 0.0, out_u8=0)
            if it.arith == 'cv2':
                code = it.code if it.code is not None else imgtables.ARITH_CV2_LINEAR          # same size: the identity tables
                d.arith = code
                if code == imgtables.ARITH_CV2_AREA_FAST:
                    d.hksize, d.vksize = imgtables.cv2_area_is_fast((w0, h0), (w, h))
                else:
                    tabs = _cv2_tables_on(device, w0, w, True, code) + _cv2_tables_on(device, h0, h, False, code)
                    d.hbounds, d.hk, d.hksize, d.vbounds, d.vk, d.vksize = P(tabs[0]), P(tabs[1]), tabs[2], P(tabs[3]), P(tabs[4]), tabs[5]
            else:
                filt = 'box' if it.code in (imgtables.ARITH_CV2_AREA, imgtables.ARITH_CV2_AREA_FAST) else 'bilinear'
                hb, hk, d.hksize = _device_tables(w0, w, device, filt)
                vb, vk, d.vksize = _device_tables(h0, h, device, filt)
                tmp = torch.empty((h0, w, c), dtype=torch.uint8, device=device)
                keep.append(tmp)
                d.arith, d.tmp, d.hbounds, d.hk, d.vbounds, d.vk = imgtables.ARITH_PILLOW, P(tmp), P(hb), P(hk), P(vb), P(vk)
            hiplib.check(lib.yh_letterbox_fwd(C.byref(d), hiplib.stream_ptr()), 'yh_letterbox_fwd')
    return out


_cv2_axis_cache = {}


def _cv2_tables_on(device, in_size, out_size, horizontal, code):
    key = (in_size, out_size, horizontal, code, str(device))
    hit = _cv2_axis_cache.get(key)
    if hit is None:
        b, k, ksize = _cv2_axis_tables(in_size, out_size, horizontal, code)
        hit = (torch.from_numpy(np.ascontiguousarray(b)).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), ksize)
        if len(_cv2_axis_cache) > 128:
            _cv2_axis_cache.clear()
        _cv2_axis_cache[key] = hit
    return hit


# ----------------------------------------------------------------------------------------------- training items (second slice)
def render_mosaic_items(items, device, out=None, dtype=torch.float32, divisor=256.0, lib=None):
    """A batch of ``utils.datasets.MosaicItem`` recipes (a ``MosaicBatch`` from the loader's ``collate_fn``, or a plain list) ->
    the network input on ``device``: ``(n, c, h, w)`` float (value / divisor, what ``imgs.to(device).float() / 256.0`` of
    train.py:345 produces from the host loader's items) or uint8 (the host loader's items themselves).  One upload of the cropped
    source frames per batch, one ``yh_mosaic_affine_hsv`` launch per item."""
    from utils.datasets import MosaicBatch
    from .hiplib import MosaicDesc
    lib = lib or hiplib.load()
    batch = items if isinstance(items, MosaicBatch) else MosaicBatch(items)
    items = batch.items
    n = len(items)
    h, w = items[0].out_hw
    c = items[0].channels
    code = {torch.uint8: 0, torch.float32: 1, torch.float16: 2}[dtype]
    if out is None:
        out = torch.empty((n, c, h, w), device=device, dtype=dtype)
    assert out.is_contiguous() and tuple(out.shape) == (n, c, h, w) and out.dtype == dtype
    if getattr(items[0], 'arith', 'pillow') == 'cv2':
        return _render_mosaic_items_cv2(batch, device, out, code, divisor, lib)
    dev_buf = batch.blob.to(device, non_blocking=True)
    base, offs = dev_buf.data_ptr(), batch.offsets
    k = 0
    with hiplib.on_device(dev_buf):
        for i, it in enumerate(items):
            d = MosaicDesc()
            for j, (shape, (x1a, y1a, x2a, y2a), (x1b, y1b)) in enumerate(it.parts):
                if shape is not None:
                    d.src[j] = base + int(offs[k])
                    d.src_h[j], d.src_w[j], d.src_pitch[j] = shape[0], shape[1], shape[1] * c
                d.x1a[j], d.y1a[j], d.x2a[j], d.y2a[j], d.x1b[j], d.y1b[j] = x1a, y1a, x2a, y2a, x1b, y1b
                k += 1
            d.dst = out[i].data_ptr()
            for q in range(6):
                d.inv[q] = float(it.inv[q])
            d.hsv = 0 if it.hsv_gains is None else 1
            for q in range(3):
                d.hsv_gain[q] = 1.0 if it.hsv_gains is None else float(it.hsv_gains[q])
            d.canvas_h, d.canvas_w = it.canvas
            d.out_h, d.out_w, d.c, d.pad_value = h, w, c, PAD_VALUE
            d.flip_lr, d.out_dtype, d.divisor = int(bool(it.flip)), code, float(divisor)
            hiplib.check(lib.yh_mosaic_affine_hsv(C.byref(d), hiplib.stream_ptr()), 'yh_mosaic_affine_hsv')
    # dev_buf may die here: the caching allocator only hands its memory to later work on this stream
    return out


_host_cv2_cache = {}


def _host_cv2_tables(in_size, out_size, horizontal):
    key = (in_size, out_size, horizontal)
    hit = _host_cv2_cache.get(key)
    if hit is None:
        if len(_host_cv2_cache) > 256:
            _host_cv2_cache.clear()
        hit = _host_cv2_cache[key] = imgtables.cv2_linear_tables(in_size, out_size, horizontal)
    return hit


def _render_mosaic_items_cv2(batch, device, out, code, divisor, lib):
    """The reference's arithmetic: every source crop is resized on the device with cv2.resize's INTER_LINEAR formulas (one
    ``yh_letterbox_fwd`` launch per part, uint8 output, only the window the warp can touch), then one ``yh_mosaic_affine_hsv``
    launch per item warps / colour-augments with warpAffine's and cvtColor's.  One upload each for the crops, the per-part tables
    and the HSV lookup tables of the batch."""
    from .hiplib import MosaicDesc
    items = batch.items
    n = len(items)
    h, w = items[0].out_hw
    c = items[0].channels
    tabs, tab_off, win_off, plan = [], 0, 0, []
    k = 0
    for it in items:
        for shape, (x1a, y1a, x2a, y2a), _, spec in it.parts:
            if shape is not None:
                h0, w0, hr, wr, wx0, wy0, c0, r0 = spec
                ww, wh = x2a - x1a, y2a - y1a
                ix, kx = _host_cv2_tables(w0, wr, True)
                iy, ky = _host_cv2_tables(h0, hr, False)
                arrays = (ix[wx0:wx0 + ww] - c0, kx[wx0:wx0 + ww], iy[wy0:wy0 + wh] - r0, ky[wy0:wy0 + wh])
                offs = []
                for a in arrays:
                    a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
                    tabs.append(a)
                    offs.append(tab_off)
                    tab_off += a.size
                plan.append((k, shape, ww, wh, offs, win_off))
                win_off += (ww * wh * c + 15) // 16 * 16
            k += 1
    luts = np.zeros((n, 3, 256), dtype=np.uint8)
    for i, it in enumerate(items):
        if it.hsv_gains is not None:
            luts[i] = imgtables.hsv_luts(it.hsv_gains)
    dev_buf = batch.blob.to(device, non_blocking=True)
    dev_tab = torch.from_numpy(np.concatenate(tabs) if tabs else np.zeros(1, dtype=np.int32)).to(device, non_blocking=True)
    dev_lut = torch.from_numpy(luts).to(device, non_blocking=True)
    windows = torch.empty(max(win_off, 16), dtype=torch.uint8, device=device)
    base, offs, tbase, wbase = dev_buf.data_ptr(), batch.offsets, dev_tab.data_ptr(), windows.data_ptr()
    where = {}
    with hiplib.on_device(dev_buf):
        for k, (ch, cw), ww, wh, toffs, woff in plan:
            d = LetterboxDesc(src=base + int(offs[k]), tmp=base, dst=wbase + woff, hbounds=tbase + 4 * toffs[0], hk=tbase + 4 * toffs[1],
                              vbounds=tbase + 4 * toffs[2], vk=tbase + 4 * toffs[3], h0=ch, w0=cw, c=c, src_pitch=cw * c, hksize=2, vksize=2,
                              new_h=wh, new_w=ww, out_h=wh, out_w=ww, top=0, left=0, pad_value=PAD_VALUE, swap_rb=0, scale=1.0, shift=0.0,
                              arith=imgtables.ARITH_CV2_LINEAR, out_u8=1)
            hiplib.check(lib.yh_letterbox_fwd(C.byref(d), hiplib.stream_ptr()), 'yh_letterbox_fwd')
            where[k] = (wbase + woff, wh, ww)
        k = 0
        for i, it in enumerate(items):
            d = MosaicDesc()
            for j, (shape, (x1a, y1a, x2a, y2a), _, spec) in enumerate(it.parts):
                if shape is not None:
                    d.src[j], d.src_h[j], d.src_w[j] = where[k]
                    d.src_pitch[j] = d.src_w[j] * c
                d.x1a[j], d.y1a[j], d.x2a[j], d.y2a[j], d.x1b[j], d.y1b[j] = x1a, y1a, x2a, y2a, 0, 0
                k += 1
            d.dst = out[i].data_ptr()
            for q in range(6):
                d.inv[q] = float(it.inv[q])
            d.hsv = 0 if it.hsv_gains is None else 1
            d.lut = dev_lut[i].data_ptr()
            d.canvas_h, d.canvas_w = it.canvas
            d.out_h, d.out_w, d.c, d.pad_value = h, w, c, PAD_VALUE
            d.flip_lr, d.out_dtype, d.divisor, d.arith = int(bool(it.flip)), code, float(divisor), imgtables.ARITH_CV2_LINEAR
            hiplib.check(lib.yh_mosaic_affine_hsv(C.byref(d), hiplib.stream_ptr()), 'yh_mosaic_affine_hsv')
    return out


def mosaic_reference(item):
    """Host restatement (numpy, float64 / float32 exactly as the kernel) of one item -> uint8 CHW.  Test infrastructure: pins the
    kernel's arithmetic to the host loader (Pillow transform + utils.datasets.augment_hsv) on the CPU tier."""
    from utils.datasets import _rgb_to_hsv, _hsv_to_rgb
    H, W = item.canvas
    c = item.channels
    canvas = np.full((H, W, c), PAD_VALUE, dtype=np.uint8)
    for crop, (x1a, y1a, x2a, y2a), (x1b, y1b) in item.parts:   # an item as the dataset returns it (before collation)
        if crop is not None:
            canvas[y1a:y2a, x1a:x2a] = crop[y1b:y1b + y2a - y1a, x1b:x1b + x2a - x1a].reshape(y2a - y1a, x2a - x1a, c)
    oh, ow = item.out_hw
    yo, xo = np.mgrid[0:oh, 0:ow].astype(np.float64)
    a = item.inv
    xin = a[0] * (xo + 0.5) + a[1] * (yo + 0.5) + a[2]
    yin = a[3] * (xo + 0.5) + a[4] * (yo + 0.5) + a[5]
    inside = ~((xin < 0) | (yin < 0) | (xin >= W) | (yin >= H))
    xin, yin = xin - 0.5, yin - 0.5
    x, y = np.floor(xin).astype(np.int64), np.floor(yin).astype(np.int64)
    dx, dy = xin - x, yin - y
    yc, x0, x1 = np.clip(y, 0, H - 1), np.clip(x, 0, W - 1), np.clip(x + 1, 0, W - 1)
    second = (y + 1 >= 0) & (y + 1 < H)
    y1c = np.clip(y + 1, 0, H - 1)
    img = np.empty((oh, ow, c), dtype=np.uint8)
    f = canvas.astype(np.float64)
    for ch in range(c):
        p = f[..., ch]
        v1 = p[yc, x0] + (p[yc, x1] - p[yc, x0]) * dx
        v2 = np.where(second, p[y1c, x0] + (p[y1c, x1] - p[y1c, x0]) * dx, v1)
        img[..., ch] = np.where(inside, (v1 + (v2 - v1) * dy).astype(np.uint8), PAD_VALUE)
    if item.hsv_gains is not None:
        h, s, v = _rgb_to_hsv(img)
        g = item.hsv_gains
        img = _hsv_to_rgb((h * g[0]) % 1.0, np.clip(s * g[1], 0, 1), np.clip(v * g[2], 0, 1))
    if item.flip:
        img = np.fliplr(img)
    return np.ascontiguousarray(img.transpose(2, 0, 1))
