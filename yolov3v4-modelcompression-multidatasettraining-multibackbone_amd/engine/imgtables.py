"""Host-side coefficient tables for the device input pipeline when it is asked for OpenCV's arithmetic (``arith='cv2'``).

The reference's loaders resize with ``cv2.resize`` and warp with ``cv2.warpAffine`` (reference utils/datasets.py:519-526, :637,
:677).  The kernels in csrc/preprocess.hip / csrc/augment.hip evaluate OpenCV's uint8 formulas; what depends only on the image
SIZES - which source samples an output sample reads and with which fixed-point weights - is computed here once per size pair, in
numpy, exactly as OpenCV's ``resize()`` prepares its ``xofs / ialpha / yofs / ibeta`` (INTER_LINEAR, 11-bit weights) and
``computeResizeAreaTab`` (INTER_AREA, float weights) arrays.  OpenCV is not installed in this image: this is a restatement of the
library's published algorithm (modules/imgproc/src/resize.cpp, imgwarp.cpp), checked against the independent restatement in
oracle/cv2_restated.py by the tests - "third-party restated", not pinned to the library itself.

No torch, no HIP here: the module is imported by the loader workers (utils/datasets.py) as well as by engine/preprocess.py.
"""
import math

import numpy as np

ARITH_PILLOW, ARITH_CV2_LINEAR, ARITH_CV2_AREA, ARITH_CV2_AREA_FAST = 0, 1, 2, 3
COEF_BITS = 11


def _rint_i32(v):
    return np.rint(v).astype(np.int32)


def _linear_axis(ssize, dsize):
    scale = 1.0 / (float(dsize) / float(ssize))                  # resize(): inv_scale = dsize / ssize, hal::resize inverts it
    f = ((np.arange(dsize, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    return s, (f - s.astype(np.float32)).astype(np.float32)


def _weights(f):
    one, sc = np.float32(1.0), np.float32(1 << COEF_BITS)
    return np.stack([_rint_i32((one - f).astype(np.float32) * sc), _rint_i32(f * sc)], 1)


def cv2_linear_tables(ssize, dsize, horizontal):
    """(idx int32 [dsize][2], coef int32 [dsize][2]): output sample d = src[idx[d, 0]] * coef[d, 0] + src[idx[d, 1]] * coef[d, 1].

    Columns pin the fraction to 0 where the left sample falls off either end (HResizeLinear's xmin / xmax handling); rows keep
    the fraction and clip the two row indices (resizeGeneric_Invoker).  Same size: the identity (2048, 0)."""
    if ssize == dsize:
        i = np.arange(dsize, dtype=np.int32)
        return np.stack([i, i], 1), np.tile(np.array([[1 << COEF_BITS, 0]], dtype=np.int32), (dsize, 1))
    s, f = _linear_axis(ssize, dsize)
    if horizontal:
        lo, hi = s < 0, s >= ssize - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, ssize - 1, s))
        idx = np.stack([s, np.minimum(s + 1, ssize - 1)], 1)
    else:
        idx = np.stack([np.clip(s, 0, ssize - 1), np.clip(s + 1, 0, ssize - 1)], 1)
    return idx.astype(np.int32), _weights(f)


def cv2_area_tables(ssize, dsize):
    """INTER_AREA, shrinking by a non-integer factor (computeResizeAreaTab): (bounds int32 [dsize][2] = (first source sample,
    count), weights float32 [dsize][ksize], ksize).  The kernel accumulates ``sum += src * w`` in float32 in this order."""
    scale = 1.0 / (float(dsize) / float(ssize))
    rows = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = int(math.ceil(f1)), int(math.floor(f2))
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        first, w = s1, []
        if s1 - f1 > 1e-3:
            first = s1 - 1
            w.append(np.float32((s1 - f1) / cell))
        w.extend([np.float32(1.0 / cell)] * (s2 - s1))
        if f2 - s2 > 1e-3:
            w.append(np.float32(min(min(f2 - s2, 1.0), cell) / cell))
        rows.append((first, w))
    ksize = max(1, max(len(w) for _, w in rows))
    bounds = np.zeros((dsize, 2), dtype=np.int32)
    coef = np.zeros((dsize, ksize), dtype=np.float32)
    for d, (first, w) in enumerate(rows):
        bounds[d] = (first, len(w))
        coef[d, :len(w)] = w
    return bounds, coef, ksize


def cv2_area_is_fast(ssize_wh, dsize_wh):
    """resize(): integer decimation factors along both axes take resizeAreaFast_; returns (scale_x, scale_y) or None."""
    sx = 1.0 / (float(dsize_wh[0]) / ssize_wh[0])
    sy = 1.0 / (float(dsize_wh[1]) / ssize_wh[1])
    ix, iy = int(np.rint(sx)), int(np.rint(sy))
    eps = np.finfo(np.float64).eps
    return (ix, iy) if abs(sx - ix) < eps and abs(sy - iy) < eps else None


def cv2_invert_affine(M):
    """warpAffine's own inversion of the forward 2x3 matrix (imgwarp.cpp), operation for operation in double precision."""
    m = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(-1)[:6]]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return np.array(m, dtype=np.float64)


def hsv_luts(gains):
    """The three 256-entry tables of augment_hsv (reference datasets.py:539-542) as one uint8 [3][256] array."""
    x = np.arange(0, 256, dtype=np.int16)
    return np.stack([((x * gains[0]) % 180).astype(np.uint8), np.clip(x * gains[1], 0, 255).astype(np.uint8),
                     np.clip(x * gains[2], 0, 255).astype(np.uint8)])


def load_image_plan(h0, w0, img_size, augment):
    """load_image's decision (reference datasets.py:519-524): ((h, w) after the resize, arithmetic code or None for 'no resize')."""
    r = img_size / max(h0, w0)
    if r < 1 or (augment and r != 1):
        h, w = int(h0 * r), int(w0 * r)
        if (h, w) == (h0, w0):
            return (h0, w0), None
        if r < 1 and not augment:
            return (h, w), (ARITH_CV2_AREA_FAST if cv2_area_is_fast((w0, h0), (w, h)) else ARITH_CV2_AREA)
        return (h, w), ARITH_CV2_LINEAR
    return (h0, w0), None
