"""OpenCV's 8-bit image arithmetic for the reference's input pipeline, restated in numpy.  TEST INFRASTRUCTURE ONLY.

THIRD-PARTY RESTATED, PARITY UNPINNED.  The reference's loaders (/root/reference/utils/datasets.py:511-716) do their pixel work
with `opencv-python` (requirements.txt pins no version; `cv2` 4.x of 2020-21): `cv2.resize(INTER_LINEAR | INTER_AREA)`,
`cv2.copyMakeBorder`, `cv2.warpAffine(INTER_LINEAR, borderValue=114)`, `cv2.cvtColor(BGR2HSV / HSV2BGR)` + `cv2.LUT`.  OpenCV is
not installed in this image and not vendored in /root/reference, and the reference holds no golden images for this path, so the
functions below restate OpenCV's published uint8 algorithms (modules/imgproc/src: resize.cpp, imgwarp.cpp, color_hsv.simd.hpp;
the C++ scalar code paths, which the SIMD paths are written to reproduce) and cannot be checked against the library here.  What
the tests do pin: the device kernels (csrc/preprocess.hip, csrc/augment.hip with arith = cv2) and the host-side coefficient
tables (engine/preprocess.py) against THIS file, bit for bit; and this file against analytic properties of the algorithms
(identity, exact 2x decimation, constant images, agreement with a float bilinear within the fixed-point error bound).

Not covered: OpenCV >= 4.11 replaced warpAffine's fixed-point bilinear by a float kernel (different low bits); IPP / OpenVX /
CUDA builds of OpenCV; INTER_AREA when enlarging.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------------------------------ helpers


def cv_round(v):
    """cvRound: SSE2 cvtsd2si under the default rounding mode = round half to even."""
    return np.rint(v).astype(np.int64)


def _saturate_u8(v):
    return np.clip(v, 0, 255).astype(np.uint8)


def _hwc(img):
    img = np.asarray(img)
    assert img.dtype == np.uint8
    return img[:, :, None] if img.ndim == 2 else img


# --------------------------------------------------------------------------------------------- cv2.resize, INTER_LINEAR, 8U
INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def linear_axis(ssize, dsize):
    """Per destination index of one axis: (first source index, weight of it, weight of the next), as resize.cpp's `resize()`
    builds xofs / ialpha (and yofs / ibeta) for INTER_LINEAR with a fixed-point (uchar) destination:

        scale = 1. / (dsize / ssize)                      (double;  hal::resize gets inv_scale and inverts it again)
        fx = (float)((dx + 0.5) * scale - 0.5);  sx = cvFloor(fx);  fx -= sx
        horizontal only:  sx < 0 -> fx = 0, sx = 0;   sx >= ssize - 1 -> fx = 0, sx = ssize - 1
        ialpha = saturate_cast<short>((1.f - fx) * 2048), saturate_cast<short>(fx * 2048)
    The vertical axis keeps fy and clips the two ROW indices instead (resizeGeneric_Invoker: clip(sy0 - ksize2 + 1 + k, 0, h)).
    Returns (s, f) with s the unclipped floor index (int64) and f the float32 fraction, before any edge rule."""
    inv_scale = float(dsize) / float(ssize)
    scale = 1.0 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coef(f):
    one = np.float32(1.0)
    c0 = cv_round(((one - f).astype(np.float32) * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.float32))
    c1 = cv_round((f * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.float32))
    return np.clip(c0, -32768, 32767), np.clip(c1, -32768, 32767)


def resize_linear(img, dsize_wh):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 images (reference datasets.py:524, :637).

    HResizeLinear<uchar, int, short, 2048>: D[dx] = S[sx] * a0 + S[sx + 1] * a1 (int; S[sx] * 2048 from xmax on), then
    VResizeLinear<uchar, int, short, ...>:  dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
    An exact 2x decimation is rerouted by resize() to the 2x2 INTER_AREA average (S00 + S01 + S10 + S11 + 2) >> 2, which is what
    the formula above already evaluates to with all four weights 1024."""
    src = _hwc(img)
    h0, w0, c = src.shape
    w, h = int(dsize_wh[0]), int(dsize_wh[1])
    if (w, h) == (w0, h0):
        return src.copy()                                      # resize(): same size -> copyTo
    sx, fx = linear_axis(w0, w)
    lo, hi = sx < 0, sx >= w0 - 1
    fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, w0 - 1, sx))
    a0, a1 = _coef(fx)
    sx1 = np.minimum(sx + 1, w0 - 1)                            # weight 0 wherever this clamps
    S = src.astype(np.int64)
    rows = S[:, sx, :] * a0[None, :, None] + S[:, sx1, :] * a1[None, :, None]          # (h0, w, c) int
    sy, fy = linear_axis(h0, h)
    b0, b1 = _coef(fy)
    y0, y1 = np.clip(sy, 0, h0 - 1), np.clip(sy + 1, 0, h0 - 1)
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return _saturate_u8(out)


# ----------------------------------------------------------------------------------------------- cv2.resize, INTER_AREA, 8U
def _area_tab(ssize, dsize, scale):
    """computeResizeAreaTab (resize.cpp): list of (di, si, alpha float32)."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(math.ceil(fsx1)), int(math.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for s in range(sx1, sx2):
            tab.append((dx, s, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area(img, dsize_wh):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_AREA) when shrinking along both axes (reference datasets.py:523-524: the
    evaluation loader, r < 1 and not augment).  Integer scale factors take resizeAreaFast_ (integer sums; 2x2: (sum + 2) >> 2,
    otherwise saturate_cast<uchar>(sum * (1.f / area))), everything else resizeArea_<uchar, float>: float32 accumulation in table
    order, first along x into `buf`, then `sum += beta * buf` over the contributing rows."""
    src = _hwc(img)
    h0, w0, c = src.shape
    w, h = int(dsize_wh[0]), int(dsize_wh[1])
    if (w, h) == (w0, h0):
        return src.copy()
    scale_x, scale_y = 1.0 / (float(w) / w0), 1.0 / (float(h) / h0)
    assert scale_x >= 1 and scale_y >= 1, 'INTER_AREA enlarging is a bilinear variant the reference never reaches'
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))
    eps = np.finfo(np.float64).eps
    if abs(scale_x - isx) < eps and abs(scale_y - isy) < eps:
        S = src.astype(np.int64)
        full_w, full_h = w0 // isx, h0 // isy                  # cells that lie wholly inside the source
        out = np.zeros((h, w, c), dtype=np.uint8)
        for dy in range(h):
            sy0 = dy * isy
            if sy0 >= h0:
                continue
            rows_full = sy0 + isy <= h0
            for dx in range(w):
                sx0 = dx * isx
                if rows_full and dx < full_w:
                    s = S[sy0:sy0 + isy, sx0:sx0 + isx].sum((0, 1))
                    if isx == 2 and isy == 2:
                        out[dy, dx] = (s + 2) >> 2
                    else:
                        v = s.astype(np.float32) * np.float32(np.float32(1.0) / np.float32(isx * isy))
                        out[dy, dx] = _saturate_u8(cv_round(v))
                elif sx0 < w0:
                    cell = S[sy0:min(sy0 + isy, h0), sx0:min(sx0 + isx, w0)]
                    v = cell.sum((0, 1)).astype(np.float32) / np.float32(cell.shape[0] * cell.shape[1])
                    out[dy, dx] = _saturate_u8(cv_round(v))
        return out
    xtab, ytab = _area_tab(w0, w, scale_x), _area_tab(h0, h, scale_y)
    S = src.astype(np.float32)
    out = np.zeros((h, w, c), dtype=np.uint8)
    acc = np.zeros((w, c), dtype=np.float32)
    prev = ytab[0][0]
    for dy, sy, beta in ytab:
        buf = np.zeros((w, c), dtype=np.float32)
        for dx, sxx, alpha in xtab:
            buf[dx] = buf[dx] + S[sy, sxx] * alpha
        if dy != prev:
            out[prev] = _saturate_u8(cv_round(acc))
            acc = (beta * buf).astype(np.float32)
            prev = dy
        else:
            acc = (acc + beta * buf).astype(np.float32)
    out[prev] = _saturate_u8(cv_round(acc))
    return out


def copy_make_border(img, top, bottom, left, right, value):
    """cv2.copyMakeBorder(..., cv2.BORDER_CONSTANT, value) (reference datasets.py:642)."""
    src = _hwc(img)
    h, w, c = src.shape
    out = np.empty((h + top + bottom, w + left + right, c), dtype=np.uint8)
    out[:] = np.asarray(value, dtype=np.uint8)[:c]
    out[top:top + h, left:left + w] = src
    return out


def letterbox(img, new_shape=(416, 416), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True):
    """The reference's letterbox (datasets.py:611-646) on the restated cv2 calls: (image, ratio, (dw, dh))."""
    shape = img.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, 64), np.mod(dh, 64)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = new_shape
        ratio = new_shape[0] / shape[1], new_shape[1] / shape[0]
    dw /= 2
    dh /= 2
    if shape[::-1] != new_unpad:
        img = resize_linear(img, new_unpad)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return copy_make_border(img, top, bottom, left, right, color), ratio, (dw, dh)


def load_image_resize(img, img_size, augment):
    """The resize of load_image (datasets.py:519-526): long side -> img_size; always shrink, enlarge only when augmenting;
    INTER_AREA when shrinking without augmentation, INTER_LINEAR otherwise."""
    h0, w0 = img.shape[:2]
    r = img_size / max(h0, w0)
    if r < 1 or (augment and r != 1):
        size = (int(w0 * r), int(h0 * r))
        return resize_area(img, size) if (r < 1 and not augment) else resize_linear(img, size)
    return _hwc(img)


# ---------------------------------------------------------------------------------------- cv2.warpAffine, INTER_LINEAR, 8U
AB_BITS, INTER_BITS = 10, 5
AB_SCALE, INTER_TAB_SIZE = 1 << AB_BITS, 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15


def invert_affine(M):
    """warpAffine without WARP_INVERSE_MAP inverts the 2x3 matrix in double precision, in this order (imgwarp.cpp)."""
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


def bilinear_tab():
    """BilinearTab_i (initInterTab2D, fixed point): w[fy * 32 + fx] = the four weights (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx
    as saturate_cast<short>(float product * 32768).  The float products are exact multiples of 2^-10, so the four always sum to
    32768 and the table's sum-repair step never fires."""
    t = np.arange(INTER_TAB_SIZE, dtype=np.float32) * np.float32(1.0 / INTER_TAB_SIZE)
    one = np.float32(1.0)
    vx = np.stack([one - t, t], 1)                                   # [32][2]
    w = (vx[:, None, :, None] * vx[None, :, None, :]).astype(np.float32)      # [fy][fx][ky][kx]
    tab = cv_round(w * np.float32(1 << INTER_REMAP_COEF_BITS)).reshape(INTER_TAB_SIZE * INTER_TAB_SIZE, 4)
    assert (tab.sum(1) == 1 << INTER_REMAP_COEF_BITS).all()
    return tab


_BILINEAR_TAB = bilinear_tab()


def warp_coordinates(Minv, dsize_wh):
    """WarpAffineInvoker: integer source coordinates and the 5 + 5 bit sub-pixel index per destination pixel."""
    w, h = int(dsize_wh[0]), int(dsize_wh[1])
    m = np.asarray(Minv, dtype=np.float64)
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    adelta = cv_round(m[0] * x * AB_SCALE)
    bdelta = cv_round(m[3] * x * AB_SCALE)
    X0 = cv_round((m[1] * y + m[2]) * AB_SCALE) + round_delta
    Y0 = cv_round((m[4] * y + m[5]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    alpha = (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))
    return sx, sy, alpha


def warp_affine(img, M, dsize_wh, border_value=(114, 114, 114)):
    """cv2.warpAffine(img, M[:2], dsize=(w, h), flags=cv2.INTER_LINEAR, borderValue=...) for uint8 (datasets.py:677), the
    fixed-point path of OpenCV 3.x - 4.10: remapBilinear<FixedPtCast<int, uchar, 15>, ..., short> with BORDER_CONSTANT:
    each of the four neighbours is the pixel if it lies inside the source, the border value otherwise, and
    dst = (sum_k v_k * w_k + 2^14) >> 15."""
    src = _hwc(img)
    H, W, c = src.shape
    sx, sy, alpha = warp_coordinates(invert_affine(M), dsize_wh)
    wt = _BILINEAR_TAB[alpha]                                      # (h, w, 4)
    cval = np.asarray(border_value, dtype=np.int64)[:c]
    S = src.astype(np.int64)
    acc = np.zeros(sx.shape + (c,), dtype=np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        xx, yy = sx + dx, sy + dy
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = np.where(ok[..., None], S[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], cval)
        acc += v * wt[..., k][..., None]
    return _saturate_u8((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS)


def rotation_matrix_2d(center_xy, angle_deg, scale):
    """cv2.getRotationMatrix2D (imgwarp.cpp): center is a Point2f."""
    a = angle_deg * (math.pi / 180)
    alpha, beta = scale * math.cos(a), scale * math.sin(a)
    cx, cy = float(np.float32(center_xy[0])), float(np.float32(center_xy[1]))
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


# ------------------------------------------------------------------------------ cv2.cvtColor BGR2HSV / HSV2BGR, 8U, + LUT
HSV_SHIFT = 12


def _div_tables():
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, dtype=np.int64)
    hdiv = np.zeros(256, dtype=np.int64)
    sdiv[1:] = cv_round((255 << HSV_SHIFT) / (1.0 * i))
    hdiv[1:] = cv_round((180 << HSV_SHIFT) / (6.0 * i))
    return sdiv, hdiv


_SDIV, _HDIV180 = _div_tables()


def to_hsv(img, order='bgr'):
    """cv2.cvtColor(img, cv2.COLOR_BGR2HSV) for uint8 (RGB2HSV_b, hrange 180; datasets.py:536): integer arithmetic with the
    12-bit reciprocal tables.  `order` names the channel order of `img` ('rgb' = COLOR_RGB2HSV); output is H, S, V."""
    x = _hwc(img).astype(np.int64)
    b, g, r = (x[..., 0], x[..., 1], x[..., 2]) if order == 'bgr' else (x[..., 2], x[..., 1], x[..., 0])
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * _SDIV[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))))
    h = (h * _HDIV180[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([_saturate_u8(h), s.astype(np.uint8), v.astype(np.uint8)], -1)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])     # tab index of b, g, r per sector


def from_hsv(hsv, order='bgr'):
    """cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR) for uint8 (HSV2RGB_b -> HSV2RGB_native, float32; datasets.py:546):
    h * (6.f / 180), s / 255, v / 255; sector = floor(h), tab = [v, v(1-s), v(1-s f), v(1-s(1-f))]; saturate_cast<uchar>(x * 255.f)."""
    x = _hwc(hsv)
    f32 = np.float32
    h = x[..., 0].astype(f32)
    s = (x[..., 1].astype(f32) * f32(1.0 / 255.0)).astype(f32)
    v = (x[..., 2].astype(f32) * f32(1.0 / 255.0)).astype(f32)
    h = (h * f32(6.0 / 180.0)).astype(f32)
    h = np.fmod(h, f32(6.0)).astype(f32)
    sector = np.floor(h).astype(np.int64)
    h = (h - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    h = np.where(bad, f32(0), h).astype(f32)
    one = f32(1.0)
    tab = np.stack([v, (v * (one - s)).astype(f32), (v * (one - (s * h).astype(f32))).astype(f32),
                    (v * (one - (s * (one - h)).astype(f32))).astype(f32)], -1)
    idx = _SECTOR[sector]                                                # (..., 3) for b, g, r
    bgr = np.take_along_axis(tab, idx, -1)
    gray = (x[..., 1] == 0)[..., None]                                   # s == 0 -> b = g = r = v
    bgr = np.where(gray, v[..., None], bgr).astype(f32)
    out = _saturate_u8(cv_round((bgr * f32(255.0)).astype(f32)))
    return out if order == 'bgr' else out[..., ::-1]


def hsv_luts(gains):
    """The three lookup tables of augment_hsv (datasets.py:539-542); gains = r of datasets.py:535."""
    x = np.arange(0, 256, dtype=np.int16)
    return (((x * gains[0]) % 180).astype(np.uint8), np.clip(x * gains[1], 0, 255).astype(np.uint8),
            np.clip(x * gains[2], 0, 255).astype(np.uint8))


def augment_hsv(img, gains, order='bgr'):
    """augment_hsv (datasets.py:534-546) for given random gains: returns the new image."""
    hsv = to_hsv(img, order)
    lh, ls, lv = hsv_luts(gains)
    return from_hsv(np.stack([lh[hsv[..., 0]], ls[hsv[..., 1]], lv[hsv[..., 2]]], -1), order)
