"""Image / label loading for detect.py, test.py and train.py — the reference's ``utils/datasets.py`` surface
(``LoadImages``, ``LoadImagesAndLabels`` + ``collate_fn``, ``letterbox``, ``augment_hsv``, ``random_affine``,
``load_mosaic``; reference lines cited per function) written on PIL + numpy: OpenCV is not part of this image.

Differences a caller can see: images are held in RGB order throughout (the reference holds BGR and flips at the very
end), so ``LoadImages`` yields ``im0`` as RGB; video files and camera / RTSP streams need OpenCV and raise here.
Tensors handed to the model are identical in layout and range: uint8 CHW RGB, normalised by the caller (``/ 256``).
"""
import glob
import math
import os
import random
from pathlib import Path

import numpy as np
import torch
from PIL import ExifTags, Image
from torch.utils.data import Dataset

from .utils import xyxy2xywh

help_url = 'https://github.com/ultralytics/yolov3/wiki/Train-Custom-Data'
img_formats = ['.bmp', '.jpg', '.jpeg', '.png', '.tif', '.tiff', '.dng']
vid_formats = ['.mov', '.avi', '.mp4', '.mpg', '.mpeg', '.m4v', '.wmv', '.mkv']
_ORIENTATION = next((k for k, v in ExifTags.TAGS.items() if v == 'Orientation'), None)
PAD_VALUE = 114


def exif_size(img):
    """(width, height) of a PIL image after its EXIF rotation (datasets.py:28-40)."""
    s = img.size
    try:
        rotation = dict(img._getexif().items())[_ORIENTATION]
        if rotation in (6, 8):
            s = (s[1], s[0])
    except Exception:
        pass
    return s


def _read(path, gray=False):
    """HWC uint8 array, RGB (or H x W x 1 for single-channel models)."""
    with Image.open(path) as im:
        arr = np.asarray(im.convert('L' if gray else 'RGB'))
    return arr[:, :, None] if gray else arr


def _resize(img, size_wh, area=False):
    gray = img.shape[2] == 1
    pil = Image.fromarray(img[:, :, 0] if gray else img)
    out = np.asarray(pil.resize(size_wh, Image.BOX if area else Image.BILINEAR))
    return out[:, :, None] if gray else out


def letterbox(img, new_shape=(416, 416), color=(PAD_VALUE,) * 3, auto=True, scaleFill=False, scaleup=True, is_gray_scale=False):
    """Aspect-preserving resize + constant border to ``new_shape`` (datasets.py:611-646).

    ``auto`` pads only to the next multiple of 64 (minimum rectangle), ``scaleFill`` stretches instead of padding,
    ``scaleup=False`` never enlarges.  Returns ``(image, (ratio_w, ratio_h), (pad_w, pad_h))`` with the padding per side."""
    h0, w0 = img.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / h0, new_shape[1] / w0)
    if not scaleup:
        r = min(r, 1.0)
    ratio = (r, r)
    unpad = (int(round(w0 * r)), int(round(h0 * r)))
    dw, dh = new_shape[1] - unpad[0], new_shape[0] - unpad[1]
    if auto:
        dw, dh = dw % 64, dh % 64
    elif scaleFill:
        dw, dh = 0.0, 0.0
        unpad = (new_shape[1], new_shape[0])
        ratio = (new_shape[1] / w0, new_shape[0] / h0)
    dw, dh = dw / 2, dh / 2
    if (w0, h0) != unpad:
        img = _resize(img, unpad)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    c = img.shape[2]
    out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, c), color[0] if c == 1 else 0, dtype=np.uint8)
    if c != 1:
        out[:] = np.asarray(color, dtype=np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, ratio, (dw, dh)


class LoadImages:
    """Iterate image files for inference (datasets.py:43-124): yields ``(path, chw_uint8_rgb, im0_hwc_rgb, None)``."""

    def __init__(self, path, img_size=416, is_gray_scale=False, rect=False, host_letterbox=True):
        path = str(Path(path))
        if os.path.isdir(path):
            files = sorted(glob.glob(os.path.join(path, '*.*')))
        elif os.path.isfile(path):
            files = [path]
        else:
            files = sorted(glob.glob(path))
        self.files = [f for f in files if os.path.splitext(f)[-1].lower() in img_formats]
        videos = [f for f in files if os.path.splitext(f)[-1].lower() in vid_formats]
        if videos:
            raise NotImplementedError('video input needs OpenCV, which this image does not ship: %s' % videos[0])
        assert self.files, 'No images found in %s. Supported formats: %s' % (path, img_formats)
        self.img_size, self.is_gray_scale, self.rect = img_size, is_gray_scale, rect
        self.host_letterbox = host_letterbox   # False: the caller letterboxes on the GPU (engine/preprocess.py); img is None
        self.nF, self.mode, self.cap = len(self.files), 'images', None

    def __iter__(self):
        self.count = 0
        return self

    def __next__(self):
        if self.count == self.nF:
            raise StopIteration
        path = self.files[self.count]
        self.count += 1
        im0 = _read(path, self.is_gray_scale)
        print('image %g/%g %s: ' % (self.count, self.nF, path), end='')
        if not self.host_letterbox:
            return path, None, im0, self.cap
        img = letterbox(im0, new_shape=self.img_size, auto=self.rect, is_gray_scale=self.is_gray_scale)[0]
        return path, np.ascontiguousarray(img.transpose(2, 0, 1)), im0, self.cap

    def __len__(self):
        return self.nF


class LoadWebcam:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError('camera input needs OpenCV, which this image does not ship')


class LoadStreams(LoadWebcam):
    pass


def load_image(self, index, is_gray_scale=False):
    """One dataset image resized so its long side is ``img_size`` (datasets.py:511-531): always shrinks, enlarges only
    when augmenting.  Returns ``(image, (h0, w0), (h, w))``."""
    img = self.imgs[index]
    if img is not None:
        return img, self.img_hw0[index], self.img_hw[index]
    img = _read(self.img_files[index], is_gray_scale)
    h0, w0 = img.shape[:2]
    r = self.img_size / max(h0, w0)
    if r < 1 or (self.augment and r != 1):
        img = _resize(img, (int(w0 * r), int(h0 * r)), area=r < 1 and not self.augment)
    return img, (h0, w0), img.shape[:2]


def load_image_lazy(self, index, is_gray_scale=False):
    """load_image with the resize left to the GPU (``arith='cv2'``): ``(original image, (h0, w0), (h, w) after cv2.resize, code)``
    where ``code`` is the OpenCV interpolation the reference would use (engine.imgtables.ARITH_CV2_*) or None for 'no resize'."""
    from engine import imgtables
    img = _read(self.img_files[index], is_gray_scale)
    h0, w0 = img.shape[:2]
    hw, code = imgtables.load_image_plan(h0, w0, self.img_size, self.augment)
    return img, (h0, w0), hw, code


def _rgb_to_hsv(img):
    x = img.astype(np.float32) / 255.0
    mx, mn = x.max(2), x.min(2)
    d = mx - mn
    h = np.zeros_like(mx)
    m = d > 0
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    i = m & (mx == r)
    h[i] = ((g - b)[i] / d[i]) % 6
    i = m & (mx == g) & ~(mx == r)
    h[i] = (b - r)[i] / d[i] + 2
    i = m & (mx == b) & ~(mx == r) & ~(mx == g)
    h[i] = (r - g)[i] / d[i] + 4
    s = np.where(mx > 0, d / np.maximum(mx, 1e-12), 0)
    return h / 6.0, s, mx


def _hsv_to_rgb(h, s, v):
    i = np.floor(h * 6.0)
    f = h * 6.0 - i
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    i = i.astype(np.int32) % 6
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return (np.stack((r, g, b), 2) * 255.0 + 0.5).clip(0, 255).astype(np.uint8)


def augment_hsv(img, hgain=0.5, sgain=0.5, vgain=0.5):
    """Random hue / saturation / value gains, in place (datasets.py:534-550)."""
    gains = np.random.uniform(-1, 1, 3) * [hgain, sgain, vgain] + 1
    h, s, v = _rgb_to_hsv(img)
    img[:] = _hsv_to_rgb((h * gains[0]) % 1.0, np.clip(s * gains[1], 0, 1), np.clip(v * gains[2], 0, 1))


def affine_matrix(img_hw, degrees=10, translate=.1, scale=.1, shear=10, border=0):
    """The random draws of ``random_affine`` (six ``random.uniform`` calls, in the reference's order, datasets.py:655-674):
    returns ``(M, s, (width, height))`` - the 3x3 forward matrix, the scale factor and the output size."""
    height, width = img_hw[0] + border * 2, img_hw[1] + border * 2
    a = math.radians(random.uniform(-degrees, degrees))
    s = random.uniform(1 - scale, 1 + scale)
    cx, cy = img_hw[1] / 2, img_hw[0] / 2
    R = np.array([[s * math.cos(a), s * math.sin(a), (1 - s * math.cos(a)) * cx - s * math.sin(a) * cy],
                  [-s * math.sin(a), s * math.cos(a), s * math.sin(a) * cx + (1 - s * math.cos(a)) * cy], [0, 0, 1]])
    T = np.eye(3)
    T[0, 2] = random.uniform(-translate, translate) * img_hw[1] + border
    T[1, 2] = random.uniform(-translate, translate) * img_hw[0] + border
    S = np.eye(3)
    S[0, 1] = math.tan(math.radians(random.uniform(-shear, shear)))
    S[1, 0] = math.tan(math.radians(random.uniform(-shear, shear)))
    return S @ T @ R, s, (width, height)


def affine_targets(targets, M, s, width, height):
    """Boxes follow their transformed corners; degenerate ones are dropped (datasets.py:688-715)."""
    n = len(targets)
    if n:
        xy = np.ones((n * 4, 3))
        xy[:, :2] = targets[:, [1, 2, 3, 4, 1, 4, 3, 2]].reshape(n * 4, 2)
        xy = (xy @ M.T)[:, :2].reshape(n, 8)
        x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
        box = np.stack((x.min(1), y.min(1), x.max(1), y.max(1)), 1)
        box[:, [0, 2]] = box[:, [0, 2]].clip(0, width)
        box[:, [1, 3]] = box[:, [1, 3]].clip(0, height)
        w, h = box[:, 2] - box[:, 0], box[:, 3] - box[:, 1]
        area0 = (targets[:, 3] - targets[:, 1]) * (targets[:, 4] - targets[:, 2])
        ar = np.maximum(w / (h + 1e-16), h / (w + 1e-16))
        keep = (w > 4) & (h > 4) & (w * h / (area0 * s + 1e-16) > 0.2) & (ar < 10)
        targets = targets[keep]
        targets[:, 1:5] = box[keep]
    return targets


def random_affine(img, targets=(), degrees=10, translate=.1, scale=.1, shear=10, border=0):
    """Random rotation / scale / translation / shear about the image centre, then crop ``border`` (negative = the mosaic's
    centre crop); boxes follow their transformed corners and degenerate ones are dropped (datasets.py:649-715)."""
    M, s, (width, height) = affine_matrix(img.shape[:2], degrees, translate, scale, shear, border)
    if border != 0 or (M != np.eye(3)).any():
        inv = np.linalg.inv(M)
        gray = img.shape[2] == 1
        pil = Image.fromarray(img[:, :, 0] if gray else img)
        fill = PAD_VALUE if gray else (PAD_VALUE,) * 3
        out = np.asarray(pil.transform((width, height), Image.AFFINE, tuple(inv[:2].reshape(-1)), Image.BILINEAR, fillcolor=fill))
        img = out[:, :, None] if gray else out
    return img, affine_targets(targets, M, s, width, height)


def mosaic_layout(self, index, is_gray_scale=False, lazy=False):
    """The random draws and the geometry of ``load_mosaic`` without touching a canvas (datasets.py:553-600): the mosaic centre,
    the three extra image indices, and per image the placement rectangle ``(x1a, y1a, x2a, y2a)`` on the 2s x 2s canvas, the
    matching source corner ``(x1b, y1b)``, the loaded image and its labels in canvas pixels.  ``lazy``: the images stay unresized
    and every part carries ``(h0, w0, h, w)`` as a fourth entry (the geometry only needs the size after the resize)."""
    s = self.img_size
    xc, yc = [int(random.uniform(s * 0.5, s * 1.5)) for _ in range(2)]
    indices = [index] + [random.randint(0, len(self.labels) - 1) for _ in range(3)]
    parts, labels4 = [], []
    for i, idx in enumerate(indices):
        if lazy:
            img, (h0, w0), (h, w), _ = load_image_lazy(self, idx, is_gray_scale)
        else:
            img, _, (h, w) = load_image(self, idx, is_gray_scale)
        if i == 0:    # top left of the centre
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
        elif i == 1:  # top right
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
            x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
        elif i == 2:  # bottom left
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
        else:         # bottom right
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
        parts.append((img, (x1a, y1a, x2a, y2a), (x1b, y1b, x2b, y2b)) + (((h0, w0, h, w),) if lazy else ()))
        padw, padh = x1a - x1b, y1a - y1b
        x = self.labels[idx]
        if x.size:
            lab = x.copy()
            lab[:, 1] = w * (x[:, 1] - x[:, 3] / 2) + padw
            lab[:, 2] = h * (x[:, 2] - x[:, 4] / 2) + padh
            lab[:, 3] = w * (x[:, 1] + x[:, 3] / 2) + padw
            lab[:, 4] = h * (x[:, 2] + x[:, 4] / 2) + padh
            labels4.append(lab)
    labels4 = np.concatenate(labels4, 0) if labels4 else np.zeros((0, 5), dtype=np.float32)
    if len(labels4):
        np.clip(labels4[:, 1:], 0, 2 * s, out=labels4[:, 1:])
    return parts, labels4


def load_mosaic(self, index, is_gray_scale=False):
    """Four images around a random centre of a 2s x 2s canvas, labels shifted with them, centre s x s kept
    (datasets.py:553-608)."""
    s = self.img_size
    parts, labels4 = mosaic_layout(self, index, is_gray_scale)
    c = 1 if is_gray_scale else 3
    canvas = np.full((s * 2, s * 2, c), PAD_VALUE, dtype=np.uint8)
    for img, (x1a, y1a, x2a, y2a), (x1b, y1b, x2b, y2b) in parts:
        canvas[y1a:y2a, x1a:x2a] = img[y1b:y2b, x1b:x2b]
    hyp = self.hyp or {}
    return random_affine(canvas, labels4, degrees=hyp.get('degrees', 0), translate=hyp.get('translate', 0),
                         scale=hyp.get('scale', 0), shear=hyp.get('shear', 0), border=-s // 2)


class MosaicItem:
    """One training item with the pixel work left undone: everything ``yh_mosaic_affine_hsv`` needs (csrc/augment.hip) plus the
    finished labels.  Produced by ``LoadImagesAndLabels.__getitem__`` when ``device_augment`` is set; consumes the random streams
    exactly like the host path, so the same seeds give the same item."""
    __slots__ = ('parts', 'canvas', 'out_hw', 'inv', 'identity', 'hsv_gains', 'flip', 'channels', 'labels', 'path', 'arith')


class MosaicBatch:
    """A collated batch of recipes: every source crop of the batch in ONE uint8 tensor (built in the loader worker, pinned by the
    DataLoader's pin-memory thread through ``pin_memory()``, uploaded with one copy) plus the per-item geometry."""

    def __init__(self, items):
        import copy
        self.items = [copy.copy(it) for it in items]      # the caller's items keep their crops
        sizes = [(0 if p[0] is None else p[0].size) for it in self.items for p in it.parts]
        self.offsets = np.concatenate([[0], np.cumsum([(n + 15) // 16 * 16 for n in sizes])]).astype(np.int64)
        self.blob = torch.empty(int(self.offsets[-1]) + 16, dtype=torch.uint8)
        flat = self.blob.numpy()
        k = 0
        for it in self.items:
            parts = []
            for part in it.parts:
                crop = part[0]
                if crop is not None:
                    flat[self.offsets[k]:self.offsets[k] + crop.size] = crop.reshape(-1)
                parts.append((None if crop is None else tuple(crop.shape[:2]),) + tuple(part[1:]))
                k += 1
            it.parts = parts          # shapes only: the pixels live in the blob

    def pin_memory(self):
        self.blob = self.blob.pin_memory()
        return self

    def __len__(self):
        return len(self.items)


def _letterbox_plan(h, w, new_shape, scaleup):
    """The numbers ``letterbox(img, new_shape, auto=False, scaleup=...)`` derives from an h x w image, with the expressions (and
    therefore the numpy / Python scalar types, which decide whether the label arithmetic below runs in float32 or float64) of
    ``letterbox`` itself: ``(r, (new_w, new_h), (dw, dh), (top, bottom, left, right))``."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / h, new_shape[1] / w)
    if not scaleup:
        r = min(r, 1.0)
    unpad = (int(round(w * r)), int(round(h * r)))
    dw, dh = new_shape[1] - unpad[0], new_shape[0] - unpad[1]
    dw, dh = dw / 2, dh / 2
    return r, unpad, (dw, dh), (int(round(dh - 0.1)), int(round(dh + 0.1)), int(round(dw - 0.1)), int(round(dw + 0.1)))


class LetterboxItem:
    """One evaluation / rect item with the pixel work left undone (``device_letterbox``): the decoded frame, the size load_image
    would resize it to and with which filter, and where letterbox would put it in the batch rectangle - what ``yh_letterbox_fwd``
    needs - plus the finished labels and the ``shapes`` tuple test.py rescales boxes with."""
    __slots__ = ('frame', 'resized_hw', 'code', 'out_hw', 'top', 'left', 'channels', 'labels', 'path', 'shapes', 'arith')


class LetterboxBatch:
    """A collated batch of ``LetterboxItem``: every frame in ONE uint8 tensor (pinned by the DataLoader, one upload)."""

    def __init__(self, items):
        import copy
        self.items = [copy.copy(it) for it in items]
        sizes = [it.frame.size for it in self.items]
        self.offsets = np.concatenate([[0], np.cumsum([(n + 15) // 16 * 16 for n in sizes])]).astype(np.int64)
        self.blob = torch.empty(int(self.offsets[-1]) + 16, dtype=torch.uint8)
        flat = self.blob.numpy()
        for k, it in enumerate(self.items):
            flat[self.offsets[k]:self.offsets[k] + it.frame.size] = it.frame.reshape(-1)
            it.frame = tuple(it.frame.shape[:2])       # shape only: the pixels live in the blob

    def pin_memory(self):
        self.blob = self.blob.pin_memory()
        return self

    def __len__(self):
        return len(self.items)


def letterbox_item(self, index):
    """``__getitem__`` of the non-augmenting path (datasets.py:480-505: load_image + letterbox + label shift) as a recipe."""
    from engine import imgtables
    it = LetterboxItem()
    it.arith, it.channels = self.arith, 1 if self.is_gray_scale else 3
    img = _read(self.img_files[index], self.is_gray_scale)
    h0, w0 = img.shape[:2]
    (h, w), code = imgtables.load_image_plan(h0, w0, self.img_size, self.augment)
    shape = self.batch_shapes[self.batch[index]] if self.rect else self.img_size
    # letterbox(img, shape, auto=False, scaleup=False): never enlarges; an image larger than its rectangle would be resized a
    # second time - the rect batch shapes are built so that this cannot happen
    r, unpad, (dw, dh), (top, bottom, left, right) = _letterbox_plan(h, w, shape, scaleup=False)
    if unpad != (w, h):
        raise NotImplementedError('%s: %dx%d does not fit its batch rectangle %s after load_image; a second resize inside letterbox '
                                  'is not built for device_letterbox' % (self.img_files[index], w, h, tuple(shape)))
    it.frame, it.resized_hw, it.code = img, (h, w), code
    it.out_hw, it.top, it.left = (h + top + bottom, w + left + right), top, left
    it.shapes = (h0, w0), ((h / h0, w / w0), (dw, dh))
    x = self.labels[index]
    labels = np.zeros((0, 5), dtype=np.float32)
    if x.size:
        labels = x.copy()
        labels[:, 1] = r * w * (x[:, 1] - x[:, 3] / 2) + dw
        labels[:, 2] = r * h * (x[:, 2] - x[:, 4] / 2) + dh
        labels[:, 3] = r * w * (x[:, 1] + x[:, 3] / 2) + dw
        labels[:, 4] = r * h * (x[:, 2] + x[:, 4] / 2) + dh
    n_l = len(labels)
    if n_l:
        labels[:, 1:5] = xyxy2xywh(labels[:, 1:5])
        labels[:, [2, 4]] /= it.out_hw[0]
        labels[:, [1, 3]] /= it.out_hw[1]
    it.labels = torch.zeros((n_l, 6))
    if n_l:
        it.labels[:, 1:] = torch.from_numpy(np.ascontiguousarray(labels))
    it.path = self.img_files[index]
    return it


def mosaic_item(self, index):
    """``__getitem__`` of the augmenting mosaic path (datasets.py:470-505) with the image left as a recipe."""
    hyp = self.hyp or {}
    s = self.img_size
    it = MosaicItem()
    it.channels = 1 if self.is_gray_scale else 3
    it.arith = self.arith
    cv2_arith = self.arith == 'cv2'
    parts, labels4 = mosaic_layout(self, index, self.is_gray_scale, lazy=cv2_arith)
    M, sc, (width, height) = affine_matrix((2 * s, 2 * s), hyp.get('degrees', 0), hyp.get('translate', 0), hyp.get('scale', 0),
                                           hyp.get('shear', 0), border=-s // 2)
    labels = affine_targets(labels4, M, sc, width, height)
    if cv2_arith:     # cv2.warpAffine inverts M itself, with its own sequence of operations
        from engine import imgtables
        it.inv = imgtables.cv2_invert_affine(M[:2])
    else:
        it.inv = np.linalg.inv(M)[:2].reshape(-1).astype(np.float64)
    it.canvas, it.out_hw = (2 * s, 2 * s), (height, width)
    it.hsv_gains = None
    if not self.is_gray_scale:   # augment_hsv: one np.random.uniform(-1, 1, 3)
        it.hsv_gains = np.random.uniform(-1, 1, 3) * [hyp.get('hsv_h', 0), hyp.get('hsv_s', 0), hyp.get('hsv_v', 0)] + 1
    n_l = len(labels)
    if n_l:
        labels[:, 1:5] = xyxy2xywh(labels[:, 1:5])
        labels[:, [2, 4]] /= height
        labels[:, [1, 3]] /= width
    it.flip = random.random() < 0.5
    if it.flip and n_l:
        labels[:, 1] = 1 - labels[:, 1]
    it.labels = torch.zeros((n_l, 6))
    if n_l:
        it.labels[:, 1:] = torch.from_numpy(np.ascontiguousarray(labels))
    it.path = self.img_files[index]
    # ship only the pixels the warp can touch: the footprint of the output rectangle on the canvas (+ the bilinear neighbour)
    inv = np.vstack([it.inv.reshape(2, 3), [0, 0, 1]])
    corners = inv @ np.array([[0, width, 0, width], [0, 0, height, height], [1, 1, 1, 1]], dtype=np.float64)
    fx0, fx1 = int(math.floor(corners[0].min())) - 2, int(math.ceil(corners[0].max())) + 2
    fy0, fy1 = int(math.floor(corners[1].min())) - 2, int(math.ceil(corners[1].max())) + 2
    it.parts = []
    for part in parts:
        img, (x1a, y1a, x2a, y2a), (x1b, y1b, x2b, y2b) = part[:3]
        cx0, cy0, cx1, cy1 = max(x1a, fx0), max(y1a, fy0), min(x2a, fx1), min(y2a, fy1)
        if cx1 <= cx0 or cy1 <= cy0:
            it.parts.append((None, (0, 0, 0, 0), (0, 0)) + ((None,) if cv2_arith else ()))
            continue
        wx0, wy0, wx1, wy1 = x1b + cx0 - x1a, y1b + cy0 - y1a, x1b + cx1 - x1a, y1b + cy1 - y1a    # window of the resized image
        if not cv2_arith:
            it.parts.append((np.ascontiguousarray(img[wy0:wy1, wx0:wx1]), (cx0, cy0, cx1, cy1), (0, 0)))
            continue
        # the resize happens on the GPU: ship the source samples this window of the resized image reads
        h0, w0, h, w = part[3]
        ix = _cv2_axis_index(w0, w, True)[wx0:wx1]
        iy = _cv2_axis_index(h0, h, False)[wy0:wy1]
        c0, c1, r0, r1 = int(ix.min()), int(ix.max()) + 1, int(iy.min()), int(iy.max()) + 1
        it.parts.append((np.ascontiguousarray(img[r0:r1, c0:c1]), (cx0, cy0, cx1, cy1), (0, 0), (h0, w0, h, w, wx0, wy0, c0, r0)))
    return it


def rect_train_item(self, index):
    """``__getitem__`` of the augmenting NON-mosaic path (rect training: datasets.py:480-505 with random_affine / augment_hsv / flip
    of the single letterboxed image) as a recipe for the same kernel: a mosaic with one part on a canvas the size of the batch
    rectangle, border 0.  Random draws in the host path's order (six ``random.uniform``, one ``np.random.uniform(-1, 1, 3)``, one
    ``random.random``)."""
    hyp = self.hyp or {}
    it = MosaicItem()
    it.channels, it.arith = 1 if self.is_gray_scale else 3, self.arith
    cv2_arith = self.arith == 'cv2'
    if cv2_arith:
        img, (h0, w0), (h, w), _ = load_image_lazy(self, index, self.is_gray_scale)
    else:
        img, (h0, w0), (h, w) = load_image(self, index, self.is_gray_scale)
    shape = self.batch_shapes[self.batch[index]] if self.rect else self.img_size
    r, unpad, (dw, dh), (top, bottom, left, right) = _letterbox_plan(h, w, shape, scaleup=True)    # letterbox(..., auto=False, scaleup=True)
    if unpad != (w, h):
        raise NotImplementedError('%s: letterbox would resize the %dx%d image a second time for the rectangle %s; not built for '
                                  'device_augment - train with --host-augment (or without --device-augment) for this data set'
                                  % (self.img_files[index], w, h, tuple(shape)))
    canvas = (h + top + bottom, w + left + right)
    x = self.labels[index]
    labels = np.zeros((0, 5), dtype=np.float32)
    if x.size:
        labels = x.copy()
        labels[:, 1] = r * w * (x[:, 1] - x[:, 3] / 2) + dw
        labels[:, 2] = r * h * (x[:, 2] - x[:, 4] / 2) + dh
        labels[:, 3] = r * w * (x[:, 1] + x[:, 3] / 2) + dw
        labels[:, 4] = r * h * (x[:, 2] + x[:, 4] / 2) + dh
    M, sc, (width, height) = affine_matrix(canvas, hyp.get('degrees', 0), hyp.get('translate', 0), hyp.get('scale', 0), hyp.get('shear', 0))
    labels = affine_targets(labels, M, sc, width, height)
    if not (M != np.eye(3)).any():                          # random_affine leaves the image alone; the unit map is exact in both arithmetics
        it.inv = np.array([1.0, 0.0, 0.0, 0.0, 1.0, 0.0])
    elif cv2_arith:
        from engine import imgtables
        it.inv = imgtables.cv2_invert_affine(M[:2])
    else:
        it.inv = np.linalg.inv(M)[:2].reshape(-1).astype(np.float64)
    it.canvas, it.out_hw = canvas, (height, width)
    it.hsv_gains = None
    if not self.is_gray_scale:
        it.hsv_gains = np.random.uniform(-1, 1, 3) * [hyp.get('hsv_h', 0), hyp.get('hsv_s', 0), hyp.get('hsv_v', 0)] + 1
    n_l = len(labels)
    if n_l:
        labels[:, 1:5] = xyxy2xywh(labels[:, 1:5])
        labels[:, [2, 4]] /= height
        labels[:, [1, 3]] /= width
    it.flip = random.random() < 0.5
    if it.flip and n_l:
        labels[:, 1] = 1 - labels[:, 1]
    it.labels = torch.zeros((n_l, 6))
    if n_l:
        it.labels[:, 1:] = torch.from_numpy(np.ascontiguousarray(labels))
    it.path = self.img_files[index]
    rect = (left, top, left + w, top + h)
    empty = (None, (0, 0, 0, 0), (0, 0)) + ((None,) if cv2_arith else ())
    if cv2_arith:
        it.parts = [(np.ascontiguousarray(img), rect, (0, 0), (h0, w0, h, w, 0, 0, 0, 0))] + [empty] * 3
    else:
        it.parts = [(np.ascontiguousarray(img), rect, (0, 0))] + [empty] * 3
    return it


_axis_index_cache = {}


def _cv2_axis_index(ssize, dsize, horizontal):
    key = (ssize, dsize, horizontal)
    hit = _axis_index_cache.get(key)
    if hit is None:
        from engine import imgtables
        if len(_axis_index_cache) > 256:
            _axis_index_cache.clear()
        hit = _axis_index_cache[key] = imgtables.cv2_linear_tables(ssize, dsize, horizontal)[0]
    return hit


class LoadImagesAndLabels(Dataset):
    """Training / evaluation dataset (datasets.py:265-508).

    ``path`` is a text file of image paths (or a directory of images); the label file of ``.../images/x.jpg`` is
    ``.../labels/x.txt`` with rows ``class x y w h`` (normalised centre format).  ``rect`` sorts by aspect ratio and
    letterboxes every batch to its own minimal multiple-of-32 rectangle (evaluation); training uses the 4-image mosaic
    unless ``rect``.  Items are ``(uint8 CHW image, labels (n, 6) = [0, class, x, y, w, h], path, shapes)``;
    ``collate_fn`` writes the image index into column 0, the format ``build_targets`` expects."""

    def __init__(self, path, img_size=416, batch_size=16, augment=False, hyp=None, rect=False, image_weights=False,
                 cache_images=False, rank=-1, is_gray_scale=False, subset_len=-1, single_cls=False, pad=0.0, device_augment=False,
                 arith=None, device_letterbox=False):
        path = str(Path(path))
        if os.path.isdir(path):
            files = sorted(glob.glob(os.path.join(path, '*.*')))
        else:
            assert os.path.isfile(path), 'File not found %s. See %s' % (path, help_url)
            with open(path) as f:
                files = [x.replace('/', os.sep) for x in f.read().splitlines()]
        self.img_files = [x for x in files if os.path.splitext(x)[-1].lower() in img_formats]
        if subset_len != -1:
            assert subset_len <= len(self.img_files)
            self.img_files = random.sample(self.img_files, subset_len)
        n = len(self.img_files)
        assert n > 0, 'No images found in %s. See %s' % (path, help_url)
        bi = np.floor(np.arange(n) / batch_size).astype(np.int64)
        nb = int(bi[-1]) + 1
        self.n, self.batch, self.img_size = n, bi, img_size
        self.augment, self.hyp, self.image_weights = augment, hyp, image_weights
        self.rect = False if image_weights else rect
        self.mosaic = self.augment and not self.rect
        # the mosaic / warp / HSV / flip pixel work on the GPU (engine/preprocess.py render_mosaic_items): items are recipes
        self.device_augment = bool(device_augment) and self.mosaic
        # rect training (augment without mosaic): the same kernel renders the single letterboxed image - a one-part "mosaic"
        self.device_rect_augment = bool(device_augment) and self.augment and not self.mosaic
        # 'cv2': the reference's OpenCV arithmetic (resize, warp, HSV) - evaluated by the GPU kernels only, so only for recipes
        self.arith = arith or os.environ.get('YOLO_IMAGE_ARITH', 'pillow')
        if self.arith not in ('pillow', 'cv2'):
            raise ValueError("arith must be 'pillow' or 'cv2', got %r" % (self.arith,))
        # evaluation / rect items as recipes: resize (load_image) + border (letterbox) + /256 + HWC->CHW on the GPU
        # (engine/preprocess.py render_letterbox_items); only without augmentation - rect TRAINING warps single images on the host
        self.device_letterbox = bool(device_letterbox) and not self.mosaic and not self.augment
        if self.arith == 'cv2' and not (self.device_augment or self.device_rect_augment or self.device_letterbox):
            raise NotImplementedError("arith='cv2' (OpenCV's arithmetic) exists on the GPU only: it needs device_augment=True (training) "
                                      "or device_letterbox=True (evaluation); this package's host loader resizes with Pillow")
        if (self.arith == 'cv2' or self.device_letterbox) and cache_images:
            raise NotImplementedError('the device recipes keep the source frames unresized; cache_images is not supported with them')
        self.is_gray_scale = is_gray_scale
        self.label_files = [x.replace('images', 'labels').replace(os.path.splitext(x)[-1], '.txt') for x in self.img_files]

        if self.rect:
            shapes = np.array([exif_size(Image.open(f)) for f in self.img_files], dtype=np.float64)   # (w, h)
            ar = shapes[:, 1] / shapes[:, 0]
            order = ar.argsort()
            self.img_files = [self.img_files[i] for i in order]
            self.label_files = [self.label_files[i] for i in order]
            self.shapes, ar = shapes[order], ar[order]
            batch_hw = [[1, 1]] * nb
            for i in range(nb):
                ari = ar[bi == i]
                if ari.max() < 1:
                    batch_hw[i] = [ari.max(), 1]
                elif ari.min() > 1:
                    batch_hw[i] = [1, 1 / ari.min()]
            self.batch_shapes = (np.ceil(np.array(batch_hw) * img_size / 32. + pad).astype(np.int64) * 32)

        self.imgs = [None] * n
        self.labels = [np.zeros((0, 5), dtype=np.float32)] * n
        nm = nf = ne = nd = 0
        for i, file in enumerate(self.label_files):
            try:
                with open(file) as f:
                    rows = np.array([x.split() for x in f.read().splitlines() if x.strip()], dtype=np.float32)
            except Exception:
                nm += 1   # missing label file
                continue
            if rows.shape[0]:
                assert rows.shape[1] == 5, '> 5 label columns: %s' % file
                assert (rows >= 0).all(), 'negative labels: %s' % file
                assert (rows[:, 1:] <= 1).all(), 'non-normalized or out of bounds coordinate labels: %s' % file
                if np.unique(rows, axis=0).shape[0] < rows.shape[0]:
                    nd += 1
                if single_cls:
                    rows[:, 0] = 0
                self.labels[i] = rows
                nf += 1
            else:
                ne += 1   # empty label file
        if rank in (-1, 0):
            print('Caching labels (%g found, %g missing, %g empty, %g duplicate, for %g images)' % (nf, nm, ne, nd, n))
        assert nf > 0 or n == 0 or not augment, 'No labels found in %s. See %s' % (os.path.dirname(self.label_files[0]), help_url)

        if cache_images:
            self.img_hw0, self.img_hw = [None] * n, [None] * n
            for i in range(n):
                self.imgs[i], self.img_hw0[i], self.img_hw[i] = load_image(self, i, is_gray_scale)

    def __len__(self):
        return len(self.img_files)

    def __getitem__(self, index):
        if self.image_weights:
            index = self.indices[index]
        hyp = self.hyp or {}
        if self.device_augment:
            return mosaic_item(self, index)
        if self.device_rect_augment:
            return rect_train_item(self, index)
        if self.device_letterbox:
            return letterbox_item(self, index)
        if self.mosaic:
            img, labels = load_mosaic(self, index, self.is_gray_scale)
            shapes = None
        else:
            img, (h0, w0), (h, w) = load_image(self, index, self.is_gray_scale)
            shape = self.batch_shapes[self.batch[index]] if self.rect else self.img_size
            img, ratio, pad = letterbox(img, shape, auto=False, scaleup=self.augment, is_gray_scale=self.is_gray_scale)
            shapes = (h0, w0), ((h / h0, w / w0), pad)
            x = self.labels[index]
            labels = np.zeros((0, 5), dtype=np.float32)
            if x.size:
                labels = x.copy()
                labels[:, 1] = ratio[0] * w * (x[:, 1] - x[:, 3] / 2) + pad[0]
                labels[:, 2] = ratio[1] * h * (x[:, 2] - x[:, 4] / 2) + pad[1]
                labels[:, 3] = ratio[0] * w * (x[:, 1] + x[:, 3] / 2) + pad[0]
                labels[:, 4] = ratio[1] * h * (x[:, 2] + x[:, 4] / 2) + pad[1]
        if self.augment:
            if not self.mosaic:
                img, labels = random_affine(img, labels, degrees=hyp.get('degrees', 0), translate=hyp.get('translate', 0),
                                            scale=hyp.get('scale', 0), shear=hyp.get('shear', 0))
            if not self.is_gray_scale:
                img = np.array(img)   # own, writable copy (PIL-backed arrays are read-only)
                augment_hsv(img, hgain=hyp.get('hsv_h', 0), sgain=hyp.get('hsv_s', 0), vgain=hyp.get('hsv_v', 0))
        n_l = len(labels)
        if n_l:
            labels[:, 1:5] = xyxy2xywh(labels[:, 1:5])
            labels[:, [2, 4]] /= img.shape[0]
            labels[:, [1, 3]] /= img.shape[1]
        if self.augment and random.random() < 0.5:   # left-right flip
            img = np.fliplr(img)
            if n_l:
                labels[:, 1] = 1 - labels[:, 1]
        labels_out = torch.zeros((n_l, 6))
        if n_l:
            labels_out[:, 1:] = torch.from_numpy(np.ascontiguousarray(labels))
        img = np.ascontiguousarray(img.transpose(2, 0, 1))
        return torch.from_numpy(img), labels_out, self.img_files[index], shapes

    @staticmethod
    def collate_fn(batch):
        if batch and isinstance(batch[0], MosaicItem):   # device_augment: (MosaicBatch, labels, paths, shapes); the train loop renders
            for i, it in enumerate(batch):
                it.labels[:, 0] = i
            return MosaicBatch(batch), torch.cat([it.labels for it in batch], 0), tuple(it.path for it in batch), (None,) * len(batch)
        if batch and isinstance(batch[0], LetterboxItem):  # device_letterbox: (LetterboxBatch, labels, paths, shapes)
            for i, it in enumerate(batch):
                it.labels[:, 0] = i
            return (LetterboxBatch(batch), torch.cat([it.labels for it in batch], 0), tuple(it.path for it in batch),
                    tuple(it.shapes for it in batch))
        img, label, path, shapes = zip(*batch)
        for i, l in enumerate(label):
            l[:, 0] = i   # image index within the batch, for build_targets()
        return torch.stack(img, 0), torch.cat(label, 0), path, shapes


def create_folder(path='./new_folder'):
    import shutil
    if os.path.exists(path):
        shutil.rmtree(path)
    os.makedirs(path)
