"""Input pipeline in the REFERENCE's arithmetic: OpenCV's uint8 resize / warpAffine / HSV formulas.

OpenCV is not installed here and the reference holds no golden images, so nothing in this file is pinned to the library itself
("third-party restated", oracle/cv2_restated.py header).  What is pinned:

CPU tier
  * the restatement against properties OpenCV's algorithms have by construction (identity, exact 2x decimation = 2x2 mean, constants,
    a float bilinear within the fixed-point error bound, integer translations of warpAffine, HSV of pure colours);
  * the host-side tables (engine/imgtables.py) + the descriptor plumbing of engine/preprocess.py, run through the host emulation
    of the C ABI (tests/fakelib.py, formula for formula from the descriptors), against the restatement computed from image sizes
    alone - letterbox, cv2.resize in its three arithmetics, and whole mosaic training items (windowed resize of four crops, warp,
    HSV tables, flip), with the random streams consumed exactly like the Pillow path.
GPU tier
  * the HIP kernels (csrc/preprocess.hip, csrc/augment.hip with arith = cv2) against the restatement, bit for bit.
"""
import random

import numpy as np
import pytest
import torch

import conftest  # noqa: F401
import fakelib
from engine import imgtables, preprocess as pp
from oracle import cv2_restated as cv
from utils import datasets


def _img(shape, seed, smooth=False):
    rs = np.random.RandomState(seed)
    if not smooth:
        return rs.randint(0, 256, size=shape, dtype=np.uint8)
    h, w, c = shape
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 90 * np.sin(xx / 7.0 + seed)[..., None] * np.cos(yy / 5.0)[..., None] + rs.randint(-20, 20, size=shape)
    return np.clip(base, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------- the restatement's own properties
def test_linear_resize_properties():
    img = _img((60, 80, 3), 0)
    assert np.array_equal(cv.resize_linear(img, (80, 60)), img)
    S = img.astype(int)
    mean2 = (S[0::2, 0::2] + S[0::2, 1::2] + S[1::2, 0::2] + S[1::2, 1::2] + 2) >> 2
    assert np.array_equal(cv.resize_linear(img, (40, 30)), mean2)           # resize() reroutes exact 2x to the 2x2 area mean
    assert np.array_equal(cv.resize_area(img, (40, 30)), mean2)
    const = np.full((33, 47, 3), 201, np.uint8)
    for size in [(31, 17), (100, 70), (47, 20)]:
        assert (cv.resize_linear(const, size) == 201).all()
    assert (cv.resize_area(const, (31, 17)) == 201).all() and (cv.resize_area(np.full((36, 48, 1), 9, np.uint8), (16, 12)) == 9).all()
    # against a float bilinear with OpenCV's sample positions: 11-bit weights and the >>4 / >>16 / >>2 chain stay within one level
    for w, h in [(100, 75), (53, 41), (80, 30), (161, 60)]:
        h0, w0 = img.shape[:2]
        x, y = (np.arange(w) + .5) * (w0 / w) - .5, (np.arange(h) + .5) * (h0 / h) - .5
        x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
        fx, fy = (x - x0)[None, :, None], (y - y0)[:, None, None]
        xa, xb, ya, yb = np.clip(x0, 0, w0 - 1), np.clip(x0 + 1, 0, w0 - 1), np.clip(y0, 0, h0 - 1), np.clip(y0 + 1, 0, h0 - 1)
        fx = np.where(((x0 < 0) | (x0 >= w0 - 1))[None, :, None], 0, fx)
        I = img.astype(float)
        want = (I[ya][:, xa] * (1 - fx) + I[ya][:, xb] * fx) * (1 - fy) + (I[yb][:, xa] * (1 - fx) + I[yb][:, xb] * fx) * fy
        assert np.abs(cv.resize_linear(img, (w, h)) - want).max() < 1.0


def test_area_resize_is_a_weighted_mean():
    img = _img((90, 120, 3), 1)
    for w, h in [(53, 41), (100, 80), (40, 30), (30, 30)]:          # general, general, 3x3 fast, 4x3 fast
        out = cv.resize_area(img, (w, h))
        assert out.shape == (h, w, 3) and abs(out.mean() - img.mean()) < 0.6
    big = np.kron(_img((10, 12, 3), 2), np.ones((3, 3, 1), np.uint8))      # 3x3 blocks of equal pixels shrink to the blocks' values
    assert np.array_equal(cv.resize_area(big, (12, 10)), _img((10, 12, 3), 2))


def test_area_fast_and_table_forms_agree_on_integer_factors():
    """resize() sends integer decimation factors to resizeAreaFast_ (integer cell sums) and everything else to the float-table form; on
    an integer factor the table form computes the same means, so the two restatements must agree - exactly for 3x3 and 4x4 cells, within
    one level for 2x2 ((s + 2) >> 2 rounds halves up, the float form rounds them to even)."""
    img = _img((96, 120, 3), 3)
    for (w, h), tol in [((60, 48), 1), ((40, 32), 0), ((30, 24), 0)]:
        fast = cv.resize_area(img, (w, h))
        xt, yt = cv._area_tab(120, w, 120 / w), cv._area_tab(96, h, 96 / h)
        S = img.astype(np.float32)
        rows = np.zeros((96, w, 3), np.float32)
        for dx, sx, alpha in xt:
            rows[:, dx] = rows[:, dx] + S[:, sx] * alpha
        table = np.zeros((h, w, 3), np.float32)
        for dy, sy, beta in yt:
            table[dy] = table[dy] + beta * rows[sy]
        d = np.abs(fast.astype(int) - np.clip(np.rint(table), 0, 255).astype(int))
        assert d.max() <= tol, ((w, h), int(d.max()))


def test_warp_affine_properties():
    img = _img((60, 80, 3), 3)
    eye = np.array([[1, 0, 0], [0, 1, 0]], float)
    assert np.array_equal(cv.warp_affine(img, eye, (80, 60)), img)
    shift = np.array([[1, 0, -10], [0, 1, -5]], float)
    assert np.array_equal(cv.warp_affine(img, shift, (60, 50)), img[5:55, 10:70])
    out = cv.warp_affine(img, np.array([[1, 0, 7], [0, 1, 3]], float), (80, 60))
    assert (out[:3] == 114).all() and (out[:, :7] == 114).all() and np.array_equal(out[3:, 7:], img[:57, :73])
    half = cv.warp_affine(img, np.array([[1, 0, 0.5], [0, 1, 0]], float), (80, 60))       # half-pixel shift = mean of neighbours
    want = (img[:, :-1].astype(int) + img[:, 1:] + 1) >> 1
    assert np.abs(half[:, 1:].astype(int) - want).max() <= 1
    assert (cv.bilinear_tab().sum(1) == 32768).all()
    M = cv.rotation_matrix_2d((40, 30), 0.0, 1.0)
    assert np.allclose(M, eye)
    Minv = cv.invert_affine(np.array([[1.1, 0.05, -10.3], [-0.04, 0.95, 4.2]]))
    assert np.allclose(np.vstack([Minv.reshape(2, 3), [0, 0, 1]]) @ np.array([[1.1, 0.05, -10.3], [-0.04, 0.95, 4.2], [0, 0, 1]]), np.eye(3))


def test_hsv_of_pure_colours_and_round_trip():
    bgr = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128], [0, 255, 255], [255, 255, 0]]], np.uint8)
    hsv = cv.to_hsv(bgr)
    assert hsv[0].tolist() == [[120, 255, 255], [60, 255, 255], [0, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128], [30, 255, 255], [90, 255, 255]]
    assert np.array_equal(cv.from_hsv(hsv), bgr)
    img = _img((40, 50, 3), 4)
    assert np.array_equal(cv.to_hsv(img[..., ::-1], 'rgb'), cv.to_hsv(img))
    back = cv.from_hsv(cv.to_hsv(img))
    assert np.abs(back.astype(int) - img).max() <= 6                      # 8-bit HSV is lossy, but only by a few levels
    assert np.array_equal(cv.augment_hsv(img, [1.0, 1.0, 1.0]), back)     # unit gains: the tables are identities
    lh, ls, lv = cv.hsv_luts([1.3, 0.5, 2.0])
    assert lh[179] == int(179 * 1.3) % 180 and ls[255] == 127 and lv[200] == 255


# -------------------------------------------------------------------------- host tables + descriptors, through the ABI emulation
@pytest.fixture
def fake(monkeypatch):
    lib = fakelib.FakeLib()
    monkeypatch.setattr(pp.hiplib, 'load', lambda: lib)
    return lib


RESIZE_CASES = [((60, 80, 3), (100, 75)), ((60, 80, 3), (53, 41)), ((37, 53, 3), (111, 159)), ((64, 64, 1), (64, 20)), ((120, 160, 3), (80, 60)),
                ((33, 47, 3), (47, 20)), ((50, 70, 3), (1, 1)), ((2, 2, 3), (7, 5))]


@pytest.mark.parametrize('shape,size', RESIZE_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_linear_tables_reproduce_the_restatement(fake, shape, size):
    img = _img(shape, 5)
    got = pp.resize_to_device(img, (size[1], size[0]), 'cpu', imgtables.ARITH_CV2_LINEAR, lib=fake)
    assert np.array_equal(got.numpy(), cv.resize_linear(img, size))


@pytest.mark.parametrize('shape,size', [((60, 80, 3), (53, 41)), ((97, 131, 3), (64, 47)), ((64, 64, 1), (40, 40)), ((90, 120, 3), (119, 89))],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_area_tables_reproduce_the_restatement(fake, shape, size):
    img = _img(shape, 6)
    assert imgtables.cv2_area_is_fast(shape[1::-1], size) is None
    got = pp.resize_to_device(img, (size[1], size[0]), 'cpu', imgtables.ARITH_CV2_AREA, lib=fake)
    assert np.array_equal(got.numpy(), cv.resize_area(img, size))


@pytest.mark.parametrize('shape,size', [((60, 80, 3), (40, 30)), ((90, 120, 3), (40, 30)), ((90, 120, 1), (30, 30)), ((64, 128, 3), (32, 16))],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_area_fast_reproduces_the_restatement(fake, shape, size):
    img = _img(shape, 7)
    assert imgtables.cv2_area_is_fast(shape[1::-1], size) is not None
    got = pp.resize_to_device(img, (size[1], size[0]), 'cpu', imgtables.ARITH_CV2_AREA_FAST, lib=fake)
    assert np.array_equal(got.numpy(), cv.resize_area(img, size))


def test_load_image_plan_follows_the_reference():
    # datasets.py:519-524: shrink always (INTER_AREA unless augmenting), enlarge only when augmenting (INTER_LINEAR)
    assert imgtables.load_image_plan(480, 640, 416, False) == ((312, 416), imgtables.ARITH_CV2_AREA)
    assert imgtables.load_image_plan(480, 640, 320, False) == ((240, 320), imgtables.ARITH_CV2_AREA_FAST)
    assert imgtables.load_image_plan(480, 640, 416, True) == ((312, 416), imgtables.ARITH_CV2_LINEAR)
    assert imgtables.load_image_plan(100, 200, 416, True) == ((208, 416), imgtables.ARITH_CV2_LINEAR)
    assert imgtables.load_image_plan(100, 200, 416, False) == ((100, 200), None)
    assert imgtables.load_image_plan(416, 300, 416, True) == ((416, 300), None)
    for h0, w0, size, aug in [(480, 640, 416, False), (480, 640, 320, False), (100, 200, 416, True), (333, 500, 608, True)]:
        img = _img((h0, w0, 3), 8)
        want = cv.load_image_resize(img, size, aug)
        (h, w), code = imgtables.load_image_plan(h0, w0, size, aug)
        assert want.shape[:2] == (h, w)


LETTERBOX_CASES = [((120, 160, 3), 128, False), ((50, 100, 3), 64, True), ((300, 200, 3), 416, False), ((97, 31, 1), 96, True),
                   ((64, 64, 3), 64, False)]


@pytest.mark.parametrize('shape,size,auto', LETTERBOX_CASES)
@pytest.mark.parametrize('maxabs', [False, True])
def test_letterbox_in_cv2_arithmetic_on_the_emulated_abi(fake, shape, size, auto, maxabs):
    img = _img(shape, 9)
    host, ratio, pad = cv.letterbox(img, size, auto=auto)
    want = torch.from_numpy(np.ascontiguousarray(host.transpose(2, 0, 1))).float() / 256.0
    if maxabs:
        want = want * 2 - 1
    got, r2, pad2 = pp.letterbox_to_device(img, size, 'cpu', auto=auto, maxabsscaler=maxabs, arith='cv2')
    assert ratio == r2 and pad == pad2 and torch.equal(got, want)
    if shape[2] == 3:        # a BGR frame, as cv2.imread hands it to the reference, with the channel swap of datasets.py:112
        got_bgr, _, _ = pp.letterbox_to_device(np.ascontiguousarray(img[:, :, ::-1]), size, 'cpu', auto=auto, maxabsscaler=maxabs,
                                               swap_rb=True, arith='cv2')
        assert torch.equal(got_bgr, want)
    # the Pillow arithmetic of the same frame: same geometry; where a resize happens the pixels differ (the switch is not a no-op),
    # but only by resampling-filter noise - random frames are the worst case, a few dozen levels at most
    pil, _, _ = pp.letterbox_to_device(img, size, 'cpu', auto=auto, maxabsscaler=maxabs, arith='pillow')
    assert pil.shape == got.shape
    if host.shape[:2] != shape[:2] and max(shape[:2]) != size:
        d = (pil - got).abs() * (128.0 if maxabs else 256.0)
        assert 0 < d.max().item() < 160 and d.mean().item() < 12


def test_arith_switches_fail_loudly(dataset_dir, monkeypatch):
    with pytest.raises(ValueError):
        pp.letterbox_to_device(_img((8, 8, 3), 0), 8, 'cpu', arith='opencv')
    monkeypatch.setenv('YOLO_IMAGE_ARITH', 'magick')
    with pytest.raises(ValueError):
        pp.default_arith()
    monkeypatch.delenv('YOLO_IMAGE_ARITH')
    kw = dict(img_size=64, batch_size=4, augment=True, hyp=HYPS[1], rect=False)
    with pytest.raises(NotImplementedError):          # the host loader has no OpenCV arithmetic: no silent Pillow stand-in
        datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), arith='cv2', **kw)
    with pytest.raises(NotImplementedError):
        datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), arith='cv2', device_augment=True, cache_images=True, **kw)
    with pytest.raises(ValueError):
        datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), arith='skimage', device_augment=True, **kw)


# ------------------------------------------------------------------------------------------------------- mosaic training items
HYPS = [dict(degrees=0, translate=0, scale=0, shear=0, hsv_h=0, hsv_s=0, hsv_v=0),
        dict(degrees=1.98, translate=0.05, scale=0.05, shear=0.641, hsv_h=0.0138, hsv_s=0.678, hsv_v=0.36),      # train.py's hyp
        dict(degrees=15.0, translate=0.2, scale=0.4, shear=8.0, hsv_h=0.3, hsv_s=0.9, hsv_v=0.9)]


def _dataset(dataset_dir, size, hyp, gray, arith):
    return datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), img_size=size, batch_size=4, augment=True, hyp=hyp, rect=False,
                                        is_gray_scale=gray, device_augment=True, arith=arith)


def _reference_item(ds, index, seed):
    """The reference's __getitem__ for a mosaic item (datasets.py:470-505, 553-608, 649-715, 534-550) on the restated cv2 calls,
    drawing from the random streams in the reference's order.  Images are RGB here (the reference holds BGR until the end): resize
    and warp are per channel, and the HSV conversion is told the order."""
    random.seed(seed)
    np.random.seed(seed)
    s, hyp = ds.img_size, ds.hyp
    parts, labels4 = datasets.mosaic_layout(ds, index, ds.is_gray_scale, lazy=True)
    c = 1 if ds.is_gray_scale else 3
    canvas = np.full((2 * s, 2 * s, c), 114, dtype=np.uint8)
    for img, (x1a, y1a, x2a, y2a), (x1b, y1b, x2b, y2b), (h0, w0, h, w) in parts:
        resized = cv.load_image_resize(img, s, True)
        assert resized.shape[:2] == (h, w)
        canvas[y1a:y2a, x1a:x2a] = resized[y1b:y2b, x1b:x2b]
    M, sc, (width, height) = datasets.affine_matrix((2 * s, 2 * s), hyp['degrees'], hyp['translate'], hyp['scale'], hyp['shear'], border=-s // 2)
    img = cv.warp_affine(canvas, M[:2], (width, height), (114, 114, 114))
    if not ds.is_gray_scale:
        gains = np.random.uniform(-1, 1, 3) * [hyp['hsv_h'], hyp['hsv_s'], hyp['hsv_v']] + 1
        img = cv.augment_hsv(img, gains, order='rgb')
    if random.random() < 0.5:
        img = np.fliplr(img)
    return np.ascontiguousarray(img.transpose(2, 0, 1))


def _recipe(ds, index, seed):
    random.seed(seed)
    np.random.seed(seed)
    item = ds[index]
    return item, (random.random(), np.random.rand())


@pytest.mark.parametrize('hyp', HYPS, ids=['crop', 'train_hyp', 'wild'])
@pytest.mark.parametrize('size,gray', [(96, False), (64, False), (128, True), (200, False)])
def test_cv2_recipes_rendered_through_the_emulated_abi_equal_the_restated_reference(dataset_dir, fake, hyp, size, gray):
    ds = _dataset(dataset_dir, size, hyp, gray, 'cv2')
    twin = _dataset(dataset_dir, size, hyp, gray, 'pillow')
    for seed in range(4):
        index = seed % len(ds)
        item, tail = _recipe(ds, index, seed)
        item_p, tail_p = _recipe(twin, index, seed)
        assert tail == tail_p and torch.equal(item.labels, item_p.labels)       # same draws, same labels as the Pillow recipe path
        want = _reference_item(ds, index, seed)
        got = pp.render_mosaic_items([item], 'cpu', dtype=torch.uint8, lib=fake)[0].numpy()
        assert got.shape == want.shape
        assert np.array_equal(got, want), (seed, int(np.abs(got.astype(int) - want).max()), int((got != want).sum()))
        # only the source samples the windows read travel: never more than the frames themselves, a few pixels of slack per crop
        assert sum(0 if p[0] is None else p[0].size for p in item.parts) <= sum(pt[0].size for pt in datasets.mosaic_layout(_reseed(ds, seed), index, gray, lazy=True)[0])


def _reseed(ds, seed):
    random.seed(seed)
    np.random.seed(seed)
    return ds


def test_cv2_recipes_collate_and_render_as_float(dataset_dir, fake):
    ds = _dataset(dataset_dir, 64, HYPS[1], False, 'cv2')
    random.seed(11)
    np.random.seed(11)
    raw = [ds[i] for i in range(3)]
    batch, labels, paths, shapes = datasets.LoadImagesAndLabels.collate_fn(raw)
    import pickle
    batch = pickle.loads(pickle.dumps(batch))                      # loader workers hand batches over by pickling
    u8 = pp.render_mosaic_items(batch, 'cpu', dtype=torch.uint8, lib=fake)
    f32 = pp.render_mosaic_items(batch, 'cpu', dtype=torch.float32, lib=fake)
    assert torch.equal(f32, u8.float() / 256.0) and labels[:, 0].unique().tolist() == [0.0, 1.0, 2.0][:len(labels[:, 0].unique())]
    for i in range(3):
        assert np.array_equal(u8[i].numpy(), pp.render_mosaic_items([raw[i]], 'cpu', dtype=torch.uint8, lib=fake)[0].numpy())


# ------------------------------------------------------------------------------------------------------------------ GPU tier
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


@pytest.mark.gpu
@pytest.mark.parametrize('shape,size', RESIZE_CASES + [((480, 640, 3), (608, 456)), ((1080, 1920, 3), (608, 342))], ids=lambda c: 'x'.join(map(str, c)))
def test_kernel_linear_resize_equals_the_restatement(shape, size):
    _gpu()
    img = _img(shape, 5)
    got = pp.resize_to_device(img, (size[1], size[0]), 'cuda', imgtables.ARITH_CV2_LINEAR)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), cv.resize_linear(img, size))


@pytest.mark.gpu
@pytest.mark.parametrize('shape,size,code', [((60, 80, 3), (53, 41), 2), ((97, 131, 3), (64, 47), 2), ((64, 64, 1), (40, 40), 2), ((480, 640, 3), (416, 312), 2),
                                             ((60, 80, 3), (40, 30), 3), ((90, 120, 3), (40, 30), 3), ((90, 120, 1), (30, 30), 3), ((480, 640, 3), (320, 240), 3)])
def test_kernel_area_resize_equals_the_restatement(shape, size, code):
    _gpu()
    img = _img(shape, 6)
    got = pp.resize_to_device(img, (size[1], size[0]), 'cuda', code)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), cv.resize_area(img, size))


@pytest.mark.gpu
@pytest.mark.parametrize('shape,size,auto', LETTERBOX_CASES + [((480, 640, 3), 608, False), ((1080, 1920, 3), 608, True)])
@pytest.mark.parametrize('maxabs', [False, True])
def test_kernel_letterbox_in_cv2_arithmetic(shape, size, auto, maxabs):
    _gpu()
    img = _img(shape, 9)
    host, ratio, pad = cv.letterbox(img, size, auto=auto)
    want = torch.from_numpy(np.ascontiguousarray(host.transpose(2, 0, 1))).float() / 256.0
    if maxabs:
        want = want * 2 - 1
    got, r2, pad2 = pp.letterbox_to_device(img, size, 'cuda', auto=auto, maxabsscaler=maxabs, arith='cv2')
    torch.cuda.synchronize()
    assert ratio == r2 and pad == pad2 and torch.equal(got.cpu(), want)


@pytest.mark.gpu
@pytest.mark.parametrize('hyp', HYPS, ids=['crop', 'train_hyp', 'wild'])
@pytest.mark.parametrize('size,gray', [(96, False), (608, False), (128, True)])
def test_kernel_mosaic_items_in_cv2_arithmetic(dataset_dir, hyp, size, gray):
    _gpu()
    ds = _dataset(dataset_dir, size, hyp, gray, 'cv2')
    items, want = [], []
    for seed in range(6):
        index = seed % len(ds)
        items.append(_recipe(ds, index, seed)[0])
        want.append(_reference_item(ds, index, seed))
    want = torch.from_numpy(np.stack(want))
    got = pp.render_mosaic_items(items, 'cuda', dtype=torch.uint8)
    torch.cuda.synchronize()
    diff = (got.cpu().int() - want.int()).abs()
    assert diff.max().item() == 0, 'uint8 items differ: %d pixels, worst %d' % ((diff > 0).sum().item(), diff.max().item())
    gotf = pp.render_mosaic_items(items, 'cuda', dtype=torch.float32)
    assert torch.equal(gotf.cpu(), want.float() / 256.0)


# ------------------------------------------------------------------------------------------ evaluation / rect items as recipes
def _reference_eval_batch(ds, idx):
    """The reference's non-augmenting __getitem__ (datasets.py:480-505) on the restated cv2 calls: load_image (INTER_AREA when it
    shrinks), letterbox to the batch rectangle (border only), HWC -> CHW."""
    out = []
    for i in idx:
        img = datasets._read(ds.img_files[i], ds.is_gray_scale)
        img = cv.load_image_resize(img, ds.img_size, False)
        shape = ds.batch_shapes[ds.batch[i]]
        lb, ratio, pad = cv.letterbox(img, tuple(int(v) for v in shape), auto=False, scaleup=False)
        assert ratio == (1.0, 1.0)
        out.append(np.ascontiguousarray(lb.transpose(2, 0, 1)))
    return torch.from_numpy(np.stack(out)).float() / 256.0


def _eval_batches(dataset_dir, size, gray, render):
    common = dict(img_size=size, batch_size=4, rect=True, is_gray_scale=gray)
    dev = datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), device_letterbox=True, arith='cv2', **common)
    twin = datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), device_letterbox=True, arith='pillow', **common)
    codes = set()
    for b in sorted(set(dev.batch.tolist())):
        idx = [i for i in range(len(dev)) if dev.batch[i] == b]
        items = [dev[i] for i in idx]
        codes |= {it.code for it in items}
        batch, labels, paths, shapes = datasets.LoadImagesAndLabels.collate_fn(items)
        _, labels_p, paths_p, shapes_p = datasets.LoadImagesAndLabels.collate_fn([twin[i] for i in idx])
        assert torch.equal(labels, labels_p) and paths == paths_p and shapes == shapes_p      # geometry does not depend on the arithmetic
        got = render(batch)
        assert torch.equal(got.cpu(), _reference_eval_batch(dev, idx))
    return codes


@pytest.mark.parametrize('size,gray', [(64, False), (96, False), (96, True), (200, False)])
def test_cv2_eval_recipes_on_the_emulated_abi_equal_the_restated_reference(dataset_dir, fake, size, gray):
    codes = _eval_batches(dataset_dir, size, gray, lambda b: pp.render_letterbox_items(b, 'cpu', lib=fake))
    assert (imgtables.ARITH_CV2_AREA in codes) if size < 160 else codes == {None}


def test_cv2_eval_recipe_with_an_integer_factor(tmp_path, fake):
    from PIL import Image
    img = _img((128, 256, 3), 12)
    (tmp_path / 'images').mkdir()
    (tmp_path / 'labels').mkdir()
    Image.fromarray(img).save(tmp_path / 'images' / 'a.png')
    (tmp_path / 'labels' / 'a.txt').write_text('0 0.5 0.5 0.25 0.25\n')
    (tmp_path / 'list.txt').write_text(str(tmp_path / 'images' / 'a.png') + '\n')
    ds = datasets.LoadImagesAndLabels(str(tmp_path / 'list.txt'), img_size=128, batch_size=1, rect=True, device_letterbox=True, arith='cv2')
    item = ds[0]
    assert item.code == imgtables.ARITH_CV2_AREA_FAST and item.resized_hw == (64, 128)
    got = pp.render_letterbox_items([item], 'cpu', lib=fake)
    assert torch.equal(got, _reference_eval_batch(ds, [0]))


@pytest.mark.gpu
@pytest.mark.parametrize('size,gray', [(64, False), (96, False), (96, True), (200, False)])
def test_kernel_eval_recipes_in_cv2_arithmetic(dataset_dir, size, gray):
    _gpu()

    def render(b):
        out = pp.render_letterbox_items(b, 'cuda')
        torch.cuda.synchronize()
        return out
    _eval_batches(dataset_dir, size, gray, render)


# ------------------------------------------------------------------------------------ rect training items in cv2 arithmetic
def _reference_rect_item(ds, index, seed):
    """The reference's augmenting non-mosaic __getitem__ (datasets.py:480-505): load_image (INTER_LINEAR), letterbox to the batch
    rectangle, random_affine (border 0), augment_hsv, flip - on the restated cv2 calls."""
    random.seed(seed)
    np.random.seed(seed)
    hyp = ds.hyp
    img = cv.load_image_resize(datasets._read(ds.img_files[index], ds.is_gray_scale), ds.img_size, True)
    shape = tuple(int(v) for v in ds.batch_shapes[ds.batch[index]])
    img, ratio, pad = cv.letterbox(img, shape, auto=False, scaleup=True)
    M, sc, (width, height) = datasets.affine_matrix(img.shape[:2], hyp['degrees'], hyp['translate'], hyp['scale'], hyp['shear'])
    if (M != np.eye(3)).any():
        img = cv.warp_affine(img, M[:2], (width, height), (114, 114, 114))
    if not ds.is_gray_scale:
        gains = np.random.uniform(-1, 1, 3) * [hyp['hsv_h'], hyp['hsv_s'], hyp['hsv_v']] + 1
        img = cv.augment_hsv(img, gains, order='rgb')
    if random.random() < 0.5:
        img = np.fliplr(img)
    return np.ascontiguousarray(img.transpose(2, 0, 1))


def _rect_ds(dataset_dir, size, hyp, gray, arith):
    return datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), img_size=size, batch_size=4, augment=True, hyp=hyp, rect=True,
                                        is_gray_scale=gray, device_augment=True, arith=arith)


@pytest.mark.parametrize('hyp', HYPS, ids=['crop', 'train_hyp', 'wild'])
@pytest.mark.parametrize('size,gray', [(96, False), (160, False), (128, True)])
def test_cv2_rect_training_recipes_on_the_emulated_abi(dataset_dir, fake, hyp, size, gray):
    ds, twin = _rect_ds(dataset_dir, size, hyp, gray, 'cv2'), _rect_ds(dataset_dir, size, hyp, gray, 'pillow')
    for seed in range(4):
        index = seed % len(ds)
        item, tail = _recipe(ds, index, seed)
        item_p, tail_p = _recipe(twin, index, seed)
        assert tail == tail_p and torch.equal(item.labels, item_p.labels)
        want = _reference_rect_item(ds, index, seed)
        got = pp.render_mosaic_items([item], 'cpu', dtype=torch.uint8, lib=fake)[0].numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (seed, int((got != want).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize('hyp', HYPS, ids=['crop', 'train_hyp', 'wild'])
@pytest.mark.parametrize('size,gray', [(96, False), (160, False)])
def test_kernel_rect_training_items_in_cv2_arithmetic(dataset_dir, hyp, size, gray):
    _gpu()
    ds = _rect_ds(dataset_dir, size, hyp, gray, 'cv2')
    for b in sorted(set(ds.batch.tolist())):
        idx = [i for i in range(len(ds)) if ds.batch[i] == b]
        items = [_recipe(ds, i, 7 * b + k)[0] for k, i in enumerate(idx)]
        want = torch.from_numpy(np.stack([_reference_rect_item(ds, i, 7 * b + k) for k, i in enumerate(idx)]))
        got = pp.render_mosaic_items(items, 'cuda', dtype=torch.uint8)
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want)
