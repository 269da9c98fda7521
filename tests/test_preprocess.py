"""Device letterbox (first slice).

CPU tier: the host-side coefficient tables + integer two-pass arithmetic (engine/preprocess.py) reproduce Pillow's BILINEAR resize
bit for bit - the resize ``utils.datasets.letterbox`` runs - for up- and down-scaling, odd sizes, 1 and 3 channels; the letterbox
geometry equals the loader's.  GPU tier: ``yh_letterbox_fwd`` against the host loader's letterbox + transpose + /256."""
import numpy as np
import pytest
import torch
from PIL import Image

import conftest  # noqa: F401
from engine import preprocess as pp
from utils import datasets

CASES = [((120, 160, 3), (96, 128)), ((50, 100, 3), (32, 64)), ((37, 53, 3), (111, 159)), ((64, 64, 3), (64, 64)), ((200, 90, 1), (100, 45)),
         ((33, 47, 3), (20, 47)), ((480, 640, 3), (456, 608))]


def _img(shape, seed):
    return np.random.RandomState(seed).randint(0, 256, size=shape, dtype=np.uint8)


@pytest.mark.parametrize('shape,out_hw', CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_tables_reproduce_pillow_bilinear(shape, out_hw):
    img = _img(shape, 1)
    want = datasets._resize(img, (out_hw[1], out_hw[0]))          # the loader's resize: Pillow BILINEAR
    got = pp.resample_reference(img, out_hw)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize('shape,size,auto', [((120, 160, 3), 128, False), ((50, 100, 3), 64, True), ((300, 200, 3), 416, False),
                                             ((97, 31, 1), 96, True)])
def test_geometry_matches_the_host_letterbox(shape, size, auto):
    img = _img(shape, 2)
    out, ratio, pad = datasets.letterbox(img, size, auto=auto)
    new_h, new_w, out_h, out_w, top, left, r2, pad2 = pp.letterbox_geometry(shape[0], shape[1], size, auto=auto)
    assert out.shape[:2] == (out_h, out_w) and ratio == r2 and pad == pad2
    assert (out[top:top + new_h, left:left + new_w] == pp.resample_reference(img, (new_h, new_w))).all()


@pytest.mark.gpu
@pytest.mark.parametrize('shape,size,auto', [((120, 160, 3), 128, False), ((50, 100, 3), 64, True), ((300, 200, 3), 416, False),
                                             ((480, 640, 3), 608, False), ((97, 31, 1), 96, True), ((1080, 1920, 3), 608, True)])
@pytest.mark.parametrize('maxabs', [False, True])
def test_device_letterbox_equals_host_loader(shape, size, auto, maxabs):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    img = _img(shape, 3)
    host, ratio, pad = datasets.letterbox(img, size, auto=auto)
    want = torch.from_numpy(np.ascontiguousarray(host.transpose(2, 0, 1))).float() / 256.0       # detect.py:99-101
    if maxabs:
        want = want * 2 - 1
    got, r2, pad2 = pp.letterbox_to_device(img, size, 'cuda', auto=auto, maxabsscaler=maxabs)
    torch.cuda.synchronize()
    assert ratio == r2 and pad == pad2 and tuple(got.shape) == tuple(want.shape)
    assert torch.equal(got.cpu(), want)
    # BGR frames (what cv2.imread hands the reference's loader) with the channel swap of datasets.py:112
    got_bgr, _, _ = pp.letterbox_to_device(np.ascontiguousarray(img[:, :, ::-1]), size, 'cuda', auto=auto, maxabsscaler=maxabs,
                                           swap_rb=shape[2] == 3)
    assert torch.equal(got_bgr.cpu(), want)


@pytest.mark.gpu
def test_device_letterbox_writes_into_a_batch_slot():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    batch = torch.zeros((3, 3, 64, 64), device='cuda')
    frames = [_img((40 + 7 * i, 80 - 5 * i, 3), 10 + i) for i in range(3)]
    for i, f in enumerate(frames):
        pp.letterbox_to_device(f, 64, 'cuda', out=batch[i], auto=False)
    torch.cuda.synchronize()
    for i, f in enumerate(frames):
        host = datasets.letterbox(f, 64, auto=False)[0]
        assert torch.equal(batch[i].cpu(), torch.from_numpy(np.ascontiguousarray(host.transpose(2, 0, 1))).float() / 256.0)


# ------------------------------------------------------------------------------------------ evaluation / rect items as recipes
def _eval_pair(dataset_dir, size, batch, gray=False, **kw):
    common = dict(img_size=size, batch_size=batch, rect=True, is_gray_scale=gray)
    host = datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), **common)
    dev = datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), device_letterbox=True, **common, **kw)
    assert dev.device_letterbox and not host.device_letterbox
    return host, dev


def _check_eval_batches(host, dev, render):
    n = len(host)
    for b0 in range(0, n, host.batch[1:].tolist().count(0) + 1 if n > 1 else 1):
        idx = [i for i in range(n) if host.batch[i] == host.batch[b0]]
        imgs, labels, paths, shapes = datasets.LoadImagesAndLabels.collate_fn([host[i] for i in idx])
        batch, labels_d, paths_d, shapes_d = datasets.LoadImagesAndLabels.collate_fn([dev[i] for i in idx])
        assert isinstance(batch, datasets.LetterboxBatch) and paths == paths_d and torch.equal(labels, labels_d)
        assert [tuple(map(tuple, (s[0], s[1][0], s[1][1]))) for s in shapes] == [tuple(map(tuple, (s[0], s[1][0], s[1][1]))) for s in shapes_d]
        got = render(batch)
        want = imgs.float() / 256.0                       # test.py:95-96
        assert tuple(got.shape) == tuple(want.shape) and torch.equal(got.cpu(), want)
        got2 = render(batch, maxabsscaler=True)
        assert torch.equal(got2.cpu(), want * 2 - 1)


@pytest.mark.parametrize('size,gray', [(64, False), (96, False), (96, True), (200, False)])
def test_eval_recipes_on_the_emulated_abi_equal_the_host_loader(dataset_dir, size, gray):
    import fakelib
    import pickle
    lib = fakelib.FakeLib()
    host, dev = _eval_pair(dataset_dir, size, 4, gray)
    assert any(it is not None for it in [dev[0].code]) or size >= 160          # the small sizes shrink (Image.BOX), 200 does not
    _check_eval_batches(host, dev, lambda b, **kw: pp.render_letterbox_items(pickle.loads(pickle.dumps(b)), 'cpu', lib=lib, **kw))


def test_box_tables_reproduce_pillow_box():
    for shape, out_hw in [((120, 160, 3), (96, 128)), ((97, 131, 3), (47, 64)), ((200, 90, 1), (100, 45)), ((64, 64, 3), (32, 32))]:
        img = _img(shape, 4)
        assert np.array_equal(pp.resample_reference(img, out_hw, 'box'), datasets._resize(img, (out_hw[1], out_hw[0]), area=True))


@pytest.mark.gpu
@pytest.mark.parametrize('size,gray', [(64, False), (96, False), (96, True), (200, False)])
def test_device_eval_recipes_equal_the_host_loader(dataset_dir, size, gray):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    host, dev = _eval_pair(dataset_dir, size, 4, gray)

    def render(b, **kw):
        out = pp.render_letterbox_items(b, 'cuda', **kw)
        torch.cuda.synchronize()
        return out
    _check_eval_batches(host, dev, render)
