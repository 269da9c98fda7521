"""Device-side training items (second slice): mosaic + affine warp + HSV + flip as one HIP kernel.

CPU tier: ``LoadImagesAndLabels(device_augment=True)`` consumes the random streams exactly like the host path and its recipe,
rendered by the numpy restatement of the kernel's arithmetic (engine/preprocess.py mosaic_reference: Pillow's double-precision
bilinear transform on a virtual canvas, numpy's float32 / float64 HSV formulas), equals the host loader's item bit for bit -
pixels and labels.  GPU tier: ``yh_mosaic_affine_hsv`` against the host loader's items, uint8 and /256 float."""
import random

import numpy as np
import pytest
import torch

import conftest  # noqa: F401
from engine import preprocess as pp
from utils import datasets

HYPS = [dict(degrees=0, translate=0, scale=0, shear=0, hsv_h=0, hsv_s=0, hsv_v=0),                      # the reference's defaults: crop only
        dict(degrees=1.98, translate=0.05, scale=0.05, shear=0.641, hsv_h=0.0138, hsv_s=0.678, hsv_v=0.36),   # train.py's hyp
        dict(degrees=15.0, translate=0.2, scale=0.4, shear=8.0, hsv_h=0.3, hsv_s=0.9, hsv_v=0.9)]             # far outside


def _pair(dataset_dir, size, hyp, gray=False):
    kw = dict(img_size=size, batch_size=4, augment=True, hyp=hyp, rect=False, is_gray_scale=gray)
    host = datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), **kw)
    dev = datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), device_augment=True, **kw)
    assert dev.device_augment and dev.mosaic
    return host, dev


def _items(ds, index, seed):
    random.seed(seed)
    np.random.seed(seed)
    item = ds[index]
    return item, (random.random(), np.random.rand())      # the next draws: both paths must leave the streams in the same state


@pytest.mark.parametrize('hyp', HYPS, ids=['crop', 'train_hyp', 'wild'])
@pytest.mark.parametrize('size,gray', [(96, False), (64, False), (128, True)])
def test_recipe_rendered_on_the_host_equals_the_host_loader(dataset_dir, hyp, size, gray):
    host, dev = _pair(dataset_dir, size, hyp, gray)
    for seed in range(6):
        index = seed % len(host)
        (img, labels, path, shapes), tail_h = _items(host, index, seed)
        item, tail_d = _items(dev, index, seed)
        assert tail_h == tail_d, 'the recipe path consumed the random streams differently'
        assert path == item.path and torch.equal(labels, item.labels)
        got = pp.mosaic_reference(item)
        assert got.shape == tuple(img.shape) and np.array_equal(got, img.numpy()), (seed, np.abs(got.astype(int) - img.numpy().astype(int)).max())
        # the shipped crops are what the warp can touch, not the whole frames
        assert sum(0 if p[0] is None else p[0].size for p in item.parts) <= 2.2 * img.numel() + 4096


def test_collate_of_recipes(dataset_dir):
    host, dev = _pair(dataset_dir, 64, HYPS[1])
    random.seed(3)
    np.random.seed(3)
    raw = [dev[i] for i in range(3)]
    rendered = np.stack([pp.mosaic_reference(it) for it in raw])
    batch, labels, paths, shapes = datasets.LoadImagesAndLabels.collate_fn(raw)
    random.seed(3)
    np.random.seed(3)
    imgs_h, labels_h, paths_h, _ = datasets.LoadImagesAndLabels.collate_fn([host[i] for i in range(3)])
    assert torch.equal(labels, labels_h) and paths == paths_h and len(batch) == 3
    assert np.array_equal(rendered, imgs_h.numpy())
    # collation moved every crop into one blob and left shapes behind; the batch survives pickling (loader workers) and pinning
    import pickle
    clone = pickle.loads(pickle.dumps(batch))
    assert torch.equal(clone.blob, batch.blob) and all(p[0] is None or len(p[0]) == 2 for it in clone.items for p in it.parts)


@pytest.mark.gpu
@pytest.mark.parametrize('hyp', HYPS, ids=['crop', 'train_hyp', 'wild'])
@pytest.mark.parametrize('size,gray', [(96, False), (608, False), (128, True)])
def test_kernel_equals_the_host_loader(dataset_dir, hyp, size, gray):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    host, dev = _pair(dataset_dir, size, hyp, gray)
    want, items = [], []
    for seed in range(8):
        index = seed % len(host)
        (img, labels, _, _), _ = _items(host, index, seed)
        item, _ = _items(dev, index, seed)
        assert torch.equal(labels, item.labels)
        want.append(img)
        items.append(item)
    want = torch.stack(want)
    got = pp.render_mosaic_items(items, 'cuda', dtype=torch.uint8)
    torch.cuda.synchronize()
    diff = (got.cpu().int() - want.int()).abs()
    assert diff.max().item() == 0, 'uint8 items differ: %d pixels, worst %d' % ((diff > 0).sum().item(), diff.max().item())
    gotf = pp.render_mosaic_items(items, 'cuda', dtype=torch.float32)
    assert torch.equal(gotf.cpu(), want.float() / 256.0)          # train.py:345 on the host items


# ----------------------------------------------------------------------------- rect training: one letterboxed image per item
def _rect_pair(dataset_dir, size, hyp, gray=False):
    kw = dict(img_size=size, batch_size=4, augment=True, hyp=hyp, rect=True, is_gray_scale=gray)
    host = datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), **kw)
    dev = datasets.LoadImagesAndLabels(str(dataset_dir / 'train.txt'), device_augment=True, **kw)
    assert dev.device_rect_augment and not dev.mosaic and not dev.device_augment
    return host, dev


@pytest.mark.parametrize('hyp', HYPS, ids=['crop', 'train_hyp', 'wild'])
@pytest.mark.parametrize('size,gray', [(96, False), (160, False), (128, True)])
def test_rect_training_recipe_rendered_on_the_host_equals_the_host_loader(dataset_dir, hyp, size, gray):
    host, dev = _rect_pair(dataset_dir, size, hyp, gray)
    for seed in range(6):
        index = seed % len(host)
        (img, labels, path, shapes), tail_h = _items(host, index, seed)
        item, tail_d = _items(dev, index, seed)
        assert tail_h == tail_d, 'the recipe path consumed the random streams differently'
        assert path == item.path and torch.equal(labels, item.labels)
        got = pp.mosaic_reference(item)
        assert got.shape == tuple(img.shape) and np.array_equal(got, img.numpy()), (seed, np.abs(got.astype(int) - img.numpy().astype(int)).max())
    # a rect batch collates like a mosaic batch (one rectangle per batch)
    random.seed(5)
    np.random.seed(5)
    idx = [i for i in range(len(dev)) if dev.batch[i] == 0]
    batch, labels, paths, _ = datasets.LoadImagesAndLabels.collate_fn([dev[i] for i in idx])
    assert len(batch) == len(idx) and len({tuple(it.out_hw) for it in batch.items}) == 1


@pytest.mark.gpu
@pytest.mark.parametrize('hyp', HYPS, ids=['crop', 'train_hyp', 'wild'])
@pytest.mark.parametrize('size,gray', [(96, False), (160, False), (128, True)])
def test_kernel_renders_rect_training_items_like_the_host_loader(dataset_dir, hyp, size, gray):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    host, dev = _rect_pair(dataset_dir, size, hyp, gray)
    for b in sorted(set(host.batch.tolist())):
        idx = [i for i in range(len(host)) if host.batch[i] == b]
        want, items = [], []
        for k, index in enumerate(idx):
            (img, labels, _, _), _ = _items(host, index, 10 * b + k)
            item, _ = _items(dev, index, 10 * b + k)
            assert torch.equal(labels, item.labels)
            want.append(img)
            items.append(item)
        want = torch.stack(want)
        got = pp.render_mosaic_items(items, 'cuda', dtype=torch.uint8)
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want)
        assert torch.equal(pp.render_mosaic_items(items, 'cuda', dtype=torch.float32).cpu(), want.float() / 256.0)
