import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'yolov3v4-modelcompression-multidatasettraining-multibackbone_amd')
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def pkg_root():
    return PKG


@pytest.fixture(scope='session')
def cfg_dir():
    return os.path.join(PKG, 'cfg')


# small synthetic image set + 21-block cfg shared by the entry-point and reference-script tests
@pytest.fixture(scope='session')
def dataset_dir(tmp_path_factory):
    """12 images of random size with 1-3 bright rectangles on a dark background; class = 0 (wide) / 1 (tall)."""
    root = tmp_path_factory.mktemp('synthdata')
    img_dir, lab_dir = root / 'images', root / 'labels'
    img_dir.mkdir()
    lab_dir.mkdir()
    rng = np.random.RandomState(0)
    files = []
    for i in range(12):
        w, h = int(rng.randint(90, 160)), int(rng.randint(70, 140))
        img = (rng.rand(h, w, 3) * 40).astype(np.uint8)
        rows = []
        for _ in range(rng.randint(1, 4)):
            bw, bh = rng.uniform(0.2, 0.5), rng.uniform(0.2, 0.5)
            cx, cy = rng.uniform(bw / 2, 1 - bw / 2), rng.uniform(bh / 2, 1 - bh / 2)
            x1, x2, y1, y2 = int((cx - bw / 2) * w), int((cx + bw / 2) * w), int((cy - bh / 2) * h), int((cy + bh / 2) * h)
            img[y1:y2, x1:x2] = rng.randint(150, 255, 3)
            rows.append('%d %.6f %.6f %.6f %.6f' % (0 if bw * w > bh * h else 1, cx, cy, bw, bh))
        p = img_dir / ('im_%02d.png' % i)
        from PIL import Image
        Image.fromarray(img).save(p)
        (lab_dir / ('im_%02d.txt' % i)).write_text('\n'.join(rows) + '\n')
        files.append(str(p))
    (root / 'train.txt').write_text('\n'.join(files[:8]) + '\n')
    (root / 'valid.txt').write_text('\n'.join(files[8:]) + '\n')
    (root / 'synth.names').write_text('wide\ntall\n')
    (root / 'synth.data').write_text('classes=2\ntrain=%s\nvalid=%s\nnames=%s\n' % (root / 'train.txt', root / 'valid.txt', root / 'synth.names'))
    return root


@pytest.fixture(scope='session')
def tiny_cfg(tmp_path_factory):
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import train_harness as th
    path = tmp_path_factory.mktemp('cfg') / 'mini2.cfg'
    path.write_text(th.mini_cfg_text())
    return str(path)
