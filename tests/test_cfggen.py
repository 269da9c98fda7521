"""Generated cfgs must parse to the same block lists as the reference's own files (container only)."""
import os

import numpy as np
import pytest

import refharness
from utils.parse_config import parse_model_cfg

PAIRS = ['yolov3/yolov3.cfg', 'yolov3tiny/yolov3-tiny.cfg', 'yolov3tiny/yolov3-tiny-hand.cfg', 'yolov4/yolov4.cfg',
         'yolov4tiny/yolov4-tiny.cfg',
         'yolov3-mobilenet/yolov3-mobilenet-coco.cfg']


def _same(a, b):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.array_equal(np.asarray(a), np.asarray(b))
    return a == b


@pytest.mark.skipif(not refharness.available(), reason='reference tree not present')
@pytest.mark.parametrize('rel', PAIRS)
def test_generated_cfg_equals_reference(rel, cfg_dir):
    mine = parse_model_cfg(os.path.join(cfg_dir, rel))
    ref = refharness.load().parse_config.parse_model_cfg(os.path.join(refharness.REF, 'cfg', rel))
    assert len(mine) == len(ref)
    for i, (m, r) in enumerate(zip(mine, ref)):
        assert set(m) == set(r), 'block %d keys differ: %s' % (i - 1, set(m) ^ set(r))
        if i == 0:  # [net]: float-looking strings may be spelled differently (0.9 vs .9)
            continue
        for k in m:
            assert _same(m[k], r[k]), 'block %d key %s: %r != %r' % (i - 1, k, m[k], r[k])


def test_generated_cfgs_parse(cfg_dir):
    for rel in PAIRS:
        blocks = parse_model_cfg(os.path.join(cfg_dir, rel))
        assert blocks[0]['type'] == 'net'
        assert sum(b['type'] == 'yolo' for b in blocks) in (2, 3)
