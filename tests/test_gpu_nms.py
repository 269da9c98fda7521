"""GPU tier: the HIP NMS kernels against the reference-generated goldens and the CPU oracle."""
import glob
import os

import numpy as np
import pytest
import torch

import synth
from oracle import darknet_oracle as oracle
from test_oracle_golden import GOLD

pytestmark = pytest.mark.gpu
NMS_FIXTURES = sorted(glob.glob(os.path.join(GOLD, 'nms_*.npz')))


@pytest.fixture(scope='module', autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _compare(got, want, tag):
    if want is None or len(want) == 0:
        assert got is None, tag
        return
    g = got.cpu().numpy()
    assert g.shape == want.shape, '%s: %s vs %s' % (tag, g.shape, want.shape)
    np.testing.assert_allclose(g[:, :4], want[:, :4], rtol=0, atol=2e-3, err_msg=tag)
    np.testing.assert_allclose(g[:, 4], want[:, 4], rtol=0, atol=1e-6, err_msg=tag)
    np.testing.assert_array_equal(g[:, 5], want[:, 5], err_msg=tag)


@pytest.mark.parametrize('path', NMS_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in NMS_FIXTURES])
def test_hip_nms_matches_reference_golden(path):
    from utils.utils import non_max_suppression
    fx = np.load(path, allow_pickle=False)
    pred = synth.nms_candidates(int(fx['n_img']), int(fx['rows']), int(fx['nc']), int(fx['seed']))
    got = non_max_suppression(pred.cuda(), conf_thres=float(fx['conf']), iou_thres=float(fx['iou']),
                              multi_label=bool(fx['multi_label']), agnostic=bool(fx['agnostic']))
    for i, n in enumerate(fx['counts']):
        _compare(got[i], fx['det%d' % i] if n else None, '%s img %d' % (os.path.basename(path), i))


@pytest.mark.parametrize('rows,nc,conf,ml,seed', [(3000, 80, 0.3, False, 51), (1500, 20, 0.05, True, 52),
                                                  (9000, 80, 0.2, False, 53), (700, 2, 0.4, True, 54)])
def test_hip_nms_matches_oracle(rows, nc, conf, ml, seed):
    """Larger seeded sets, including n >= 3000 candidates where the reference skips the merge step."""
    from utils.utils import non_max_suppression
    pred = synth.nms_candidates(2, rows, nc, seed, n_clusters=40, hot=0.9 if rows >= 9000 else 0.35)
    want = oracle.non_max_suppression(pred.numpy(), conf, 0.6, multi_label=ml)
    got = non_max_suppression(pred.cuda(), conf, 0.6, multi_label=ml)
    for i in range(2):
        _compare(got[i], want[i], 'img %d' % i)


def test_hip_nms_class_filter_and_idempotence():
    from utils.utils import non_max_suppression
    pred = synth.nms_candidates(1, 800, 10, 55)
    want = oracle.non_max_suppression(pred.numpy(), 0.3, 0.6, multi_label=False, classes=[1, 4])
    got = non_max_suppression(pred.cuda(), 0.3, 0.6, multi_label=False, classes=[1, 4])
    _compare(got[0], want[0], 'class filter')
    # kept boxes of one class do not overlap above the threshold (property of greedy NMS before merging)
    raw = non_max_suppression(pred.cuda(), 0.3, 0.6, multi_label=False)[0]
    assert raw is not None and torch.all(raw[:-1, 4] >= raw[1:, 4]), 'output must be in descending score order'
