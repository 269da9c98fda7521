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


def test_hip_nms_bound_forgets_outliers_and_respects_the_mask_budget(monkeypatch):
    """the candidate bound used to be a grow-only power of two per input shape, so one dense batch pinned a
    quadratic IoU mask for the rest of the process.  Now: (a) the bound follows the candidate density of the last few calls,
    (b) a guessed bound whose mask exceeds the budget takes the count-first path - and every path returns the same boxes."""
    from engine import nms as hnms
    from utils.utils import non_max_suppression
    monkeypatch.setattr(hnms, '_density', {})
    dense = synth.nms_candidates(2, 4000, 20, 61, n_clusters=40, hot=0.9).cuda()
    sparse = synth.nms_candidates(2, 4000, 20, 62, n_clusters=10, hot=0.02).cuda()
    want_dense = oracle.non_max_suppression(dense.cpu().numpy(), 0.05, 0.6, multi_label=True)
    want_sparse = oracle.non_max_suppression(sparse.cpu().numpy(), 0.3, 0.6, multi_label=False)
    got = non_max_suppression(dense, 0.05, 0.6, multi_label=True)          # first call: overflow of the 256 default, repeated exactly
    for i in range(2):
        _compare(got[i], want_dense[i], 'dense img %d' % i)
    d0 = max(hnms._density[1])
    assert d0 > 0.5                                                           # thousands of candidates per image
    for _ in range(hnms._HINT_WINDOW):                                        # sparse single-label calls never see the dense hint
        got = non_max_suppression(sparse, 0.3, 0.6, multi_label=False)
    for i in range(2):
        _compare(got[i], want_sparse[i], 'sparse img %d' % i)
    assert max(hnms._density[0]) < 0.1 and max(hnms._density[1]) == d0       # keyed by label mode
    for _ in range(hnms._HINT_WINDOW):                                        # multi-label at a high threshold: the outlier ages out
        non_max_suppression(sparse, 0.6, 0.6, multi_label=True)
    assert max(hnms._density[1]) < d0
    # budget path: a hint this dense with a tiny budget must count first and still agree
    monkeypatch.setattr(hnms, '_MASK_BUDGET', 1 << 16)
    hnms._density[1].append(d0)
    got = non_max_suppression(dense, 0.05, 0.6, multi_label=True)
    for i in range(2):
        _compare(got[i], want_dense[i], 'budget path img %d' % i)


def test_hip_nms_processes_large_batches_in_chunks(monkeypatch):
    """test.py settings (conf 0.001, multi-label): thousands of candidates per image and an IoU bit mask quadratic in them.  With a
    small work budget the batch is processed in image chunks; same boxes as one pass and as the oracle; a single image beyond the
    budget raises MemoryError naming the candidate count."""
    from engine import nms as hnms
    from utils.utils import non_max_suppression
    monkeypatch.setattr(hnms, '_density', {})
    pred = synth.nms_candidates(5, 3000, 20, 71, n_clusters=40, hot=0.9).cuda()
    want = oracle.non_max_suppression(pred.cpu().numpy(), 0.05, 0.6, multi_label=True)
    one = non_max_suppression(pred, 0.05, 0.6, multi_label=True)
    mmax = int(max(hnms._density[1]) * 3000)
    assert mmax > 2000
    monkeypatch.setattr(hnms, '_density', {})
    monkeypatch.setattr(hnms, '_WORK_BUDGET', 2 * (mmax * 2) * ((mmax * 2 + 63) // 64) * 8)      # room for about two images
    chunked = non_max_suppression(pred, 0.05, 0.6, multi_label=True)
    for i in range(5):
        _compare(one[i], want[i], 'one pass img %d' % i)
        _compare(chunked[i], want[i], 'chunked img %d' % i)
    monkeypatch.setattr(hnms, '_WORK_BUDGET', 1 << 16)
    with pytest.raises(MemoryError):
        non_max_suppression(pred, 0.05, 0.6, multi_label=True)
