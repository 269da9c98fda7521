"""GPU tier: the HIP NMS kernels against the reference-generated goldens and the CPU oracle."""
import glob
import os

import numpy as np
import pytest
import torch

import synth
from oracle import darknet_oracle as oracle
from test_oracle_golden import GOLD, build_mirror

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


SEG = pytest.mark.parametrize('seg', ['', '1', '0'], ids=['auto', 'by-class', 'bit-mask'])


@SEG
@pytest.mark.parametrize('path', NMS_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in NMS_FIXTURES])
def test_hip_nms_matches_reference_golden(path, seg, monkeypatch):
    from utils.utils import non_max_suppression
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', seg)      # '': class by class for multi-label calls; 1: whenever not agnostic; 0: never
    fx = np.load(path, allow_pickle=False)
    pred = synth.nms_candidates(int(fx['n_img']), int(fx['rows']), int(fx['nc']), int(fx['seed']))
    got = non_max_suppression(pred.cuda(), conf_thres=float(fx['conf']), iou_thres=float(fx['iou']),
                              multi_label=bool(fx['multi_label']), agnostic=bool(fx['agnostic']))
    for i, n in enumerate(fx['counts']):
        _compare(got[i], fx['det%d' % i] if n else None, '%s img %d' % (os.path.basename(path), i))


@pytest.mark.parametrize('rows,nc,conf,ml,seed', [(3000, 80, 0.3, False, 51), (1500, 20, 0.05, True, 52),
                                                  (9000, 80, 0.2, False, 53), (700, 2, 0.4, True, 54)])
@SEG
def test_hip_nms_matches_oracle(rows, nc, conf, ml, seed, seg, monkeypatch):
    """Larger seeded sets, including n >= 3000 candidates where the reference skips the merge step."""
    from utils.utils import non_max_suppression
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', seg)
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
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', '0')      # the mask budget is the bit-mask form's (the class-by-class form has no mask)
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


@pytest.mark.parametrize('seg', ['0'], ids=['bit-mask'])
def test_hip_nms_processes_large_batches_in_chunks(monkeypatch, seg):
    """test.py settings (conf 0.001, multi-label): thousands of candidates per image and an IoU bit mask quadratic in them.  With a
    small work budget the batch is processed in image chunks; same boxes as one pass and as the oracle; a single image beyond the
    budget raises MemoryError naming the candidate count."""
    from engine import nms as hnms
    from utils.utils import non_max_suppression
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', seg)      # the budget arithmetic of this test is the bit-mask form's
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


def _same(a, b, tag):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x is None) == (y is None), '%s img %d' % (tag, i)
        if x is not None:
            assert torch.equal(x, y), '%s img %d: %d vs %d boxes' % (tag, i, x.shape[0], y.shape[0])


def test_hip_nms_class_by_class_equals_the_bit_mask_form_at_evaluation_settings(monkeypatch):
    """Round 5.  test.py's settings (conf 0.001, multi-label) on clustered predictions: 10^4 candidates per image in
    80 classes, most of them suppressed.  One workgroup per (image, class) runs the greedy scan on that class's boxes alone
    (yh_nms_class_scan) - BIT-identical output to the bit-mask form over all candidates, and to the oracle on the smaller images;
    the single host read reports that no image needed the general form."""
    from engine import nms as hnms
    from utils.utils import non_max_suppression
    monkeypatch.setattr(hnms, '_density', {})
    pred = synth.nms_candidates(4, 2500, 80, 81, n_clusters=30, hot=0.5)
    pred[0, 260:, 4] = 0                                     # ~ 260 rows with objectness, nearly all of their 80 class scores pass
    pred[1, 120:, 4] = 0
    pred[2, 30:, 4] = 0                                      # an image with few candidates (merge applies: < 3000)
    pred[3, :, 4] = 0                                        # and one with none
    want = oracle.non_max_suppression(pred[1:].numpy(), 0.001, 0.6, multi_label=True)
    dev = pred.cuda()
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', '0')
    general = non_max_suppression(dev, 0.001, 0.6, multi_label=True)
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', '')
    reads = []
    orig = torch.Tensor.cpu
    monkeypatch.setattr(torch.Tensor, 'cpu', lambda t, *a, **k: (reads.append(tuple(t.shape)), orig(t, *a, **k))[1])
    by_class = non_max_suppression(dev, 0.001, 0.6, multi_label=True)
    by_class = non_max_suppression(dev, 0.001, 0.6, multi_label=True)      # second call: the bound is known
    cands = hnms._density[1][-1] * 2500
    monkeypatch.undo()
    assert reads[-1] == (10, 4) and len([r for r in reads if r == (10, 4)]) <= 3, reads      # ONE read in the steady state
    _same(by_class, general, 'class by class vs bit mask')
    m = [0 if d is None else d.shape[0] for d in by_class]
    assert cands > 15000 and 1000 < m[0] < 0.5 * cands and m[3] == 0, (cands, m)      # most candidates are suppressed
    _compare(by_class[1], want[0], '~ 9000 candidates')
    _compare(by_class[2], want[1], 'few candidates, merged')
    assert by_class[3] is None and want[2] is None


def test_hip_nms_class_by_class_hands_over_what_it_does_not_cover(monkeypatch):
    """(a) A class with more than 2048 candidates, (b) candidates spread over more than 4096 pixels in x and y - where the reference's
    offset boxes of DIFFERENT classes can overlap: a class-0 box at (5000, 5000) and a class-1 box at (904, 904) are the same box after
    `+ cls * 4096` (utils.py:840) and the reference keeps only the better one.  Both are detected on the device, the batch takes the
    bit-mask form, and the result is the oracle's."""
    from engine import nms as hnms
    from utils.utils import non_max_suppression
    monkeypatch.setattr(hnms, '_density', {})
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', '')
    big = synth.nms_candidates(2, 4000, 2, 91, n_clusters=25, hot=0.9)      # two classes, ~3500 candidates each
    want = oracle.non_max_suppression(big.numpy(), 0.02, 0.6, multi_label=True)
    got = non_max_suppression(big.cuda(), 0.02, 0.6, multi_label=True)
    for i in range(2):
        _compare(got[i], want[i], 'class overflow img %d' % i)
    far = torch.zeros(2, 64, 5 + 3)
    far[:, :, 4] = 0.0
    far[0, 0] = torch.tensor([5000., 5000., 1500., 1500., 0.9, 0.95, 0.01, 0.01])
    far[0, 1] = torch.tensor([904., 904., 1500., 1500., 0.8, 0.01, 0.95, 0.01])
    far[0, 2] = torch.tensor([100., 100., 50., 60., 0.7, 0.01, 0.01, 0.95])
    far[1, 0] = torch.tensor([300., 300., 80., 80., 0.9, 0.95, 0.9, 0.01])
    far[1, 1] = torch.tensor([302., 301., 80., 80., 0.8, 0.95, 0.01, 0.01])
    want = oracle.non_max_suppression(far.numpy(), 0.3, 0.6, multi_label=True)
    assert len(want[0]) == 2 and len(want[1]) == 2           # the cross-class suppression happened; image 1: 3 candidates -> 2
    got = non_max_suppression(far.cuda(), 0.3, 0.6, multi_label=True)
    for i in range(2):
        _compare(got[i], want[i], 'far-out img %d' % i)
    # without the far-out boxes the same call stays class by class (and still agrees)
    far[0, 0, 4] = 0.0
    want = oracle.non_max_suppression(far.numpy(), 0.3, 0.6, multi_label=True)
    got = non_max_suppression(far.cuda(), 0.3, 0.6, multi_label=True)
    for i in range(2):
        _compare(got[i], want[i], 'near img %d' % i)


@pytest.mark.parametrize('rel,size,precision', [('yolov3/yolov3.cfg', 320, 'fp16'), ('yolov3tiny/yolov3-tiny-hand.cfg', 416, 'fp32'),
                                                ('yolov4tiny/yolov4-tiny.cfg', 416, 'fp16')], ids=['yolov3-320-fp16', 'tiny-hand-416-fp32', 'v4tiny-416-fp16'])
@pytest.mark.parametrize('ml', [False, True], ids=['best-class', 'multi-label'])
def test_decode_fused_into_the_candidate_pass_gives_the_two_pass_detections(rel, size, precision, ml, cfg_dir, monkeypatch):
    """Round 5: models.Darknet.hip_detect = forward + NMS as one engine call.  The plan stops at the head
    convolutions; yh_yolo_decode_candidates decodes only the rows whose objectness passes the threshold, straight into the candidate
    records - the (N, rows, 5 + nc) tensor of models.py:418 / :554 is never written.  Same list of (n_i, 6) tensors as
    non_max_suppression(model(x)[0], ...), BIT for bit; also with a class filter, and when the batch is split into image chunks."""
    from engine import nms as hnms
    from utils.utils import non_max_suppression
    if not os.path.exists(os.path.join(cfg_dir, rel)):
        pytest.skip('cfg not in this checkout')
    model = build_mirror(cfg_dir, rel, size).cuda()
    model.hip_precision = precision
    x = synth.image_batch(5, size, seed=33).cuda()
    with torch.no_grad():
        inf = model(x)[0].clone()
    if ml and inf.shape[2] > 6:      # random heads: every score sits near 0.25 - put the threshold under the 1500 best (row, class) scores
        conf = float((inf[..., 5:] * inf[..., 4:5]).flatten().topk(1500).values[-1])
    else:
        conf = float(inf[..., 4].flatten().float().quantile(0.97))      # ~ 3 % of the rows carry objectness
    monkeypatch.setattr(hnms, '_density', {})
    two = non_max_suppression(inf, conf, 0.6, multi_label=ml)
    with torch.no_grad():
        one = model.hip_detect(x, conf, 0.6, multi_label=ml)
        _same(one, two, 'fused vs two-pass')
        assert sum(0 if d is None else d.shape[0] for d in one) >= 20
        nc = inf.shape[2] - 5
        if nc > 1:
            pick = [0, nc - 1]
            _same(model.hip_detect(x, conf, 0.6, multi_label=ml, classes=pick), non_max_suppression(inf, conf, 0.6, multi_label=ml, classes=pick),
                  'class filter')
        monkeypatch.setattr(hnms, '_density', {})
        mmax = max(int((inf[i, :, 4] > conf).sum()) for i in range(5)) * (nc if ml else 1)
        cap = hnms._pow2_at_least(max(mmax, 1))
        monkeypatch.setattr(hnms, '_WORK_BUDGET', 2 * (cap * ((cap + 63) // 64) * 8 + cap * 23 * 4 + 14 * cap) + 4096)      # two images per pass
        _same(model.hip_detect(x, conf, 0.6, multi_label=ml), two, 'image chunks')
        monkeypatch.undo()
        monkeypatch.setenv('YOLO_HIP_FUSED_DETECT', '0')
        _same(model.hip_detect(x, conf, 0.6, multi_label=ml), two, 'switched off')


def test_fused_detection_on_the_int8_engine_and_on_the_eager_path():
    """The int8 engine's heads are fp32 too (the head convolution dequantises): hip_detect equals the two calls there as well; on a CPU
    model (eager modules) hip_detect IS the two calls."""
    from utils.utils import non_max_suppression
    from test_ptq import build_qmodel
    from ptq_minicfg import SIZE
    m, _ = build_qmodel()
    x = synth.image_batch(3, SIZE, seed=7)
    with torch.no_grad():
        cpu_inf = m(x)[0]
        conf = float(cpu_inf[..., 4].flatten().quantile(0.95)) * 0.999
        _same(m.hip_detect(x, conf, 0.6), non_max_suppression(cpu_inf, conf, 0.6, multi_label=False), 'eager')
        m.cuda()
        inf = m(x.cuda())[0].clone()
        assert m.__dict__['_hip_engine'].precision == 'int8'
        two = non_max_suppression(inf, conf, 0.6, multi_label=False)
        assert sum(0 if d is None else d.shape[0] for d in two) >= 5
        _same(m.hip_detect(x.cuda(), conf, 0.6), two, 'int8 fused vs two-pass')
