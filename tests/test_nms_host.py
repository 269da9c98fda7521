"""CPU tier: the host side of the NMS (engine/nms.py - candidate bound, the single read, class-by-class form and its hand-over, image
chunks, the head-output source of DarknetEngine.detect) replayed through the host emulation of the C ABI (tests/fakelib.py), against the
reference-generated goldens and the oracle.  The kernels themselves are checked on the GPU tier (tests/test_gpu_nms.py)."""
import glob
import os

import numpy as np
import pytest
import torch

import fakelib
import synth
from oracle import darknet_oracle as oracle
from test_oracle_golden import GOLD, build_mirror

NMS_FIXTURES = sorted(glob.glob(os.path.join(GOLD, 'nms_*.npz')))
SEG = pytest.mark.parametrize('seg', ['', '1', '0'], ids=['auto', 'by-class', 'bit-mask'])


@pytest.fixture()
def fake(monkeypatch):
    from engine import hiplib
    from engine import nms as hnms
    lib = fakelib.FakeLib()
    monkeypatch.setattr(hiplib, 'load', lambda: lib)
    monkeypatch.setattr(hnms, '_density', {})
    return lib


def _compare(got, want, tag):
    if want is None or len(want) == 0:
        assert got is None, tag
        return
    g = got.numpy()
    assert g.shape == want.shape, '%s: %s vs %s' % (tag, g.shape, want.shape)
    np.testing.assert_allclose(g[:, :4], want[:, :4], rtol=0, atol=2e-3, err_msg=tag)
    np.testing.assert_allclose(g[:, 4], want[:, 4], rtol=0, atol=1e-6, err_msg=tag)
    np.testing.assert_array_equal(g[:, 5], want[:, 5], err_msg=tag)


def _same(a, b, tag):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x is None) == (y is None), '%s img %d' % (tag, i)
        if x is not None:
            assert x.shape == y.shape and torch.allclose(x, y, rtol=0, atol=2e-3), '%s img %d' % (tag, i)


@SEG
@pytest.mark.parametrize('path', NMS_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in NMS_FIXTURES])
def test_host_flow_reproduces_the_reference_goldens(path, seg, fake, monkeypatch):
    from engine import nms as hnms
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', seg)
    fx = np.load(path, allow_pickle=False)
    if seg == '0' and bool(fx['multi_label']) and int(fx['nc']) >= 80:
        pytest.skip('an all-pairs IoU matrix over 10^4 candidates in numpy: minutes; the GPU tier runs this case')
    pred = synth.nms_candidates(int(fx['n_img']), int(fx['rows']), int(fx['nc']), int(fx['seed']))
    got = hnms.non_max_suppression(pred, conf_thres=float(fx['conf']), iou_thres=float(fx['iou']), multi_label=bool(fx['multi_label']),
                                   agnostic=bool(fx['agnostic']))
    for i, n in enumerate(fx['counts']):
        _compare(got[i], fx['det%d' % i] if n else None, '%s img %d' % (os.path.basename(path), i))
    by_class = 'nms_class_scan' in fake.calls
    want = (seg == '1' or (seg == '' and bool(fx['multi_label']) and int(fx['nc']) > 1)) and not bool(fx['agnostic'])
    assert by_class == want, (seg, fake.calls)


def test_class_by_class_form_hands_over_and_reads_once(fake, monkeypatch):
    """One host read per call in the steady state; an image the class-by-class form does not cover (a class above 2048 candidates; boxes
    spread over more than 4096 pixels in x and y) sends the batch through the bit-mask steps - a second read - with the oracle's result."""
    from engine import nms as hnms
    reads = []
    orig = torch.Tensor.cpu
    monkeypatch.setattr(torch.Tensor, 'cpu', lambda t, *a, **k: (reads.append(tuple(t.shape)), orig(t, *a, **k))[1])
    pred = synth.nms_candidates(2, 600, 20, 81, n_clusters=15, hot=0.5)
    want = oracle.non_max_suppression(pred.numpy(), 0.05, 0.6, multi_label=True)
    hnms.non_max_suppression(pred, 0.05, 0.6, multi_label=True)
    del reads[:], fake.calls[:]
    got = hnms.non_max_suppression(pred, 0.05, 0.6, multi_label=True)
    assert reads == [(10, 2)] and 'nms_mask' not in fake.calls and 'nms_class_scan' in fake.calls, (reads, fake.calls)
    for i in range(2):
        _compare(got[i], want[i], 'steady state img %d' % i)
    big = synth.nms_candidates(1, 2600, 2, 91, n_clusters=25, hot=0.9)      # two classes, > 2048 candidates each
    want = oracle.non_max_suppression(big.numpy(), 0.02, 0.6, multi_label=True)
    hnms.non_max_suppression(big, 0.02, 0.6, multi_label=True)
    del reads[:], fake.calls[:]
    got = hnms.non_max_suppression(big, 0.02, 0.6, multi_label=True)
    assert reads == [(10, 1), (10, 1)] and fake.calls.count('nms_mask') == 1 and 'nms_class_scan' in fake.calls, (reads, fake.calls)
    _compare(got[0], want[0], 'class overflow')
    far = torch.zeros(2, 64, 8)
    far[0, 0] = torch.tensor([5000., 5000., 1500., 1500., 0.9, 0.95, 0.01, 0.01])
    far[0, 1] = torch.tensor([904., 904., 1500., 1500., 0.8, 0.01, 0.95, 0.01])      # the same box as row 0 after + cls * 4096 (utils.py:840)
    far[0, 2] = torch.tensor([100., 100., 50., 60., 0.7, 0.01, 0.01, 0.95])
    far[1, 0] = torch.tensor([300., 300., 80., 80., 0.9, 0.95, 0.9, 0.01])
    far[1, 1] = torch.tensor([302., 301., 80., 80., 0.8, 0.95, 0.01, 0.01])
    want = oracle.non_max_suppression(far.numpy(), 0.3, 0.6, multi_label=True)
    assert len(want[0]) == 2
    del fake.calls[:]
    got = hnms.non_max_suppression(far, 0.3, 0.6, multi_label=True)
    assert 'nms_mask' in fake.calls
    for i in range(2):
        _compare(got[i], want[i], 'far-out img %d' % i)


@pytest.mark.parametrize('seg', ['', '0'], ids=['by-class', 'bit-mask'])
def test_batches_beyond_the_budget_go_in_image_chunks(seg, fake, monkeypatch):
    from engine import nms as hnms
    monkeypatch.setenv('YOLO_HIP_NMS_SEGMENTED', seg)
    pred = synth.nms_candidates(5, 400, 20, 71, n_clusters=20, hot=0.6)
    want = oracle.non_max_suppression(pred.numpy(), 0.05, 0.6, multi_label=True)
    one = hnms.non_max_suppression(pred, 0.05, 0.6, multi_label=True)
    mmax = int(max(hnms._density[1]) * 400)
    cap = hnms._pow2_at_least(mmax)
    general = seg == '0'
    each = (cap * ((cap + 63) // 64) * 8 if general else 14 * cap) + cap * 23 * 4
    monkeypatch.setattr(hnms, '_density', {})
    monkeypatch.setattr(hnms, '_WORK_BUDGET', 2 * each + 64)
    del fake.calls[:]
    chunked = hnms.non_max_suppression(pred, 0.05, 0.6, multi_label=True)
    assert fake.calls.count('nms_candidates') >= 1 + 3      # the count pass, then three chunks of at most two images
    for i in range(5):
        _compare(one[i], want[i], 'one pass img %d' % i)
        _compare(chunked[i], want[i], 'chunked img %d' % i)
    monkeypatch.setattr(hnms, '_WORK_BUDGET', 1 << 12)
    with pytest.raises(MemoryError):
        hnms.non_max_suppression(pred, 0.05, 0.6, multi_label=True)


def test_hand_over_under_a_tight_budget_terminates(fake, monkeypatch):
    """ADVICE r5 (engine/nms.py:201): the class-by-class form hands a batch over to the bit-mask form; with a density history the bound
    in use is inflated (>= 2 x pow2(mmax)), and when the budget lies between n x per_image(pow2(mmax)) and n x per_image(inflated bound)
    the whole batch used to come back as ONE chunk whose sub-call picked the same inflated bound - RecursionError.  The advisor's case:
    two images, 3000 rows, six classes, one class above 2048 candidates, budget = 2 x per_image(8192) + 1024."""
    from engine import nms as hnms
    pred = synth.nms_candidates(2, 3000, 6, 93, n_clusters=25, hot=0.9)
    want = oracle.non_max_suppression(pred.numpy(), 0.02, 0.6, multi_label=True)
    monkeypatch.setattr(hnms, '_density', {})
    hnms.non_max_suppression(pred, 0.02, 0.6, multi_label=True)          # fills the density history: the next call starts inflated
    mmax = int(max(hnms._density[1]) * 3000)
    cap = hnms._pow2_at_least(mmax)
    assert mmax > 2048 * 2 and hnms._pow2_at_least(int(2 * max(hnms._density[1]) * 3000) + 1) > cap
    each = cap * ((cap + 63) // 64) * 8 + cap * 23 * 4
    monkeypatch.setattr(hnms, '_WORK_BUDGET', 2 * each + 1024)
    del fake.calls[:]
    got = hnms.non_max_suppression(pred, 0.02, 0.6, multi_label=True)
    assert 'nms_mask' in fake.calls and fake.calls.count('nms_class_scan') <= 4, fake.calls
    for i in range(2):
        _compare(got[i], want[i], 'tight budget img %d' % i)
    # a single image whose bit mask does not fit: MemoryError, not recursion
    monkeypatch.setattr(hnms, '_WORK_BUDGET', each // 2)
    with pytest.raises(MemoryError):
        hnms.non_max_suppression(pred, 0.02, 0.6, multi_label=True)


@pytest.mark.parametrize('rel,size', [('yolov3tiny/yolov3-tiny-hand.cfg', 416), ('yolov4tiny/yolov4-tiny.cfg', 416)], ids=['tiny-hand', 'v4tiny'])
@pytest.mark.parametrize('ml', [False, True], ids=['best-class', 'multi-label'])
def test_detect_from_the_head_outputs_equals_decode_then_nms(rel, size, ml, cfg_dir, fake, monkeypatch):
    """DarknetEngine.detect: the plan stops in front of its decode ops, the candidates come from the head convolutions' outputs
    (yh_yolo_decode_candidates with the plan's descriptors: row offsets, pitches, anchors), whole batch and image chunks."""
    from engine import nms as hnms
    from engine.plan import DarknetEngine
    model = build_mirror(cfg_dir, rel, size)
    x = synth.image_batch(3, size, seed=33)
    eng = DarknetEngine(model, precision='fp32', lib=fake)
    io = eng(x)[0]
    nc = io.shape[2] - 5
    if ml and nc > 1:
        conf = float((io[..., 5:] * io[..., 4:5]).flatten().topk(400).values[-1])
    else:
        conf = float(io[..., 4].flatten().quantile(0.985))
    two = hnms.non_max_suppression(io, conf, 0.6, multi_label=ml)
    assert sum(0 if d is None else d.shape[0] for d in two) >= 10
    del fake.calls[:]
    one = eng.detect(x, conf, 0.6, multi_label=ml)
    assert 'decode_candidates' in fake.calls and 'nms_candidates' not in fake.calls
    _same(one, two, 'heads vs decoded tensor')
    mmax = max(int(max(v)) for v in hnms._density.values()) * io.shape[1] + 1
    cap = hnms._pow2_at_least(int(mmax))
    monkeypatch.setattr(hnms, '_density', {})
    monkeypatch.setattr(hnms, '_WORK_BUDGET', 2 * (cap * ((cap + 63) // 64) * 8 + cap * 23 * 4) + 64)
    _same(eng.detect(x, conf, 0.6, multi_label=ml), two, 'image chunks')
    monkeypatch.setenv('YOLO_HIP_FUSED_DETECT', '0')
    del fake.calls[:]
    _same(eng.detect(x, conf, 0.6, multi_label=ml), two, 'switched off')
    assert 'decode_candidates' not in fake.calls
