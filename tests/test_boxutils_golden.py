"""the host-side box utilities against goldens produced by the REFERENCE (tests/golden/make_golden_boxutils.py).

These functions stay host / torch code (the per-box arithmetic that matters for throughput lives inside the NMS and loss
kernels); what a drop-in needs from them is the reference's exact values, in-place semantics included.  Bit-equal: they are
a handful of IEEE operations in the reference's order.  The GPU variant runs the same torch code on CUDA tensors (what
detect.py / test.py hand them after the HIP NMS) and allows the one ulp a different divide / atan implementation may cost.
"""
import os

import numpy as np
import pytest
import torch

from utils import utils as U

FX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'box_utils.npz'))
T = lambda k: torch.from_numpy(FX[k])


def _check(got, want, exact=True):
    got = got.detach().cpu() if torch.is_tensor(got) else torch.as_tensor(got)
    want = torch.as_tensor(want)
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    # a zero-area box makes the reference's CIoU / GIoU terms 0 / 0: the NaNs must sit in the same places
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    got, want = torch.nan_to_num(got, nan=0.0), torch.nan_to_num(want, nan=0.0)
    if exact:
        assert torch.equal(got, want), float((got.double() - want.double()).abs().max())
    else:
        assert torch.allclose(got, want, rtol=2e-6, atol=1e-6), float((got.double() - want.double()).abs().max())


def _run(dev, exact):
    b = T('xyxy').to(dev)
    _check(U.xyxy2xywh(b.clone()), FX['xyxy2xywh'], exact)
    _check(U.xywh2xyxy(U.xyxy2xywh(b.clone())), FX['xywh2xyxy'], exact)
    det = T('det').to(dev)
    d = det.clone()
    r = U.scale_coords((320, 416), d, (1080, 1920, 3))
    assert r is d, 'scale_coords edits its argument in place and returns it (reference utils.py:138-151)'
    _check(r, FX['scale_coords'], exact)
    _check(U.scale_coords((320, 416), det.clone(), (1080, 1920, 3), ratio_pad=((0.2166, 0.2166), (0.0, 43.0))), FX['scale_coords_pad'], exact)
    _check(U.scale_coords((416, 320), det.clone(), (1333, 750, 3)), FX['scale_coords_portrait'], exact)
    c = det.clone()
    assert U.clip_coords(c, (300, 400)) is None
    _check(c, FX['clip_coords'], exact)
    b1, b2 = T('b1').to(dev), T('b2').to(dev)
    _check(U.box_iou(b1, b2), FX['box_iou'], exact)
    for flag in ('IoU', 'GIoU', 'DIoU', 'CIoU'):
        kw = {} if flag == 'IoU' else {flag: True}
        _check(U.bbox_iou(b1[0], b2, x1y1x2y2=True, **kw), FX['bbox_iou_xyxy_' + flag], exact)
        _check(U.bbox_iou(U.xyxy2xywh(b1)[:40].t(), U.xyxy2xywh(b2)[:40], x1y1x2y2=False, **kw), FX['bbox_iou_xywh_' + flag], exact)
    _check(U.wh_iou(T('wh1').to(dev), T('wh2').to(dev)), FX['wh_iou'], exact)


def test_box_utilities_equal_the_reference_on_cpu():
    _run('cpu', exact=True)
    got = U.xyxy2xywh(FX['xyxy'].copy())           # numpy input stays numpy (reference utils.py:100)
    assert isinstance(got, np.ndarray) and np.array_equal(got, FX['xyxy2xywh_np'])


def test_ap_functions_equal_the_reference():
    p, r, ap, f1, cls = U.ap_per_class(FX['ap_tp'], FX['ap_conf'], FX['ap_pred_cls'], FX['ap_target_cls'])
    for got, key in ((p, 'ap_p'), (r, 'ap_r'), (ap, 'ap_ap'), (f1, 'ap_f1'), (cls, 'ap_cls')):
        assert np.array_equal(np.asarray(got), FX[key]), key
    assert U.compute_ap(FX['ca_recall'], FX['ca_precision']) == float(FX['compute_ap'])


@pytest.mark.gpu
def test_box_utilities_on_cuda_tensors():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    _run('cuda', exact=False)
