"""Container-only tier: every cfg family the reference ships (reference cfg/*/*.cfg) goes through this repo's builder and
the emulated HIP lowering and is compared with the reference's own ``Darknet`` on the same weights and input.

The reference tree does not exist on the GPU box, so the whole module skips there; the networks the GPU tier runs come
from the committed goldens.  All 39 runnable cfgs are swept (about half a minute); ``YOLO_TEST_ALL_CFGS=0`` keeps one
representative per family that the goldens do not already cover.
"""
import glob
import os

import pytest
import torch

import fakelib
import refharness
import models
from engine.plan import DarknetEngine

pytestmark = pytest.mark.skipif(not refharness.available(), reason='needs the reference tree (build container only)')

ROOT = os.path.join(refharness.REF, 'cfg')
SUBSET = ['yolov2/yolov2.cfg', 'yolov2/yolov2-tiny.cfg', 'yolov3-ghostnet/yolov3-ghost-coco.cfg', 'yolov3/yolov3-spp3.cfg',
          'yolov3tiny-mobilenet-small/yolov3tiny-mobilenet-small-coco.cfg', 'yolov3tiny/yolov3-tiny3.cfg', 'yolov4/yolov4-relu.cfg',
          'yolov3-singlechannel/yolov3-singlechannel.cfg']
BROKEN_IN_REFERENCE = ['yolov3/yolov3-asff.cfg', 'yolov3/yolov3-spp-pan-scale.cfg', 'yolov3/yolov3-spp-matrix.cfg',
                       'yolov3tiny-efficientnetB0/yolov3tiny-efficientnetB0.cfg']


def _cases():
    if os.environ.get('YOLO_TEST_ALL_CFGS', '1') == '1' and os.path.isdir(ROOT):
        every = sorted(os.path.relpath(p, ROOT) for p in glob.glob(os.path.join(ROOT, '*', '*.cfg')))
        return [c for c in every if c not in BROKEN_IN_REFERENCE]
    return SUBSET


def _randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)


@pytest.mark.parametrize('rel', _cases())
def test_cfg_builds_and_lowers_like_the_reference(rel):
    ref = refharness.load()
    cfg = os.path.join(ROOT, rel)
    kw = {'is_gray_scale': True} if 'singlechannel' in rel else {}
    torch.manual_seed(0)
    theirs = ref.models.Darknet(cfg, (128, 128), **kw).eval()
    ours = models.Darknet(cfg, (128, 128), **kw).eval()
    assert [type(m).__name__ for m in ours.module_list] == [type(m).__name__ for m in theirs.module_list]
    assert list(ours.state_dict()) == list(theirs.state_dict()), '.pt compatibility: same keys in the same order'
    ours.load_state_dict(theirs.state_dict())
    _randomize_bn(theirs, 1)
    _randomize_bn(ours, 1)
    x = torch.rand(2, 1 if kw else 3, 128, 128, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want, want_raw, _ = theirs(x)
        got, got_raw, _ = ours(x)
    assert torch.equal(got, want) and all(torch.equal(a, b) for a, b in zip(got_raw, want_raw))
    io, raws, _ = DarknetEngine(ours, precision='fp32', lib=fakelib.FakeLib())(x)      # no NotImplementedError: fully lowered
    assert (io[..., :4] - want[..., :4]).abs().max().item() <= 1e-3 and (io[..., 4:] - want[..., 4:]).abs().max().item() <= 1e-5


@pytest.mark.parametrize('rel', BROKEN_IN_REFERENCE)
def test_cfgs_the_reference_cannot_run_fail_the_same_way(rel):
    """Two cfgs use fields the parser rejects, two build but break in forward (ASFF head shape, PAN route sizes): the error
    type and message are part of the contract too."""
    ref = refharness.load()
    cfg = os.path.join(ROOT, rel)
    outcome = []
    for mod in (ref.models, models):
        try:
            torch.manual_seed(0)
            m = mod.Darknet(cfg, (128, 128)).eval()
            with torch.no_grad():
                m(torch.zeros(1, 3, 128, 128))
            outcome.append(('ran', ''))
        except (AssertionError, RuntimeError) as e:
            outcome.append((type(e).__name__, str(e)))
    assert outcome[0] == outcome[1] and outcome[0][0] != 'ran'
