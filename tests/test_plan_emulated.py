"""CPU tier: the graph lowering (engine/plan.py) replayed through the host emulation of the C ABI.

Validates fusion decisions, zero-copy concat placement, channel padding maps, the packed weight layout
and plan replay against (a) the reference-generated goldens and (b) the eager mirror, without a GPU.
The HIP kernels themselves are checked on the GPU tier (tests/test_gpu_*.py).
"""
import os

import numpy as np
import pytest
import torch

import fakelib
import synth
from engine.plan import DarknetEngine
from test_oracle_golden import GOLD, build_mirror

CASES = ['tiny_hand_416', 'yolov3_320', 'yolov4_320', 'mobilenet_224', 'yolov4tiny_416']


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_lowered_network_matches_reference_golden(name, prec, cfg_dir):
    fx = np.load(os.path.join(GOLD, 'net_%s.npz' % name))
    rel, size, batch, rs = str(fx['cfg']), int(fx['size']), int(fx['batch']), int(fx['row_stride'])
    model = build_mirror(cfg_dir, rel, size)
    x = synth.image_batch(batch, size, seed=0)
    eng = DarknetEngine(model, precision=prec, lib=fakelib.FakeLib())
    io, raws, feats = eng(x)
    assert feats == [] and tuple(io.shape) == tuple(fx['inf_shape'])
    d = (io[:, ::rs] - torch.from_numpy(fx['inf_rows'])).abs()
    box_tol, conf_tol = (1e-3, 1e-5) if prec == 'fp32' else (0.25, 2e-3)
    assert d[..., :4].max().item() <= box_tol and d[..., 4:].max().item() <= conf_tol
    with torch.no_grad():
        _, raws_ref, _ = model(x)  # eager mirror == reference (test_oracle_golden)
    for r, rr in zip(raws, raws_ref):
        assert r.shape == rr.shape and r.is_contiguous()
        assert (r - rr).abs().max().item() <= (2e-4 if prec == 'fp32' else 0.05)
    plan = next(iter(eng._plans.values()))
    kinds = {''.join(c for c in what if not c.isdigit()) for what, _ in plan['ops']}
    allowed = {'stem', 'conv', 'pool', 'yolo', 'dw', 'se'}
    if 'yolov4tiny' in name:   # its stride-16 feature map sits in two concats: the second one has to copy it
        allowed = allowed | {'cat'}
    assert kinds <= allowed, 'shortcut/route/upsample must all be fused: %s' % kinds


def _odd_cfg():
    """Widths that are not multiples of 8, a tensor routed twice, an unfusable shortcut, a lone upsample."""
    net = {'type': 'net', 'width': 64, 'height': 64, 'channels': 3}
    conv = lambda f, k, s=1, act='leaky', bn=1: {'type': 'convolutional', 'batch_normalize': bn, 'filters': f, 'size': k,
                                                 'stride': s, 'pad': 1, 'activation': act}
    anchors = np.array([[10., 13.], [16., 30.], [33., 23.]])
    return [net,
            conv(13, 3),                      # 0  stem, 13 channels -> padded to 16
            conv(21, 3, 2, 'mish'),           # 1
            conv(10, 1, 1, 'relu6'),          # 2
            {'type': 'route', 'layers': [-1, -2]},   # 3  concat 10 + 21 (padded segments)
            conv(21, 3, 1, 'h_swish'),        # 4
            {'type': 'shortcut', 'from': [-4], 'activation': 'linear'},   # 5  fused into conv 4
            conv(21, 1, 1, 'relu'),           # 6  routed below -> its shortcut cannot be fused
            {'type': 'shortcut', 'from': [-2], 'activation': 'linear'},   # 7  explicit add
            {'type': 'route', 'layers': [6, 3]},     # 8  conv 6 and the earlier concat (nested -> copy)
            {'type': 'maxpool', 'size': 2, 'stride': 2},   # 9
            {'type': 'upsample', 'stride': 2},       # 10 lone upsample (input is a pool)
            {'type': 'route', 'layers': [-1, 7]},    # 11
            {'type': 'maxpool', 'size': 5, 'stride': 1},   # 12
            conv(24, 1, 1, 'linear', 0),      # 13 head (3 * (3 + 5))
            {'type': 'yolo', 'mask': [0, 1, 2], 'anchors': anchors, 'classes': 3, 'num': 3}]


@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_odd_widths_and_unfused_paths(prec):
    import models
    torch.manual_seed(5)
    model = models.Darknet(_odd_cfg(), (64, 64))
    model.load_state_dict(synth.randomize_bn_(model.state_dict(), seed=6))
    model.eval()
    x = synth.image_batch(2, 64, seed=7)
    with torch.no_grad():
        ref, raws_ref, _ = model(x)
    eng = DarknetEngine(model, precision=prec, lib=fakelib.FakeLib())
    io, raws, _ = eng(x)
    tol = 2e-4 if prec == 'fp32' else 0.08
    assert (io - ref).abs().max().item() <= tol
    assert (raws[0] - raws_ref[0]).abs().max().item() <= tol
    plan = next(iter(eng._plans.values()))
    kinds = [''.join(c for c in what if not c.isdigit()) for what, _ in plan['ops']]
    assert 'add' in kinds and 'ups' in kinds and 'cat' in kinds  # the non-fused forms were exercised


def _odd_mobile_cfg():
    """Depthwise + squeeze-excite on channel counts that need padding (13, 20) and a 5x5 kernel."""
    net = {'type': 'net', 'width': 64, 'height': 64, 'channels': 3}
    conv = lambda f, k, s=1, act='leaky', bn=1: {'type': 'convolutional', 'batch_normalize': bn, 'filters': f, 'size': k,
                                                 'stride': s, 'pad': 1, 'activation': act}
    dw = lambda f, k, s, act: {'type': 'depthwise', 'batch_normalize': 1, 'filters': f, 'size': k, 'stride': s, 'pad': 1,
                                'activation': act}
    anchors = np.array([[10., 13.], [16., 30.], [33., 23.]])
    return [net, conv(13, 3, 2, 'h_swish'), conv(20, 1, 1, 'relu6'), dw(20, 5, 1, 'relu6'), {'type': 'se', 'filters': 20},
            conv(13, 1, 1, 'linear'), {'type': 'shortcut', 'from': [-5], 'activation': 'linear'},
            conv(24, 1, 1, 'h_swish'), dw(24, 3, 2, 'h_swish'), {'type': 'route', 'layers': [-1, -1]},
            conv(24, 1, 1, 'linear', 0), {'type': 'yolo', 'mask': [0, 1, 2], 'anchors': anchors, 'classes': 3, 'num': 3}]


@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_depthwise_and_se_on_padded_channels(prec):
    import models
    torch.manual_seed(11)
    model = models.Darknet(_odd_mobile_cfg(), (64, 64))
    model.load_state_dict(synth.randomize_bn_(model.state_dict(), seed=12))
    model.eval()
    x = synth.image_batch(2, 64, seed=13)
    with torch.no_grad():
        ref, raws_ref, _ = model(x)
    eng = DarknetEngine(model, precision=prec, lib=fakelib.FakeLib())
    io, raws, _ = eng(x)
    tol = 2e-4 if prec == 'fp32' else 0.08
    assert (io - ref).abs().max().item() <= tol


def _ghost_like_cfg():
    """GhostNet idioms (reference cfg/yolov3-ghostnet): grouped 3x3 convs written as ``convolutional`` + ``groups``, a
    shortcut whose operand is a concat of padded pieces, and shortcuts between tensors of different widths (the sum runs
    over the leading min(Ca, Cb) channels, reference layers.py:65-70)."""
    net = {'type': 'net', 'width': 64, 'height': 64, 'channels': 3}
    conv = lambda f, k, s=1, act='relu', bn=1, g=1: {'type': 'convolutional', 'batch_normalize': bn, 'filters': f, 'size': k,
                                                     'stride': s, 'pad': 1, 'groups': g, 'activation': act}
    anchors = np.array([[10., 13.], [16., 30.], [33., 23.]])
    return [net,
            conv(16, 3, 2),                                   # 0
            conv(12, 1),                                      # 1  primary half of a ghost module
            conv(12, 3, 1, 'relu', 1, 12),                    # 2  cheap half: per-channel 3x3
            {'type': 'route', 'layers': [-1, 1]},             # 3  12 + 12 -> pieces padded to 16 each
            conv(24, 1, 1, 'linear'),                         # 4
            {'type': 'shortcut', 'from': [-2], 'activation': 'linear'},   # 5  operand is the padded concat
            conv(10, 1, 1, 'linear'),                         # 6
            {'type': 'shortcut', 'from': [4], 'activation': 'linear'},    # 7  10 + first 10 of 24
            conv(24, 1),                                      # 8
            {'type': 'shortcut', 'from': [6], 'activation': 'linear'},    # 9  first 10 of 24 += 10
            conv(24, 1, 1, 'linear', 0),                      # 10 head
            {'type': 'yolo', 'mask': [0, 1, 2], 'anchors': anchors, 'classes': 3, 'num': 3}]


@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_ghost_idioms_are_lowered(prec):
    import models
    torch.manual_seed(21)
    model = models.Darknet(_ghost_like_cfg(), (64, 64))
    model.load_state_dict(synth.randomize_bn_(model.state_dict(), seed=22))
    model.eval()
    x = synth.image_batch(2, 64, seed=23)
    with torch.no_grad():
        ref, raws_ref, _ = model(x)
    eng = DarknetEngine(model, precision=prec, lib=fakelib.FakeLib())
    io, raws, _ = eng(x)
    tol = 2e-4 if prec == 'fp32' else 0.08
    assert (io - ref).abs().max().item() <= tol
    assert (raws[0] - raws_ref[0]).abs().max().item() <= tol
    plan = next(iter(eng._plans.values()))
    mapped = [d for what, d in plan['ops'] if what.startswith('add') and d.amap]
    assert len(mapped) == 3


def test_feature_out_on_request(cfg_dir):
    """eval returns (inf_out, raw_p, feature_out); feature_out holds every conv block that does not feed a yolo layer
    (reference models.py:540-543).  The engine only copies them out when asked (Darknet.hip_return_features)."""
    model = build_mirror(cfg_dir, 'yolov3tiny/yolov3-tiny-hand.cfg', 64)
    x = synth.image_batch(2, 64, seed=4)
    with torch.no_grad():
        _, _, want = model(x)
    eng = DarknetEngine(model, precision='fp32', lib=fakelib.FakeLib())
    assert eng(x)[2] == []
    eng.return_features = True
    got = eng(x)[2]
    assert len(got) == len(want) > 0
    for a, b in zip(got, want):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())


def test_raw_head_copies_only_on_request(cfg_dir):
    """The second return value of an eval forward (the raw head maps) is a copy the decode kernel makes; detect.py reads
    model(img)[0] only and switches it off (Darknet.hip_return_raw = False): same detections, no raw tensors, and back on again."""
    model = build_mirror(cfg_dir, 'yolov3tiny/yolov3-tiny-hand.cfg', 64)
    x = synth.image_batch(2, 64, seed=4)
    eng = DarknetEngine(model, precision='fp32', lib=fakelib.FakeLib())
    io, raws, _ = eng(x)
    assert len(raws) == 2 and all(r.dim() == 5 for r in raws)
    eng.want_raw = False
    io2, raws2, _ = eng(x)
    assert len(raws2) == 0 and torch.equal(io, io2)
    eng.want_raw = True
    io3, raws3, _ = eng(x)
    assert torch.equal(io, io3) and all(torch.equal(a, b) for a, b in zip(raws, raws3))


def test_weight_edits_are_picked_up():
    """In-place parameter updates (optimizer step, prune script) must reach the packed weights."""
    import models
    torch.manual_seed(5)
    model = models.Darknet(_odd_cfg(), (64, 64)).eval()
    x = synth.image_batch(1, 64, seed=8)
    eng = DarknetEngine(model, precision='fp32', lib=fakelib.FakeLib())
    a = eng(x)[0].clone()
    with torch.no_grad():
        model.module_list[4][0].weight.mul_(0.5)          # version bump
        model.module_list[6][1].bias.data = model.module_list[6][1].bias.data + 1.0   # rebinding .data
    b = eng(x)[0]
    with torch.no_grad():
        ref = model(x)[0]
    assert (a - b).abs().max().item() > 1e-3
    assert (b - ref).abs().max().item() <= 2e-4


def test_batch_and_shape_change_build_new_plans():
    import models
    torch.manual_seed(5)
    model = models.Darknet(_odd_cfg(), (64, 64)).eval()
    eng = DarknetEngine(model, precision='fp32', lib=fakelib.FakeLib())
    for n, s in ((1, 64), (3, 96), (1, 64)):
        x = synth.image_batch(n, s, seed=9)
        with torch.no_grad():
            ref = model(x)[0]
        assert (eng(x)[0] - ref).abs().max().item() <= 2e-4
    assert len(eng._plans) == 2


def test_plan_cache_is_bounded(cfg_dir):
    """Rectangular evaluation feeds many input shapes: only the most recently used plans (and their buffers) stay."""
    import torch
    import fakelib
    from engine.plan import DarknetEngine
    from models import Darknet
    model = Darknet(os.path.join(cfg_dir, 'yolov3tiny', 'yolov3-tiny-hand.cfg'), (64, 64)).eval()
    eng = DarknetEngine(model, 'fp32', lib=fakelib.FakeLib())
    eng.max_plans = 2
    outs = {}
    for hw in ((64, 64), (64, 96), (96, 64), (64, 64)):
        outs.setdefault(hw, []).append(eng(torch.rand(1, 3, *hw, generator=torch.Generator().manual_seed(0)))[0])
        assert len(eng._plans) <= 2
    assert list(eng._plans)[-1] == (1, 3, 64, 64)
    assert torch.equal(outs[(64, 64)][0], outs[(64, 64)][1])   # the rebuilt plan reproduces the evicted one
