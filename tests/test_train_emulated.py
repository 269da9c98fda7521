"""CPU tier: the training-step lowering (engine/train.py) replayed on the host emulation of the C ABI.

What this pins down without a GPU: the forward/backward op sequences, buffer placement (zero-copy concat slices for
activations AND their gradients), write-vs-accumulate decisions for multi-consumer gradients, the dgrad weight image
and the stride-2 dilation geometry, the gradient arena slicing, running-stat updates, and the autograd glue in
``models.Darknet`` — all against eager fp32 autograd.  The kernels themselves are checked on the GPU tier.
"""
import os

import pytest
import torch

import conftest
import fakelib
import synth
import train_harness as th


@pytest.fixture(scope='module')
def mini():
    path = th.write_cfg(th.mini_cfg_text())
    yield path
    os.unlink(path)


@pytest.mark.parametrize('size', [64, 52], ids=['64', '52_odd_grid'])
def test_mini_train_step_matches_eager_autograd(mini, size):
    model = th.build(mini, size)
    x = synth.image_batch(4, size, seed=0)
    raws_ref, grads_ref, m_ref, ws = th.eager_step(model, x)
    raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
    for a, b in zip(raws, raws_ref):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    assert set(grads) == set(grads_ref)
    for k in grads_ref:
        assert th.rel_l2(grads[k], grads_ref[k]) < 2e-5, k
    # BatchNorm running statistics and the step counter follow nn.BatchNorm2d (momentum 0.1, unbiased variance)
    for (k, a), (_, b) in zip(m.state_dict().items(), m_ref.state_dict().items()):
        if 'running' in k:
            assert (a - b).abs().max().item() <= 1e-5 * (b.abs().max().item() + 1), k
        if 'num_batches_tracked' in k:
            assert int(a) == int(b) == 1, k


def test_mini_fp16_storage_stays_close(mini):
    """fp16 activations/gradients with fp32 accumulation: statistical agreement (leaky kinks flip under rounding)."""
    model = th.build(mini, 64)
    x = synth.image_batch(4, 64, seed=0)
    raws_ref, grads_ref, _, ws = th.eager_step(model, x)
    raws, grads, _ = th.engine_step(model, x, ws, 'fp16', lib=fakelib.FakeLib())
    for a, b in zip(raws, raws_ref):
        assert (a - b).abs().max().item() <= 5e-3 * b.abs().max().item()
    for k in grads_ref:
        assert th.cosine(grads[k], grads_ref[k]) > 0.97, k


@pytest.fixture(scope='module')
def mini16():
    """The mini cfg with 16 filters in its first conv (the one-pass first-block backward covers 16 / 32 output channels)."""
    path = th.write_cfg(th.mini_cfg_text().replace('filters=8', 'filters=16', 1))
    yield path
    os.unlink(path)


def test_first_block_backward_in_one_pass_equals_the_four_launch_form(mini16, monkeypatch):
    """csrc/stem_bwd.hip (emulated here): BatchNorm backward + weight gradient of block 0 from ONE pass over dy and z, through the
    closed form dW = gamma invstd (Q - S1/P SX - S2/P R).  Same fp16 step with YOLO_HIP_STEM_BWD=0 (reduce, apply, image copy,
    weight gradient): every gradient of the first block agrees to the fp16 rounding of dz the four-launch form stores (the
    one-pass form keeps dz in fp32), every other gradient is identical (nothing downstream of block 0 changes)."""
    model = th.build(mini16, 52)          # 52: segments of 32 pixels with a 20-pixel tail
    x = synth.image_batch(4, 52, seed=0)
    _, _, _, ws = th.eager_step(model, x)
    raws_a, grads_a, m = th.engine_step(model, x, ws, 'fp16', lib=fakelib.FakeLib())
    plan = m.__dict__['_hip_train_engine']._current
    assert any(what.startswith('stembwd') for what, _ in plan['bwd_ops'])
    assert not any(what in ('dbn0', 'dbnx0', 'image0', 'wgrad0') for what, _ in plan['bwd_ops'])
    monkeypatch.setenv('YOLO_HIP_STEM_BWD', '0')
    raws_b, grads_b, m = th.engine_step(model, x, ws, 'fp16', lib=fakelib.FakeLib())
    plan = m.__dict__['_hip_train_engine']._current
    assert all(what in [w for w, _ in plan['bwd_ops']] for what in ('dbn0', 'dbnx0', 'image0', 'wgrad0'))
    first = [k for k in grads_a if k.startswith('module_list.0.')]
    assert len(first) == 3
    for k in grads_a:
        if k in first:
            assert th.rel_l2(grads_a[k], grads_b[k]) <= 2e-3, (k, th.rel_l2(grads_a[k], grads_b[k]))
        else:
            assert torch.equal(grads_a[k], grads_b[k]), k
    # and against eager fp32 autograd the one-pass form is at least as close as the four-launch form
    _, grads_ref, _, _ = th.eager_step(model, x, ws)
    for k in first:
        assert th.rel_l2(grads_a[k], grads_ref[k]) <= 1.05 * th.rel_l2(grads_b[k], grads_ref[k]) + 1e-4, k


def test_first_layer_statistics_ride_in_its_epilogue(mini16, monkeypatch):
    """csrc/conv_stem_mfma.hip (emulated here): the first-layer kernel emits the BatchNorm partial sums of what it stores
    (yh_conv2d_stem_stats_rows rows of [2][cout], summed by yh_bn_finalize like a conv's stats_ws rows), so the plan has no
    yh_bn_stats pass over the largest tensor of the net (reference models.py:100 computes these statistics inside BatchNorm2d).
    YOLO_HIP_STEM_STATS=0 keeps the separate pass: on fp32 storage both plans produce the same step up to the order of the sums."""
    model = th.build(mini16, 52)
    x = synth.image_batch(4, 52, seed=0)
    _, _, _, ws = th.eager_step(model, x)
    raws_a, grads_a, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
    plan = m.__dict__['_hip_train_engine']._current
    names = [w for w, _ in plan['fwd_ops']]
    assert 'stem0' in names and 'bnfin0' in names and 'bnstat0' not in names
    stem = dict(plan['fwd_ops'])['stem0']
    fin = dict(plan['fwd_ops'])['bnfin0']
    assert stem.stats_ws_floats == fin.nparts * 2 * stem.cout and fin.nparts == 2      # FakeLib reports two rows
    monkeypatch.setenv('YOLO_HIP_STEM_STATS', '0')
    raws_b, grads_b, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
    names = [w for w, _ in m.__dict__['_hip_train_engine']._current['fwd_ops']]
    assert 'bnstat0' in names
    for a, b in zip(raws_a, raws_b):
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
    for k in grads_a:
        assert th.rel_l2(grads_a[k], grads_b[k]) <= 1e-4, (k, th.rel_l2(grads_a[k], grads_b[k]))


def test_next_data_gradient_fused_into_the_first_block_backward(monkeypatch):
    """Darknet-53's opening (3x3 32, then 3x3 / s2 64 as the stem's only consumer): the data gradient of the second conv is computed
    inside the first block's backward pass (yh_stem_bwd dz1 / w1) - no dgrad launch, dy of block 0 never exists.  Same fp16 step with
    YOLO_HIP_STEM_DGRAD=0 (data gradient as a launch of its own, ups = 4 form): identical plans otherwise, gradients equal to the
    rounding of the emulated conv-transpose."""
    c = lambda f, k, st: th._CONV % (f, k, st, 'leaky')
    text = ('[net]\nbatch=1\nwidth=64\nheight=64\nchannels=3\n\n' + c(32, 3, 1) + c(64, 3, 2) + c(32, 1, 1) + c(64, 3, 1)
            + '[shortcut]\nfrom=-3\nactivation=linear\n\n' + th._HEAD + th._YOLO % '3,4,5')
    path = th.write_cfg(text)
    try:
        model = th.build(path, 52)
        x = synth.image_batch(3, 52, seed=0)
        _, grads_ref, _, ws = th.eager_step(model, x)
        raws_a, grads_a, m = th.engine_step(model, x, ws, 'fp16', lib=fakelib.FakeLib())
        ops = [w for w, _ in m.__dict__['_hip_train_engine']._current['bwd_ops']]
        assert 'stembwd_dgrad0' in ops and 'dgrad1' not in ops and 'wgrad1' in ops
        monkeypatch.setenv('YOLO_HIP_STEM_DGRAD', '0')
        raws_b, grads_b, m = th.engine_step(model, x, ws, 'fp16', lib=fakelib.FakeLib())
        ops = [w for w, _ in m.__dict__['_hip_train_engine']._current['bwd_ops']]
        assert 'stembwd0' in ops and 'dgrad1' in ops
        for k in grads_a:
            assert th.rel_l2(grads_a[k], grads_b[k]) <= 1e-3, (k, th.rel_l2(grads_a[k], grads_b[k]))
            if not k.startswith('module_list.0.'):
                assert torch.equal(grads_a[k], grads_b[k]), k
        for k in grads_ref:
            assert th.cosine(grads_a[k], grads_ref[k]) > 0.97, k
    finally:
        os.unlink(path)


def test_loss_backward_writes_the_head_tensors_own_gradient(mini):
    """engine/loss.py takes the NHWC head tensors the raw heads are views of as its autograd inputs (`_head_bases`): the fused loss
    backward then writes the head tensor's own gradient and autograd's slice / view backward (zero fill + strided copy per head)
    never runs.  Same loss, same parameter gradients as with plain view inputs."""
    import copy
    from engine import loss as hloss
    from engine.padded import make_train_engine
    from utils.utils import compute_loss
    model = th.build(mini, 64)
    model.nc, model.gr = 2, 1.0
    model.hyp = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'fl_gamma': 0.0}
    x = synth.image_batch(2, 64, seed=0)
    targets = torch.tensor([[0, 1, 0.5, 0.5, 0.3, 0.4], [1, 0, 0.3, 0.6, 0.2, 0.2], [1, 1, 0.7, 0.2, 0.5, 0.3]])
    outs = []
    for use_bases in (True, False):
        m = copy.deepcopy(model).train()
        os.environ['YOLO_HIP_TRAIN_PRECISION'] = 'fp32'
        try:
            m.__dict__['_hip_train_engine'] = make_train_engine(m, 'fp32', x, lib=fakelib.FakeLib())
            hloss._LIB_OVERRIDE = fakelib.FakeLib()
            raws, _ = m._forward_hip_train(x)
            bases = hloss._head_bases(raws)
            assert bases is not None and all(b.dim() == 4 for b, _, _ in bases)
            if not use_bases:
                raws = [r * 1.0 for r in raws]          # fresh tensors: no longer views of the head tensors
                assert hloss._head_bases(raws) is None
            loss, items = compute_loss(raws, targets, m)
            loss.backward()
        finally:
            hloss._LIB_OVERRIDE = None
            del os.environ['YOLO_HIP_TRAIN_PRECISION']
        outs.append((float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def _gray_model():
    import models
    # linear activations: a leaky kink that flips under a different fp32 summation order moves every upstream gradient by
    # ~1e-3, which would hide what this cfg is for (the 1-channel stem and its weight gradient)
    path = th.write_cfg(th.mini_cfg_text('linear').replace('channels=3', 'channels=1'))
    torch.manual_seed(0)
    model = models.Darknet(path, (64, 64), is_gray_scale=True)
    state = model.state_dict()
    synth.randomize_bn_(state, seed=1)
    model.load_state_dict(state)
    return model.train()


def test_single_channel_input_trains_on_the_hip_path():
    """cfg/yolov3-singlechannel (Darknet(..., is_gray_scale=True)): the first conv reads one image channel; its weight
    gradient runs through the 8-channel NHWC copy with cin_w = 1."""
    model = _gray_model()
    x = torch.rand(4, 1, 64, 64, generator=torch.Generator().manual_seed(3))
    raws_ref, grads_ref, _, ws = th.eager_step(model, x)
    raws, grads, _ = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
    for a, b in zip(raws, raws_ref):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    assert_grads_close(grads, grads_ref, 2e-5)


def test_narrow_net_with_a_64_channel_stride2_conv_builds_and_trains():
    """ADVICE r5 (engine/train.py:322): the one-pass stride-2 data gradient for <= 64 input channels has 4 * cin = 256 rows, but the
    shared zero bias row was sized by the widest conv INPUT rounded to 128 - a narrow (pruned) net whose widest input is <= 128 and
    that has a stride-2 conv with 33 .. 64 inputs failed at plan build.  Such a net: 16 -> 64 -> (s2) 64 -> 32 -> head."""
    c = lambda f, k, s: th._CONV % (f, k, s, 'leaky')
    text = ('[net]\nbatch=1\nwidth=64\nheight=64\nchannels=3\n\n' + c(16, 3, 1) + c(64, 3, 2) + c(64, 3, 2) + c(32, 1, 1) + c(64, 3, 2)
            + th._HEAD + th._YOLO % '3,4,5')
    path = th.write_cfg(text)
    try:
        model = th.build(path, 64)
        x = synth.image_batch(2, 64, seed=0)
        raws_ref, grads_ref, _, ws = th.eager_step(model, x)
        raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
        eng = m.__dict__['_hip_train_engine']
        fused = [v.tpack['fused_m_pad'] for v in eng._current['values'] if v.kind == 'conv' and 'fused_m_pad' in getattr(v, 'tpack', {})]
        assert 256 in fused, fused            # the case the row was too short for
        for a, b in zip(raws, raws_ref):
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
        assert_grads_close(grads, grads_ref, 2e-5)
    finally:
        os.unlink(path)


def test_backward_sums_ride_in_the_data_gradient_that_completes_dy(mini, monkeypatch):
    """Round 6 (yh_conv_desc.bwd_z): when a 1x1 / stride-1 conv's data gradient writes the LAST contribution to the gradient of a
    BatchNorm block, that launch also takes the block's backward sums (sum g, sum g xhat) from the rows it stores, and the block's
    own reduction pass over dy and z is not emitted.  On the host emulation (which carries the sums on every such launch): the plan
    has such launches, each followed at once by the op that adds its rows into dbeta / dgamma; the blocks they serve have no
    reduction pass of their own; and the gradients equal those of the unfused plan (YOLO_HIP_FUSE_DBN=0) to summation order."""
    from engine import hiplib
    model = th.build(mini, 64)
    x = synth.image_batch(4, 64, seed=0)
    raws_ref, grads_ref, m_ref, ws = th.eager_step(model, x)
    raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
    ops = m.__dict__['_hip_train_engine']._current['bwd_ops']
    fused = [i for i, (what, d) in enumerate(ops) if isinstance(d, hiplib.ConvDesc) and d.bwd_z]
    assert fused, 'no data gradient carries backward sums'
    served = set()
    for i in fused:
        what, d = ops[i + 1]
        assert isinstance(d, hiplib.BnBwdReduceDesc) and d.nparts > 0 and what.startswith('dbn'), (what, d.nparts)
        served.add(what)
    for what, d in ops:
        if isinstance(d, hiplib.BnBwdReduceDesc) and what in served:
            assert d.nparts > 0, 'block %s still runs its own reduction pass' % what
    monkeypatch.setenv('YOLO_HIP_FUSE_DBN', '0')
    _, grads_plain, m2 = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
    ops2 = m2.__dict__['_hip_train_engine']._current['bwd_ops']
    assert not [1 for what, d in ops2 if isinstance(d, hiplib.ConvDesc) and d.bwd_z]
    assert len(ops2) == len(ops)            # one small summing op in place of one reduction pass, per served block
    for k in grads_ref:
        assert th.rel_l2(grads[k], grads_plain[k]) < 2e-6, k
    assert_grads_close(grads, grads_ref, 2e-5)


def assert_grads_close(grads, grads_ref, tol):
    """rel-l2 per parameter; gradients that are analytically zero (a BatchNorm bias feeding another BatchNorm through linear
    layers) are compared against the scale of the largest gradient instead of their own round-off."""
    top = max(g.norm().item() for g in grads_ref.values())
    for k, ref in grads_ref.items():
        if ref.norm().item() < 1e-4 * top:
            assert (grads[k] - ref).norm().item() < tol * top, k
        else:
            assert th.rel_l2(grads[k], ref) < tol, k


def test_gradient_accumulation_and_stale_backward(mini):
    from engine.train import TrainEngine
    model = th.build(mini, 64)
    x = synth.image_batch(2, 64, seed=0)
    model.__dict__['_hip_train_engine'] = TrainEngine(model, 'fp32', lib=fakelib.FakeLib())
    ws = None
    for _ in range(2):  # two forward/backward pairs accumulate into .grad like eager autograd
        raws, _ = model._forward_hip_train(x)
        ws = ws or th.loss_weights(raws)
        th.toy_loss(raws, ws).backward()
    g2 = {k: p.grad.clone() for k, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    raws, _ = model._forward_hip_train(x)
    th.toy_loss(raws, ws).backward()
    # BN statistics moved between the passes only through running stats (not used in train mode): grads are 2x
    for k, p in model.named_parameters():
        assert th.rel_l2(g2[k], 2 * p.grad) < 1e-5, k
    # a backward whose forward buffers were overwritten must fail loudly
    raws_a, _ = model._forward_hip_train(x)
    raws_b, _ = model._forward_hip_train(x)
    with pytest.raises(RuntimeError, match='overwritten'):
        th.toy_loss(raws_a, ws).backward()


def test_unsupported_cfg_reports_not_implemented():
    from engine.train import TrainEngine
    from models import Darknet
    cfg_text = th.mini_cfg_text().replace('filters=8\nsize=3\nstride=1', 'filters=12\nsize=3\nstride=1', 1)   # 12 channels: not a multiple of 8
    path = th.write_cfg(cfg_text)
    try:
        model = Darknet(path, (64, 64)).train()
        eng = TrainEngine(model, 'fp32', lib=fakelib.FakeLib())
        with pytest.raises(NotImplementedError, match='multiples'):
            eng._get_plan(torch.zeros(1, 3, 64, 64))
    finally:
        os.unlink(path)


@pytest.mark.parametrize('rel,size', [('yolov3tiny/yolov3-tiny.cfg', 64), ('yolov4/yolov4.cfg', 64), ('yolov4tiny/yolov4-tiny.cfg', 64),
                                      ('yolov3-mobilenet/yolov3-mobilenet-coco.cfg', 128)],
                         ids=['yolov3-tiny', 'yolov4', 'yolov4-tiny', 'mobilenet'])
def test_maxpool_and_mish_graphs_train_on_the_hip_path(rel, size):
    """yolov3-tiny (2/2 and 2/1 zero-edge maxpools) and YOLOv4 (mish, SPP 5/9/13, PAN routes) against eager autograd.
    Maxpool routes gradient by argmax, which flips under round-off when two window entries are within an ulp, so the
    comparison is against an fp64 run like the deep YOLOv3 test."""
    cfg = os.path.join(conftest.PKG, 'cfg', rel)
    model = th.build(cfg, size)
    x = synth.image_batch(4, size, seed=0)
    raws64, grads64, _, ws = th.eager_step(model, x, dtype=torch.float64)
    _, grads32, _, _ = th.eager_step(model, x, ws=ws)
    raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
    for a, b in zip(raws, raws64):
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item()
    num = sum((grads[k] - grads64[k]).norm().item() ** 2 for k in grads64) ** 0.5
    den = sum((grads32[k] - grads64[k]).norm().item() ** 2 for k in grads64) ** 0.5
    tot = sum(grads64[k].norm().item() ** 2 for k in grads64) ** 0.5
    assert num <= 4 * den + 2e-4 * tot, (num / tot, den / tot)
    kinds = [w.rstrip('0123456789') for w, _ in m.__dict__['_hip_train_engine']._current['bwd_ops']]
    assert kinds.count('dpool') == {'yolov3tiny/yolov3-tiny.cfg': 6, 'yolov4/yolov4.cfg': 3, 'yolov4tiny/yolov4-tiny.cfg': 3}.get(rel, 0)
    if 'mobilenet' in rel:
        assert kinds.count('dwwgrad') == 15 and kinds.count('dse') == 8


def test_yolov3_train_plan_against_fp64(mini):
    """Full YOLOv3 graph at 64x64: random-weight fp32 gradients are ill-conditioned (eager fp32 vs fp64 differ by
    ~1e-2), so the engine's error against an fp64 eager run is bounded by a multiple of eager fp32's own error."""
    cfg = os.path.join(conftest.PKG, 'cfg', 'yolov3', 'yolov3.cfg')
    model = th.build(cfg, 64)
    x = synth.image_batch(4, 64, seed=0)
    raws64, grads64, _, ws = th.eager_step(model, x, dtype=torch.float64)
    raws32, grads32, _, _ = th.eager_step(model, x, ws=ws)
    raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
    assert len(grads) == 222
    for a, b in zip(raws, raws64):
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()
    num = sum((grads[k] - grads64[k]).norm().item() ** 2 for k in grads64) ** 0.5
    den = sum((grads32[k] - grads64[k]).norm().item() ** 2 for k in grads64) ** 0.5
    tot = sum(grads64[k].norm().item() ** 2 for k in grads64) ** 0.5
    assert num <= 4 * den + 1e-4 * tot, (num / tot, den / tot)
    plan = m.__dict__['_hip_train_engine']._current
    assert plan['zero_list'] == []  # every gradient buffer's first writer covers it: no memsets in the step
    kinds = [w.rstrip('0123456789') for w, _ in plan['bwd_ops']]
    # 69 stride-1 data gradients + 5 stride-2 layers x 4 parity phases
    # five stride-2 layers: the first two (32 and 64 input channels, memory bound) run their four parity phases as one pass (ups = 4;
    # round 5: 64 channels measured 0.642 against 0.704 ms, profiles/r05_fused_dgrad_ab.txt)
    assert kinds.count('wgrad') == 75 and kinds.count('dgrad') == 69 + 3 * 4 + 2
    assert sum(1 for what, d in plan['bwd_ops'] if what.startswith('dgrad') and d.ups == 4) == 2


def test_backward_ranges_hand_gradients_over_early(mini, monkeypatch):
    """The backward plan is cut into ranges (engine/train.py ``_make_segments``): same gradients as one range, and the
    gradients of the last layers reach autograd (hence DDP's bucket hooks) before the first layers' backward has run."""
    from engine.train import TrainEngine
    model = th.build(mini, 64)
    x = synth.image_batch(2, 64, seed=0)
    results = {}
    for nseg in (1, 4):
        monkeypatch.setenv('YOLO_HIP_TRAIN_SEGMENTS', str(nseg))
        m = __import__('copy').deepcopy(model)
        lib = fakelib.FakeLib()
        m.__dict__['_hip_train_engine'] = TrainEngine(m, 'fp32', lib=lib)
        raws, _ = m._forward_hip_train(x)
        plan = m.__dict__['_hip_train_engine']._current
        assert len(plan['segments']) == nseg
        assert [s['ops'][0] for s in plan['segments']] == sorted((s['ops'][0] for s in plan['segments']), reverse=True)
        params = list(m.parameters())
        fired = {}
        n_fwd = len(lib.calls)
        def mark(name, fired=fired, lib=lib, n_fwd=n_fwd):
            def hook(g):
                fired.setdefault(name, len(lib.calls) - n_fwd)
            return hook
        params[-1].register_hook(mark('last'))
        params[0].register_hook(mark('first'))
        ws = th.loss_weights(raws)
        th.toy_loss(raws, ws).backward()
        results[nseg] = ({k: p.grad.clone() for k, p in m.named_parameters()}, fired, len(plan['bwd_ops']))
    g1, _, _ = results[1]
    g4, fired, nops = results[4]
    for k in g1:
        assert torch.equal(g1[k], g4[k]), k
    assert fired['first'] == nops                      # the first layer's gradient needs the whole backward plan
    assert 0 < fired['last'] < nops // 2               # the last layer's gradient is out after the first range


TRAIN_GOLD = [('tinyhand', 'yolov3tiny/yolov3-tiny-hand.cfg', 128, 1), ('v4tiny', 'yolov4tiny/yolov4-tiny.cfg', 128, 80)]
GOLD_HYP = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'fl_gamma': 0.0}


def _golden_step(tag, rel, size, nc, path='eager', device='cpu'):
    """The recipe of tests/golden/make_golden.py train_fixture on this package's Darknet."""
    import numpy as np
    from models import Darknet
    from utils.utils import compute_loss
    from engine.train import TrainEngine
    torch.manual_seed(0)
    model = Darknet(os.path.join(conftest.PKG, 'cfg', rel), (size, size))
    state = model.state_dict()
    synth.randomize_bn_(state, seed=1)
    model.load_state_dict(state)
    model.train().to(device)
    model.nc, model.hyp, model.gr = nc, dict(GOLD_HYP), 1.0
    x = synth.image_batch(4, size, seed=0).to(device)
    targets = synth.loss_inputs(model, size, batch=4, seed=9, labels_per_image=5)[1].to(device)
    if path == 'eager':
        pred, _ = model._forward_eager(x)
    else:
        if path == 'emulated':
            model.__dict__['_hip_train_engine'] = TrainEngine(model, 'fp32', lib=fakelib.FakeLib())
        os.environ['YOLO_HIP_TRAIN_PRECISION'] = 'fp32'
        try:
            pred, _ = model._forward_hip_train(x) if path == 'emulated' else model(x)
        finally:
            del os.environ['YOLO_HIP_TRAIN_PRECISION']
    loss, items = compute_loss(pred, targets, model, fused=False if device == 'cpu' else None)
    loss.backward()
    return model, pred, items


@pytest.mark.parametrize('path', ['eager', 'emulated'])
@pytest.mark.parametrize('tag,rel,size,nc', TRAIN_GOLD, ids=[t[0] for t in TRAIN_GOLD])
def test_training_step_matches_reference_golden(tag, rel, size, nc, path):
    """Row T pinned by the reference itself: train-mode forward + compute_loss + backward of the REFERENCE on CPU
    (tests/golden/train_step.npz) vs this package's eager modules (the training oracle of the GPU tier; 1e-5) and vs the
    HIP lowering replayed on the emulated C ABI (3e-3 of each gradient's scale: a different summation order through
    maxpool argmax routing and 16-sample BatchNorm statistics at this size)."""
    import numpy as np
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'train_step.npz'))
    model, pred, items = _golden_step(tag, rel, size, nc, path)
    tight = path == 'eager'
    np.testing.assert_allclose(items.detach().cpu().numpy(), gold[tag + '_items'], rtol=1e-6 if tight else 2e-5)

    def chk(t):
        t = t.detach().double().cpu()
        return np.array([t.sum().item(), t.abs().sum().item(), t.abs().max().item()])
    for i, p in enumerate(pred):
        np.testing.assert_allclose(chk(p)[1:], gold['%s_raw%d_checks' % (tag, i)][1:], rtol=1e-6 if tight else 2e-5)
    names = [str(n) for n in gold[tag + '_param_names']]
    params = dict(model.named_parameters())
    assert names == list(params)
    for k, want in zip(names, gold[tag + '_grad_checks']):
        g = params[k].grad
        rows = gold['%s_grow_%s' % (tag, k)]
        scale = want[2] + 1e-12
        got = g.reshape(-1)[::211].cpu().numpy()
        if tight:
            np.testing.assert_allclose(got, rows, rtol=0, atol=1e-5 * scale, err_msg=k)
        else:   # single argmax / kink flips move isolated elements: bound the error in the l2 sense
            assert np.linalg.norm(got - rows) <= 3e-3 * np.linalg.norm(rows) + 2e-3 * scale * len(rows) ** 0.5, k
        assert abs(chk(g)[1] - want[1]) <= (1e-5 if tight else 3e-3) * want[1] + 1e-12, k
    rs = {k: v for k, v in model.state_dict().items() if 'running' in k}
    assert [str(n) for n in gold[tag + '_running_names']] == list(rs)
    for v, want in zip(rs.values(), gold[tag + '_running_checks']):
        assert abs(chk(v)[1] - want[1]) <= 1e-5 * want[1] + 1e-9


# ------------------------------------------------------------------------------------------- widths that are not multiples of 8
def _pruned_like_cfg_text():
    """The mini cfg after a slim_prune-style channel prune: arbitrary widths (>= 1), shortcut-connected layers pruned alike
    (prune_utils.merge_mask), route + upsample + concat of two odd-width tensors."""
    c = lambda f, k, s: th._CONV % (f, k, s, 'leaky')
    sc = '[shortcut]\nfrom=-3\nactivation=linear\n\n'
    return ('[net]\nbatch=1\nwidth=64\nheight=64\nchannels=3\n\n'
            + c(5, 3, 1) + c(13, 3, 2) + c(3, 1, 1) + c(13, 3, 1) + sc            # 0-4
            + c(27, 3, 2) + c(9, 1, 1) + c(27, 3, 1) + sc                         # 5-8
            + c(11, 1, 1) + c(30, 3, 1) + th._HEAD + th._YOLO % '3,4,5'           # 9-12
            + '[route]\nlayers = -4\n\n' + c(6, 1, 1) + '[upsample]\nstride=2\n\n'  # 13-15
            + '[route]\nlayers = -1, 4\n\n' + c(10, 1, 1) + c(19, 3, 1) + th._HEAD + th._YOLO % '0,1,2')


def test_odd_width_graph_trains_through_the_padded_twin():
    from engine.padded import PaddedTrainEngine
    path = th.write_cfg(_pruned_like_cfg_text())
    try:
        model = th.build(path, 64)
        x = synth.image_batch(3, 64, seed=0)
        raws_ref, grads_ref, m_ref, ws = th.eager_step(model, x)
        raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
        assert isinstance(m.__dict__['_hip_train_engine'], PaddedTrainEngine)
        for a, b in zip(raws, raws_ref):
            assert a.shape == b.shape
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
        assert set(grads) == set(grads_ref)
        for k in grads_ref:
            assert grads[k].shape == grads_ref[k].shape
            assert th.rel_l2(grads[k], grads_ref[k]) < 5e-5, k
        for (k, a), (_, b) in zip(m.state_dict().items(), m_ref.state_dict().items()):
            if 'running' in k:
                assert (a - b).abs().max().item() <= 1e-5 * (b.abs().max().item() + 1), k
            if 'num_batches_tracked' in k:
                assert int(a) == int(b) == 1, k
        # pad lanes of the twin stay exact zeros after the step (weights, biases, BatchNorm shifts)
        pad = m.__dict__['_hip_train_engine'].pad
        mask = torch.ones(pad.twin_flat.numel(), dtype=torch.bool)
        mask[(pad.push_index < pad.real_numel).cpu()] = False
        assert pad.twin_flat.cpu()[mask].abs().max().item() == 0.0
    finally:
        os.unlink(path)


def test_two_sgd_steps_on_an_odd_width_graph_track_eager():
    import copy
    path = th.write_cfg(_pruned_like_cfg_text())
    try:
        model = th.build(path, 64)
        ref = copy.deepcopy(model)
        from engine.padded import make_train_engine
        dev = copy.deepcopy(model)
        opt_r = torch.optim.SGD(ref.parameters(), lr=1e-3, momentum=0.9)
        opt_d = torch.optim.SGD(dev.parameters(), lr=1e-3, momentum=0.9)
        os.environ['YOLO_HIP_TRAIN_PRECISION'] = 'fp32'
        try:
            for step in range(2):
                x = synth.image_batch(3, 64, seed=10 + step)
                raws = ref._forward_eager(x)[0]
                ws = th.loss_weights(raws)
                opt_r.zero_grad()
                th.toy_loss(raws, ws).backward()
                opt_r.step()
                if dev.__dict__.get('_hip_train_engine') is None:
                    dev.__dict__['_hip_train_engine'] = make_train_engine(dev, 'fp32', x, lib=fakelib.FakeLib())
                raws_d, _ = dev._forward_hip_train(x)
                opt_d.zero_grad()
                th.toy_loss(raws_d, ws).backward()
                opt_d.step()
        finally:
            del os.environ['YOLO_HIP_TRAIN_PRECISION']
        for (k, a), (_, b) in zip(dev.state_dict().items(), ref.state_dict().items()):
            if a.dtype.is_floating_point:
                assert (a - b).abs().max().item() <= 1e-3 * (b.abs().max().item() + 1e-3), k
    finally:
        os.unlink(path)


def test_gray_scale_odd_width_graph_trains_through_the_padded_twin():
    """A slim-pruned single-channel net: the twin must be built gray-scale too (first conv reads ONE image channel); with a
    3-channel twin the stem weight image, the layout descriptor and the first conv's gradient map would disagree."""
    import models
    from engine.padded import PaddedTrainEngine
    text = _pruned_like_cfg_text().replace('channels=3', 'channels=1').replace('activation=leaky', 'activation=linear')
    path = th.write_cfg(text)
    try:
        torch.manual_seed(0)
        model = models.Darknet(path, (64, 64), is_gray_scale=True)
        state = model.state_dict()
        synth.randomize_bn_(state, seed=1)
        model.load_state_dict(state)
        model.train()
        assert model.module_list[0][0].in_channels == 1
        x = torch.rand(3, 1, 64, 64, generator=torch.Generator().manual_seed(3))
        raws_ref, grads_ref, _, ws = th.eager_step(model, x)
        raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
        eng = m.__dict__['_hip_train_engine']
        assert isinstance(eng, PaddedTrainEngine)
        assert eng.pad.twin.is_gray_scale and eng.pad.twin.module_list[0][0].in_channels == 1
        for a, b in zip(raws, raws_ref):
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
        assert_grads_close(grads, grads_ref, 5e-5)
    finally:
        os.unlink(path)


def test_live_running_statistics_edited_between_steps_reach_the_padded_twin():
    """DDP's broadcast_buffers (ranks > 0) and scripts that reset running_mean edit the LIVE buffers in place between steps;
    the twin must start the next step from them instead of overwriting them with its own copy."""
    from engine.padded import make_train_engine
    path = th.write_cfg(_pruned_like_cfg_text())
    try:
        model = th.build(path, 64)
        x = synth.image_batch(3, 64, seed=0)
        model.__dict__['_hip_train_engine'] = make_train_engine(model, 'fp32', x, lib=fakelib.FakeLib())
        model._forward_hip_train(x)
        bn = model.module_list[0][1]
        with torch.no_grad():
            bn.running_mean.fill_(7.0)
            bn.running_var.fill_(3.0)
            bn.num_batches_tracked.zero_()          # a live reset of the step counter must reach the twin as well
        ref_mean = 0.9 * bn.running_mean.clone()
        model._forward_hip_train(x)
        assert int(bn.num_batches_tracked) == 1
        # new = 0.9 * edited + 0.1 * batch statistic: with the edit reverted it would sit near 0.1 * batch, far from 6.3
        assert (bn.running_mean - ref_mean).abs().max().item() < 0.5
        assert bn.running_var.min().item() > 2.0
    finally:
        os.unlink(path)


def _ghost_like_text(second_join=False):
    c = lambda f, k, s: th._CONV % (f, k, s, 'leaky')
    text = ('[net]\nbatch=1\nwidth=32\nheight=32\nchannels=3\n\n' + c(24, 3, 1) + c(12, 1, 1) + c(12, 3, 1) + '[route]\nlayers = -1, -2\n\n'
            + '[shortcut]\nfrom=-4\nactivation=linear\n\n')
    if second_join:   # the same 24-channel conv (block 0) joined to an 8 + 16 concat as well: it cannot have both layouts
        text += c(8, 1, 1) + c(16, 3, 1) + '[route]\nlayers = -1, -2\n\n' + '[shortcut]\nfrom=0\nactivation=linear\n\n'
    return text + th._HEAD + th._YOLO % '0,1,2'


def test_conv_adopts_the_layout_of_the_concat_it_is_added_to():
    """GhostNet-style: a 12 + 12 channel concat (24 of 32 padded lanes) added to a 24-channel conv.  The conv's output layout is
    free, so the twin gives it the concat's layout (engine/padded.py plan_layouts); gradients match eager autograd."""
    import models
    from engine.padded import PaddedTrainEngine
    path = th.write_cfg(_ghost_like_text())
    try:
        torch.manual_seed(0)
        model = models.Darknet(path, (32, 32))
        model.load_state_dict(synth.randomize_bn_(model.state_dict(), seed=2))
        model.train()
        x = synth.image_batch(3, 32, seed=3)
        raws_ref, grads_ref, m_ref, ws = th.eager_step(model, x)
        raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
        eng = m.__dict__['_hip_train_engine']
        assert isinstance(eng, PaddedTrainEngine)
        total = sum(g.norm().item() ** 2 for g in grads_ref.values()) ** 0.5
        for a, b in zip(raws, raws_ref):
            assert (a - b).abs().max().item() <= 5e-5 * b.abs().max().item()
        for k in grads_ref:
            assert (grads[k] - grads_ref[k]).norm().item() <= 1e-4 * grads_ref[k].norm().item() + 1e-6 * total, k
    finally:
        os.unlink(path)


def test_shortcut_between_different_padded_layouts_raises():
    """A conv that two shortcuts would need in two different padded layouts: no twin exists and there is no eager fallback."""
    import models
    from engine.padded import make_train_engine
    path = th.write_cfg(_ghost_like_text(second_join=True))
    try:
        torch.manual_seed(0)
        model = models.Darknet(path, (32, 32)).train()
        with pytest.raises(NotImplementedError, match='layouts'):
            make_train_engine(model, 'fp32', synth.image_batch(2, 32), lib=fakelib.FakeLib())
    finally:
        os.unlink(path)


def test_shortcut_whose_operand_is_a_concat_keeps_the_parts_gradients():
    """conv added to a route of two convs (GhostNet's ghost-module join, here with aligned widths).  The step shares the residual's
    gradient buffer with dy when it can; a concat's gradient buffer is also addressed by the parts produced into it and must not
    be re-pointed (regression: the parts read a stale buffer and every upstream gradient was wrong by O(1))."""
    import models
    c = lambda f, k, s: th._CONV % (f, k, s, 'leaky')
    text = ('[net]\nbatch=1\nwidth=32\nheight=32\nchannels=3\n\n' + c(16, 3, 1) + c(16, 1, 1) + c(16, 3, 1) + '[route]\nlayers = -1, -2\n\n'
            + c(32, 3, 1) + '[shortcut]\nfrom=-2\nactivation=linear\n\n' + th._HEAD + th._YOLO % '0,1,2')
    path = th.write_cfg(text)
    try:
        torch.manual_seed(0)
        model = models.Darknet(path, (32, 32))
        model.load_state_dict(synth.randomize_bn_(model.state_dict(), seed=2))
        model.train()
        x = synth.image_batch(3, 32, seed=3)
        raws_ref, grads_ref, m_ref, ws = th.eager_step(model, x)
        raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=fakelib.FakeLib())
        total = sum(g.norm().item() ** 2 for g in grads_ref.values()) ** 0.5
        for k in grads_ref:
            assert (grads[k] - grads_ref[k]).norm().item() <= 1e-4 * grads_ref[k].norm().item() + 1e-6 * total, k
    finally:
        os.unlink(path)


def test_two_lane_backward_declares_its_dependencies(monkeypatch):
    """The backward plan can run the weight gradients on a side lane (yh_plan_set_lane / yh_plan_add_dep; YOLO_HIP_WGRAD_LANE=1,
    off by default: measured gain < 1 %).  The host emulation
    replays the LATEST schedule the declared dependencies allow; with them the step matches eager autograd, and with the
    write-after-read dependencies of the alternating dz buffers dropped it does not - so the check has teeth."""
    path = th.write_cfg(th.mini_cfg_text())
    try:
        model = th.build(path, 64)
        x = synth.image_batch(3, 64, seed=3)
        raws_ref, grads_ref, m_ref, ws = th.eager_step(model, x)
        total = sum(g.norm().item() ** 2 for g in grads_ref.values()) ** 0.5
        errs = {}
        for mode in ('declared', 'dropped', 'one_lane'):
            lib = fakelib.FakeLib()
            if mode == 'dropped':
                orig = lib.yh_plan_add_dep
                lib.yh_plan_add_dep = lambda h, op, dep, orig=orig, lib=lib: \
                    orig(h, op, dep) if lib.plans[fakelib._addr(h)].get('lane', {}).get(op, 0) == 1 else 0
            monkeypatch.setenv('YOLO_HIP_WGRAD_LANE', '0' if mode == 'one_lane' else '1')
            raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=lib)
            errs[mode] = sum((grads[k] - grads_ref[k]).norm().item() ** 2 for k in grads_ref) ** 0.5 / total
            eng = m.__dict__['_hip_train_engine']
            plan = eng._current
            lanes = lib.plans[fakelib._addr(plan['bwd'])].get('lane', {})
            assert (sum(lanes.values()) > 0) == (mode != 'one_lane')
        assert errs['declared'] <= 1e-5 and errs['one_lane'] <= 1e-5 and errs['dropped'] >= 1e-2, errs
    finally:
        os.unlink(path)
