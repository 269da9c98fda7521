"""COS-PTQ calibration with this package's own ``utils/quantized/quantized_ptq_cos.py``.

* everywhere: the recipe of ``tests/golden/make_golden_ptq.py`` (which ran the REFERENCE's calibration) repeated with the
  native modules must reproduce the stored scales, grid weights / biases and eval outputs bit for bit;
* everywhere: nets the reference cannot calibrate (max-pools: yolov3-tiny, yolov4-tiny) calibrate here and the calibrated
  graph lowers to the int8 engine (host emulation of the C ABI);
* build container only: side by side with the reference's module on the same weights and batches, every tensor of the
  ``state_dict`` (incl. the bias-corrected float biases and all ranges) and the eval output are identical.
"""
import copy
import os

import numpy as np
import pytest
import torch

import conftest
import fakelib
import refharness
import synth
from ptq_minicfg import SIZE, mini_cfg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ptq_mini.npz')


@pytest.fixture(autouse=True)
def _native_module():
    """tests/test_ptq.py installs an eval-only stand-in under the same module name; make sure the package's own is used."""
    import sys
    for name in ('utils.quantized.quantized_ptq_cos', 'utils.quantized'):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, '__file__', '').startswith(conftest.PKG):
            del sys.modules[name]
    import utils
    if hasattr(utils, 'quantized') and not getattr(utils.quantized, '__file__', '').startswith(conftest.PKG):
        del utils.quantized
    import utils.quantized.quantized_ptq_cos as q
    assert q.__file__.startswith(conftest.PKG)
    yield


def _copy_float_weights(fm, qm):
    """What load_darknet_weights(quant=True) does (reference models.py:610-628): BN tensors land on the conv itself."""
    for f, q in zip(fm.module_list, qm.module_list):
        if isinstance(f, torch.nn.Sequential) and len(f) and isinstance(f[0], torch.nn.Conv2d):
            qc = q[0]
            qc.weight.data.copy_(f[0].weight.data)
            if len(f) > 1 and isinstance(f[1], torch.nn.BatchNorm2d):
                qc.gamma.data.copy_(f[1].weight.data)
                qc.beta.data.copy_(f[1].bias.data)
                qc.running_mean.copy_(f[1].running_mean)
                qc.running_var.copy_(f[1].running_var)
            else:
                qc.bias.data.copy_(f[0].bias.data)
                qc.gamma.data.zero_()
                qc.beta.data.zero_()


def test_native_calibration_reproduces_the_reference_fixture():
    import models
    fx = np.load(GOLD)
    torch.set_num_threads(8)
    torch.manual_seed(0)
    fm = models.Darknet(mini_cfg(), (SIZE, SIZE))
    state = synth.randomize_bn_(fm.state_dict(), seed=1)
    synth.trained_like_heads_(state, fm.module_defs)
    fm.load_state_dict(state)
    qm = models.Darknet(mini_cfg(), (SIZE, SIZE), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    _copy_float_weights(fm, qm)
    qm.train()
    with torch.no_grad():
        for it in range(6):
            qm(synth.image_batch(4, SIZE, seed=100 + it))
    qm.eval()
    sd = qm.state_dict()
    keys = [k[3:] for k in fx.files if k.startswith('sd.')]
    assert len(keys) > 40
    for k in keys:
        assert np.array_equal(sd[k].numpy(), fx['sd.' + k].reshape(sd[k].shape)), k
    with torch.no_grad():
        inf, raws, _ = qm(synth.image_batch(2, SIZE, seed=7))
    assert torch.equal(inf, torch.from_numpy(fx['inf']))
    for i, r in enumerate(raws):
        assert torch.equal(r, torch.from_numpy(fx['raw%d' % i]))


@pytest.mark.parametrize('rel', ['yolov3tiny/yolov3-tiny.cfg', 'yolov4tiny/yolov4-tiny.cfg'])
def test_maxpool_nets_calibrate_and_lower_to_int8(rel, cfg_dir):
    """The reference raises on these (max_pool2d gets the [quantised, float] list)."""
    import models
    from engine.plan import DarknetEngine
    cfg = os.path.join(cfg_dir, rel)
    torch.manual_seed(0)
    fm = models.Darknet(cfg, (96, 96))
    fm.load_state_dict(synth.randomize_bn_(fm.state_dict(), seed=1))
    qm = models.Darknet(cfg, (96, 96), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    _copy_float_weights(fm, qm)
    qm.train()
    with torch.no_grad():
        for it in range(3):
            out = qm(synth.image_batch(2, 96, seed=10 + it))
    assert len(out[0]) == 2 and all(torch.isfinite(p).all() for p in out[0])
    qm.eval()
    x = synth.image_batch(2, 96, seed=99)
    with torch.no_grad():
        want = qm(x)[0]
        flt = fm.eval()(x)[0]
    assert torch.isfinite(want).all()
    assert (want[..., 4:] - flt[..., 4:]).abs().max().item() < 0.1         # a sane int8 version of the float detector
    io = DarknetEngine(qm, precision='int8', lib=fakelib.FakeLib())(x)[0]
    d = (io - want).abs()
    assert (d[..., :4] > 0.05).float().mean().item() <= 0.02 and (d[..., 4:] > 2e-3).float().mean().item() <= 0.02


@pytest.mark.skipif(not refharness.available(), reason='needs the reference tree (build container only)')
@pytest.mark.parametrize('way', [1, 2])
def test_native_calibration_matches_the_reference_module(way):
    import models
    ref = refharness.load()
    torch.manual_seed(0)
    theirs = ref.models.Darknet(mini_cfg(), (SIZE, SIZE), quantized=3, a_bit=8, w_bit=8, shortcut_way=way)
    ours = models.Darknet(mini_cfg(), (SIZE, SIZE), quantized=3, a_bit=8, w_bit=8, shortcut_way=way)
    assert type(theirs.module_list[0][0]).__module__ != type(ours.module_list[0][0]).__module__ or \
        type(theirs.module_list[0][0]) is not type(ours.module_list[0][0])
    assert list(ours.state_dict()) == list(theirs.state_dict())
    g = torch.Generator().manual_seed(1)
    sd = theirs.state_dict()
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            if k.endswith('gamma'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith('beta') or k.endswith('running_mean'):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif k.endswith('running_var'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    theirs.load_state_dict(sd)
    ours.load_state_dict(copy.deepcopy(sd))
    theirs.train()
    ours.train()
    with torch.no_grad():
        for it in range(4):
            x = synth.image_batch(2, SIZE, seed=200 + it)
            theirs(x)
            ours(x)
    a, b = theirs.state_dict(), ours.state_dict()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for m1, m2 in zip(theirs.modules(), ours.modules()):      # the vote histograms are not in the state_dict
        for name in ('scale_list', 'scale_list_x', 'scale_list_a', 'scale_list_sum'):
            if hasattr(m1, name):
                assert getattr(m1, name) == getattr(m2, name), name
    theirs.eval()
    ours.eval()
    x = synth.image_batch(2, SIZE, seed=7)
    with torch.no_grad():
        assert torch.equal(theirs(x)[0], ours(x)[0])


@pytest.mark.gpu
def test_calibration_on_the_gpu_then_int8_engine(cfg_dir):
    """The PTQ.py flow on a GPU: calibrate a max-pool net with train-mode forwards on CUDA tensors, switch to eval (the int8
    MFMA engine takes over) and compare with the same calibrated modules evaluated on the CPU."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import models
    cfg = os.path.join(cfg_dir, 'yolov3tiny/yolov3-tiny.cfg')
    torch.manual_seed(0)
    fm = models.Darknet(cfg, (96, 96))
    fm.load_state_dict(synth.randomize_bn_(fm.state_dict(), seed=1))
    qm = models.Darknet(cfg, (96, 96), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    _copy_float_weights(fm, qm)
    host = copy.deepcopy(qm).train()             # the same calibration on the host, for the scale decisions
    qm.cuda().train()
    with torch.no_grad():
        for it in range(3):
            qm(synth.image_batch(2, 96, seed=10 + it).cuda())
            host(synth.image_batch(2, 96, seed=10 + it))
    # device-calibrated against host-calibrated state: every scale is a power-of-two decision taken from float tensors that differ
    # in their last bits between the two (MIOpen / oneDNN convolutions), so all but a few boundary votes must coincide
    sd_d, sd_h = qm.state_dict(), host.state_dict()
    scales = [k for k in sd_h if k.endswith('scale') or 'scale_' in k.rsplit('.', 1)[-1]]
    assert len(scales) >= 20
    same = sum(bool(torch.equal(sd_d[k].cpu(), sd_h[k])) for k in scales)
    ratio = max(float((sd_d[k].cpu().double() / sd_h[k].double()).max()) for k in scales), \
        min(float((sd_d[k].cpu().double() / sd_h[k].double()).min()) for k in scales)
    print('calibration on the GPU vs on the host: %d of %d scale decisions identical, ratio range %.3g .. %.3g' % (same, len(scales), ratio[1], ratio[0]))
    # measured: 39 of 41 identical; the two others are near-ties of the cosine search between candidates up to two steps apart
    assert same >= 0.9 * len(scales) and ratio[0] <= 4.0 and ratio[1] >= 0.25
    qm.eval()
    x = synth.image_batch(2, 96, seed=99)
    with torch.no_grad():
        got = qm(x.cuda())[0].cpu()
    eng = qm.__dict__['_hip_engine']
    assert eng is not None and eng.precision == 'int8'
    cpu = copy.deepcopy(qm).cpu()
    cpu.__dict__['_hip_engine'] = None
    with torch.no_grad():
        want = cpu(x)[0]
    d = (got - want).abs()
    assert (d[..., :4] > 0.05).float().mean().item() <= 0.02 and (d[..., 4:] > 2e-3).float().mean().item() <= 0.02


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['v3_608', 'v4_640'])
def test_device_calibration_at_the_baseline_shapes_then_int8_engine(tag):
    """`PTQ.py`-style calibration of the BASELINE networks on the GPU - YOLOv3-608 and YOLOv4-640, batch 2,
    3 calibration batches, every scale search and every calibration-mode convolution through engine/calib.py (csrc/calib.hip,
    yh_conv2d_fwd fp32) - against the scale decisions of the SAME recipe on the host (tests/golden/ptq_calib_<tag>.npz, written by
    tests/golden/make_golden_ptq608.py with the module's host loop, i.e. the reference's loop quantized_ptq_cos.py:64-93): all but
    the documented cosine near-ties coincide.  Then the calibrated graph in eval mode: the int8 MFMA engine's raw heads equal the
    calibrated modules' own CPU evaluation bit for bit on dyadic frames (the stem's fp32 conv is exact there)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_ptq608', os.path.join(os.path.dirname(GOLD), 'make_golden_ptq608.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rel, size = gen.CASES[tag]
    gold = np.load(os.path.join(os.path.dirname(GOLD), 'ptq_calib_%s.npz' % tag))
    # every search of the run is ALSO evaluated by the reference's loop (fp32 torch reductions) on the same device tensor: a decision
    # that differs must be a tie of that loop's own cosines
    import utils.quantized.quantized_ptq_cos as q
    from engine import calib
    log = []
    dev_search = q._search

    def both(t, first_step, n, bits):
        j_dev, cos_dev = dev_search(t, first_step, n, bits)
        os.environ['YOLO_PTQ_HOST_SEARCH'] = '1'
        try:
            j_host, cos_host = dev_search(t, first_step, n, bits)
        finally:
            del os.environ['YOLO_PTQ_HOST_SEARCH']
        log.append((j_dev, j_host, float(cos_host[j_dev]), float(cos_host[j_host]), float(max(cos_dev))))
        return j_dev, cos_dev
    q._search = both
    try:
        qm, scales = gen.calibrated_scales(rel, size, device='cuda')
    finally:
        q._search = dev_search
    differ = [(a, b, ca, cb) for a, b, ca, cb, _ in log if a != b]
    gap = max([abs(ca - cb) for _, _, ca, cb in differ] or [0.0])
    print('%s: %d cosine searches on the device, %d decided differently by the fp32 loop on the same tensor; largest cosine gap between the '
          'two choices %.2g' % (tag, len(log), len(differ), gap))
    assert len(log) >= 400 and len(differ) <= 0.1 * len(log) and gap <= 2e-5
    assert set(scales) == set(gold.files) and len(scales) >= 500
    same = sum(bool(np.array_equal(scales[k], gold[k])) for k in scales)
    worst = max(float(np.max(np.maximum(scales[k] / gold[k], gold[k] / scales[k]))) for k in scales if np.all(gold[k] > 0) and np.all(scales[k] > 0))
    steps = np.concatenate([np.abs(np.log2(scales[k] / gold[k])).reshape(-1) for k in scales if np.all(gold[k] > 0) and np.all(scales[k] > 0)])
    print('%s calibration on the GPU vs the host-calibrated golden: %d of %d scale tensors identical, largest ratio %.3g; per scale value: %.1f %% '
          'equal, %.1f %% one power-of-two step apart, %.1f %% two' % (tag, same, len(scales), worst, 100 * np.mean(steps == 0),
                                                                       100 * np.mean(steps == 1), 100 * np.mean(steps == 2)))
    # Against the golden (written in the build container: oneDNN convolutions, fp32 cosine sums) the comparison is statistical: votes
    # over power-of-two candidates whose cosines tie to 1e-6 flip with the last bits of the tensors, and in a 75 / 110-layer net one
    # flipped activation scale changes every tensor behind it.  Measured: ~80 % of the tensors identical, the rest one or two steps.
    # thresholds at 1.5 x the measured share of differing tensors (v3_608: 78.0 % identical, v4_640: 86.1 %)
    assert same >= {'v3_608': 0.67, 'v4_640': 0.79}.get(tag, 0.6) * len(scales) and worst <= 4.0
    qm.eval()
    x = synth.dyadic_frames(synth.image_batch(1, size, seed=77))
    with torch.no_grad():
        io, raws, _ = qm(x.cuda())
    eng = qm.__dict__['_hip_engine']
    assert eng is not None and eng.precision == 'int8'
    cpu = copy.deepcopy(qm).cpu()
    cpu.__dict__['_hip_engine'] = None
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        _, raws_cpu, _ = cpu(x)
    off = [((a.cpu() != b).float().mean().item(), (a.cpu() - b).abs().max().item()) for a, b in zip(raws, raws_cpu)]
    print('%s int8 engine vs the calibrated modules on the CPU, dyadic frame: fraction of head values that differ %s, largest difference %s'
          % (tag, ['%.2g' % f for f, _ in off], ['%.3g' % d for _, d in off]))
    if tag == 'v3_608':      # leaky ReLU: integer arithmetic end to end -> bit for bit
        for a, b in zip(raws, raws_cpu):
            assert torch.equal(a.cpu(), b), (a.cpu() - b).abs().max().item()
    else:
        # Mish is evaluated in fp32 on both sides, by different formulas (the kernel: v n / (n + 2) with n = e^v (e^v + 2) on expf;
        # torch on the CPU: v tanh(softplus(v)) through log1pf / expf / tanhf, ~3 ulp from the true value): a value whose Mish lies
        # within that error of a rounding tie of the next grid comes out one step apart, and a 110-layer quantised net carries such a
        # flip to its heads.  Round 5 measured that the kernel side is not the lever: deciding the tie band by the EXACTLY rounded
        # double form instead (common.h mish_f64, -DYH_QMISH_TIE_F64) left the share of differing head values at the same 0.6 % / 1.4 % /
        # 1.2 %, one grid step at most, and cost the int8 Mish kernels 17 - 40 registers - bit-equality with torch's CPU libm is not
        # reachable by any formula on the GPU.  Bounded at ~2 x the measured share (the synthetic power-of-two state of
        # tests/test_ptq_large.py is bit-equal; the reference itself cannot calibrate this cfg at all).
        assert all(f <= 0.025 and d <= 0.0625 for f, d in off), off


@pytest.mark.gpu
def test_map_protocol_on_the_device_calibrated_yolov4_640_state(cfg_dir):
    """north_star's "mAP@0.5 within +-0.2 pt" for BASELINE config 4 (YOLOv4-640 int8) on a CALIBRATED state.  The
    float graph is conditioned like a trained detector's (layer gains equalised, one favourite class per anchor: synth.py) so that the
    protocol measures box / score fidelity rather than tie-breaking; PTQ.py-style calibration then runs ON THE DEVICE (3 batches of 2,
    engine/calib.py); in eval mode the int8 MFMA engine's detections on two frames are scored by tests/map_protocol.py against the
    detections of the SAME calibrated modules evaluated on the CPU - the reference's arithmetic (quantized_ptq_cos.py:288-296,717;
    this package's modules are bit-equal to it wherever the reference runs, tests/test_ptq_large.py).  Frames on the k / 256 grid:
    the stem convolution is then exact in any order, so what is left is the fp32 Mish by two formulas (0.6 - 1.4 % of the head values
    one grid step apart, test above).  Asserted: |mAP - mAP(reference vs itself)| <= 0.002."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import models
    from map_protocol import map50
    from utils.utils import non_max_suppression
    size = 640
    cfg = os.path.join(cfg_dir, 'yolov4/yolov4.cfg')
    torch.manual_seed(0)
    fm = models.Darknet(cfg, (size, size))
    fm.load_state_dict(synth.randomize_bn_(fm.state_dict(), seed=1))
    x = synth.dyadic_frames(synth.image_batch(2, size, seed=0))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    synth.equalize_bn_gain_(fm.eval(), x)
    fm.load_state_dict(synth.trained_like_heads_(fm.state_dict(), fm.module_defs))
    qm = models.Darknet(cfg, (size, size), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    _copy_float_weights(fm, qm)
    del fm
    qm.cuda().train()
    with torch.no_grad():
        for it in range(3):
            qm(synth.dyadic_frames(synth.image_batch(2, size, seed=10 + it)).cuda())
    qm.eval()
    with torch.no_grad():
        io, raws, _ = qm(x.cuda())
    eng = qm.__dict__['_hip_engine']
    assert eng is not None and eng.precision == 'int8'
    cpu = copy.deepcopy(qm).cpu()
    cpu.__dict__['_hip_engine'] = None
    with torch.no_grad():
        ref_full, raws_cpu, _ = cpu(x)
    off = [(a.cpu() != b).float().mean().item() for a, b in zip(raws, raws_cpu)]
    conf = float(torch.quantile(ref_full[..., 4].flatten(), 0.985)) * 0.999
    gt = non_max_suppression(ref_full.clone(), conf, 0.6, multi_label=False)
    n_gt = sum(0 if g is None else len(g) for g in gt)
    det = non_max_suppression(io, conf * 0.9, 0.6, multi_label=False)
    det_self = non_max_suppression(ref_full.cuda(), conf * 0.9, 0.6, multi_label=False)
    score, perfect = map50(gt, det), map50(gt, det_self)
    print('synthetic mAP@0.5 on the device-calibrated YOLOv4-640 int8 state: engine %.4f, calibrated modules on the CPU against themselves %.4f '
          '(%d ground-truth boxes on %d frames; fraction of head values that differ %s)' % (score, perfect, n_gt, len(gt), ['%.2g' % f for f in off]))
    assert n_gt >= 10
    assert abs(score - perfect) <= 0.002, (score, perfect)


# ------------------------------------------------------------------------------ device calibration services (csrc/calib.hip)
def _calibrate(qm, way_device, monkeypatch, batches=3):
    import utils.quantized.quantized_ptq_cos as q
    from engine import calib
    if way_device:      # route the CPU tensors through the device services, on the host emulation of the C ABI
        monkeypatch.setattr(q, '_on_device', lambda t: torch.is_tensor(t))
        monkeypatch.setattr(calib, '_lib_override', fakelib.FakeLib())
    qm.train()
    with torch.no_grad():
        for it in range(batches):
            qm(synth.image_batch(2, SIZE, seed=10 + it))
    monkeypatch.undo()
    return qm


@pytest.mark.parametrize('way', [1, 2], ids=['shortcut_min', 'shortcut_max'])
def test_device_calibration_services_take_the_same_decisions(way, monkeypatch):
    """The whole calibration (scale votes of every quantiser, shortcut searches, concat ranges, bias correction) through
    engine/calib.py - one-pass cosine search with double sums, abs-max, fp32 HIP convolutions, here all on the emulated C ABI -
    against the reference's loop on the same weights and batches: every scale decision identical."""
    import models
    torch.manual_seed(0)
    fm = models.Darknet(mini_cfg(), (SIZE, SIZE))
    fm.load_state_dict(synth.randomize_bn_(fm.state_dict(), seed=1))
    qm = models.Darknet(mini_cfg(), (SIZE, SIZE), quantized=3, a_bit=8, w_bit=8, shortcut_way=way)
    _copy_float_weights(fm, qm)
    host = _calibrate(copy.deepcopy(qm), False, monkeypatch)
    dev = _calibrate(copy.deepcopy(qm), True, monkeypatch)
    sd_h, sd_d = host.state_dict(), dev.state_dict()
    scales = [k for k in sd_h if k.endswith('scale') or 'scale_' in k.rsplit('.', 1)[-1] or 'float_range' in k]
    assert len(scales) >= 20
    for k in scales:
        assert torch.equal(sd_h[k], sd_d[k]), k
    for k in sd_h:      # grids and corrected biases follow (the emulated fp32 convolution is torch's own)
        if sd_h[k].dtype.is_floating_point:
            assert torch.allclose(sd_h[k], sd_d[k], rtol=1e-5, atol=1e-6), k


@pytest.mark.gpu
def test_cosine_search_kernel_against_the_reference_loop():
    """yh_ptq_cos_search on the MI355X against the modules' own loop (fp32 torch on the CPU) on tensors of several sizes, layouts
    and alignments: same winner unless the loop's two best cosines tie to within fp32 summation noise; cosines within 1e-6."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import utils.quantized.quantized_ptq_cos as q
    from engine import calib
    g = torch.Generator().manual_seed(5)
    cases = []
    for shape, spread in (((2, 32, 52, 52), 1.0), ((1, 255, 13, 13), 6.0), ((3, 7, 5, 11), 0.02), ((1000003,), 3.0), ((2, 16, 40, 40), 40.0)):
        cases.append(torch.randn(*shape, generator=g) * spread)
    cases.append((torch.randn(2, 20, 20, 24, generator=g) * 2).permute(0, 3, 1, 2))     # NHWC-backed NCHW view (dense, not contiguous)
    cases.append(torch.randn(4099, generator=g)[3:])                                     # 4-byte aligned only
    cases.append(torch.zeros(2, 8, 4, 4))                                                # all zero: every cosine 0, first candidate wins
    def exact_cosines(t, first, n, bits):
        """The candidates' cosines with the modules' fp32 element arithmetic and float64 sums (what the kernel computes)."""
        out = []
        for j in range(n):
            scale = torch.zeros(1).add_(2 ** (first + j)) / float(1 << (bits - 1))
            qd = q._fake_quant(t, scale, bits).double().reshape(-1)
            td = t.double().reshape(-1)
            nq, nt = float(qd.norm()), float(td.norm())
            out.append(float(td @ qd) / (nt * nq) if nt > 0 and nq > 0 else 0.0)
        return out

    for t in cases:
        for first, n, bits in ((-5, 15, 8), (0, 8, 8)):
            loop_j, loop_cos = q._search(t, first, n, bits)             # the reference's loop: fp32 sums (torch.cosine_similarity)
            loop_cos = [float(c) for c in loop_cos]
            want_cos = exact_cosines(t, first, n, bits)
            got_j, got_cos = calib.cos_search(t.cuda(), 2.0 ** first / 128.0, n, bits)
            # the kernel against the exact sums: double round-off only
            assert max(abs(a - b) for a, b in zip(got_cos, want_cos)) <= 1e-10, (tuple(t.shape), first)
            assert got_j == max(range(n), key=lambda j: (want_cos[j], -j)), (tuple(t.shape), first)
            # the fp32 loop itself is only accurate to its summation noise (measured: up to 6e-5 on 170 k elements); when it picks
            # another candidate, the two must tie to within that noise
            if got_j != loop_j:
                assert abs(want_cos[got_j] - want_cos[loop_j]) <= 2e-4, (tuple(t.shape), first, got_j, loop_j)
    t = torch.randn(3, 40, 17, 9, generator=g) * 5
    assert float(calib.absmax(t.cuda())) == float(t.abs().max())


def _mobilenet_calibration(monkeypatch, device, size=64):
    """Host calibration and device calibration of yolov3-mobilenet-coco (depthwise + squeeze-excite blocks) on the same batches;
    ``device`` = 'emulated' (CPU tensors through engine/calib.py on the host emulation of the C ABI) or 'cuda'."""
    import models
    import utils.quantized.quantized_ptq_cos as q
    from engine import calib
    cfg = os.path.join(conftest.PKG, 'cfg', 'yolov3-mobilenet', 'yolov3-mobilenet-coco.cfg')
    torch.manual_seed(0)
    fm = models.Darknet(cfg, (size, size))
    fm.load_state_dict(synth.randomize_bn_(fm.state_dict(), seed=1))
    qm = models.Darknet(cfg, (size, size), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    _copy_float_weights(fm, qm)
    host, dev = copy.deepcopy(qm).train(), copy.deepcopy(qm).train()
    if device == 'emulated':
        monkeypatch.setattr(q, '_on_device', lambda t: torch.is_tensor(t))
        monkeypatch.setattr(calib, '_lib_override', fakelib.FakeLib())
    else:
        dev.cuda()
    with torch.no_grad():
        for it in range(2):
            x = synth.image_batch(2, size, seed=10 + it)
            dev(x.cuda() if device == 'cuda' else x)
    monkeypatch.undo()
    with torch.no_grad():
        for it in range(2):
            host(synth.image_batch(2, size, seed=10 + it))
    sd_h, sd_d = host.state_dict(), dev.state_dict()
    scales = [k for k in sd_h if k.endswith('scale') or 'scale_' in k.rsplit('.', 1)[-1]]
    same = sum(bool(torch.equal(sd_d[k].cpu(), sd_h[k])) for k in scales)
    return same, len(scales)


def test_depthwise_graph_calibrates_through_the_device_services(monkeypatch):
    """with the calibration convolutions on engine/calib.py, `PTQ.py --device 0` aborted on every cfg with
    depthwise blocks.  The whole Mobilenetv3 backbone now calibrates through the device services (emulated ABI here, the kernels
    in the GPU tier) and takes the host loop's decisions."""
    same, total = _mobilenet_calibration(monkeypatch, 'emulated')
    # 238 of 244 here: the device search sums the cosines in double, the host loop in fp32 (torch.cosine_similarity) - candidates
    # whose cosines agree to ~1e-7 can swap (documented divergence; YOLO_PTQ_HOST_SEARCH=1 forces the reference's loop)
    assert total >= 60 and same >= 0.95 * total, (same, total)


@pytest.mark.gpu
def test_depthwise_graph_calibrates_on_the_gpu(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    same, total = _mobilenet_calibration(monkeypatch, 'cuda', size=96)
    print('mobilenet calibration on the GPU vs on the host: %d of %d scale decisions identical' % (same, total))
    assert total >= 60 and same >= 0.9 * total


def test_calibration_depthwise_convolution_on_the_emulated_abi(monkeypatch):
    """BNFold_COSPTQuantizedConv2d_For_FPGA is also the class of the depthwise blocks (reference models.py:115-160), so
    ``calib.conv2d`` sees groups == channels on the Mobilenet / GhostNet cfgs; it lowers them onto yh_dwconv2d_fwd (here: the host
    emulation of that entry point) instead of raising."""
    import torch.nn.functional as F
    from engine import calib
    monkeypatch.setattr(calib, '_lib_override', fakelib.FakeLib())
    g = torch.Generator().manual_seed(11)
    for (n, c, h, k, s) in ((2, 16, 20, 3, 1), (1, 40, 13, 5, 2), (2, 12, 9, 3, 2)):
        x = torch.randn(n, c, h, h, generator=g)
        w = torch.randn(c, 1, k, k, generator=g) * (k * k) ** -0.5
        b = torch.randn(c, generator=g)
        want = F.conv2d(x, w, b, s, (k - 1) // 2, 1, c)
        got = calib.conv2d(x, w, b, (s, s), ((k - 1) // 2,) * 2, (1, 1), c)
        assert got.shape == want.shape and torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    with pytest.raises(NotImplementedError):
        calib.conv2d(torch.zeros(1, 8, 8, 8), torch.zeros(8, 2, 3, 3), None, 1, 1, 1, groups=4)


@pytest.mark.gpu
def test_calibration_convolution_runs_on_the_hip_kernels():
    """engine.calib.conv2d (what the calibration-mode modules call for CUDA tensors) against torch's CPU convolution."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.nn.functional as F
    from engine import calib
    g = torch.Generator().manual_seed(9)
    for (n, cin, cout, h, k, s) in ((2, 3, 16, 40, 3, 1), (2, 16, 32, 33, 3, 2), (1, 64, 21, 13, 1, 1), (2, 32, 64, 20, 3, 1)):
        x = torch.randn(n, cin, h, h, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5
        b = torch.randn(cout, generator=g)
        want = F.conv2d(x, w, b, s, (k - 1) // 2)
        got = calib.conv2d(x.cuda(), w.cuda(), b.cuda(), (s, s), ((k - 1) // 2,) * 2)
        assert got.shape == want.shape
        assert (got.cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    # depthwise blocks (PTQ.py --device 0 on the Mobilenet cfgs): groups == channels goes through yh_dwconv2d_fwd
    for (n, c, h, k, s) in ((2, 16, 40, 3, 1), (1, 72, 21, 5, 2), (2, 12, 19, 3, 2)):
        x = torch.randn(n, c, h, h, generator=g)
        w = torch.randn(c, 1, k, k, generator=g) * (k * k) ** -0.5
        b = torch.randn(c, generator=g)
        want = F.conv2d(x, w, b, s, (k - 1) // 2, 1, c)
        got = calib.conv2d(x.cuda(), w.cuda(), b.cuda(), (s, s), ((k - 1) // 2,) * 2, (1, 1), c)
        assert got.shape == want.shape
        assert (got.cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    with pytest.raises(NotImplementedError):     # grouped, not depthwise: no cfg of the reference has one
        calib.conv2d(torch.zeros(1, 8, 8, 8).cuda(), torch.zeros(8, 2, 3, 3).cuda(), None, 1, 1, 1, groups=4)
