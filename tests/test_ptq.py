"""int8 PTQ eval path.

CPU tier: (1) the eval stand-ins of the reference's COS-PTQ modules reproduce the reference's own outputs from the
calibrated fixture bit for bit; (2) the int8 lowering, replayed through the host emulation of the C ABI, tracks
that eager path.  GPU tier: the MFMA-i8 engine against both.
"""
import os

import numpy as np
import pytest
import torch

import fakelib
import ptq_standin
import synth
from ptq_minicfg import SIZE, mini_cfg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ptq_mini.npz')


def build_qmodel():
    import models
    ptq_standin.install()
    torch.manual_seed(0)
    m = models.Darknet(mini_cfg(), (SIZE, SIZE), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    fx = np.load(GOLD)
    return ptq_standin.load_fixture(m, fx).eval(), fx


def _grid_mismatch(a, b, tol):
    d = (a - b).abs()
    return d.max().item(), (d > tol).float().mean().item()


def test_standin_reproduces_reference_eval_bit_exactly():
    m, fx = build_qmodel()
    x = synth.image_batch(2, SIZE, seed=7)
    with torch.no_grad():
        inf, raws, _ = m(x)
    assert torch.equal(inf, torch.from_numpy(fx['inf']))
    for i, r in enumerate(raws):
        assert torch.equal(r, torch.from_numpy(fx['raw%d' % i]))
    # and the calibrated int8 model is a sane detector: close to the float model it was calibrated from
    assert (inf[..., :4] - torch.from_numpy(fx['inf_float'])[..., :4]).abs().max().item() < 4.0


def test_int8_lowering_tracks_reference_eval():
    from engine.plan import DarknetEngine
    m, fx = build_qmodel()
    x = synth.image_batch(2, SIZE, seed=7)
    eng = DarknetEngine(m, precision='int8', lib=fakelib.FakeLib())
    io, raws, _ = eng(x)
    ref = torch.from_numpy(fx['inf'])
    # exact integer accumulation vs the reference's fp32 conv: identical up to rare rounding ties (one grid step)
    mx, frac = _grid_mismatch(io[..., :4], ref[..., :4], 0.05)
    assert mx <= 1.5 and frac <= 0.02, (mx, frac)
    mx, frac = _grid_mismatch(io[..., 4:], ref[..., 4:], 2e-3)
    assert mx <= 0.05 and frac <= 0.02, (mx, frac)
    plan = next(iter(eng._plans.values()))
    kinds = [''.join(c for c in w if not c.isdigit()) for w, _ in plan['ops']]
    # the three quantised shortcuts ride in their producing convs' epilogues (YOLO_HIP_FUSE_QADD=0 keeps them as yh_qadd launches)
    fused = [d for w, d in plan['ops'] if w.startswith('conv') and d.res]
    assert kinds.count('qadd') == 0 and len(fused) == 3 and 'stem' in kinds and 'yolo' in kinds


@pytest.mark.gpu
def test_hip_int8_engine_matches_reference_eval():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from map_protocol import map50
    from utils.utils import non_max_suppression
    m, fx = build_qmodel()
    x = synth.image_batch(2, SIZE, seed=7)
    m.cuda()
    with torch.no_grad():
        io, raws, _ = m(x.cuda())
    torch.cuda.synchronize()
    eng = m.__dict__['_hip_engine']
    assert eng is not None and eng.precision == 'int8'
    io = io.cpu()
    ref = torch.from_numpy(fx['inf'])
    mx, frac = _grid_mismatch(io[..., :4], ref[..., :4], 0.05)
    assert mx <= 1.5 and frac <= 0.02, (mx, frac)
    mx, frac = _grid_mismatch(io[..., 4:], ref[..., 4:], 2e-3)
    assert mx <= 0.05 and frac <= 0.02, (mx, frac)
    # objectness lives on the int8 grid (many ties): put the threshold just below a grid level that keeps ~3 %
    conf = float(torch.quantile(ref[..., 4].flatten(), 0.97)) * 0.999
    gt = non_max_suppression(ref.clone(), conf, 0.6, multi_label=False)
    assert sum(0 if g is None else len(g) for g in gt) >= 10
    det = non_max_suppression(io.cuda(), conf * 0.9, 0.6, multi_label=False)
    assert abs(map50(gt, det) - map50(gt, gt)) <= 0.002


def test_fused_and_unfused_shortcut_lowerings_agree(monkeypatch):
    """The quantised shortcut inside the conv epilogue is yh_qadd's arithmetic on the value the conv would have stored: both plans
    must produce identical int8 graphs (host emulation)."""
    from engine.plan import DarknetEngine
    x = synth.image_batch(2, SIZE, seed=7)
    outs = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('YOLO_HIP_FUSE_QADD', flag)
        m, fx = build_qmodel()
        eng = DarknetEngine(m, precision='int8', lib=fakelib.FakeLib())
        io, raws, _ = eng(x)
        plan = next(iter(eng._plans.values()))
        kinds = [''.join(c for c in w if not c.isdigit()) for w, _ in plan['ops']]
        assert kinds.count('qadd') == (0 if flag == '1' else 3)
        outs[flag] = (io, raws)
    assert torch.equal(outs['1'][0], outs['0'][0])
    for a, b in zip(outs['1'][1], outs['0'][1]):
        assert torch.equal(a, b)
