"""Drop-in boundary: the reference's UNMODIFIED compression scripts run on top of this package.

`slim_prune.py` and `layer_prune.py` of a reference checkout are executed with `runpy` (which, unlike `python script.py`,
does not put the script's directory on sys.path) from a scratch working directory, with this package first on sys.path:
`from models import *`, `from test import test`, `from utils.utils import *` then resolve to THIS package, while
`utils.prune_utils` (out of scope) resolves to the reference's own file through utils/__init__.py.  The scripts
hard-code `.cuda()` on a few masks (slim_prune.py:55,113; prune_utils.py:267,269,441): patched to the identity, everything
runs on the CPU eager path.  Container-only: skipped where /root/reference does not exist (the GPU box).

Checked: the script runs to completion, writes the compact cfg + darknet .weights, the compact cfg builds in this package with
fewer parameters, and the saved file loads back into it.
"""
import glob
import os
import subprocess
import sys

import pytest
import torch

import conftest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'slim_prune.py')), reason='needs the reference checkout')

RUNNER = r'''
import runpy, sys, torch
sys.path.insert(0, %(pkg)r)
torch.Tensor.cuda = lambda self, *a, **k: self          # the scripts call .cuda() on masks unconditionally
torch.manual_seed(0)
sys.argv = %(argv)r
runpy.run_path(%(script)r, run_name='__main__')
'''


def _prepare(tmp_path, tiny_cfg, seed):
    """Scratch cwd with cfg/mini.cfg and a weights file whose BN gammas are spread out (so a global threshold prunes some)."""
    import models
    (tmp_path / 'cfg' / 'mini').mkdir(parents=True)   # the scripts derive their output names from a cfg/<family>/<net>.cfg path
    cfg = tmp_path / 'cfg' / 'mini' / 'mini.cfg'
    cfg.write_text(open(tiny_cfg).read())
    torch.manual_seed(seed)
    model = models.Darknet(str(cfg), (64, 64))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('BatchNorm2d.weight'):
                v.copy_(torch.rand(v.shape, generator=g) * 1.5 + 0.01)
            elif k.endswith('running_var'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith('BatchNorm2d.bias') or k.endswith('running_mean'):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    (tmp_path / 'weights').mkdir()
    wfile = tmp_path / 'weights' / 'mini.weights'
    models.save_weights(model, str(wfile))
    return model, 'cfg/mini/mini.cfg', 'weights/mini.weights'


def _run(script, argv, cwd):
    code = RUNNER % dict(pkg=conftest.PKG, argv=[script] + argv, script=os.path.join(REF, script))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, '-c', code], cwd=str(cwd), capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def _check_outputs(tmp_path, original, pattern_cfg, pattern_w):
    import models
    cfgs = [p for p in glob.glob(str(tmp_path / 'cfg' / '**' / '*.cfg'), recursive=True) if pattern_cfg in os.path.basename(p)]
    weights = [p for p in glob.glob(str(tmp_path / 'weights' / '**' / '*.weights'), recursive=True) if pattern_w in os.path.basename(p)]
    assert len(cfgs) == 1 and len(weights) == 1, (cfgs, weights)
    compact = models.Darknet(cfgs[0], (64, 64))
    n0 = sum(p.numel() for p in original.parameters())
    n1 = sum(p.numel() for p in compact.parameters())
    assert n1 < n0, 'the pruned graph must be smaller (%d vs %d parameters)' % (n1, n0)
    before = [p.clone() for p in compact.parameters()]
    models.load_darknet_weights(compact, weights[0])
    assert any(not torch.equal(a, b) for a, b in zip(before, compact.parameters())), 'the saved weights did not load'
    # the file holds exactly the compact model: header + every parameter / BN statistic, nothing left over
    n_float = sum(p.numel() for p in compact.parameters()) + sum(b.numel() for k, b in compact.named_buffers() if 'running_' in k)
    assert os.path.getsize(weights[0]) == 5 * 4 + 4 * n_float
    with torch.no_grad():
        inf, _, _ = compact.eval()(torch.rand(1, 3, 64, 64))
    assert torch.isfinite(inf).all()
    return compact


def test_reference_slim_prune_runs_unmodified(tmp_path, tiny_cfg, dataset_dir):
    model, cfg, wfile = _prepare(tmp_path, tiny_cfg, seed=0)
    out = _run('slim_prune.py', ['--cfg', cfg, '--data', str(dataset_dir / 'synth.data'), '--weights', wfile, '--percent', '0.5',
                                 '--img-size', '64', '--batch-size', '4'], tmp_path)
    assert 'Compact model has been saved' in out
    compact = _check_outputs(tmp_path, model, 'slim_prune_0.5', 'slim_prune_0.5')
    widths = [m[0].out_channels for m in compact.module_list if isinstance(m, torch.nn.Sequential) and hasattr(m[0], 'out_channels')]
    assert any(w % 8 for w in widths[:-1]), 'slim_prune leaves arbitrary channel counts (%s): the odd-width paths are exercised' % widths
    # BASELINE config 5 continues with a fine-tune of exactly this graph: it must train on the HIP path (through the channel-padded
    # twin, here on the host emulation of the C ABI) and agree with eager autograd on the compact modules
    sys.path.insert(0, os.path.join(conftest.REPO, 'tests'))
    import fakelib
    import synth
    import train_harness as th
    from engine.padded import PaddedTrainEngine
    compact.train()
    x = synth.image_batch(3, 64, seed=3)
    raws_ref, grads_ref, m_ref, ws = th.eager_step(compact, x)
    raws, grads, m = th.engine_step(compact, x, ws, 'fp32', lib=fakelib.FakeLib())
    assert isinstance(m.__dict__['_hip_train_engine'], PaddedTrainEngine)
    total = sum(g.norm().item() ** 2 for g in grads_ref.values()) ** 0.5
    for k in grads_ref:
        assert (grads[k] - grads_ref[k]).norm().item() <= 1e-4 * grads_ref[k].norm().item() + 1e-6 * total, k


def test_reference_layer_prune_runs_unmodified(tmp_path, tiny_cfg, dataset_dir):
    model, cfg, wfile = _prepare(tmp_path, tiny_cfg, seed=2)
    out = _run('layer_prune.py', ['--cfg', cfg, '--data', str(dataset_dir / 'synth.data'), '--weights', wfile, '--shortcuts', '1',
                                  '--img-size', '64', '--batch-size', '4'], tmp_path)
    assert 'Compact model has been saved' in out
    _check_outputs(tmp_path, model, 'layer_prune', 'layer_prune')


def test_reference_ptq_runs_unmodified(tmp_path, tiny_cfg, dataset_dir):
    """The reference's PTQ.py (PTQ.py:12-131), unmodified: float evaluation through this package's test.test, COS-PTQ calibration
    by train-mode forwards of Darknet(quantized=3) (this package's utils/quantized/quantized_ptq_cos.py behind the reference's
    class names), evaluation of the quantised graph, weights/PTQ.pt.  Its star imports (`from models import *`,
    `from utils.datasets import *`, `from utils.utils import *`) must hand it every name it uses."""
    model, cfg, wfile = _prepare(tmp_path, tiny_cfg, seed=4)
    (tmp_path / 'data').mkdir()
    (tmp_path / 'data' / 'synth.data').write_text((dataset_dir / 'synth.data').read_text())
    out = _run('PTQ.py', ['--cfg', cfg, '--data', 'data/synth.data', '--weights', wfile, '--img-size', '64', '--batch-size', '4',
                          '--device', 'cpu'], tmp_path)
    assert 'test original model' in out and 'test quantized model' in out
    saved = tmp_path / 'weights' / 'PTQ.pt'
    assert saved.is_file()
    import models
    state = torch.load(str(saved), map_location='cpu', weights_only=False)['model']
    q = models.Darknet(str(tmp_path / cfg), (64, 64), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    q.load_state_dict(state)
    scales = {k: float(v) for k, v in state.items() if k.endswith('activation_quantizer.scale')}
    assert scales and all(s > 0 for s in scales.values()), 'every activation quantiser settled on a scale'
    with torch.no_grad():
        inf, _, _ = q.eval()(torch.rand(1, 3, 64, 64))
    assert torch.isfinite(inf).all()
