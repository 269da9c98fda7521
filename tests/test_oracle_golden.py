"""Pin the CPU oracle (and the host-side eager mirror) against the reference-generated fixtures.

Runs without a GPU and without the reference tree: the fixtures under tests/golden/ were produced by
tests/golden/make_golden.py from the reference's own code in the build container.
"""
import glob
import hashlib
import os

import numpy as np
import pytest
import torch

import synth
from oracle import darknet_oracle as oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _digest(state):
    h = hashlib.sha256()
    for k, v in state.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def _checks(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), t.abs().max().item()])


def build_mirror(cfg_dir, rel, size):
    """Seeded model of this repo with the fixture recipe (seed 0 init, seed 1 BN)."""
    import models
    torch.manual_seed(0)
    m = models.Darknet(os.path.join(cfg_dir, rel), (size, size))
    state = synth.randomize_bn_(m.state_dict(), seed=1)
    m.load_state_dict(state)
    return m.eval()


NET_FIXTURES = sorted(glob.glob(os.path.join(GOLD, 'net_*.npz')))


@pytest.mark.parametrize('path', NET_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in NET_FIXTURES])
def test_network_forward_matches_reference(path, cfg_dir):
    fx = np.load(path, allow_pickle=False)
    rel, size, batch, rs = str(fx['cfg']), int(fx['size']), int(fx['batch']), int(fx['row_stride'])
    if size >= 608 and os.environ.get('YOLO_FAST_TESTS'):
        pytest.skip('fast mode')
    model = build_mirror(cfg_dir, rel, size)
    state = model.state_dict()
    assert _digest(state) == str(fx['weights_sha256']), 'seeded weight recipe drifted from the fixture'
    x = synth.image_batch(batch, size, seed=0)

    # (1) the oracle restatement, BN folded like the HIP packer does
    inf_o, raws_o = oracle.forward(model.module_defs, state, x, fold=True)
    assert tuple(inf_o.shape) == tuple(fx['inf_shape'])
    ref_rows = torch.from_numpy(fx['inf_rows'])
    d = (inf_o[:, ::rs] - ref_rows).abs()
    assert d[..., :4].max().item() <= 1e-3, 'box drift %g px' % d[..., :4].max().item()
    assert d[..., 4:].max().item() <= 1e-5, 'conf drift %g' % d[..., 4:].max().item()
    np.testing.assert_allclose(_checks(inf_o), fx['inf_checks'], rtol=2e-6)
    for i, r in enumerate(raws_o):
        assert tuple(r.shape) == tuple(fx['raw%d_shape' % i])
        np.testing.assert_allclose(_checks(r)[1:], fx['raw%d_checks' % i][1:], rtol=2e-5)

    # (2) the unfolded oracle (conv, then BN) is the reference's exact op order
    inf_u, _ = oracle.forward(model.module_defs, state, x, fold=False)
    du = (inf_u[:, ::rs] - ref_rows).abs()
    assert du[..., :4].max().item() <= 1e-3 and du[..., 4:].max().item() <= 1e-5

    # (3) the eager nn.Module mirror reproduces the reference bit for bit
    with torch.no_grad():
        inf_m, raws_m, _ = model(x)
    assert torch.equal(inf_m[:, ::rs], ref_rows)


NMS_FIXTURES = sorted(glob.glob(os.path.join(GOLD, 'nms_*.npz')))


@pytest.mark.parametrize('path', NMS_FIXTURES, ids=[os.path.basename(p)[4:-4] for p in NMS_FIXTURES])
def test_nms_matches_reference(path):
    from utils.utils import non_max_suppression
    fx = np.load(path, allow_pickle=False)
    pred = synth.nms_candidates(int(fx['n_img']), int(fx['rows']), int(fx['nc']), int(fx['seed']))
    kw = dict(conf_thres=float(fx['conf']), iou_thres=float(fx['iou']), multi_label=bool(fx['multi_label']),
              agnostic=bool(fx['agnostic']))
    got_oracle = oracle.non_max_suppression(pred.numpy(), **kw)
    got_host = non_max_suppression(pred.clone(), **kw)
    for i, n in enumerate(fx['counts']):
        want = fx['det%d' % i]
        for tag, got in (('oracle', got_oracle[i]), ('host', got_host[i])):
            if n == 0:
                assert got is None, tag
                continue
            g = got if isinstance(got, np.ndarray) else got.numpy()
            assert g.shape == want.shape, '%s: %s vs %s' % (tag, g.shape, want.shape)
            np.testing.assert_allclose(g[:, :4], want[:, :4], rtol=0, atol=2e-3, err_msg=tag)
            np.testing.assert_allclose(g[:, 4], want[:, 4], rtol=0, atol=1e-6, err_msg=tag)
            np.testing.assert_array_equal(g[:, 5], want[:, 5], err_msg=tag)


def test_fold_bn_matches_reference():
    from utils.torch_utils import fold_bn
    fx = np.load(os.path.join(GOLD, 'fuse_conv_bn.npz'))
    t = {k: torch.from_numpy(fx[k]) for k in ('w', 'gamma', 'beta', 'mean', 'var')}
    for fn in (oracle.fold_bn, fold_bn):
        w, b = fn(t['w'], None, t['gamma'], t['beta'], t['mean'], t['var'], float(fx['eps']))
        np.testing.assert_allclose(w.numpy(), fx['fused_w'], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(b.numpy(), fx['fused_b'], rtol=1e-6, atol=1e-7)


def test_yolo_decode_matches_reference():
    from models import YOLOLayer
    fx = np.load(os.path.join(GOLD, 'yolo_decode.npz'))
    p = torch.from_numpy(fx['p'])
    io, raw = oracle.yolo_decode(p, fx['anchors'], int(fx['stride']), int(fx['nc']))
    np.testing.assert_allclose(io.numpy(), fx['io'], rtol=1e-6, atol=1e-5)
    np.testing.assert_array_equal(raw.numpy(), fx['raw'])
    layer = YOLOLayer(fx['anchors'], int(fx['nc']), (224, 160), 0, [], int(fx['stride'])).eval()
    io2, raw2 = layer(p.clone(), None)
    np.testing.assert_array_equal(io2.numpy(), fx['io'])
    np.testing.assert_array_equal(raw2.numpy(), fx['raw'])
