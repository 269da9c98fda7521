"""Darknet .weights I/O (reference models.py:587-782): byte format pinned to the reference's own writer.

tests/golden/weights_sha.json holds the SHA-256 of files written by the REFERENCE for three nets (plain conv+BN, depthwise +
SE, CSP/Mish) from a key-seeded state; this package must produce the same bytes from the same state, load them back exactly,
and (in the build container) exchange files with the reference in both directions."""
import hashlib
import json
import os
import sys

import pytest
import torch

import conftest

sys.path.insert(0, os.path.join(conftest.REPO, 'tests', 'golden'))
import make_golden_weights as mk  # noqa: E402
import refharness  # noqa: E402

GOLD = json.load(open(os.path.join(conftest.REPO, 'tests', 'golden', 'weights_sha.json')))


def _ours(name):
    import models
    torch.manual_seed(0)
    return mk.keyed_state_(models.Darknet(os.path.join(conftest.PKG, 'cfg', GOLD[name]['cfg']), (416, 416)))


@pytest.mark.parametrize('name', sorted(GOLD))
def test_save_weights_bytes_match_reference_writer(name, tmp_path):
    import models
    model = _ours(name)
    path = str(tmp_path / (name + '.weights'))
    models.save_weights(model, path)
    data = open(path, 'rb').read()
    assert len(data) == GOLD[name]['bytes']
    assert hashlib.sha256(data).hexdigest() == GOLD[name]['sha256'], 'file differs from the reference writer\'s'
    # and the loader reads every tensor back exactly
    torch.manual_seed(123)
    back = models.Darknet(os.path.join(conftest.PKG, 'cfg', GOLD[name]['cfg']), (416, 416))
    models.load_darknet_weights(back, path)
    for (k, a), (_, b) in zip(back.state_dict().items(), model.state_dict().items()):
        if a.dtype.is_floating_point:
            assert torch.equal(a, b), k


def test_header_and_cutoff(tmp_path):
    import numpy as np
    import models
    model = _ours('yolov3-tiny')
    model.seen[0] = 12345
    path = str(tmp_path / 'yolov3-tiny.conv.15')      # the file NAME selects cutoff 15 on load (models.py:591-595)
    models.save_weights(model, path, cutoff=15)
    raw = np.fromfile(path, dtype=np.int32, count=5)
    assert list(raw[:3]) == list(model.version) and int(np.frombuffer(raw[3:5].tobytes(), dtype=np.int64)[0]) == 12345
    torch.manual_seed(5)
    fresh = models.Darknet(os.path.join(conftest.PKG, 'cfg', GOLD['yolov3-tiny']['cfg']), (416, 416))
    keep = {k: v.clone() for k, v in fresh.state_dict().items()}
    models.load_darknet_weights(fresh, path)
    for k, v in fresh.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        idx = int(k.split('.')[1])
        if idx < 15:
            assert torch.equal(v, model.state_dict()[k]), k
        else:
            assert torch.equal(v, keep[k]), 'block %d is past the cutoff and must stay untouched' % idx


@pytest.mark.skipif(not refharness.available(), reason='needs the reference checkout')
@pytest.mark.parametrize('name', sorted(GOLD))
def test_files_cross_load_with_the_reference(name, tmp_path):
    import models
    ref = refharness.load()
    ours = _ours(name)
    torch.manual_seed(0)
    theirs = mk.keyed_state_(ref.models.Darknet(os.path.join(ref.root, 'cfg', GOLD[name]['cfg']), (416, 416)))
    f_ours, f_theirs = str(tmp_path / 'ours.weights'), str(tmp_path / 'theirs.weights')
    models.save_weights(ours, f_ours)
    ref.models.save_weights(theirs, f_theirs)
    assert open(f_ours, 'rb').read() == open(f_theirs, 'rb').read()
    torch.manual_seed(9)
    a = models.Darknet(os.path.join(conftest.PKG, 'cfg', GOLD[name]['cfg']), (416, 416))
    b = ref.models.Darknet(os.path.join(ref.root, 'cfg', GOLD[name]['cfg']), (416, 416))
    models.load_darknet_weights(a, f_theirs)
    ref.models.load_darknet_weights(b, f_ours)
    for (k, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k == kb
        if va.dtype.is_floating_point:
            assert torch.equal(va, vb), k
