"""Shared drivers for the training-path tests — TEST INFRASTRUCTURE ONLY.

``MINI_CFG`` is a 21-block cfg with every structure YOLOv3 trains through (3x3 / 1x1 / stride-2 convs with train-mode
BatchNorm + leaky, residual shortcuts, a multi-consumer tensor, route + 2x upsample + concat, two yolo heads) that is
shallow enough for fp32 gradients to be well conditioned: the eager fp32 autograd result and the HIP path must then
agree to round-off.  (Through the full 75-conv YOLOv3 with random weights the leaky-ReLU kinks make eager fp32 itself
differ from an fp64 run by ~1e-2 relative, so deep-net checks compare error against that eager-vs-fp64 error instead.)
"""
import copy
import os
import tempfile

import torch

import conftest  # noqa: F401
import synth
from models import Darknet

_CONV = '[convolutional]\nbatch_normalize=1\nfilters=%d\nsize=%d\nstride=%d\npad=1\nactivation=%s\n\n'
_HEAD = '[convolutional]\nsize=1\nstride=1\npad=1\nfilters=21\nactivation=linear\n\n'
_YOLO = ('[yolo]\nmask = %s\nanchors = 4,5, 8,10, 13,16, 30,33, 42,40, 50,55\nclasses=2\nnum=6\njitter=.3\n'
         'ignore_thresh = .7\ntruth_thresh = 1\nrandom=1\n\n')


def mini_cfg_text(act='leaky'):
    c = lambda f, k, s: _CONV % (f, k, s, act)
    sc = '[shortcut]\nfrom=-3\nactivation=linear\n\n'
    return ('[net]\nbatch=1\nwidth=64\nheight=64\nchannels=3\n\n'
            + c(8, 3, 1) + c(16, 3, 2) + c(8, 1, 1) + c(16, 3, 1) + sc            # 0-4
            + c(32, 3, 2) + c(16, 1, 1) + c(32, 3, 1) + sc                        # 5-8
            + c(16, 1, 1) + c(32, 3, 1) + _HEAD + _YOLO % '3,4,5'                 # 9-12
            + '[route]\nlayers = -4\n\n' + c(8, 1, 1) + '[upsample]\nstride=2\n\n'  # 13-15
            + '[route]\nlayers = -1, 4\n\n' + c(16, 1, 1) + c(32, 3, 1) + _HEAD + _YOLO % '0,1,2')


def write_cfg(text):
    f = tempfile.NamedTemporaryFile('w', suffix='.cfg', delete=False)
    f.write(text)
    f.close()
    return f.name


def build(cfg_path, size, seed=0):
    torch.manual_seed(seed)
    model = Darknet(cfg_path, (size, size))
    state = model.state_dict()
    synth.randomize_bn_(state, seed=1)
    model.load_state_dict(state)
    return model.train()


def loss_weights(raws, seed=5):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(tuple(p.shape), generator=g) for p in raws]


def toy_loss(raws, ws):
    """Smooth, dense function of every raw head element (linear + small quadratic term)."""
    return sum((p * w.to(p.device, p.dtype)).sum() + 0.005 * (p * p).sum() for p, w in zip(raws, ws))


def eager_step(model, x, ws=None, dtype=torch.float32):
    """Reference: the eager modules + autograd on ``model``'s device (CPU).  Returns raws, grads (by parameter name)."""
    m = copy.deepcopy(model).to(dtype).train()
    for p in m.parameters():
        p.grad = None
    raws = m._forward_eager(x.to(dtype))[0]
    ws = ws or loss_weights(raws)
    toy_loss(raws, ws).backward()
    grads = {k: (p.grad.float() if p.grad is not None else torch.zeros_like(p).float()) for k, p in m.named_parameters()}
    return [r.detach().float() for r in raws], grads, m, ws


def engine_step(model, x, ws, precision, lib=None, device='cpu'):
    """The HIP training path through ``Darknet._forward_hip_train`` (``lib`` = FakeLib on CPU, None = the real library)."""
    from engine.padded import make_train_engine
    m = copy.deepcopy(model).to(device).train()
    for p in m.parameters():
        p.grad = None
    os.environ['YOLO_HIP_TRAIN_PRECISION'] = precision
    try:
        if lib is not None:
            m.__dict__['_hip_train_engine'] = make_train_engine(m, precision, x.to(device), lib=lib)
        raws, feats = m._forward_hip_train(x.to(device))
        toy_loss(raws, ws).backward()
    finally:
        del os.environ['YOLO_HIP_TRAIN_PRECISION']
    grads = {k: (p.grad.float().cpu() if p.grad is not None else torch.zeros_like(p).float().cpu())
             for k, p in m.named_parameters()}
    return [r.detach().float().cpu() for r in raws], grads, m


def rel_l2(a, b):
    return (a - b).norm().item() / (b.norm().item() + 1e-20)


def cosine(a, b):
    return (a.flatten() @ b.flatten()).item() / (a.norm().item() * b.norm().item() + 1e-20)
