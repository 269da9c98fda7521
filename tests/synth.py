"""Seeded synthetic weights / inputs / NMS candidates shared by the golden generator and the tests.

Everything is derived from ``torch.Generator().manual_seed(seed)`` on the CPU so that the build
container (where the reference is importable) and the GPU box produce identical tensors.
"""
import math

import numpy as np
import torch


def randomize_bn_(state, seed=1):
    """Make BN non-trivial in a state_dict, in key order: mean~N(0,.1) var~U(.5,1.5) g~U(.5,1.5) b~N(0,.1)."""
    g = torch.Generator().manual_seed(seed)
    for k, v in state.items():
        if k.endswith('running_mean'):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        elif k.endswith('running_var'):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif k.endswith('BatchNorm2d.weight'):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif k.endswith('BatchNorm2d.bias'):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    return state


def calm_bn_(state, gamma_gain=0.05, beta_gain=5.0):
    """The WELL-CONDITIONED training state (round 6).  A deep batch-statistics BatchNorm net with random weights amplifies any perturbation
    exponentially with depth: with ``randomize_bn_``'s gains (gamma ~ U(.5, 1.5)) rounding the weights and frames of Darknet-53 to fp16
    - nothing else - turns the 608 x 608 batch-2 gradient by 20 degrees (rel l2 0.34), YOLOv4's (110 convolutions, Mish) by 70 (rel l2
    1.5): no fp16 engine can be held to anything on such a state, and fp32 engines differ from each other by percents.  Scaling every
    BatchNorm gain by ``gamma_gain`` and every BatchNorm bias by ``beta_gain`` (gamma ~ U(.025, .075), beta ~ N(0, .5)) makes each
    pre-activation gamma xhat + beta a small variation around its channel's bias: the activation works on an almost fixed operating
    point per channel, perturbations are no longer amplified (measured with the reference's own modules: the same fp16 rounding moves
    the gradient by rel l2 1.3e-3 on Darknet-53, 1.8e-3 on YOLOv4), while every kernel still sees both branches of the activation, real
    batch statistics and gradients of ordinary size.  Applied after ``randomize_bn_``; same recipe on the reference's model and ours."""
    for k, v in state.items():
        if k.endswith('BatchNorm2d.weight'):
            v.mul_(gamma_gain)
        elif k.endswith('BatchNorm2d.bias'):
            v.mul_(beta_gain)
    return state


def trained_like_heads_(state, module_defs, margin=8.0):
    """Give every (head, anchor) a clear favourite class, like a trained detector has.

    With random weights the 80 class logits of a cell are near-ties, so a 1e-3 relative perturbation
    (fp16 storage) flips the arg-max class of a few percent of boxes and the mAP protocol would measure
    tie-breaking noise instead of box/score fidelity.  Adding ``margin`` to one class bias per anchor
    makes the class decision robust while leaving boxes, objectness and every conv untouched.
    """
    defs = [d for d in module_defs if d['type'] != 'net']
    head = 0
    for i, d in enumerate(defs):
        if d['type'] != 'yolo':
            continue
        nc = int(d['classes'])
        b = state['module_list.%d.Conv2d.bias' % (i - 1)].view(len(d['mask']), nc + 5)
        for a in range(len(d['mask'])):
            b[a, 5 + ((3 * head + a) * 7) % nc] += margin
        head += 1
    return state


def equalize_bn_gain_(model, x):
    """Rescale every BatchNorm's (gamma, beta) by ONE factor per layer so that its eval-mode output has unit standard deviation
    on the batch ``x`` (single forward pass, in place).  Deep random-weight nets with arbitrary eval statistics lose the signal
    layer by layer (YOLOv4's 110 convs end with every cell of an anchor within 1e-4 of the same objectness, so any rounding
    reorders the whole score ranking); with the layer gains equalised the activations stay O(1) and the outputs vary over the
    image like a trained detector's, while the per-channel statistics keep the well-conditioned ranges of ``randomize_bn_``.
    The factor is rounded to 7 mantissa bits (within 0.4 % of the measured value): the measured standard deviation differs in its
    last bits between CPUs, the rounded factor - and with it every rescaled parameter - does not, so goldens built from this state
    are reproducible on the GPU box."""
    import torch.nn as nn
    hooks = []

    def hook(mod, inp, out):
        mant, expo = math.frexp(max(float(out.double().std()), 1e-6))
        s = math.ldexp(round(mant * 256) / 256, expo)
        mod.weight.data.div_(s)
        mod.bias.data.div_(s)
        return out / s
    for m in model.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            hooks.append(m.register_forward_hook(hook))
    was_training = model.training
    model.eval()
    with torch.no_grad():
        model._forward_eager(x) if hasattr(model, '_forward_eager') else model(x)
    for h in hooks:
        h.remove()
    model.train(was_training)
    return model


def spread_objectness_(state, module_defs, gain=40.0):
    """Scale the objectness rows of every head conv: a random-weight detector's objectness logits sit within +-0.15 of the
    -4.5 bias (all scores within 30 % of each other), so rounding-level drift reorders the whole ranking and the mAP protocol
    would measure tie-breaking.  With the rows scaled the logits spread over several units, like a trained head's."""
    defs = [d for d in module_defs if d['type'] != 'net']
    for i, d in enumerate(defs):
        if d['type'] != 'yolo':
            continue
        nc = int(d['classes'])
        w = state['module_list.%d.Conv2d.weight' % (i - 1)]
        w.view(len(d['mask']), nc + 5, -1)[:, 4] *= gain
    return state


def image_batch(n, size, seed=0, channels=3):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, channels, size, size, generator=g)


def dyadic_frames(x, bits=8):
    """Frames on the grid k / 2**bits (what an 8-bit image scaled by 1/256 is).  With the PTQ weight / bias grids (powers of two)
    every product and partial sum of the stem's fp32 convolution is then exactly representable, so ANY summation order gives the
    same bits: the one fp32 conv of the int8 path stops being implementation-defined and the whole path can be compared bit for bit."""
    return (x * 2 ** bits).round() / 2 ** bits


def tensor_digest(t):
    """sha256 of a tensor's fp32 bytes (+0.0 first: -0.0 and 0.0 are the same value)."""
    import hashlib
    return hashlib.sha256((t.detach().float().cpu() + 0.0).contiguous().numpy().tobytes()).hexdigest()


def nms_candidates(n_img, rows, nc, seed, n_clusters=12, img=608, hot=0.35):
    """Decoded-head-like tensor (n_img, rows, 5+nc): clustered boxes, tied scores, tiny/huge boxes."""
    g = torch.Generator().manual_seed(seed)
    out = torch.zeros(n_img, rows, 5 + nc)
    for b in range(n_img):
        centers = torch.rand(n_clusters, 2, generator=g) * img
        sizes = torch.rand(n_clusters, 2, generator=g) * 150 + 20
        which = torch.randint(0, n_clusters, (rows,), generator=g)
        out[b, :, 0:2] = centers[which] + torch.randn(rows, 2, generator=g) * 6
        out[b, :, 2:4] = sizes[which] * (1 + torch.randn(rows, 2, generator=g) * 0.08)
        obj = torch.rand(rows, generator=g)
        obj = torch.where(torch.rand(rows, generator=g) < hot, obj, obj * 0.05)
        obj = (obj * 64).round() / 64  # quantise -> plenty of exactly tied scores
        out[b, :, 4] = obj
        cls = torch.rand(rows, nc, generator=g)
        fav = torch.randint(0, nc, (n_clusters,), generator=g)[which]
        cls[torch.arange(rows), fav] = 0.9 + 0.1 * torch.rand(rows, generator=g)
        out[b, :, 5:] = cls
        if rows >= 8:  # degenerate rows: too small, too large, non-finite
            out[b, 0, 2:4] = 1.0
            out[b, 1, 2:4] = 5000.0
            out[b, 2, 4] = 0.99
            out[b, 2, 0] = float('inf')
    return out


def loss_inputs(model, size, batch=3, seed=21, labels_per_image=6):
    """Seeded raw head tensors (bs, na, ny, nx, no) for ``model``'s yolo layers at ``size`` and a (nt, 6) label tensor
    (image, class, x, y, w, h normalised).  Works on the reference's and this package's Darknet alike."""
    g = torch.Generator().manual_seed(seed)
    raws = []
    for j in model.yolo_layers:
        m = model.module_list[j]
        stride = int(m.stride)
        ny = nx = size // stride
        raws.append(torch.randn(batch, m.na, ny, nx, m.no, generator=g) * 0.8)
    rows = []
    for b in range(batch):
        for _ in range(labels_per_image):
            cls = int(torch.randint(0, model.nc, (1,), generator=g))
            wh = torch.rand(2, generator=g) * 0.5 + 0.03
            xy = torch.rand(2, generator=g) * (1 - wh) + wh / 2
            rows.append([b, cls, xy[0].item(), xy[1].item(), wh[0].item(), wh[1].item()])
    return raws, torch.tensor(rows, dtype=torch.float32)
