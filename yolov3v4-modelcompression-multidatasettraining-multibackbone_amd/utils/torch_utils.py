"""Device / model helpers.

Interface mirror of the reference's ``utils/torch_utils.py`` (init_seeds :9, select_device :16,
time_synchronized :43, fuse_conv_and_bn :65-89, model_info :92, scale_img :130, ModelEMA :141).

``fold_bn`` is the one piece that is on the hot path: it is the host-side BN->conv folding
(``W' = diag(g/sqrt(var+eps)) W``, ``b' = beta - g*mu/sqrt(var+eps) + scale*b``) that both the eager
``Darknet.fuse()`` and the HIP engine's weight packer (``engine/plan.py``) use.
"""
import math
import os
import time
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.functional as F


def init_seeds(seed=0):
    torch.manual_seed(seed)
    if seed == 0:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def select_device(device='', batch_size=None):
    """'cpu', '' (auto) or a comma list of GPU ordinals -> torch.device."""
    want_cpu = device.lower() == 'cpu'
    if device and not want_cpu:
        os.environ['CUDA_VISIBLE_DEVICES'] = device
        assert torch.cuda.is_available(), 'CUDA unavailable, invalid device %s requested' % device
    use_gpu = (not want_cpu) and torch.cuda.is_available()
    if use_gpu:
        count = torch.cuda.device_count()
        if count > 1 and batch_size:
            assert batch_size % count == 0, 'batch-size %g not multiple of GPU count %g' % (batch_size, count)
        for i in range(count):
            prop = torch.cuda.get_device_properties(i)
            lead = 'Using CUDA ' if i == 0 else ' ' * 11
            print("%sdevice%g _CudaDeviceProperties(name='%s', total_memory=%dMB)" %
                  (lead, i, prop.name, prop.total_memory / 1024 ** 2))
    else:
        print('Using CPU')
    print('')
    return torch.device('cuda:0' if use_gpu else 'cpu')


def time_synchronized():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def initialize_weights(model):
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps = 1e-4
            m.momentum = 0.03
        elif type(m) in (nn.LeakyReLU, nn.ReLU, nn.ReLU6):
            m.inplace = True


def find_modules(model, mclass=nn.Conv2d):
    return [i for i, m in enumerate(model.module_list) if isinstance(m, mclass)]


def fold_bn(weight, bias, gamma, beta, mean, var, eps):
    """Fold eval-mode BatchNorm into the preceding conv; returns (weight', bias') in weight's dtype.

    weight: (Cout, Cin/g, kh, kw); bias: (Cout,) or None.  Pure tensor math, device agnostic.
    """
    scale = gamma / torch.sqrt(var + eps)
    w = weight * scale.view(-1, 1, 1, 1)
    shift = beta - mean * scale
    b = shift if bias is None else shift + bias * scale
    return w, b


def fuse_conv_and_bn(conv, bn):
    """New bias-carrying ``nn.Conv2d`` equal to ``bn(conv(x))`` in eval mode."""
    with torch.no_grad():
        fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size,
                          stride=conv.stride, padding=conv.padding, groups=conv.groups, bias=True)
        fused = fused.to(conv.weight.device)
        w, b = fold_bn(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        fused.weight.copy_(w)
        fused.bias.copy_(b)
    return fused


def model_info(model, verbose=False):
    params = list(model.parameters())
    n_p = sum(p.numel() for p in params)
    n_g = sum(p.numel() for p in params if p.requires_grad)
    if verbose:
        print('%5s %40s %9s %12s %20s %10s %10s' % ('layer', 'name', 'gradient', 'parameters', 'shape', 'mu', 'sigma'))
        for i, (name, p) in enumerate(model.named_parameters()):
            print('%5g %40s %9s %12g %20s %10.3g %10.3g' % (
                i, name.replace('module_list.', ''), p.requires_grad, p.numel(), list(p.shape), p.mean(), p.std()))
    print('Model Summary: %g layers, %g parameters, %g gradients' % (len(params), n_p, n_g))


def scale_img(img, ratio=1.0, same_shape=True):
    """Bilinear rescale of an (N,C,H,W) batch, padded back (value 0.447) to H,W or to a /64 grid."""
    h, w = img.shape[2:]
    new = (int(h * ratio), int(w * ratio))
    img = F.interpolate(img, size=new, mode='bilinear', align_corners=False)
    if not same_shape:
        h, w = [math.ceil(v * ratio / 64) * 64 for v in (h, w)]
    return F.pad(img, [0, w - new[1], 0, h - new[0]], value=0.447)


class ModelEMA:
    """Exponential moving average of a model's state_dict with a ramped decay."""

    def __init__(self, model, decay=0.9999, device=''):
        self.ema = deepcopy(model)
        self.ema.eval()
        self.updates = 0
        self.decay = lambda n: decay * (1 - math.exp(-n / 2000))
        self.device = device
        if device:
            self.ema.to(device=device)
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        self.updates += 1
        d = self.decay(self.updates)
        wrapped = type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)
        with torch.no_grad():
            src = (model.module if wrapped else model).state_dict()
            dst = (self.ema.module if wrapped else self.ema).state_dict()
            for k, v in dst.items():
                if v.dtype.is_floating_point:
                    v.mul_(d).add_(src[k].detach(), alpha=1. - d)

    def update_attr(self, model):
        for k, v in model.__dict__.items():
            if not k.startswith('_'):
                setattr(self.ema, k, v)
