"""Eval-mode stand-ins for the reference's COS-PTQ modules — TEST INFRASTRUCTURE ONLY.

The reference's ``utils/quantized/quantized_ptq_cos.py`` is not present on the GPU box, but the int8 HIP engine
lowers its module classes by name and attribute (``q_weight``, ``q_bias``, ``*_quantizer.scale``, ``scale_x`` ...).
These classes carry the same names, constructor signatures, buffers and **eval** arithmetic
(quantized_ptq_cos.py: Round :14-20, Quantizer :23-113, conv eval :288-296,543-567,717, shortcut :877-912,1029,
concat :1540-1553) so that a ``Darknet(cfg, quantized=3)`` can be built and loaded from the committed fixture
``tests/golden/ptq_mini.npz`` (a calibrated ``state_dict`` produced by the reference itself).  Calibration
(train mode) is deliberately not restated: it stays with the reference.  ``tests/test_ptq.py`` checks that the
stand-in reproduces the reference's eval outputs bit for bit.
"""
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


def _round_away(t):
    return torch.sign(t) * torch.floor(torch.abs(t) + 0.5)


class Quantizer(nn.Module):
    def __init__(self, bits, out_channels):
        super().__init__()
        self.bits = bits
        self.register_buffer('scale', torch.zeros(1))
        self.register_buffer('float_range', torch.zeros(1))

    def forward(self, x):
        if self.training:
            raise RuntimeError('stand-in quantizer is eval-only (calibration stays with the reference)')
        q = _round_away(x / self.scale)
        q = torch.clamp(q, -(1 << (self.bits - 1)), (1 << (self.bits - 1)) - 1)
        return q * self.scale


class BNFold_COSPTQuantizedConv2d_For_FPGA(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False, eps=1e-5,
                 momentum=0.1, a_bits=8, w_bits=8, bn=0, activate='leaky', quantizer_output=False, reorder=False, TM=32, TN=32,
                 name='', layer_idx=-1, maxabsscaler=False):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation, groups=groups,
                         bias=bias)
        self.bn = bn
        if not bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        self.activate = activate
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(out_channels))
        self.beta = nn.Parameter(torch.zeros(out_channels))
        self.register_buffer('running_mean', torch.zeros(out_channels))
        self.register_buffer('running_var', torch.zeros(out_channels))
        self.register_buffer('q_bias', torch.zeros(out_channels))
        self.register_buffer('q_weight', torch.zeros(self.weight.shape))
        self.quantized = False
        self.maxabsscaler = maxabsscaler
        self.a_bits, self.w_bits = a_bits, w_bits
        self.activation_quantizer = Quantizer(a_bits, -1)
        self.weight_quantizer = Quantizer(w_bits, -1)
        self.bias_quantizer = Quantizer(w_bits, -1)

    def forward(self, x):
        if self.training or not self.quantized:
            raise RuntimeError('stand-in conv is eval-only and needs a calibrated state (load the fixture, set .quantized)')
        y = F.conv2d(x, self.q_weight, self.q_bias, self.stride, self.padding, self.dilation, self.groups)
        a = self.activate
        if a == 'leaky':
            y = F.leaky_relu(y, 0.25 if self.maxabsscaler else 0.1, inplace=True)
        elif a == 'relu6':
            y = F.relu6(y, inplace=True)
        elif a == 'h_swish':
            y = y * (F.relu6(y + 3.0, inplace=True) / 6.0)
        elif a == 'relu':
            y = F.relu(y, inplace=True)
        elif a == 'mish':
            y = y * F.softplus(y).tanh()
        return self.activation_quantizer(y)


class _QShortcut(nn.Module):
    def __init__(self, layers, weight=False, bits=8, quantizer_output=False, reorder=False, TM=32, TN=32, name='', layer_idx=-1):
        super().__init__()
        self.layers, self.weight, self.n, self.bits = layers, weight, len(layers) + 1, bits
        for tag in ('x', 'a', 'sum'):
            self.register_buffer('scale_' + tag, torch.zeros(1))
            self.register_buffer('float_range_' + tag, torch.zeros(1))

    def forward(self, x, outputs):
        if self.training or self.weight:
            raise RuntimeError('stand-in shortcut: eval-only, unweighted')
        a = outputs[self.layers[0]]
        x = _round_away(x / self.scale_x) * self.scale_x
        a = _round_away(a / self.scale_a) * self.scale_a
        s = x + a
        q = torch.clamp(_round_away(s / self.scale_sum), -(1 << (self.bits - 1)), (1 << (self.bits - 1)) - 1)
        return q * self.scale_sum


class COSPTQuantizedShortcut_min(_QShortcut):
    pass


class COSPTQuantizedShortcut_max(_QShortcut):
    pass


class COSPTQuantizedFeatureConcat(nn.Module):
    def __init__(self, layers, groups, bits=8, quantizer_output=False, reorder=False, TM=32, TN=32, name='', layer_idx=-1):
        super().__init__()
        self.layers, self.groups, self.multiple, self.bits = layers, groups, len(layers) > 1, bits
        self.register_buffer('scale', torch.zeros(1))
        self.register_buffer('float_max_list', torch.zeros(len(layers)))

    def forward(self, x, outputs):
        if self.training:
            raise RuntimeError('stand-in concat is eval-only')
        if self.multiple:
            lo, hi = -(1 << (self.bits - 1)), (1 << (self.bits - 1)) - 1
            for i in self.layers:  # the reference re-quantises the cached outputs in place (:1540-1545)
                outputs[i] = torch.clamp(_round_away(outputs[i] / self.scale), lo, hi) * self.scale
            return torch.cat([outputs[i] for i in self.layers], 1)
        if self.groups:
            return x[:, (x.shape[1] // 2):]
        return outputs[self.layers[0]]


def install():
    """Make ``import utils.quantized.quantized_ptq_cos`` resolve to this module when the reference is absent."""
    import utils
    pkg = sys.modules.get('utils.quantized')
    if pkg is None:
        pkg = types.ModuleType('utils.quantized')
        pkg.__path__ = []
        sys.modules['utils.quantized'] = pkg
        utils.quantized = pkg
    me = sys.modules[__name__]
    sys.modules['utils.quantized.quantized_ptq_cos'] = me
    pkg.quantized_ptq_cos = me
    return me


def load_fixture(model, fx):
    """Load the calibrated tensors of ``ptq_mini.npz`` into a stand-in ``Darknet(quantized=3)``."""
    sd = model.state_dict()
    for k in sd:
        key = 'sd.' + k
        if key in fx:
            sd[k].copy_(torch.from_numpy(fx[key]).reshape(sd[k].shape))
    model.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, BNFold_COSPTQuantizedConv2d_For_FPGA):
            m.quantized = True
    return model
