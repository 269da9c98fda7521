"""Graph operators that sit between the fused conv blocks of a Darknet cfg.

Interface mirror of the reference's ``utils/layers.py`` (FeatureConcat :26-40, Shortcut :43-72,
Mish :146-148, ReLU6 :151-156, HardSwish :159-164, HardSigmoid :167-173, SE :176-192).  The class
*names* matter: ``Darknet.forward_once`` dispatches on ``module.__class__.__name__`` and the prune
scripts test ``activation.__class__.__name__``.

These ``nn.Module`` bodies are the eager (autograd / CPU-tensor) semantics.  On a CUDA tensor in eval
mode ``Darknet`` does not call them at all: the HIP engine (``engine/plan.py``) lowers shortcut to a
conv-epilogue residual add, route/concat to producer-side channel-slice writes, and the activations
to the conv epilogue.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def make_divisible(v, divisor):
    return math.ceil(v / divisor) * divisor


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)


class FeatureConcat(nn.Module):
    """``route``: channel concat of earlier outputs; single index = alias; ``groups`` = 2nd half."""

    def __init__(self, layers, groups):
        super().__init__()
        self.layers = layers
        self.groups = groups
        self.multiple = len(layers) > 1

    def forward(self, x, outputs):
        if self.multiple:
            return torch.cat([outputs[i] for i in self.layers], 1)
        if self.groups:
            half = x.shape[1] // 2
            return x[:, half:]
        return outputs[self.layers[0]]


class Shortcut(nn.Module):
    """``shortcut``: x + outputs[from...] with optional learned sigmoid weights.

    Channel mismatch follows the reference (layers.py:65-70): the narrower operand decides how many
    leading channels are summed.
    """

    def __init__(self, layers, weight=False):
        super().__init__()
        self.layers = layers
        self.weight = weight
        self.n = len(layers) + 1
        if weight:
            self.w = nn.Parameter(torch.zeros(self.n), requires_grad=True)

    def forward(self, x, outputs):
        w = None
        if self.weight:
            w = torch.sigmoid(self.w) * (2 / self.n)
            x = x * w[0]
        cx = x.shape[1]
        for k, idx in enumerate(self.layers):
            a = outputs[idx]
            if w is not None:
                a = a * w[k + 1]
            ca = a.shape[1]
            if cx == ca:
                x = x + a
            elif cx > ca:
                x[:, :ca] = x[:, :ca] + a
            else:
                x = x + a[:, :cx]
        return x


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class Mish(nn.Module):
    def forward(self, x):
        return x * torch.tanh(F.softplus(x))


class ReLU6(nn.Module):
    def forward(self, x):
        return F.relu6(x, inplace=True)


class HardSwish(nn.Module):
    def forward(self, x):
        return x * (F.relu6(x + 3.0, inplace=True) / 6.0)


class HardSigmoid(nn.Module):
    def forward(self, x):
        return F.relu6(x + 3.0, inplace=True) / 6.0


class SE(nn.Module):
    """Squeeze-excite: x * hsigmoid(W2 relu(W1 avgpool(x))); both Linear layers bias-free."""

    def __init__(self, channel, reduction=4):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(
            nn.Linear(channel, channel // reduction, bias=False),
            nn.ReLU(inplace=True),
            nn.Linear(channel // reduction, channel, bias=False),
            HardSigmoid())

    def forward(self, x):
        b, c = x.shape[:2]
        s = self.fc(self.avg_pool(x).view(b, c)).view(b, c, 1, 1)
        return x * s.expand_as(x)
