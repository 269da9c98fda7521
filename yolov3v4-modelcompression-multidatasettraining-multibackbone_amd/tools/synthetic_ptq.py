"""A COS-PTQ graph with a SYNTHETIC calibrated state, for timing and large-graph parity of the int8 eval path.

Real calibration of YOLOv3-608 / YOLOv4-640 needs images (and the reference cannot calibrate cfgs with max-pools at all,
SURVEY 8c), but the EVAL arithmetic (reference quantized_ptq_cos.py:288-296, 543-567, 717: grid conv + grid bias ->
activation -> round/clamp onto the activation grid; shortcut / concat re-scaling) runs with any scale buffers.  This fills
them deterministically: weight / bias grids from the BN-folded float weights with power-of-two max-abs scales, every
activation, shortcut and concat scale a fixed power of two.  Works on this package's modules and, attribute for attribute,
on the reference's (tests/golden/make_golden_ptq.py uses it to produce reference-side eval goldens).
"""
import math

import torch


def _fold(w, b, g, beta, mean, var, eps):
    """BN folded into the conv, in float64: the grid decisions below (floor(|w| / s + 0.5)) then do not depend on the last fp32
    bit of the fold, which differs between hosts (measured: the fp32 fold gave different grid weights on an EPYC 9575F and on the
    build container - same float state, same scales - and the goldens stopped being reproducible on the GPU box)."""
    w, g, beta, mean, var = (t.detach().double() for t in (w, g, beta, mean, var))
    s = g / torch.sqrt(var + eps)
    shift = beta - mean * s
    return w * s.view(-1, 1, 1, 1), (shift if b is None else shift + b.detach().double() * s)


def _to_grid(t, scale):
    """Round-half-away onto the int8 grid of ``scale`` (reference quantized_ptq_cos.py:14-20), computed in float64, stored as fp32
    (grid values k * 2^e with |k| <= 128 are exact in fp32)."""
    t = t.detach().double()
    return ((torch.sign(t) * torch.floor(t.abs() / scale + 0.5)).clamp(-128, 127) * scale).float()


def pow2_scale(t, levels=127.0):
    """Smallest power of two s with max|t| / s <= levels."""
    return 2.0 ** math.ceil(math.log2(max(float(t), 1e-12) / levels))


def measure_ranges(float_model, x):
    """max|output| of every block of ``float_model`` (eval, float) on the batch ``x``: the data a one-batch max-abs calibration
    would see.  Forward hooks only, so it works on any Darknet implementation with a ``module_list``."""
    peaks = [0.0] * len(float_model.module_list)
    hooks = []
    for i, m in enumerate(float_model.module_list):
        def hook(mod, inp, out, i=i):
            t = out[0] if isinstance(out, (tuple, list)) else out
            if torch.is_tensor(t):
                peaks[i] = float(t.detach().abs().max())
        hooks.append(m.register_forward_hook(hook))
    was_training = float_model.training
    try:
        with torch.no_grad():
            float_model.eval()(x)
    finally:
        for h in hooks:
            h.remove()
        float_model.train(was_training)
    return peaks


def fill_synthetic_state(float_model, q_model, act_scale=2.0 ** -4, sum_scale=2.0 ** -3, ranges=None):
    """Copy ``float_model``'s (seeded) weights onto ``q_model``'s grids; both are Darknet graphs of the same cfg, ``q_model`` built
    with ``quantized=3``.  ``ranges`` (from ``measure_ranges``) puts every activation / shortcut / concat on the power-of-two
    grid that just covers its float range (a live int8 detector); without it every scale is the fixed ``act_scale`` (timing
    only: deep random-weight nets then clamp or die).  Returns ``q_model`` in eval mode."""
    defs = [d for d in getattr(float_model, 'module_defs', []) if d.get('type') != 'net']
    grid = (lambda i: pow2_scale(ranges[i], 127.0)) if ranges is not None else (lambda i: act_scale)
    with torch.no_grad():
        for i, (f, q) in enumerate(zip(float_model.module_list, q_model.module_list)):
            name = q.__class__.__name__
            if isinstance(f, torch.nn.Sequential) and len(f) and isinstance(f[0], torch.nn.Conv2d):
                conv = f[0]
                bn = f[1] if len(f) > 1 and isinstance(f[1], torch.nn.modules.batchnorm.BatchNorm2d) else None
                w, b = (conv.weight.detach().double(), conv.bias.detach().double()) if bn is None else _fold(conv.weight, conv.bias, bn.weight, bn.bias,
                                                                         bn.running_mean, bn.running_var, bn.eps)
                qc = q[0]
                sw, sb = pow2_scale(w.abs().max()), pow2_scale(b.abs().max())
                qc.weight_quantizer.scale.fill_(sw)
                qc.bias_quantizer.scale.fill_(sb)
                qc.activation_quantizer.scale.fill_(grid(i))
                qc.q_weight.copy_(_to_grid(w, sw))
                qc.q_bias.copy_(_to_grid(b, sb))
                qc.quantized = True      # BN already folded, grids in place: eval uses q_weight / q_bias as they are
            elif name.startswith('COSPTQuantizedShortcut'):
                if ranges is None:
                    sx = sa = act_scale
                    ss = sum_scale
                else:   # both operands on the finer of their two grids (what the _min search settles on), the sum on its own
                    frm = [int(k) for k in (defs[i]['from'] if isinstance(defs[i]['from'], (list, tuple)) else [defs[i]['from']])]
                    src = frm[0] if frm[0] >= 0 else i + frm[0]
                    sx = sa = min(grid(i - 1), grid(src))
                    ss = grid(i)
                q.scale_x.fill_(sx)
                q.scale_a.fill_(sa)
                q.scale_sum.fill_(ss)
            elif name == 'COSPTQuantizedFeatureConcat':
                q.scale.fill_(grid(i))
    return q_model.eval()
