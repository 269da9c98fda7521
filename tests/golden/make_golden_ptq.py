#!/usr/bin/env python3
"""Golden fixture for the int8 PTQ eval path, produced by the REFERENCE's own code.

Runs in the build container only.  Builds a small maxpool-free detector (the reference's calibration crashes on
cfgs with maxpool), copies seeded float weights into the reference's ``Darknet(quantized=3,
shortcut_way=1)``, calibrates it with the reference's train-mode forward on seeded batches (PTQ.py:76-88), then
stores the calibrated ``state_dict`` (scales, q_weight, q_bias ...) and the reference's eval outputs.

    python tests/golden/make_golden_ptq.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401
import refharness
import synth
from ptq_minicfg import mini_cfg, SIZE


def mini_fixture(ref, way, fname):
    """Calibrated-by-the-reference fixture of the mini net; way 1 = COSPTQuantizedShortcut_min, 2 = _max (models.py:276-290)."""
    torch.manual_seed(0)
    fm = ref.models.Darknet(mini_cfg(), (SIZE, SIZE))
    state = synth.randomize_bn_(fm.state_dict(), seed=1)
    synth.trained_like_heads_(state, fm.module_defs)
    fm.load_state_dict(state)
    qm = ref.models.Darknet(mini_cfg(), (SIZE, SIZE), quantized=3, a_bit=8, w_bit=8, shortcut_way=way)
    # what load_darknet_weights(quant=True) does (models.py:610-628): BN tensors land on the conv itself
    for f, q in zip(fm.module_list, qm.module_list):
        if isinstance(f, torch.nn.Sequential) and len(f) and isinstance(f[0], torch.nn.Conv2d):
            qc = q[0]
            qc.weight.data.copy_(f[0].weight.data)
            if len(f) > 1 and isinstance(f[1], torch.nn.BatchNorm2d):
                qc.gamma.data.copy_(f[1].weight.data)
                qc.beta.data.copy_(f[1].bias.data)
                qc.running_mean.copy_(f[1].running_mean)
                qc.running_var.copy_(f[1].running_var)
            else:
                qc.bias.data.copy_(f[0].bias.data)
    qm.train()
    with torch.no_grad():
        for it in range(6):
            qm(synth.image_batch(4, SIZE, seed=100 + it))
    qm.eval()
    x = synth.image_batch(2, SIZE, seed=7)
    with torch.no_grad():
        inf, raws, _ = qm(x)
        inf_float = fm.eval()(x)[0]
    out = {'inf': inf.numpy(), 'inf_float': inf_float.numpy()}
    for i, r in enumerate(raws):
        out['raw%d' % i] = r.numpy()
    for k, v in qm.state_dict().items():  # only what the eval path reads: grid weights/biases and the scales
        if k.split('.')[-1] in ('q_weight', 'q_bias', 'scale', 'scale_x', 'scale_a', 'scale_sum'):
            out['sd.' + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    scales = {k: float(v) for k, v in qm.state_dict().items() if k.endswith('scale') or 'scale_' in k.split('.')[-1]}
    print('stored', len(out), 'arrays;', len(scales), 'scales; int8-vs-float max box diff %.3f px' %
          (inf[..., :4] - inf_float[..., :4]).abs().max().item())
    print({k.replace('module_list.', ''): v for k, v in list(scales.items())[:12]})


EVAL_CASES = [   # name, cfg, size, batch, row stride: the BASELINE graphs the int8 bench times (configs 2 and 4)
    ('yolov3_608', 'yolov3/yolov3.cfg', 608, 1, 64, 'plain'),
    # plain seeded BN statistics collapse YOLOv4's 110 convs (every cell of an anchor within 1e-4 of the same objectness): the layer
    # gains are equalised first (synth.equalize_bn_gain_), as in the fp16 mAP protocol
    ('yolov4_640', 'yolov4/yolov4.cfg', 640, 1, 64, 'equalized'),
    # the same two graphs on DYADIC frames (k / 256, synth.dyadic_frames): the stem conv - the only fp32 convolution of the path - is
    # then exact in any summation order, and the int8 engine must reproduce the reference BIT FOR BIT (whole-tensor digests stored)
    ('yolov3_608_dyadic', 'yolov3/yolov3.cfg', 608, 1, 64, 'plain'),
    ('yolov4_640_dyadic', 'yolov4/yolov4.cfg', 640, 1, 64, 'equalized'),
]


def eval_fixture(ref, name, rel, size, batch, row_stride, conditioning):
    """EVAL-mode int8 golden on a full BASELINE graph: the reference's quantized=3 modules with the synthetic power-of-two state
    of tools/synthetic_ptq.py (the reference cannot CALIBRATE max-pool cfgs, SURVEY 8c, but its eval branch
    quantized_ptq_cos.py:288-296,717 runs with any scale buffers).  Stored: row subset + whole-tensor checksums."""
    from tools.synthetic_ptq import fill_synthetic_state, measure_ranges
    from make_golden import checks
    cfg = os.path.join(refharness.REF, 'cfg', rel)
    torch.manual_seed(0)
    fm = ref.models.Darknet(cfg, (size, size))
    state = synth.randomize_bn_(fm.state_dict(), seed=1)
    fm.load_state_dict(state)
    x = synth.image_batch(batch, size, seed=0)
    frames = 'dyadic' if name.endswith('_dyadic') else 'float'
    if frames == 'dyadic':
        x = synth.dyadic_frames(x)
    if conditioning == 'equalized':
        synth.equalize_bn_gain_(fm.eval(), x)
    state = synth.trained_like_heads_(fm.state_dict(), fm.module_defs)
    fm.load_state_dict(state)
    torch.manual_seed(0)
    qm = ref.models.Darknet(cfg, (size, size), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    fill_synthetic_state(fm, qm, ranges=measure_ranges(fm, x))
    with torch.no_grad():
        inf, raws, _ = qm(x)
        inf_float = fm.eval()(x)[0]
    out = dict(cfg=rel, size=size, batch=batch, row_stride=row_stride, conditioning=conditioning, frames=frames,
               inf_rows=inf[:, ::row_stride].numpy().astype(np.float32), inf_checks=checks(inf), inf_shape=np.array(inf.shape))
    for i, r in enumerate(raws):
        out['raw%d_checks' % i] = checks(r)
        out['raw%d_rows' % i] = r.reshape(-1)[::997].numpy().astype(np.float32)
        out['raw%d_sha256' % i] = synth.tensor_digest(r)
    np.savez_compressed(os.path.join(HERE, 'ptq_eval_%s.npz' % name), **out)
    print('eval fixture', name, tuple(inf.shape), 'obj range %.3g..%.3g; int8-vs-float: box %.3g px, obj %.3g' % (
        inf[..., 4].min().item(), inf[..., 4].max().item(), (inf[..., :4] - inf_float[..., :4]).abs().max().item(),
        (inf[..., 4] - inf_float[..., 4]).abs().max().item()))


def main():
    ref = refharness.load()
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if not only or 'mini' in only:
        mini_fixture(ref, 1, 'ptq_mini.npz')
    if not only or 'mini_max' in only:
        mini_fixture(ref, 2, 'ptq_mini_max.npz')
    for case in EVAL_CASES:
        if not only or case[0] in only:
            eval_fixture(ref, *case)


if __name__ == '__main__':
    main()
