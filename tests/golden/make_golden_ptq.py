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


def main():
    ref = refharness.load()
    torch.set_num_threads(8)
    torch.manual_seed(0)
    fm = ref.models.Darknet(mini_cfg(), (SIZE, SIZE))
    state = synth.randomize_bn_(fm.state_dict(), seed=1)
    synth.trained_like_heads_(state, fm.module_defs)
    fm.load_state_dict(state)
    qm = ref.models.Darknet(mini_cfg(), (SIZE, SIZE), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
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
    np.savez_compressed(os.path.join(HERE, 'ptq_mini.npz'), **out)
    scales = {k: float(v) for k, v in qm.state_dict().items() if k.endswith('scale') or 'scale_' in k.split('.')[-1]}
    print('stored', len(out), 'arrays;', len(scales), 'scales; int8-vs-float max box diff %.3f px' %
          (inf[..., :4] - inf_float[..., :4]).abs().max().item())
    print({k.replace('module_list.', ''): v for k, v in list(scales.items())[:12]})


if __name__ == '__main__':
    main()
