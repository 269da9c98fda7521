#!/usr/bin/env python3
"""Scale decisions of a HOST COS-PTQ calibration at the BASELINE shapes: YOLOv3-608 and YOLOv4-640,
batch 2, 3 calibration batches, this package's utils/quantized/quantized_ptq_cos.py on CPU tensors - the reference's loop
(quantized_ptq_cos.py:64-93), which the module reproduces decision for decision where the reference can run
(tests/test_ptq_calibration.py; the reference itself crashes on yolov4.cfg).  Stored: every `scale` / `float_range`
buffer of the calibrated state (a few hundred numbers per net) in tests/golden/ptq_calib_<tag>.npz.

    python tests/golden/make_golden_ptq608.py [v3_608] [v4_640]      # tens of minutes per net on 8 threads
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401
import synth

CASES = {'v3_608': ('yolov3/yolov3.cfg', 608), 'v4_640': ('yolov4/yolov4.cfg', 640)}


def calibrated_scales(rel, size, device='cpu', batches=3, batch=2):
    import models
    from test_ptq_calibration import _copy_float_weights
    cfg = os.path.join(conftest.PKG, 'cfg', rel)
    torch.manual_seed(0)
    fm = models.Darknet(cfg, (size, size))
    fm.load_state_dict(synth.randomize_bn_(fm.state_dict(), seed=1))
    qm = models.Darknet(cfg, (size, size), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    _copy_float_weights(fm, qm)
    del fm
    qm.to(device).train()
    with torch.no_grad():
        for it in range(batches):
            qm(synth.image_batch(batch, size, seed=10 + it).to(device))
    sd = qm.state_dict()
    keys = [k for k in sd if k.endswith('scale') or 'scale_' in k.rsplit('.', 1)[-1] or 'float_range' in k]
    return qm, {k: sd[k].detach().float().cpu().numpy() for k in keys}


def main():
    torch.set_num_threads(int(os.environ.get('GOLDEN_THREADS', '8')))
    for tag in (sys.argv[1:] or list(CASES)):
        rel, size = CASES[tag]
        t0 = time.time()
        _, scales = calibrated_scales(rel, size)
        np.savez_compressed(os.path.join(HERE, 'ptq_calib_%s.npz' % tag), **scales)
        print('%s: %d scale tensors, %.0f s' % (tag, len(scales), time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
