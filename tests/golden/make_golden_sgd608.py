#!/usr/bin/env python3
"""Golden of the first TEN optimisation steps of the REFERENCE's training loop at the headline shape:
YOLOv3 (Darknet-53) at 608 x 608, batch 2, fp32 on the CPU, the loop of /root/reference/train.py:344-455 with its burn-in
schedule and its three-group nesterov SGD (tests/sgd_protocol.py states it once for both sides), on the reference's own
`Darknet` and `compute_loss` imported from /root/reference (tests/refharness.py).  Stored in tests/golden/sgd_608.npz: the four
loss items, the parameter norm and the norm of the displacement from the initial point after every step, which steps the
optimizer fired on (accumulate grows during burn-in), and checksums of the final parameters and running statistics.  State: synth.calm_bn_ (well-conditioned: rounding the
weights to fp16 moves this net's gradient by 1e-3, not by 0.3).

    python tests/golden/make_golden_sgd608.py            # about a minute on 8 threads
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401
import refharness
import sgd_protocol
import synth
from make_golden import REFCFG, checks

SIZE, BATCH, STEPS = 608, 2, 10


def main():
    ref = refharness.load()
    torch.set_num_threads(int(os.environ.get('GOLDEN_THREADS', '8')))
    torch.manual_seed(0)
    model = ref.models.Darknet(os.path.join(REFCFG, 'yolov3', 'yolov3.cfg'), (SIZE, SIZE))
    state = model.state_dict()
    synth.calm_bn_(synth.randomize_bn_(state, seed=1))      # the well-conditioned state (synth.calm_bn_): a trajectory in the chaotic one pins nothing
    model.load_state_dict(state)
    tr = sgd_protocol.run(model, ref.utils.compute_loss, STEPS, SIZE, BATCH,
                          on_step=lambda ni, it, pn, dn: print('step %d items %s |p| %.6f |p - p0| %.6f' % (ni, it, pn, dn), flush=True))
    out = dict(tr, size=SIZE, batch=BATCH, steps=STEPS)
    sd = model.state_dict()
    keys = [k for k, v in sd.items() if v.dtype.is_floating_point]
    out['state_names'] = np.array(keys)
    out['state_checks'] = np.stack([checks(sd[k]) for k in keys])
    np.savez_compressed(os.path.join(HERE, 'sgd_608.npz'), **out)
    print('sgd 608 fixture ok:', STEPS, 'steps, optimizer fired on', int(tr['stepped'].sum()))


if __name__ == '__main__':
    main()
