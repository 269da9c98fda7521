"""Give a random-weight detector the head statistics of a trained one, for TIMING the detect path honestly.

With seeded random weights and the default head bias every objectness logit sits within +-0.15 of the bias (-4.5), so at
detect.py's settings (conf 0.3) not a single cell passes the threshold: `non_max_suppression` returns after its count pass and a
"forward + NMS" step measures the forward only (the reference's README has NMS at 1.7 of 14 ms, README.md:228).  A trained
detector at those settings hands NMS on the order of a hundred candidates per image, in spatial clusters.

``detector_like_heads_`` edits ONLY the three head convolutions (reference models.py:92-113 conv blocks feeding YOLOLayer,
models.py:406-418), in place and deterministically for a given batch:

* the objectness rows of the head weights are scaled so that the logits spread over a few units (a trained head's range);
* the objectness bias is shifted so that the ``per_image``-th largest logit of an image sits just above logit(conf);
* every (head, anchor) gets one favourite class (+``margin`` on that class bias), so score = obj * cls stays above conf.

Boxes, every other convolution and the amount of convolution work are untouched.  bench.py reports the candidate / survivor
counts it then measures, next to the NMS time.
"""
import math

import torch


def _yolo_heads(model):
    defs = [d for d in model.module_defs if d.get('type') != 'net']
    for i, d in enumerate(defs):
        if d['type'] == 'yolo':
            yield i, d


def _objectness_logits(raws):
    # raw p of a head: (bs, na, ny, nx, no); objectness logit is channel 4 (reference models.py:406-418)
    return torch.cat([p[..., 4].reshape(p.shape[0], -1).float() for p in raws], 1)


@torch.no_grad()
def detector_like_heads_(model, x, per_image=100, conf=0.3, spread=2.5, margin=8.0):
    """In place; ``model`` in eval mode on the device of ``x``.  Returns the statistics it settled on."""
    was_training = model.training
    model.eval()
    heads = list(_yolo_heads(model))
    convs = [model.module_list[i - 1][0] for i, _ in heads]

    def logits():
        _, raws, *_ = model(x)
        return _objectness_logits(raws)

    lg = logits()
    sigma = float(lg.std())
    gain = spread / max(sigma, 1e-6)
    for (i, d), conv in zip(heads, convs):
        na, no = len(d['mask']), int(d['classes']) + 5
        conv.weight.data.view(na, no, -1)[:, 4] *= gain
    if hasattr(model, 'hip_refresh'):
        model.hip_refresh()
    lg = logits()
    # the threshold goes BETWEEN the k-th and the (k+1)-th largest logit (pooled over the calibration frames): after the gain the
    # nine (head, anchor) groups have different means, and the top of the distribution sits inside the densest group - a margin of
    # a fraction of a logit above the k-th value already lets thousands of cells through
    k = min(per_image * lg.shape[0], lg.numel() - 1)
    top = lg.flatten().topk(k + 1).values
    kth = float(top[-2] + top[-1]) / 2
    shift = math.log(conf / (1 - conf)) - kth
    for h, ((i, d), conv) in enumerate(zip(heads, convs)):
        na, nc = len(d['mask']), int(d['classes'])
        b = conv.bias.data.view(na, nc + 5)
        b[:, 4] += shift
        for a in range(na):
            b[a, 5 + ((3 * h + a) * 7) % nc] += margin
    if hasattr(model, 'hip_refresh'):
        model.hip_refresh()
    model.train(was_training)
    return {'objectness_gain': round(gain, 3), 'objectness_shift': round(shift, 3), 'logit_sigma_before': round(sigma, 4)}
