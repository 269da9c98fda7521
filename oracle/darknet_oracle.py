"""CPU oracle for the Darknet hot path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the *checker* (or the timed CPU baseline), never as something the product path
routes through.  The product path is the HIP library behind ``models.Darknet`` and raises when that
library is missing.

What it is: a functional restatement — tensors in, tensors out, no ``nn.Module`` — of the
reference's float forward path, written from the reference's algorithm:

    block walk / out[] cache ........ models.py:508-561 (forward_once)
    conv block (conv, BN eval, act) .. models.py:92-113 ; BN eps 1e-5 ; leaky slope 0.1
    depthwise block .................. models.py:176-197
    BN folding ....................... utils/torch_utils.py:65-89 (fuse_conv_and_bn)
    maxpool (+tiny zero pad) ......... models.py:207-215
    upsample nearest ................. models.py:225
    route / group split .............. utils/layers.py:33-40
    shortcut (+channel mismatch) ..... utils/layers.py:52-72
    SE ............................... utils/layers.py:188-192
    activations ...................... utils/layers.py:146-173
    yolo decode ...................... models.py:406-418 (+ grid :367-378, anchors/stride :362)
    NMS + merge ...................... utils/utils.py:782-860
    torchvision.ops.boxes.nms ........ third party, unpinned in requirements.txt:9 and absent from
                                       /root/reference; restated from its published contract
                                       (score-descending greedy, suppress IoU > thr)

Arithmetic is floating point, so per the task rules this oracle keeps torch's CPU fp32 (or fp64)
conv as its contraction primitive; everything around it is explicit tensor algebra.

Pinning: the reference ships no golden vectors, KATs or tests, so the oracle is pinned
against outputs of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (which imports /root/reference) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every fixture.  The NMS boundary inherits the "unpinned
third-party nms" caveat above: both the fixture generator and this file restate torchvision's contract.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
MIN_WH, MAX_WH = 2.0, 4096.0


# ------------------------------------------------------------------------------------------ operators
def act(x, name, leaky_slope=0.1):
    if name == 'leaky':
        return torch.where(x > 0, x, x * leaky_slope)
    if name == 'relu':
        return x.clamp(min=0)
    if name == 'relu6':
        return x.clamp(0, 6)
    if name == 'h_swish':
        return x * (x + 3.0).clamp(0, 6) / 6.0
    if name == 'mish':
        return x * torch.tanh(torch.log1p(torch.exp(-x.abs())) + x.clamp(min=0))  # stable softplus
    return x  # linear / none / unknown strings


def fold_bn(w, b, gamma, beta, mean, var, eps=BN_EPS):
    """W' = diag(g/sqrt(var+eps)) W ; b' = beta - g*mean/sqrt(var+eps) (+ scaled conv bias)."""
    s = gamma / torch.sqrt(var + eps)
    wf = w * s.reshape(-1, 1, 1, 1)
    bf = beta - mean * s
    if b is not None:
        bf = bf + b * s
    return wf, bf


def conv_block(x, mdef, st, prefix, depthwise=False, fold=True, leaky_slope=0.1):
    """One [convolutional]/[depthwise] block from its state_dict entries."""
    cname = 'DepthWise2d' if depthwise else 'Conv2d'
    w = st[prefix + cname + '.weight'].to(x.dtype)
    b = st.get(prefix + cname + '.bias')
    b = None if b is None else b.to(x.dtype)
    k = int(mdef['size'])
    pad = (k - 1) // 2 if int(mdef['pad']) else 0
    stride = int(mdef['stride'])
    groups = x.shape[1] if depthwise else int(mdef.get('groups', 1))
    has_bn = (prefix + 'BatchNorm2d.weight') in st
    if has_bn:
        g, be = st[prefix + 'BatchNorm2d.weight'].to(x.dtype), st[prefix + 'BatchNorm2d.bias'].to(x.dtype)
        mu, var = st[prefix + 'BatchNorm2d.running_mean'].to(x.dtype), st[prefix + 'BatchNorm2d.running_var'].to(x.dtype)
        if fold:
            w, b = fold_bn(w, b, g, be, mu, var)
            y = F.conv2d(x, w, b, stride=stride, padding=pad, groups=groups)
        else:
            y = F.conv2d(x, w, b, stride=stride, padding=pad, groups=groups)
            y = (y - mu.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS) * g.view(1, -1, 1, 1) \
                + be.view(1, -1, 1, 1)
    else:
        y = F.conv2d(x, w, b, stride=stride, padding=pad, groups=groups)
    return act(y, mdef['activation'], leaky_slope)


def maxpool(x, size, stride):
    if size == 2 and stride == 1:  # tiny: zero border right/bottom, then an unpadded 2x2 window
        x = F.pad(x, (0, 1, 0, 1), value=0.0)
        return F.max_pool2d(x, 2, 1, 0)
    return F.max_pool2d(x, size, stride, (size - 1) // 2)


def upsample_nearest(x, s):
    return x.repeat_interleave(s, dim=2).repeat_interleave(s, dim=3)


def shortcut(x, others, weights=None):
    if weights is not None:
        x = x * weights[0]
    cx = x.shape[1]
    for k, a in enumerate(others):
        if weights is not None:
            a = a * weights[k + 1]
        ca = a.shape[1]
        if cx == ca:
            x = x + a
        elif cx > ca:
            x = torch.cat((x[:, :ca] + a, x[:, ca:]), 1)
        else:
            x = x + a[:, :cx]
    return x


def se_block(x, w1, w2):
    s = x.mean(dim=(2, 3))
    s = (s @ w1.t()).clamp(min=0) @ w2.t()
    s = (s + 3.0).clamp(0, 6) / 6.0
    return x * s[:, :, None, None]


def yolo_decode(p, anchors_px, stride, nc):
    """(bs, na*no, ny, nx) head map -> (io (bs, na*ny*nx, no), raw (bs, na, ny, nx, no))."""
    bs, _, ny, nx = p.shape
    na, no = len(anchors_px), nc + 5
    raw = p.reshape(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
    gx = torch.arange(nx, dtype=p.dtype).view(1, 1, 1, nx)
    gy = torch.arange(ny, dtype=p.dtype).view(1, 1, ny, 1)
    anchor = torch.as_tensor(np.asarray(anchors_px), dtype=p.dtype) / stride  # grid units
    io = torch.empty_like(raw)
    io[..., 0] = (torch.sigmoid(raw[..., 0]) + gx) * stride
    io[..., 1] = (torch.sigmoid(raw[..., 1]) + gy) * stride
    io[..., 2] = torch.exp(raw[..., 2]) * anchor[:, 0].view(1, na, 1, 1) * stride
    io[..., 3] = torch.exp(raw[..., 3]) * anchor[:, 1].view(1, na, 1, 1) * stride
    io[..., 4:] = torch.sigmoid(raw[..., 4:])
    return io.reshape(bs, -1, no), raw


# ---------------------------------------------------------------------------------------------- graph
def head_strides(layer_defs):
    """Cumulative down-sampling factor at every block output (used for the yolo strides)."""
    scales, hist = [], [1.0]
    for d in layer_defs:
        t, s = d['type'], hist[-1]
        if t in ('convolutional', 'depthwise', 'maxpool'):
            s = s * int(d['stride'])
        elif t == 'upsample':
            s = s / int(d['stride'])
        elif t == 'route':
            l = d['layers'][0]
            s = hist[l + 1 if l > 0 else l]
        hist.append(s)
        scales.append(s)
    return scales


def forward(module_defs, state, x, fold=True, dtype=torch.float32, leaky_slope=0.1, return_layers=False):
    """Eval-mode forward of a cfg (list of block dicts, with or without the leading [net]).

    ``state`` maps reference state_dict names (``module_list.{i}.Conv2d.weight`` ...) to tensors.
    Returns ``(inf_out, [raw_p...])`` (+ every block output when ``return_layers``).
    """
    defs = [d for d in module_defs if d['type'] != 'net']
    scales = head_strides(defs)
    x = x.to(dtype)
    outs, ios, raws = [], [], []
    for i, d in enumerate(defs):
        t = d['type']
        pre = 'module_list.%d.' % i
        if t == 'convolutional':
            x = conv_block(x, d, state, pre, False, fold, leaky_slope)
        elif t == 'depthwise':
            x = conv_block(x, d, state, pre, True, fold, leaky_slope)
        elif t == 'maxpool':
            x = maxpool(x, int(d['size']), int(d['stride']))
        elif t == 'upsample':
            x = upsample_nearest(x, int(d['stride']))
        elif t == 'route':
            srcs = [outs[l] for l in d['layers']]  # negative = relative, positive = absolute (python indexing)
            if len(srcs) > 1:
                x = torch.cat(srcs, 1)
            elif 'groups' in d:
                x = x[:, x.shape[1] // 2:]
            else:
                x = srcs[0]
        elif t == 'shortcut':
            w = None
            if (pre + 'w') in state:
                w = torch.sigmoid(state[pre + 'w'].to(dtype)) * (2.0 / (len(d['from']) + 1))
            x = shortcut(x, [outs[l] for l in d['from']], w)
        elif t == 'se':
            x = se_block(x, state[pre + 'se.fc.0.weight'].to(dtype), state[pre + 'se.fc.2.weight'].to(dtype))
        elif t == 'yolo':
            anchors = np.asarray(d['anchors'])[d['mask']]
            io, raw = yolo_decode(x, anchors, scales[i], int(d['classes']))
            ios.append(io)
            raws.append(raw)
        elif t == 'reorg3d':
            pass
        else:
            raise ValueError('oracle: unsupported block type ' + t)
        outs.append(x)
    res = (torch.cat(ios, 1), raws)
    return res + (outs,) if return_layers else res


# ------------------------------------------------------------------------------------------------ NMS
def _iou_1_to_n(b, bs):
    x1 = np.maximum(b[0], bs[:, 0])
    y1 = np.maximum(b[1], bs[:, 1])
    x2 = np.minimum(b[2], bs[:, 2])
    y2 = np.minimum(b[3], bs[:, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    a = (b[2] - b[0]) * (b[3] - b[1])
    an = (bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1])
    return inter / (a + an - inter)


def nms_indices(boxes, scores, thr):
    """torchvision.ops.boxes.nms contract on numpy fp32 arrays; returns indices in score order."""
    order = np.argsort(-scores, kind='stable')
    keep, dead = [], np.zeros(len(order), dtype=bool)
    b = boxes[order]
    for i in range(len(order)):
        if dead[i]:
            continue
        keep.append(order[i])
        dead |= _iou_1_to_n(b[i], b) > thr
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(pred, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None, agnostic=False):
    """Reference NMS (utils/utils.py:782-860) on a numpy (N, rows, 5+nc) fp32 array.

    Returns a list with one (n_i, 6) fp32 array [x1, y1, x2, y2, conf, cls] or None per image.
    """
    pred = np.asarray(pred, dtype=np.float32)
    nc = pred.shape[2] - 5
    multi_label = bool(multi_label) and nc > 1
    result = []
    for x in pred:
        x = x[x[:, 4] > conf_thres]
        x = x[((x[:, 2:4] > MIN_WH) & (x[:, 2:4] < MAX_WH)).all(1)]
        if len(x) == 0:
            result.append(None)
            continue
        x = x.copy()
        x[:, 5:] *= x[:, 4:5]
        box = np.stack((x[:, 0] - x[:, 2] / 2, x[:, 1] - x[:, 3] / 2, x[:, 0] + x[:, 2] / 2, x[:, 1] + x[:, 3] / 2), 1)
        if multi_label:
            r, c = np.nonzero(x[:, 5:] > conf_thres)
            det = np.concatenate((box[r], x[r, c + 5][:, None], c[:, None].astype(np.float32)), 1)
        else:
            c = x[:, 5:].argmax(1)
            det = np.concatenate((box, x[np.arange(len(x)), c + 5][:, None], c[:, None].astype(np.float32)), 1)
        if classes:
            det = det[np.isin(det[:, 5], np.asarray(classes, dtype=np.float32))]
        det = det[np.isfinite(det).all(1)].astype(np.float32)
        n = len(det)
        if n == 0:
            result.append(None)
            continue
        offs = det[:, 5:6] * (0 if agnostic else 1)
        boxes = det[:, :4] + offs * np.float32(MAX_WH)
        scores = det[:, 4]
        keep = nms_indices(boxes, scores, iou_thres)
        if 1 < n < 3000:  # merge: every kept box becomes the score-weighted mean of the boxes it overlaps
            src = det[:, :4].copy()
            for i in keep:
                w = (_iou_1_to_n(boxes[i], boxes) > iou_thres).astype(np.float32) * scores
                det[i, :4] = (w[None, :] @ src).astype(np.float32)[0] / w.sum(dtype=np.float32)
        result.append(det[keep])
    return result
