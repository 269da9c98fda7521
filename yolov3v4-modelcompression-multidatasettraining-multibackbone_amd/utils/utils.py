"""Box algebra, NMS and detection metrics.

Interface mirror of the hot-path part of the reference's ``utils/utils.py``: ``xyxy2xywh`` :98,
``xywh2xyxy`` :108, ``scale_coords`` :138, ``clip_coords`` :154, ``ap_per_class`` :162,
``compute_ap`` :225, ``bbox_iou`` :254, ``box_iou`` :300, ``wh_iou`` :325,
``non_max_suppression`` :782-860.  Star-importing this module also provides the std names the
reference scripts expect from it (``torch``, ``nn``, ``np``, ``os``, ``math``, ``glob`` ...).

``non_max_suppression`` has two executions of the same algorithm:

* CUDA tensor  -> the HIP kernels of ``csrc/nms.hip`` through ``engine.nms`` (candidate compaction,
  IoU bit-mask matrix, greedy scan, merge).  No silent fallback: a missing library raises.
* CPU tensor   -> the eager host path below (what runs in the CPU-only container and in host tests).

``torchvision.ops.boxes.nms`` (the reference's call at utils.py:843) is not available in this image;
its published contract is restated in ``nms_greedy``: order by score descending, keep a box unless an
already-kept box overlaps it with IoU **>** threshold, return kept indices in score order.
"""
import glob
import math
import os
import random
import shutil
import subprocess
import time
from pathlib import Path
from sys import platform

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

try:  # tqdm is plumbing only
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, *a, **k):
        return x

try:  # opencv is optional in this image; image I/O paths need it, tensor paths do not
    import cv2
    cv2.setNumThreads(0)
except Exception:  # pragma: no cover
    cv2 = None

from . import torch_utils

torch.set_printoptions(linewidth=320, precision=5, profile='long')
np.set_printoptions(linewidth=320, formatter={'float_kind': '{:11.5g}'.format})

MIN_WH, MAX_WH = 2, 4096  # NMS box-size window and the per-class coordinate offset (utils.py:790)


def init_seeds(seed=0):
    random.seed(seed)
    np.random.seed(seed)
    torch_utils.init_seeds(seed=seed)


def load_classes(path):
    with open(path, 'r') as fh:
        return [n for n in fh.read().split('\n') if n]


def coco80_to_coco91_class():
    skipped = {12, 26, 29, 30, 45, 66, 68, 69, 71, 83}
    return [i for i in range(1, 91) if i not in skipped]


# ------------------------------------------------------------------------------------------------ boxes
def _empty_like(x):
    return torch.zeros_like(x) if isinstance(x, torch.Tensor) else np.zeros_like(x)


def xyxy2xywh(x):
    y = _empty_like(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def xywh2xyxy(x):
    y = _empty_like(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def clip_coords(boxes, img_shape):
    boxes[:, 0].clamp_(0, img_shape[1])
    boxes[:, 1].clamp_(0, img_shape[0])
    boxes[:, 2].clamp_(0, img_shape[1])
    boxes[:, 3].clamp_(0, img_shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Map xyxy boxes from the letterboxed network input back to the original image, in place."""
    if ratio_pad is None:
        gain = max(img1_shape) / max(img0_shape)
        pad = ((img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2)
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords


def box_iou(box1, box2):
    """(N,4) x (M,4) xyxy -> (N,M) IoU."""
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    lt = torch.max(box1[:, None, :2], box2[:, :2])
    rb = torch.min(box1[:, None, 2:], box2[:, 2:])
    inter = (rb - lt).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def wh_iou(wh1, wh2):
    wh1 = wh1[:, None]
    wh2 = wh2[None]
    inter = torch.min(wh1, wh2).prod(2)
    return inter / (wh1.prod(2) + wh2.prod(2) - inter)


def bbox_iou(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False):
    """IoU (or G/D/C-IoU) of one box (4,) against n boxes (n,4)."""
    box2 = box2.t()
    if x1y1x2y2:
        ax1, ay1, ax2, ay2 = box1[0], box1[1], box1[2], box1[3]
        bx1, by1, bx2, by2 = box2[0], box2[1], box2[2], box2[3]
    else:
        ax1, ax2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
        ay1, ay2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
        bx1, bx2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
        by1, by2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.min(ax2, bx2) - torch.max(ax1, bx1)).clamp(0) * \
            (torch.min(ay2, by2) - torch.max(ay1, by1)).clamp(0)
    w1, h1 = ax2 - ax1, ay2 - ay1
    w2, h2 = bx2 - bx1, by2 - by1
    union = (w1 * h1 + 1e-16) + w2 * h2 - inter
    iou = inter / union
    if not (GIoU or DIoU or CIoU):
        return iou
    cw = torch.max(ax2, bx2) - torch.min(ax1, bx1)
    ch = torch.max(ay2, by2) - torch.min(ay1, by1)
    if GIoU:
        hull = cw * ch + 1e-16
        return iou - (hull - union) / hull
    diag2 = cw ** 2 + ch ** 2 + 1e-16
    rho2 = ((bx1 + bx2) - (ax1 + ax2)) ** 2 / 4 + ((by1 + by2) - (ay1 + ay2)) ** 2 / 4
    if DIoU:
        return iou - rho2 / diag2
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (1 - iou + v)
    return iou - (rho2 / diag2 + v * alpha)


# ------------------------------------------------------------------------------------------------- loss
class FocalLoss(nn.Module):
    """Focal modulation around an ``nn.BCEWithLogitsLoss`` (reference utils/utils.py:333-360, the TF-addons form)."""

    def __init__(self, loss_fcn, gamma=1.5, alpha=0.25):
        super().__init__()
        self.loss_fcn, self.gamma, self.alpha = loss_fcn, gamma, alpha
        self.reduction = loss_fcn.reduction
        self.loss_fcn.reduction = 'none'

    def forward(self, pred, true):
        loss = self.loss_fcn(pred, true)
        prob = torch.sigmoid(pred)
        p_t = true * prob + (1 - true) * (1 - prob)
        loss = loss * (true * self.alpha + (1 - true) * (1 - self.alpha)) * (1.0 - p_t) ** self.gamma
        if self.reduction == 'mean':
            return loss.mean()
        return loss.sum() if self.reduction == 'sum' else loss


def smooth_BCE(eps=0.1):
    """(positive, negative) BCE targets under label smoothing (utils.py:363-365)."""
    return 1.0 - 0.5 * eps, 0.5 * eps


def _yolo_modules(model):
    core = model.module if type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel) else model
    return [core.module_list[j] for j in core.yolo_layers]


def build_targets(p, targets, model):
    """Assign every label to every anchor of every head whose wh-IoU with it exceeds ``hyp['iou_t']``.

    ``targets`` is (nt, 6) = image, class, x, y, w, h (normalised).  Per head: classes, boxes (cell-relative xy,
    grid-unit wh), index tuple (image, anchor, gy, gx) and the matching anchor vectors (reference utils.py:725-779)."""
    nt = targets.shape[0]
    dev = targets.device
    tcls, tbox, indices, av = [], [], [], []
    for i, layer in enumerate(_yolo_modules(model)):
        anchors = layer.anchor_vec
        ny, nx = p[i].shape[2], p[i].shape[3]
        gain = torch.tensor([1, 1, nx, ny, nx, ny], device=dev, dtype=targets.dtype)
        t = targets * gain
        a = torch.zeros(0, dtype=torch.long, device=dev)
        if nt:
            na = anchors.shape[0]
            iou = wh_iou(anchors.to(dev), t[:, 4:6])                    # (na, nt)
            a = torch.arange(na, device=dev).view(-1, 1).repeat(1, nt).view(-1)
            t = t.repeat(na, 1)
            keep = iou.view(-1) > model.hyp['iou_t']
            t, a = t[keep], a[keep]
        b, c = t[:, :2].long().t()
        gxy, gwh = t[:, 2:4], t[:, 4:6]
        gi, gj = gxy.long().t()
        indices.append((b, a, gj, gi))
        tbox.append(torch.cat((gxy - gxy.floor(), gwh), 1))
        av.append(anchors.to(dev)[a])
        tcls.append(c)
        if c.shape[0] and int(c.max()) >= model.nc:
            raise AssertionError('Model accepts %g classes labeled from 0-%g, however you labelled a class %g. '
                                 % (model.nc, model.nc - 1, int(c.max())))
    return tcls, tbox, indices, av


def _fused_loss_override():
    try:
        from engine import loss as hip_loss
    except Exception:
        return False
    return hip_loss._LIB_OVERRIDE is not None


def compute_loss(p, targets, model, fused=None):
    """GIoU box loss + objectness BCE + class BCE over the raw head tensors (reference utils.py:368-432).

    Returns ``(loss, detached [lbox, lobj, lcls, loss])``; mean reduction, gains from ``model.hyp``, objectness
    target ``(1 - gr) + gr * giou`` with ``model.gr``.  CUDA fp32 raw heads without focal loss go through the fused HIP
    kernels (engine/loss.py, csrc/loss.hip: same values, gradient written directly); everything else runs the torch
    ops below."""
    if fused is not False and (p[0].is_cuda or _fused_loss_override()):
        from engine import loss as hip_loss
        if hip_loss.usable(p, model):
            return hip_loss.compute_loss(p, targets, model, _yolo_modules, smooth_BCE)
    dev = p[0].device
    z = lambda: torch.zeros(1, device=dev)
    lcls, lbox, lobj = z(), z(), z()
    tcls, tbox, indices, anchor_vec = build_targets(p, targets, model)
    h = model.hyp
    bce_cls = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([h['cls_pw']], device=dev), reduction='mean')
    bce_obj = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([h['obj_pw']], device=dev), reduction='mean')
    cp, cn = smooth_BCE(eps=0.0)
    if h['fl_gamma'] > 0:
        bce_cls, bce_obj = FocalLoss(bce_cls, h['fl_gamma']), FocalLoss(bce_obj, h['fl_gamma'])
    for i, pi in enumerate(p):
        b, a, gj, gi = indices[i]
        tobj = torch.zeros_like(pi[..., 0])
        nb = len(b)
        if nb:
            ps = pi[b, a, gj, gi]
            pxy = torch.sigmoid(ps[:, 0:2])
            pwh = torch.exp(ps[:, 2:4]).clamp(max=1E3) * anchor_vec[i]
            giou = bbox_iou(torch.cat((pxy, pwh), 1).t(), tbox[i], x1y1x2y2=False, GIoU=True)
            lbox = lbox + (1.0 - giou).mean()
            tobj[b, a, gj, gi] = (1.0 - model.gr) + model.gr * giou.detach().clamp(0).type(tobj.dtype)
            if model.nc > 1:
                t = torch.full_like(ps[:, 5:], cn)
                t[range(nb), tcls[i]] = cp
                lcls = lcls + bce_cls(ps[:, 5:], t)
        lobj = lobj + bce_obj(pi[..., 4], tobj)
    lbox = lbox * h['giou']
    lobj = lobj * h['obj']
    lcls = lcls * h['cls']
    loss = lbox + lobj + lcls
    return loss, torch.cat((lbox, lobj, lcls, loss)).detach()


def compute_lost_KD(output_s, output_t, num_classes, batch_size):
    """Hinton soft-target distillation between student and teacher raw heads (reference utils.py:435-444)."""
    T, weight = 3.0, 0.001
    s = torch.cat([o.reshape(-1, num_classes + 5) for o in output_s])
    t = torch.cat([o.reshape(-1, num_classes + 5) for o in output_t])
    kl = nn.KLDivLoss(reduction='sum')(F.log_softmax(s / T, dim=1), F.softmax(t / T, dim=1))
    return kl * (T * T) / batch_size * weight


# -------------------------------------------------------------------------------------------------- NMS
def nms_greedy(boxes, scores, iou_thres):
    """Host restatement of torchvision.ops.boxes.nms (see module docstring). Returns LongTensor."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].detach().cpu().float().numpy()
    # one IoU ROW per kept box (box_iou's arithmetic in fp32) instead of the n x n matrix: evaluation settings (test.py: conf 0.001,
    # multi-label) give 10^4 candidates per image, 2 GB of matrix and a minute of Python per image in the quadratic form
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    alive = np.ones(n, dtype=bool)
    thr = np.float32(iou_thres)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        w = np.minimum(b[i, 2], b[i + 1:, 2]) - np.maximum(b[i, 0], b[i + 1:, 0])
        h = np.minimum(b[i, 3], b[i + 1:, 3]) - np.maximum(b[i, 1], b[i + 1:, 1])
        inter = np.maximum(w, np.float32(0)) * np.maximum(h, np.float32(0))
        with np.errstate(divide='ignore', invalid='ignore'):      # degenerate boxes: 0 / 0 = NaN compares False, as in torch
            alive[i + 1:] &= ~(inter / (area[i] + area[i + 1:] - inter) > thr)
    return order[torch.tensor(keep, dtype=torch.long, device=boxes.device)]


def _nms_one_image_host(x, conf_thres, iou_thres, multi_label, classes, agnostic):
    """Reference algorithm for one image's (rows, 5+nc) prediction block; returns (n,6) or None."""
    x = x[x[:, 4] > conf_thres]
    wh = x[:, 2:4]
    x = x[((wh > MIN_WH) & (wh < MAX_WH)).all(1)]
    if x.shape[0] == 0:
        return None
    x[:, 5:] *= x[:, 4:5]
    box = xywh2xyxy(x[:, :4])
    if multi_label:
        rows, cls = (x[:, 5:] > conf_thres).nonzero(as_tuple=False).t()
        x = torch.cat((box[rows], x[rows, cls + 5].unsqueeze(1), cls.float().unsqueeze(1)), 1)
    else:
        conf, cls = x[:, 5:].max(1)
        x = torch.cat((box, conf.unsqueeze(1), cls.float().unsqueeze(1)), 1)
    if classes:
        x = x[(cls.view(-1, 1) == torch.tensor(classes, device=cls.device)).any(1)]
    finite = torch.isfinite(x).all(1)
    if not finite.all():
        x = x[finite]
    n = x.shape[0]
    if n == 0:
        return None
    offs = x[:, 5] * 0 if agnostic else x[:, 5]
    boxes = x[:, :4].clone() + offs.view(-1, 1) * MAX_WH
    scores = x[:, 4]
    keep = nms_greedy(boxes, scores, iou_thres)
    if 1 < n < 3000:  # merge-NMS: kept boxes become score-weighted means of what they suppress
        w = (box_iou(boxes[keep], boxes) > iou_thres) * scores[None]
        x[keep, :4] = torch.mm(w, x[:, :4]).float() / w.sum(1, keepdim=True)
    return x[keep]


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None, agnostic=False):
    """(N, rows, 5+nc) decoded head output -> list of (n_i, 6) [x1,y1,x2,y2,conf,cls] or None."""
    nc = prediction[0].shape[1] - 5
    multi_label = bool(multi_label) and nc > 1
    if isinstance(prediction, torch.Tensor) and prediction.is_cuda:
        from engine import nms as hip_nms  # HIP path; raises if the library is missing
        return hip_nms.non_max_suppression(prediction, conf_thres, iou_thres, multi_label, classes, agnostic)
    out = [None] * len(prediction)
    for k, x in enumerate(prediction):
        out[k] = _nms_one_image_host(x, conf_thres, iou_thres, multi_label, classes, agnostic)
    return out


# ----------------------------------------------------------------------------------------------- metrics
def compute_ap(recall, precision):
    """101-point interpolated AP of one PR curve (COCO style)."""
    mrec = np.concatenate(([0.], recall, [min(recall[-1] + 1E-3, 1.)]))
    mpre = np.concatenate(([0.], precision, [0.]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    trapz = getattr(np, 'trapezoid', None) or np.trapz
    return trapz(np.interp(x, mrec, mpre), x)


def ap_per_class(tp, conf, pred_cls, target_cls):
    """Per-class P, R, AP, F1 from per-detection TP flags (n, n_iou), scores and classes."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    unique_classes = np.unique(target_cls)
    pr_score = 0.1
    shape = [len(unique_classes), tp.shape[1]]
    ap, p, r = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for ci, c in enumerate(unique_classes):
        sel = pred_cls == c
        n_gt = (target_cls == c).sum()
        if sel.sum() == 0 or n_gt == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_gt + 1e-16)
        precision = tpc / (tpc + fpc)
        r[ci] = np.interp(-pr_score, -conf[sel], recall[:, 0])
        p[ci] = np.interp(-pr_score, -conf[sel], precision[:, 0])
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p * r / (p + r + 1e-16)
    return p, r, ap, f1, unique_classes.astype('int32')


def fitness(x):
    w = [0.0, 0.0, 0.8, 0.2]  # weights for [P, R, mAP, F1]
    return (x[:, :4] * w).sum(1)


def get_yolo_layers(model):
    return [i for i, d in enumerate(model.module_defs) if d['type'] == 'yolo']


# ------------------------------------------------------------------------------------ training / reporting helpers
def labels_to_class_weights(labels, nc=80):
    """Inverse-frequency class weights from the dataset's label arrays (reference utils.py:44-58)."""
    if len(labels) == 0 or labels[0] is None:
        return torch.Tensor()
    classes = np.concatenate(labels, 0)[:, 0].astype(np.int64) if len(labels) else np.zeros(0, np.int64)
    weights = np.bincount(classes, minlength=nc).astype(np.float64)
    weights[weights == 0] = 1
    weights = 1 / weights
    return torch.from_numpy(weights / weights.sum()).float()


def labels_to_image_weights(labels, nc=80, class_weights=np.ones(80)):
    """Per-image sampling weight = class weights dotted with the image's class histogram (reference utils.py:61-68)."""
    counts = np.array([np.bincount(l[:, 0].astype(np.int64), minlength=nc) for l in labels])
    return (np.asarray(class_weights).reshape(1, nc) * counts).sum(1)


def output_to_target(output, width, height):
    """NMS output list -> (n, 7) rows [image, class, x, y, w, h, conf] normalised, for plotting (reference utils.py:1011-1036)."""
    rows = []
    for i, o in enumerate(output):
        if o is None:
            continue
        o = o.detach().cpu().numpy() if isinstance(o, torch.Tensor) else np.asarray(o)
        for x1, y1, x2, y2, conf, cls in o[:, :6]:
            rows.append([i, cls, ((x1 + x2) / 2) / width, ((y1 + y2) / 2) / height, (x2 - x1) / width, (y2 - y1) / height, conf])
    return np.array(rows, dtype=np.float32).reshape(-1, 7)


def plot_one_box(x, img, color=None, label=None, line_thickness=None):
    """Draw one xyxy box (and label) into an HWC uint8 RGB array, in place (reference utils.py:962-975; PIL instead of cv2)."""
    from PIL import Image, ImageDraw
    color = tuple(int(c) for c in (color or [random.randint(0, 255) for _ in range(3)]))
    tl = line_thickness or max(round(0.002 * (img.shape[0] + img.shape[1]) / 2), 1)
    pil = Image.fromarray(img)
    draw = ImageDraw.Draw(pil)
    box = [int(v) for v in x[:4]]
    draw.rectangle(box, outline=color, width=int(tl))
    if label:
        w, h = 6 * len(label) + 4, 12
        draw.rectangle([box[0], max(box[1] - h, 0), box[0] + w, max(box[1] - h, 0) + h], fill=color)
        draw.text((box[0] + 2, max(box[1] - h, 0)), label, fill=(255, 255, 255))
    img[:] = np.asarray(pil)


def plot_images(images, targets, paths=None, fname='images.jpg', names=None, max_size=640, max_subplots=16, is_gray_scale=False):
    """Mosaic of a batch with its boxes, written to ``fname`` (reference utils.py:1039-1111; PIL instead of cv2)."""
    from PIL import Image
    if os.path.isfile(fname):
        return None
    imgs = images.detach().cpu().float().numpy() if isinstance(images, torch.Tensor) else np.asarray(images, dtype=np.float32)
    tg = targets.detach().cpu().numpy() if isinstance(targets, torch.Tensor) else np.asarray(targets)
    if imgs.max() <= 1.0 + 1e-6:
        imgs = imgs * 255
    bs, c, h, w = imgs.shape
    bs = min(bs, max_subplots)
    ns = int(np.ceil(bs ** 0.5))
    scale = min(max_size / max(h, w), 1.0)
    hh, ww = int(math.ceil(h * scale)), int(math.ceil(w * scale))
    canvas = np.full((ns * hh, ns * ww, 3), 255, dtype=np.uint8)
    for i in range(bs):
        im = imgs[i].transpose(1, 2, 0).clip(0, 255).astype(np.uint8)
        if im.shape[2] == 1:
            im = np.repeat(im, 3, 2)
        if scale < 1:
            im = np.asarray(Image.fromarray(im).resize((ww, hh)))
        im = np.ascontiguousarray(im)
        rows = tg[tg[:, 0] == i] if len(tg) else []
        for r in rows:
            cls = int(r[1])
            cx, cy, bw, bh = r[2] * ww, r[3] * hh, r[4] * ww, r[5] * hh
            conf = r[6] if len(r) > 6 else None
            if conf is None or conf > 0.3:
                label = (names[cls] if names else str(cls)) + ('' if conf is None else ' %.1f' % conf)
                plot_one_box([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], im, color=[(37 * cls) % 255, (91 * cls) % 255, 200],
                             label=label, line_thickness=1)
        y0, x0 = (i // ns) * hh, (i % ns) * ww
        canvas[y0:y0 + hh, x0:x0 + ww] = im
    if fname:
        Image.fromarray(canvas).save(fname)
    return canvas


def strip_optimizer(f='weights/best.pt'):
    """Drop optimizer state from a checkpoint so it can be shared (reference utils.py:862-868)."""
    x = torch.load(f, map_location='cpu', weights_only=False)
    x['optimizer'] = None
    torch.save(x, f)


def print_mutation(hyp, results, bucket=''):
    """Append one hyper-parameter evolution record to evolve.txt (reference utils.py:921-939; no cloud bucket here)."""
    a = '%10s' * len(hyp) % tuple(hyp.keys())
    b = '%10.3g' * len(hyp) % tuple(hyp.values())
    c = '%10.4g' * len(results) % tuple(results)
    print('\n%s\n%s\nEvolved fitness: %s\n' % (a, b, c))
    with open('evolve.txt', 'a') as f:
        f.write(c + b + '\n')


def plot_results(start=0, stop=0, bucket='', id=()):
    """results*.txt curves -> results.png when matplotlib is importable (reference utils.py:1187-1218); otherwise a no-op."""
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
    except Exception:
        return
    files = sorted(glob.glob('results*.txt'))
    if not files:
        return
    titles = ['GIoU', 'Objectness', 'Classification', 'Precision', 'Recall', 'val GIoU', 'val Objectness', 'val Classification', 'mAP@0.5', 'F1']
    fig, ax = plt.subplots(2, 5, figsize=(14, 7))
    ax = ax.ravel()
    for f in files:
        try:
            res = np.loadtxt(f, usecols=[2, 3, 4, 8, 9, 12, 13, 14, 10, 11], ndmin=2).T
        except Exception:
            continue
        x = range(start, min(stop, res.shape[1]) if stop else res.shape[1])
        for i in range(10):
            ax[i].plot(x, res[i, x], marker='.', label=Path(f).stem)
            ax[i].set_title(titles[i])
    ax[1].legend()
    fig.tight_layout()
    fig.savefig('results.png', dpi=150)
    plt.close(fig)
