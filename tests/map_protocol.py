"""Synthetic mAP@0.5 protocol: detections of the fp32 CPU reference path are the ground
truth; a candidate engine's detections are scored with the reference's matcher (test.py:156-182) and
``ap_per_class`` (utils/utils.py:162).  The reference path scores 1.0 against itself by construction."""
import numpy as np
import torch

from utils.utils import ap_per_class, box_iou


def map50(gt_dets, cand_dets):
    """gt_dets / cand_dets: per-image list of (n,6) [x1,y1,x2,y2,conf,cls] tensors or None."""
    stats = []
    for gt, pred in zip(gt_dets, cand_dets):
        nl = 0 if gt is None else len(gt)
        tcls = gt[:, 5].tolist() if nl else []
        if pred is None or len(pred) == 0:
            if nl:
                stats.append((np.zeros((0, 1), dtype=bool), np.zeros(0), np.zeros(0), tcls))
            continue
        pred = pred.detach().float().cpu()
        correct = torch.zeros(pred.shape[0], 1, dtype=torch.bool)
        if nl:
            g = gt.detach().float().cpu()
            detected = []
            for cls in torch.unique(g[:, 5]):
                ti = (cls == g[:, 5]).nonzero(as_tuple=False).view(-1)
                pi = (cls == pred[:, 5]).nonzero(as_tuple=False).view(-1)
                if pi.shape[0]:
                    ious, i = box_iou(pred[pi, :4], g[ti, :4]).max(1)
                    for j in (ious > 0.5).nonzero(as_tuple=False):
                        d = ti[i[j]]
                        if d.item() not in detected:
                            detected.append(d.item())
                            correct[pi[j]] = True
                            if len(detected) == nl:
                                break
        stats.append((correct.numpy(), pred[:, 4].numpy(), pred[:, 5].numpy(), tcls))
    if not stats:
        return float('nan')
    stats = [np.concatenate(x, 0) for x in zip(*stats)]
    p, r, ap, f1, _ = ap_per_class(*stats)
    return float(ap[:, 0].mean())
