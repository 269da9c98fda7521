"""Batched NMS on the HIP kernels of ``csrc/nms.hip`` (reference: utils/utils.py:782-860).

Two host synchronisations are inherent to the reference's contract (a Python list of variable-length
tensors): one to size the candidate buffers, one to learn how many boxes survived.
"""
import torch

from . import hiplib

MERGE_LO, MERGE_HI = 1, 3000  # merge-NMS applies for 1 < n < 3000 (utils.py:845)


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None, agnostic=False):
    with hiplib.on_device(prediction):
        return _non_max_suppression(prediction, conf_thres, iou_thres, multi_label, classes, agnostic)


def _non_max_suppression(prediction, conf_thres, iou_thres, multi_label, classes, agnostic):
    lib = hiplib.load()
    if prediction.dim() != 3:
        raise ValueError('expected (N, rows, 5 + nc)')
    pred = prediction.contiguous()
    if pred.dtype != torch.float32:
        pred = pred.float()
    n, rows, no = pred.shape
    nc = no - 5
    dev = pred.device
    P, S = hiplib.ptr, hiplib.stream_ptr()
    cmask = None
    if classes:
        cmask = torch.zeros(nc, dtype=torch.uint8, device=dev)
        cmask[torch.as_tensor(list(classes), dtype=torch.long, device=dev)] = 1
    ml = 1 if (multi_label and nc > 1) else 0

    count = torch.zeros(n, dtype=torch.int32, device=dev)
    hiplib.check(lib.yh_nms_candidates(P(pred), n, rows, nc, conf_thres, ml, P(cmask), None, P(count), 0, S), 'nms count')
    counts = count.cpu()
    mmax = int(counts.max())
    out = [None] * n
    if mmax == 0:
        return out
    cap = mmax
    cand = torch.empty((n, cap, 8), dtype=torch.float32, device=dev)
    count.zero_()
    hiplib.check(lib.yh_nms_candidates(P(pred), n, rows, nc, conf_thres, ml, P(cmask), P(cand), P(count), cap, S), 'nms cand')
    srt = torch.empty_like(cand)
    hiplib.check(lib.yh_nms_sort(P(cand), P(count), n, cap, mmax, P(srt), S), 'nms sort')
    words = (mmax + 63) // 64
    mask = torch.empty((n, mmax, words), dtype=torch.int64, device=dev)
    hiplib.check(lib.yh_nms_mask(P(srt), P(count), n, cap, mmax, iou_thres, 1 if agnostic else 0, P(mask), S), 'nms mask')
    keep_idx = torch.empty((n, cap), dtype=torch.int32, device=dev)
    n_keep = torch.zeros(n, dtype=torch.int32, device=dev)
    hiplib.check(lib.yh_nms_reduce(P(mask), P(count), n, cap, mmax, P(keep_idx), P(n_keep), S), 'nms reduce')
    kept = n_keep.cpu()
    kmax = int(kept.max())
    if kmax == 0:
        return out
    res = torch.empty((n, cap, 6), dtype=torch.float32, device=dev)
    hiplib.check(lib.yh_nms_merge(P(srt), P(count), P(keep_idx), P(n_keep), n, cap, kmax, iou_thres, 1 if agnostic else 0,
                                  MERGE_LO, MERGE_HI, P(res), S), 'nms merge')
    for i in range(n):
        k = int(kept[i])
        if k > 0:
            out[i] = res[i, :k]
    return out
