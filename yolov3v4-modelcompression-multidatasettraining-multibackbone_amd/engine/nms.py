"""Batched NMS on the HIP kernels of ``csrc/nms.hip`` (reference: utils/utils.py:782-860).

The reference's contract is a Python list of variable-length tensors, so ONE device-to-host read is inherent (how many boxes
survived per image).  The candidate buffers are therefore sized by an upper bound instead of by a first read-back of the exact
counts; every kernel takes the per-image counts from device memory.  The single read returns candidate counts and survivor
counts together; only when an image had more candidates than the bound the pass is repeated with an exact bound (a second read).

Evaluation settings (test.py: conf 0.001, multi-label) produce 10^3 - 10^5 candidates per image, and the IoU bit mask is quadratic
in the bound: when the buffers of a batch would exceed ``_WORK_BUDGET`` the images are processed in chunks (the reference's loop is
per image anyway, utils.py:797); a single image beyond the budget raises MemoryError with the candidate count (torchvision's
bit-mask NMS, which the reference calls, fails the same way).

The bound is a power of two with 2x head-room over the candidate DENSITY (candidates per prediction row) of the last few calls:
keyed on the label mode only, so the rectangular shapes of an evaluation run share it, and windowed, so one outlier batch (or
a test.py call at conf 0.001 before a detect.py call at conf 0.3) stops costing memory after ``_HINT_WINDOW`` calls.  The IoU
bit mask is quadratic in the bound (n * cap * cap / 8 bytes); above ``_MASK_BUDGET`` the bound is not trusted and the call takes
the exact path: a count-only pass, one extra read, buffers sized by the true maximum.
"""
import collections

import torch

from . import hiplib

MERGE_LO, MERGE_HI = 1, 3000  # merge-NMS applies for 1 < n < 3000 (utils.py:845)
_CAP_MIN = 256
_HINT_WINDOW = 8              # calls a density observation is remembered for
_MASK_BUDGET = 256 << 20      # bytes of IoU bit mask a *guessed* bound may cost; above it the counts are read first
_WORK_BUDGET = 8 << 30        # bytes of candidate / mask buffers one pass may hold: larger batches are processed in image chunks
_density = {}                 # multi_label -> deque of the last candidate densities (max candidates of an image / rows)


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None, agnostic=False):
    with hiplib.on_device(prediction):
        return _non_max_suppression(prediction, conf_thres, iou_thres, multi_label, classes, agnostic)


def _pow2_at_least(v):
    c = _CAP_MIN
    while c < v:
        c *= 2
    return c


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
    most = rows * (nc if ml else 1)           # no image can emit more candidates than this
    seen = _density.get(ml)
    guess = _CAP_MIN if not seen else _pow2_at_least(int(2 * max(seen) * rows) + 1)
    cap = min(guess, _pow2_at_least(most))
    ag = 1 if agnostic else 0
    exact = False
    work_budget = {}      # resolved at most once per call, and only when a pass is large enough to ask (budget())

    def per_image_bytes(c):
        return c * ((c + 63) // 64) * 8 + c * (8 + 8 + 6 + 1) * 4

    def budget():
        """Bytes the buffers of one pass may take: the fixed ceiling, and never more than half of what the device has free right now (an
        evaluation inside a training run shares the GPU with ~40 GB of step buffers; smaller parts have less than the ceiling in all)."""
        b = work_budget.get('v')
        if b is None:
            try:
                free = torch.cuda.mem_get_info(dev)[0] if dev.type == 'cuda' else _WORK_BUDGET
            except Exception:
                free = _WORK_BUDGET
            b = work_budget['v'] = int(min(_WORK_BUDGET, max(free // 2, 64 << 20)))
        return b

    def in_chunks(c):
        """Run the images in groups whose buffers fit the budget and concatenate the results."""
        if per_image_bytes(c) > budget():
            raise MemoryError('non_max_suppression: %d candidates in one image need a %.1f GB IoU bit mask (conf_thres %g, %s): raise '
                              'conf_thres' % (c, per_image_bytes(c) / 1e9, conf_thres, 'multi-label' if ml else 'best class'))
        step = max(1, int(budget() // per_image_bytes(c)))
        out = []
        for i in range(0, n, step):
            out += _non_max_suppression(pred[i:i + step], conf_thres, iou_thres, multi_label, classes, agnostic)
        return out

    if n * cap * ((cap + 63) // 64) * 8 > _MASK_BUDGET:
        # a guessed bound this large is not worth its mask: count first (one extra 4n-byte read), then size exactly
        count = torch.zeros(n, dtype=torch.int32, device=dev)
        hiplib.check(lib.yh_nms_candidates(P(pred), n, rows, nc, conf_thres, ml, P(cmask), None, P(count), 0, S), 'nms count')
        cmax = max(int(count.max()), 1)
        cap, exact = _pow2_at_least(cmax), True
        if n * per_image_bytes(cap) > budget() and (n > 1 or per_image_bytes(cap) > budget()):
            _density.setdefault(ml, collections.deque(maxlen=_HINT_WINDOW)).append(cmax / float(rows))     # the TRUE count
            return in_chunks(cap)
    while True:
        words = (cap + 63) // 64
        counts = torch.zeros((2, n), dtype=torch.int32, device=dev)     # row 0: candidates per image, row 1: survivors
        count, n_keep = counts[0], counts[1]
        cand = torch.empty((n, cap, 8), dtype=torch.float32, device=dev)
        hiplib.check(lib.yh_nms_candidates(P(pred), n, rows, nc, conf_thres, ml, P(cmask), P(cand), P(count), cap, S), 'nms cand')
        srt = torch.empty_like(cand)
        hiplib.check(lib.yh_nms_sort(P(cand), P(count), n, cap, cap, P(srt), S), 'nms sort')
        mask = torch.empty((n, cap, words), dtype=torch.int64, device=dev)
        hiplib.check(lib.yh_nms_mask(P(srt), P(count), n, cap, cap, iou_thres, ag, P(mask), S), 'nms mask')
        keep_idx = torch.empty((n, cap), dtype=torch.int32, device=dev)
        hiplib.check(lib.yh_nms_reduce(P(mask), P(count), n, cap, cap, P(keep_idx), P(n_keep), S), 'nms reduce')
        res = torch.empty((n, cap, 6), dtype=torch.float32, device=dev)
        hiplib.check(lib.yh_nms_merge(P(srt), P(count), P(keep_idx), P(n_keep), n, cap, cap, iou_thres, ag,
                                      MERGE_LO, MERGE_HI, P(res), S), 'nms merge')
        host = counts.cpu()                    # the one device-to-host read of the call
        mmax = int(host[0].max())
        if mmax <= cap:
            break
        assert not exact, 'candidate count changed between the count pass and the emit pass'
        cap = _pow2_at_least(mmax)             # an image overflowed the bound: repeat with one that holds every candidate
        if n * per_image_bytes(cap) > budget() and (n > 1 or per_image_bytes(cap) > budget()):
            _density.setdefault(ml, collections.deque(maxlen=_HINT_WINDOW)).append(mmax / float(rows))
            return in_chunks(cap)
    _density.setdefault(ml, collections.deque(maxlen=_HINT_WINDOW)).append(mmax / float(rows))
    out = [None] * n
    for i in range(n):
        k = int(host[1, i])
        if k > 0:
            out[i] = res[i, :k]
    return out
