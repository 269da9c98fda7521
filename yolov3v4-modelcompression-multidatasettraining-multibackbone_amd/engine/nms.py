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

Round 5: with per-class offsets (not agnostic) and multi-label candidates the suppression runs class by class
(``yh_nms_class_scan``: one workgroup per (image, class), boxes in LDS, no bit mask - csrc/nms.hip says when that is the
reference's result bit for bit and checks it on the device).  Its buffers are linear in the bound, so the mask budget does not
apply; the single read also returns the per-image verdict, and only a batch with an image that needs the general form (a class
above 2048 candidates, or candidates spread over more than 4096 pixels in x and y) takes the mask path afterwards.
``YOLO_HIP_NMS_SEGMENTED`` = 0 (never) / 1 (whenever not agnostic) / unset (multi-label only) selects it; its sort is the tile
sort of ``yh_nms_sort_tiles`` (``YOLO_HIP_NMS_SORT=count``: the counting sort of the bit-mask form).
"""
import collections
import ctypes as C
import os

import torch

from . import hiplib

MERGE_LO, MERGE_HI = 1, 3000  # merge-NMS applies for 1 < n < 3000 (utils.py:845)
_CAP_MIN = 256
_HINT_WINDOW = 8              # calls a density observation is remembered for
_MASK_BUDGET = 256 << 20      # bytes of IoU bit mask a *guessed* bound may cost; above it the counts are read first
_WORK_BUDGET = 8 << 30        # bytes of candidate / mask buffers one pass may hold: larger batches are processed in image chunks
_density = {}                 # multi_label -> deque of the last candidate densities (max candidates of an image / rows)


class _Rows:
    """Where the candidates come from: the decoded (N, rows, 5 + nc) tensor (yh_nms_candidates) ..."""

    def __init__(self, pred):
        if pred.dim() != 3:
            raise ValueError('expected (N, rows, 5 + nc)')
        pred = pred.contiguous()
        self.pred = pred if pred.dtype == torch.float32 else pred.float()
        self.n, self.rows, no = self.pred.shape
        self.nc, self.device = no - 5, self.pred.device

    def candidates(self, lib, conf, ml, cmask, cand, count, cap, stream):
        hiplib.check(lib.yh_nms_candidates(hiplib.ptr(self.pred), self.n, self.rows, self.nc, conf, ml, hiplib.ptr(cmask), hiplib.ptr(cand),
                                           hiplib.ptr(count), cap, stream), 'nms candidates')

    def images(self, i0, i1):
        return _Rows(self.pred[i0:i1])


class _Heads:
    """... or the head convolutions' outputs themselves (yh_yolo_decode_candidates: decode and filter in one pass, DarknetEngine.detect)."""

    def __init__(self, descs, n, rows, nc, device):
        self.descs, self.n, self.rows, self.nc, self.device = descs, n, rows, nc, device

    def candidates(self, lib, conf, ml, cmask, cand, count, cap, stream):
        for d in self.descs:
            hiplib.check(lib.yh_yolo_decode_candidates(C.byref(d), conf, ml, hiplib.ptr(cmask), hiplib.ptr(cand), hiplib.ptr(count), cap,
                                                       stream), 'decode + candidates')

    def images(self, i0, i1):
        part = []
        for d in self.descs:
            e = hiplib.DecodeDesc.from_buffer_copy(d)
            e.p = d.p + i0 * d.ny * d.nx * d.ldp * 4          # fp32 NHWC with pitch ldp, image-major
            e.n = i1 - i0
            part.append(e)
        return _Heads(part, i1 - i0, self.rows, self.nc, self.device)


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None, agnostic=False):
    with hiplib.on_device(prediction):
        return _non_max_suppression(_Rows(prediction), conf_thres, iou_thres, multi_label, classes, agnostic)


def non_max_suppression_heads(descs, n, rows, nc, device, conf_thres=0.1, iou_thres=0.6, multi_label=True, classes=None, agnostic=False):
    """The same call on the yolo heads' raw outputs (``descs``: one yh_decode_desc per head, as the plan holds them)."""
    return _non_max_suppression(_Heads(descs, n, rows, nc, device), conf_thres, iou_thres, multi_label, classes, agnostic)


def _pow2_at_least(v):
    c = _CAP_MIN
    while c < v:
        c *= 2
    return c


def _non_max_suppression(src, conf_thres, iou_thres, multi_label, classes, agnostic, cap_exact=None):
    """cap_exact: a bound that is KNOWN to hold every image's candidates (set by `in_chunks` for its sub-calls: the chunks were sized
    with it, so the sub-call must not pick a larger one from the density history - ADVICE r5: with the inflated guess a hand-over
    re-chunked the whole batch as one 'chunk' for ever)."""
    lib = hiplib.load()
    n, rows, nc, dev = src.n, src.rows, src.nc, src.device
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
    how = os.environ.get('YOLO_HIP_NMS_SEGMENTED', '')
    seg = (not ag) and nc <= 255 and n <= 65535 and how != '0' and (ml == 1 or how == '1')
    exact = False
    work_budget = {}      # resolved at most once per call, and only when a pass is large enough to ask (budget())

    def per_image_bytes(c, general=True):
        return (c * ((c + 63) // 64) * 8 if general else 14 * c) + c * (8 + 8 + 6 + 1) * 4

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

    def too_big(c, general):
        """Do the buffers of one pass over the whole batch exceed the budget?  (Asks the device only when they are large at all.)"""
        need = n * per_image_bytes(c, general)
        return need > min(64 << 20, _WORK_BUDGET) and need > budget()

    def in_chunks(c, general):
        """Run the images in groups whose buffers fit the budget and concatenate the results."""
        if cap_exact is None or c < cap_exact:
            _density.setdefault(ml, collections.deque(maxlen=_HINT_WINDOW)).append(c / float(rows))     # a TRUE count
        capc = _pow2_at_least(c)
        each = per_image_bytes(capc, general)
        if each > budget():
            raise MemoryError('non_max_suppression: %d candidates in one image need %.1f GB of buffers%s (conf_thres %g, %s): raise '
                              'conf_thres' % (c, each / 1e9, ', most of it the IoU bit mask' if general else '', conf_thres,
                                              'multi-label' if ml else 'best class'))
        step = max(1, int(budget() // each))
        if step >= n and cap_exact == capc:
            # this call WAS sized with capc and still does not fit (the device has less free memory than when the caller measured it):
            # halve rather than hand the same batch down again
            step = max(1, n // 2)
            if n == 1:
                raise MemoryError('non_max_suppression: the buffers of one image (%.1f GB) do not fit the free device memory' % (each / 1e9))
        out = []
        for i in range(0, n, step):
            out += _non_max_suppression(src.images(i, min(n, i + step)), conf_thres, iou_thres, multi_label, classes, agnostic, cap_exact=capc)
        return out

    if cap_exact is not None:
        cap, exact = cap_exact, True          # the caller's chunks were sized with this bound
        if too_big(cap, not seg):
            return in_chunks(cap, not seg)
    elif (not seg and n * cap * ((cap + 63) // 64) * 8 > _MASK_BUDGET) or too_big(cap, not seg):
        # a guessed bound this large is not worth its mask: count first (one extra 4n-byte read), then size exactly
        count = torch.zeros(n, dtype=torch.int32, device=dev)
        src.candidates(lib, conf_thres, ml, cmask, None, count, 0, S)
        cmax = max(int(count.max()), 1)
        cap, exact = _pow2_at_least(cmax), True
        if too_big(cap, not seg):
            return in_chunks(cmax, not seg)
    while True:
        words = (cap + 63) // 64
        ctl = torch.zeros((10, n), dtype=torch.int32, device=dev)     # row 0: candidates per image, row 1: survivors, rest: state [n][8]
        count, n_keep, state = ctl[0], ctl[1], ctl[2:].view(n, 8)
        cand = torch.empty((n, cap, 8), dtype=torch.float32, device=dev)
        src.candidates(lib, conf_thres, ml, cmask, cand, count, cap, S)
        srt = torch.empty_like(cand)
        keep_idx = torch.empty((n, cap), dtype=torch.int32, device=dev)
        res = torch.empty((n, cap, 6), dtype=torch.float32, device=dev)

        def general_steps():
            mask = torch.empty((n, cap, words), dtype=torch.int64, device=dev)
            hiplib.check(lib.yh_nms_mask(P(srt), P(count), n, cap, cap, iou_thres, ag, P(mask), S), 'nms mask')
            hiplib.check(lib.yh_nms_reduce(P(mask), P(count), n, cap, cap, P(keep_idx), P(n_keep), S), 'nms reduce')

        def merge_step():
            hiplib.check(lib.yh_nms_merge(P(srt), P(count), P(keep_idx), P(n_keep), n, cap, cap, iou_thres, ag,
                                          MERGE_LO, MERGE_HI, P(res), S), 'nms merge')
        if seg:
            state[:, 1:3] = -1       # minima start at the largest encoding
            cls8 = torch.empty((n, cap), dtype=torch.uint8, device=dev)
            keep8 = torch.empty((n, cap), dtype=torch.uint8, device=dev)
            if os.environ.get('YOLO_HIP_NMS_SORT', 'tiles') == 'tiles':      # O(m log m): tile sort in LDS + rank by binary search
                ws = torch.empty(n * cap * 12, dtype=torch.uint8, device=dev)
                hiplib.check(lib.yh_nms_sort_tiles(P(cand), P(count), n, cap, cap, P(srt), P(cls8), P(ws), ws.numel(), S), 'nms sort')
            else:                                                               # 'count': the O(m^2) counting sort (A/B)
                hiplib.check(lib.yh_nms_sort_cls(P(cand), P(count), n, cap, cap, P(srt), P(cls8), S), 'nms sort')
            hiplib.check(lib.yh_nms_class_scan(P(srt), P(cls8), P(count), n, cap, nc, iou_thres, P(keep8), P(state), P(keep_idx),
                                               P(n_keep), S), 'nms class scan')
        else:
            hiplib.check(lib.yh_nms_sort(P(cand), P(count), n, cap, cap, P(srt), S), 'nms sort')
            general_steps()
        merge_step()
        host = ctl.cpu()                       # the one device-to-host read of the call
        mmax = int(host[0].max())
        if mmax <= cap:
            if seg and bool(host[2:].view(n, 8)[:, 5].any()):
                # an image the class-by-class form does not cover: the bit-mask form on the sorted records that are already there
                # (a second read; its mask is quadratic in the bound, so the budget decides between one pass and image chunks)
                if too_big(cap, True):
                    return in_chunks(mmax, True)
                general_steps()
                merge_step()
                host = ctl.cpu()
            break
        assert not exact, 'candidate count changed between the count pass and the emit pass'
        cap = _pow2_at_least(mmax)             # an image overflowed the bound: repeat with one that holds every candidate
        if too_big(cap, not seg):
            return in_chunks(mmax, not seg)
    _density.setdefault(ml, collections.deque(maxlen=_HINT_WINDOW)).append(mmax / float(rows))
    out = [None] * n
    for i in range(n):
        k = int(host[1, i])
        if k > 0:
            out[i] = res[i, :k]
    return out
