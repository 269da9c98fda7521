"""Per-stage GPU time of the evaluation-settings NMS (reference test.py:15-16, 91: conf 0.001, iou 0.6, multi-label), both forms side by
side on the same sorted records: the IoU bit mask + one greedy scan per image (yh_nms_mask / yh_nms_reduce) against one scan per
(image, class) in LDS (yh_nms_class_scan).  HIP events around the C-ABI calls; the kept lists of the two forms are compared.
Output of record: profiles/r05_nms_stages.txt."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, '..'), os.path.join(HERE, '..', '..', 'tests')]
import synth  # noqa: E402
from engine import hiplib  # noqa: E402

lib = hiplib.load()
P = hiplib.ptr
for (n, rows, nc, hot, conf) in ((16, 22743, 80, 0.0135, 0.06), (16, 22743, 80, 0.004, 0.06), (64, 22743, 80, 0.0044, 0.3)):
    pred = synth.nms_candidates(n, rows, nc, 5, n_clusters=60, hot=hot).cuda()
    ml = 1 if conf < 0.1 else 0      # cold rows carry objectness <= 0.05: conf 0.06 keeps the hot rows only
    S = hiplib.stream_ptr()
    count = torch.zeros(n, dtype=torch.int32, device='cuda')
    lib.yh_nms_candidates(P(pred), n, rows, nc, conf, ml, None, None, P(count), 0, S)
    mmax = int(count.max())
    cap = 256
    while cap < mmax:
        cap *= 2
    words = (cap + 63) // 64
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    ws = torch.empty(n * cap * 12, dtype=torch.uint8, device='cuda')
    srt2 = torch.empty((n, cap, 8), device='cuda')
    cls82 = torch.empty((n, cap), dtype=torch.uint8, device='cuda')
    for rep in range(3):
        ctl = torch.zeros((11, n), dtype=torch.int32, device='cuda')
        cnt, nk, nk2, state = ctl[0], ctl[1], ctl[2], ctl[3:].view(n, 8)
        state[:, 1:3] = -1
        cand = torch.empty((n, cap, 8), device='cuda')
        srt = torch.empty_like(cand)
        cls8 = torch.empty((n, cap), dtype=torch.uint8, device='cuda')
        keep8 = torch.empty((n, cap), dtype=torch.uint8, device='cuda')
        mask = torch.empty((n, cap, words), dtype=torch.int64, device='cuda')
        keep = torch.empty((n, cap), dtype=torch.int32, device='cuda')
        keep2 = torch.empty((n, cap), dtype=torch.int32, device='cuda')
        res = torch.empty((n, cap, 6), device='cuda')
        ev[0].record(); lib.yh_nms_candidates(P(pred), n, rows, nc, conf, ml, None, P(cand), P(cnt), cap, S)
        ev[1].record(); lib.yh_nms_sort_cls(P(cand), P(cnt), n, cap, cap, P(srt), P(cls8), S)
        ev[2].record(); lib.yh_nms_mask(P(srt), P(cnt), n, cap, cap, 0.6, 0, P(mask), S)
        ev[3].record(); lib.yh_nms_reduce(P(mask), P(cnt), n, cap, cap, P(keep), P(nk), S)
        ev[4].record(); lib.yh_nms_class_scan(P(srt), P(cls8), P(cnt), n, cap, nc, 0.6, P(keep8), P(state), P(keep2), P(nk2), S)
        ev[5].record(); lib.yh_nms_merge(P(srt), P(cnt), P(keep2), P(nk2), n, cap, cap, 0.6, 0, 1, 3000, P(res), S)
        ev[6].record(); lib.yh_nms_sort_tiles(P(cand), P(cnt), n, cap, cap, P(srt2), P(cls82), P(ws), ws.numel(), S)
        ev[7].record(); torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) for i in range(7)]
    h = ctl.cpu()
    same = bool((h[1] == h[2]).all()) and all(torch.equal(keep[i, :h[1, i]], keep2[i, :h[1, i]]) for i in range(n))
    print('n %d candidates max %d (cap %d) conf %g ml %d: candidates %.3f  sort %.3f | bit mask %.3f + scan %.3f | by class %.3f | merge %.3f ms;'
          ' survivors %.0f / image, %d images handed over, kept lists %s' % (n, mmax, cap, conf, ml, *t[:6], h[1].float().mean().item(),
                                                                               int(h[3:].view(n, 8)[:, 5].sum()), 'IDENTICAL' if same else 'DIFFER'))
    mm = [int(v) for v in h[0]]
    same_sort = all(torch.equal(srt[i, :mm[i]], srt2[i, :mm[i]]) and torch.equal(cls8[i, :mm[i]], cls82[i, :mm[i]]) for i in range(n))
    print('   tile sort %.3f ms (sorted records %s the counting sort\'s); per call: bit-mask form %.3f ms, class by class %.3f ms, class by class on '
          'the tile sort %.3f ms' % (t[6], 'EQUAL' if same_sort else 'DIFFER FROM', t[0] + t[1] + t[2] + t[3] + t[5], t[0] + t[1] + t[4] + t[5],
                                      t[0] + t[6] + t[4] + t[5]))
