#!/usr/bin/env python3
"""Frame-at-a-time latency of the detect path (detect.py's loop: one letterboxed frame per forward + NMS).

    python tools/latency.py [--size 608] [--cfg ...] [--precision fp16]

Reports milliseconds per forward and per forward+NMS for batches 1..8 with the hipGraph replay on and off."""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.dirname(PKG))

import torch  # noqa: E402


def main():
    from bench import build_model, build_qmodel_synthetic
    from utils.utils import non_max_suppression
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default=os.path.join(PKG, 'cfg', 'yolov3', 'yolov3.cfg'))
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--precision', default='fp16')
    ap.add_argument('--iters', type=int, default=200)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    model = build_qmodel_synthetic(args.cfg, args.size, dev) if args.precision == 'int8' else \
        build_model(args.cfg, args.size, args.precision, dev)
    print('%-6s %-6s %12s %12s %10s' % ('batch', 'graph', 'fwd ms', 'fwd+nms ms', 'frames/s'))
    for batch in (1, 2, 4, 8):
        x = torch.rand(batch, 3, args.size, args.size, device=dev)
        for graph in (0, 8):
            model.hip_refresh()
            os.environ['YOLO_HIP_GRAPH_BATCH'] = str(graph)
            res = []
            for with_nms in (False, True):
                def step():
                    with torch.no_grad():
                        inf = model(x)[0]
                    if with_nms:
                        non_max_suppression(inf, 0.3, 0.6, multi_label=False)
                for _ in range(20):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.iters):
                    step()
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / args.iters * 1e3)
            print('%-6d %-6s %12.4f %12.4f %10.1f' % (batch, 'on' if graph else 'off', res[0], res[1], batch / res[1] * 1e3))


if __name__ == '__main__':
    main()
