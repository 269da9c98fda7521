#!/usr/bin/env python3
"""A/B of the 3x3 halo ping-pong kernel (tile 43, csrc/conv_halo_pp.hip) with four barriers per K step (YH_HPP_BARRIERS=4) against one
(YH_HPP_BARRIERS=1) on the 3x3 / s1 layer shapes of YOLOv3-608 batch 64 and YOLOv4-640 batch 32: bit-equality of the outputs and
interleaved timing rounds (the clock state drifts over a process by more than the differences looked for)."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
sys.path[:0] = [PKG, os.path.join(os.path.dirname(PKG), 'tests')]
import torch  # noqa: E402
from engine import hiplib  # noqa: E402
import ops_harness as oh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--variants', nargs='*', default=['YH_HPP_BARRIERS=4', 'YH_HPP_BARRIERS=1'])
    args = ap.parse_args()
    lib = hiplib.load()
    torch.manual_seed(0)
    shapes = [('fwd 152 64->128 stats', 64, 152, 64, 128, False, True, 0), ('fwd 76 128->256 stats', 64, 76, 128, 256, False, True, 0),
              ('fwd 38 256->512 stats', 64, 38, 256, 512, False, True, 0), ('fwd 19 512->1024 stats', 64, 19, 512, 1024, False, True, 0),
              ('fwd 76 128->256 leaky', 64, 76, 128, 256, False, False, 1), ('fwd 38 256->512 leaky', 64, 38, 256, 512, False, False, 1),
              ('fwd 19 512->1024 leaky', 64, 19, 512, 1024, False, False, 1),
              ('dgrad 76 256->128 res', 64, 76, 256, 128, True, False, 0), ('dgrad 38 512->256 res', 64, 38, 512, 256, True, False, 0),
              ('dgrad 19 1024->512 res', 64, 19, 1024, 512, True, False, 0), ('dgrad 152 128->64 res', 64, 152, 128, 64, True, False, 0),
              ('v4 b32 fwd 80 128->256 mish', 32, 80, 128, 256, False, False, 5), ('v4 b32 fwd 40 256->512 mish', 32, 40, 256, 512, False, False, 5)]
    print('%-30s %s   bit-equal' % ('layer', '   '.join('%-22s' % v for v in args.variants)))
    tot = [0.0] * len(args.variants)
    for tag, N, HW, cin, cout, res, stats, act in shapes:
        x = (torch.randn(N, HW, HW, cin, device='cuda') * 0.5).half()
        w = torch.randn(cout, cin, 3, 3, device='cuda') * (1.0 / (cin * 9) ** 0.5)
        packed, bias, cin_k, m_pad = oh.pack_conv(lib, 0, w, None, None, cin_phys=cin)
        r = (torch.randn(N, HW, HW, cout, device='cuda') * 0.5).half() if res else None
        flops = 2.0 * N * HW * HW * cout * cin * 9
        outs, times = [], [[] for _ in args.variants]

        def go(y=None):
            sd = {} if stats else None
            return oh.conv(lib, 0, x, packed, bias, cin_k, m_pad, cout, 3, 1, 1, act=act, tile=43, res=r, stats=sd, y=y)
        for v in args.variants:
            k, val = v.split('=')
            os.environ[k] = val
            outs.append(go().clone())
        torch.cuda.synchronize()
        y = torch.empty_like(outs[0])
        for _ in range(args.rounds):
            for i, v in enumerate(args.variants):
                k, val = v.split('=')
                os.environ[k] = val
                go(y)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    go(y)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / args.reps)
        cells = []
        for i in range(len(args.variants)):
            ms = sorted(times[i])[len(times[i]) // 2]
            tot[i] += ms
            cells.append('%.4f ms %6.0f TF/s' % (ms, flops / ms / 1e9))
        print('%-30s %s   %s' % (tag, '   '.join('%-22s' % c for c in cells), all(torch.equal(o, outs[0]) for o in outs)), flush=True)
        del x, w, r, outs, y
        torch.cuda.empty_cache()
    print('%-30s %s' % ('sum', '   '.join('%-22s' % ('%.4f ms' % t) for t in tot)))


if __name__ == '__main__':
    main()
