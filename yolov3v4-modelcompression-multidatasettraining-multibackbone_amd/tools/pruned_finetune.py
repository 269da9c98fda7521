"""BASELINE.json configs[4], step 2 (GPU box): the compact graph the reference's slim_prune.py produced (tools/make_pruned.py;
cfg text committed as tests/golden/slim_prune_0.5_yolov3-mobilenet-coco.cfg) fine-tunes on the HIP path.

    python tools/pruned_finetune.py [--cfg CFG] [--weights FILE.weights] [--size 416] [--bench]

37 of its 70 conv widths are not multiples of 8, so the step runs through the channel-padded twin (engine/padded.py).  Printed per
sample: the total-gradient error of eager fp32 autograd and of the HIP step against eager fp64 autograd, next to the sample's own
KINK SENSITIVITY - the change of the fp64 gradient when the frames move by one fp32 ulp.  relu6 / h-swish / leaky have kinks; a
pre-activation within rounding distance of one lands on either side depending on the summation order, and on random weights one
flipped unit moves every upstream gradient by ~1e-3 .. 1e-2.  An fp32 implementation can only be expected to match fp64 down to
that sensitivity; with Mish in place of the kinked activations (same graph, same weights) it has to match to round-off.
"""
import argparse
import json
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPO = os.path.dirname(PKG)
for p in (PKG, os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)

import torch  # noqa: E402

GOLD_CFG = os.path.join(REPO, 'tests', 'golden', 'slim_prune_0.5_yolov3-mobilenet-coco.cfg')


def total_error(g, g64):
    total = sum(v.norm().item() ** 2 for v in g64.values()) ** 0.5
    return sum((g[k] - g64[k].float()).norm().item() ** 2 for k in g64) ** 0.5 / total


def ulp_perturbed(x, seed):
    """Frames moved by at most one fp32 ulp (relative 2^-23), seeded."""
    g = torch.Generator().manual_seed(seed)
    return x * (1 + (torch.rand(x.shape, generator=g) * 2 - 1) * 2.0 ** -23)


def sample_report(model, x, device):
    """(eager fp32 error, HIP error, kink sensitivity) of one batch, all against eager fp64 autograd."""
    import train_harness as th
    _, g64, _, ws = th.eager_step(model, x, dtype=torch.float64)
    _, g32, _, _ = th.eager_step(model, x, ws=ws)
    _, ghip, m = th.engine_step(model, x, ws, 'fp32', device=device)
    sens = 0.0
    for seed in (11, 12, 13):
        _, gp, _, _ = th.eager_step(model, ulp_perturbed(x, seed), ws=ws, dtype=torch.float64)
        sens = max(sens, total_error(gp, g64))
    return total_error(g32, g64), total_error(ghip, g64), sens, m


def smooth_twin(model, cfg, size):
    """The same graph and weights with every kinked activation replaced by Mish."""
    import models
    from utils.parse_config import parse_model_cfg
    defs = parse_model_cfg(cfg)
    for d in defs[1:]:
        if d['type'] in ('convolutional', 'depthwise') and d.get('activation') in ('leaky', 'relu', 'relu6', 'h_swish'):
            d['activation'] = 'mish'
    torch.manual_seed(0)
    twin = models.Darknet(defs, (size, size), verbose=False)
    twin.load_state_dict(model.state_dict())
    return twin.train()


def build(cfg, weights, size):
    import models
    import synth
    torch.manual_seed(0)
    m = models.Darknet(cfg, (size, size), verbose=False)
    if weights:
        models.load_darknet_weights(m, weights)
    else:
        state = m.state_dict()
        synth.randomize_bn_(state, seed=1)
        m.load_state_dict(state)
    return m.train()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default=GOLD_CFG)
    ap.add_argument('--weights', default='')
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--bench', action='store_true', help='also time bench.py --mode train / detect at 416, batch 64, beside the unpruned model')
    args = ap.parse_args()
    from engine.padded import PaddedTrainEngine
    m = build(args.cfg, args.weights, args.size)
    widths = [b[0].out_channels for b in m.module_list if isinstance(b, torch.nn.Sequential) and hasattr(b[0], 'out_channels')]
    print('compact model: %d conv blocks, %d with a width that is not a multiple of 8, %.1f M parameters, %d px'
          % (len(widths), sum(1 for c in widths if c % 8), sum(p.numel() for p in m.parameters()) / 1e6, args.size))
    for name, net in (('as pruned (relu6 / h-swish / leaky)', m), ('Mish in place of the kinked activations', smooth_twin(m, args.cfg, args.size))):
        for seed in (4, 5, 6):
            x = torch.rand(args.batch, 3, args.size, args.size, generator=torch.Generator().manual_seed(seed))
            e32, ehip, sens, mm = sample_report(net, x, 'cuda')
            assert isinstance(mm.__dict__['_hip_train_engine'], PaddedTrainEngine)
            print('%-40s sample %d: eager fp32 %.2e, HIP padded twin %.2e, fp64 kink sensitivity (frames + 1 ulp) %.2e' % (name, seed, e32, ehip, sens))
    if args.bench:
        full = os.path.join(PKG, 'cfg', 'yolov3-mobilenet', 'yolov3-mobilenet-coco.cfg')
        for name, c in (('unpruned yolov3-mobilenet-coco', full), ('slim_prune 0.5 (reference script)', args.cfg)):
            for mode in ('train', 'detect'):
                out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--mode', mode, '--cfg', c, '--size', '416', '--batch', '64',
                                      '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-v4'], capture_output=True, text=True).stdout
                line = [l for l in out.splitlines() if l.startswith('{')][-1]
                d = json.loads(line)
                print('%-36s %-6s 416 b64: %8.1f images/s  %.2f ms/step' % (name, mode, d['value'], d['ms_per_step']))


if __name__ == '__main__':
    main()
