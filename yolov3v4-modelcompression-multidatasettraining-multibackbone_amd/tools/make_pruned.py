"""BASELINE.json configs[4], step 1 (build container only - needs a reference checkout): run the reference's UNMODIFIED
`slim_prune.py --percent 0.5` (reference slim_prune.py:97-207, utils/prune_utils.py:212-258) on this package's
yolov3-mobilenet-coco.cfg and keep what it writes.

    python tools/make_pruned.py [--out DIR] [--reference /root/reference]

The script is executed with `runpy` on top of THIS package (`from models import *`, `from test import test` ... resolve here;
`utils.prune_utils` resolves to the reference's own file), on seeded weights whose BatchNorm gammas are spread out so that the
global threshold prunes every layer differently, and on a four-image synthetic validation set (slim_prune evaluates the model
before and after).  Output: the compact cfg - 70 conv blocks, 37 of them with a width that is not a multiple of 8 (13, 15, 33,
75, 85, 126, 534 ...) - and its darknet .weights.  The cfg TEXT is committed as tests/golden/slim_prune_0.5_yolov3-mobilenet-coco.cfg
(the 33 MB weights file is not: the GPU test trains the compact graph from seeded weights, tools/pruned_finetune.py takes either).
"""
import argparse
import glob
import os
import shutil
import subprocess
import sys
import tempfile

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, PKG)

RUNNER = '''
import runpy, sys, torch
sys.path.insert(0, %r)
torch.Tensor.cuda = lambda self, *a, **k: self          # slim_prune.py:55,113 call .cuda() on masks unconditionally
torch.manual_seed(0)
sys.argv = %r
runpy.run_path(%r, run_name='__main__')
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default=os.environ.get('YOLO_REFERENCE_ROOT', '/root/reference'))
    ap.add_argument('--out', default=os.path.join(os.path.dirname(PKG), 'gpurun_out', 'pruned'))
    ap.add_argument('--percent', default='0.5')
    args = ap.parse_args()
    import numpy as np
    import torch
    from PIL import Image
    import models
    work = tempfile.mkdtemp()
    for d in ('cfg/yolov3-mobilenet', 'weights', 'data/images', 'data/labels'):
        os.makedirs(os.path.join(work, d))
    rel = 'cfg/yolov3-mobilenet/yolov3-mobilenet-coco.cfg'
    shutil.copy(os.path.join(PKG, rel), os.path.join(work, rel))
    rng = np.random.RandomState(0)
    files = []
    for i in range(4):
        p = '%s/data/images/im_%d.png' % (work, i)
        Image.fromarray((rng.rand(320, 416, 3) * 255).astype(np.uint8)).save(p)
        open('%s/data/labels/im_%d.txt' % (work, i), 'w').write('0 0.5 0.5 0.3 0.3\n17 0.3 0.6 0.2 0.4\n')
        files.append(p)
    open(work + '/data/valid.txt', 'w').write('\n'.join(files) + '\n')
    open(work + '/data/coco.names', 'w').write('\n'.join('c%d' % i for i in range(80)) + '\n')
    open(work + '/data/synth.data', 'w').write('classes=80\ntrain=%s/data/valid.txt\nvalid=%s/data/valid.txt\nnames=%s/data/coco.names\n'
                                               % (work, work, work))
    torch.manual_seed(0)
    model = models.Darknet(os.path.join(work, rel), (416, 416))
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith('BatchNorm2d.weight'):
                v.copy_(torch.rand(v.shape, generator=g) * 1.5 + 0.01)
            elif k.endswith('running_var'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith('BatchNorm2d.bias') or k.endswith('running_mean'):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    models.save_weights(model, work + '/weights/m.weights')
    argv = ['slim_prune.py', '--cfg', rel, '--data', work + '/data/synth.data', '--weights', 'weights/m.weights',
            '--percent', args.percent, '--img-size', '416', '--batch-size', '2']
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, '-c', RUNNER % (PKG, argv, os.path.join(args.reference, 'slim_prune.py'))], cwd=work,
                       capture_output=True, text=True, env=env)
    print(r.stdout[-1500:])
    if r.returncode:
        print(r.stderr[-3000:])
        raise SystemExit(r.returncode)
    os.makedirs(args.out, exist_ok=True)
    for f in glob.glob(work + '/cfg/**/*slim_prune*.cfg', recursive=True) + glob.glob(work + '/weights/**/*slim_prune*.weights', recursive=True):
        shutil.copy(f, args.out)
        print('kept', os.path.join(args.out, os.path.basename(f)))


if __name__ == '__main__':
    main()
