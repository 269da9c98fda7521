"""Pins the darknet .weights byte format to the REFERENCE's writer.

Run in the build container (needs /root/reference):  python tests/golden/make_golden_weights.py
For each net: the reference's `Darknet(cfg)` gets a state derived from a fixed seed per state_dict key (independent of
module construction order and RNG consumption), `ref.models.save_weights` writes the file (models.py:738-782), and the
SHA-256 + size go to weights_sha.json.  tests/test_weights_format.py rebuilds the same state in THIS package's modules,
writes with this package's `save_weights` and compares the digest (runs anywhere, also on the GPU box); in the container
it additionally cross-loads both files with both loaders.
"""
import hashlib
import json
import os
import sys
import tempfile
import zlib

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: E402,F401
import refharness  # noqa: E402

NETS = {  # name -> cfg path relative to the cfg root (same relative layout in the reference and in this package)
    'yolov3-tiny': 'yolov3tiny/yolov3-tiny.cfg',
    'yolov3-mobilenet-coco': 'yolov3-mobilenet/yolov3-mobilenet-coco.cfg',
    'yolov4': 'yolov4/yolov4.cfg',
}


def keyed_state_(model):
    """Fill every floating tensor of the state_dict from a generator seeded by the key's CRC32 (positive variances)."""
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
            if k.endswith('running_var'):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            else:
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    return model


def digest(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        h.update(f.read())
    return h.hexdigest(), os.path.getsize(path)


def main():
    ref = refharness.load()
    out = {}
    for name, rel in NETS.items():
        torch.manual_seed(0)
        model = keyed_state_(ref.models.Darknet(os.path.join(ref.root, 'cfg', rel), (416, 416)))
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, name + '.weights')
            ref.models.save_weights(model, path)
            sha, size = digest(path)
        out[name] = {'cfg': rel, 'sha256': sha, 'bytes': size, 'writer': 'reference models.save_weights (models.py:738-782)'}
        print(name, size, sha)
    with open(os.path.join(HERE, 'weights_sha.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
