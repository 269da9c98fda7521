"""Darknet ``.cfg`` / ``.data`` readers.

Mirrors the reference's ``utils/parse_config.py`` interface (``parse_model_cfg`` :6-51,
``parse_data_cfg`` :54-70) so that every caller (``models.Darknet``, the prune scripts, ``PTQ.py``)
keeps working.  Written from the observed behaviour, not from the reference text:

* a block header ``[type]`` opens a dict ``{'type': type}``; ``[convolutional]`` blocks are
  pre-seeded with ``batch_normalize = 0``;
* ``anchors`` becomes an ``(n, 2)`` float ndarray; ``from`` / ``layers`` / ``mask`` (and a
  comma-carrying ``size``) become ``list[int]``;
* any other value that is a pure digit string becomes ``int``; everything else (``.7``, ``-1``,
  ``leaky`` ...) stays ``str``;
* keys outside the allow-list abort with an ``AssertionError``.
"""
import os

import numpy as np

# keys the graph builder knows about (same allow-list as the reference, parse_config.py:40-43)
_KNOWN_KEYS = frozenset((
    'type', 'batch_normalize', 'filters', 'size', 'stride', 'pad', 'activation', 'layers', 'groups',
    'reduction', 'from', 'mask', 'anchors', 'classes', 'num', 'jitter', 'ignore_thresh',
    'truth_thresh', 'random', 'stride_x', 'stride_y', 'weights_type', 'weights_normalization',
    'scale_x_y', 'beta_nms', 'nms_kind', 'iou_loss', 'iou_normalizer', 'cls_normalizer',
    'iou_thresh', 'group_id', 'resize'))

_INT_LIST_KEYS = ('from', 'layers', 'mask')


def _resolve(path, suffix, folder):
    if suffix and not path.endswith(suffix):
        path += suffix
    if not os.path.exists(path):
        alt = os.path.join(folder, path)
        if os.path.exists(alt):
            path = alt
    return path


def _convert(key, raw):
    if key == 'anchors':
        return np.array([float(v) for v in raw.split(',')]).reshape((-1, 2))
    if key in _INT_LIST_KEYS or (key == 'size' and ',' in raw):
        return [int(v) for v in raw.split(',')]
    raw = raw.strip()
    if raw.isnumeric():
        return int(raw)
    return raw


def parse_model_cfg(path):
    """Return the list of block dicts of a darknet cfg; element 0 is the ``[net]`` block."""
    path = _resolve(path, '.cfg', 'cfg')
    with open(path, 'r') as fh:
        text = fh.read()

    blocks = []
    for line in text.split('\n'):
        line = line.strip()
        if not line or line.startswith('#'):
            continue
        if line.startswith('['):
            kind = line[1:-1].rstrip()
            block = {'type': kind}
            if kind == 'convolutional':
                block['batch_normalize'] = 0
            blocks.append(block)
            continue
        key, raw = line.split('=')
        key = key.rstrip()
        blocks[-1][key] = _convert(key, raw)

    unknown = []
    for block in blocks[1:]:
        for key in block:
            if key not in _KNOWN_KEYS and key not in unknown:
                unknown.append(key)
    assert not unknown, "Unsupported fields %s in %s. See https://github.com/ultralytics/yolov3/issues/631" % (
        unknown, path)
    return blocks


def parse_data_cfg(path):
    """Return the ``key = value`` pairs of a ``.data`` file as a dict of strings."""
    path = _resolve(path, '', 'data')
    options = {}
    with open(path, 'r') as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith('#'):
                continue
            key, val = line.split('=')
            options[key.strip()] = val.strip()
    return options
