"""A small maxpool-free YOLOv3-style detector used by the int8 PTQ parity tests (list-of-dicts cfg)."""
import copy

import numpy as np

SIZE = 128


def mini_cfg():
    """Heads sit at true strides 32 and 16, so the reference's fixed stride table (models.py:312) and the
    graph-derived strides of this repo agree for a list cfg."""
    net = {'type': 'net', 'width': SIZE, 'height': SIZE, 'channels': 3}
    conv = lambda f, k, s=1, act='leaky', bn=1: {'type': 'convolutional', 'batch_normalize': bn, 'filters': f, 'size': k,
                                                 'stride': s, 'pad': 1, 'activation': act}
    sc = {'type': 'shortcut', 'from': [-3], 'activation': 'linear'}
    anchors = np.array([[6., 8.], [10., 14.], [16., 12.], [20., 30.], [32., 24.], [40., 44.]])
    yolo = lambda mask: {'type': 'yolo', 'mask': mask, 'anchors': anchors, 'classes': 3, 'num': 6}
    blocks = [net,
              conv(16, 3),                                  # 0            stride 1
              conv(32, 3, 2),                               # 1            2
              conv(16, 1), conv(32, 3), dict(sc),           # 2 3 4
              conv(64, 3, 2),                               # 5            4
              conv(32, 1), conv(64, 3), dict(sc),           # 6 7 8
              conv(64, 3, 2),                               # 9            8
              conv(128, 3, 2),                              # 10           16
              conv(64, 1), conv(128, 3), dict(sc),          # 11 12 13
              conv(128, 3, 2),                              # 14           32
              conv(64, 1), conv(128, 3, 1, 'mish'),         # 15 16
              conv(24, 1, 1, 'linear', 0), yolo([3, 4, 5]),  # 17 18       head at stride 32
              {'type': 'route', 'layers': [-4]},            # 19 -> block 15
              conv(32, 1), {'type': 'upsample', 'stride': 2},   # 20 21
              {'type': 'route', 'layers': [-1, 13]},        # 22           concat with the stride-16 shortcut
              conv(64, 3), conv(24, 1, 1, 'linear', 0), yolo([0, 1, 2])]   # 23 24 25   head at stride 16
    return copy.deepcopy(blocks)
