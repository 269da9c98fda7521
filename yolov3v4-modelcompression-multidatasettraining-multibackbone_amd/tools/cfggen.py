#!/usr/bin/env python3
"""Programmatic writers for the darknet cfgs the BASELINE configs name.

The GPU box has no copy of the reference tree, so the network definitions the benchmark and the
parity tests need are *generated* here from compact architecture descriptions (Darknet-53 stage
table, CSPDarknet-53 stage table, tiny).  ``tests/test_cfggen.py`` checks, in the build container,
that every generated file parses to block lists equal to the reference's own cfg of the same name.

    python tools/cfggen.py            # (re)write cfg/*.cfg next to this package
"""
import os

COCO_ANCHORS_V3 = '10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326'
COCO_ANCHORS_V4 = '12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401'
TINY_ANCHORS = '10,14,  23,27,  37,58,  81,82,  135,169,  344,319'
TINY_HAND_ANCHORS = '9,13,  16,22,  27,38,  28,27,  44,49,  79,83'


class CfgWriter:
    def __init__(self):
        self.lines = []
        self.n = -1  # index of the last emitted layer block ([net] is not counted)

    def block(self, kind, **kv):
        self.lines.append('[%s]' % kind)
        for k, v in kv.items():
            self.lines.append('%s=%s' % (k.rstrip('_'), v))
        self.lines.append('')
        if kind != 'net':
            self.n += 1
        return self.n

    def net(self, **kv):
        self.block('net', **kv)

    def conv(self, filters, size, stride=1, act='leaky', bn=1):
        kv = {}
        if bn:
            kv['batch_normalize'] = 1
        kv.update(filters=filters, size=size, stride=stride, pad=1, activation=act)
        return self.block('convolutional', **kv)

    def depthwise(self, filters, size, stride=1, act='relu6'):
        return self.block('depthwise', batch_normalize=1, filters=filters, size=size, stride=stride, pad=1, activation=act)

    def se(self, filters):
        return self.block('se', filters=filters)

    def shortcut(self, frm=-3):
        return self.block('shortcut', from_=frm, activation='linear')

    def route(self, *layers):
        return self.block('route', layers=','.join(str(l) for l in layers))

    def maxpool(self, size, stride):
        return self.block('maxpool', size=size, stride=stride)

    def upsample(self, stride=2):
        return self.block('upsample', stride=stride)

    def yolo(self, mask, anchors, classes, num, **extra):
        kv = dict(mask=','.join(str(m) for m in mask), anchors=anchors, classes=classes, num=num, jitter='.3',
                  ignore_thresh='.7', truth_thresh=1)
        kv.update(extra)
        return self.block('yolo', **kv)

    def text(self):
        return '\n'.join(self.lines)


def _net_v3(w, size, steps='400000,450000', scales='.1,.1'):
    w.net(batch=16, subdivisions=1, width=size, height=size, channels=3, momentum=0.9, decay=0.0005, angle=0,
          saturation=1.5, exposure=1.5, hue='.1', learning_rate=0.001, burn_in=1000, max_batches=500200,
          policy='steps', steps=steps, scales=scales)


def yolov3(classes=80, size=416, anchors=COCO_ANCHORS_V3):
    """Darknet-53 backbone (stage widths 64..1024, residual counts 1,2,8,8,4) + 3-scale FPN head."""
    w = CfgWriter()
    _net_v3(w, size)
    head_filters = 3 * (classes + 5)
    w.conv(32, 3)
    stage_end = {}
    for width, repeats in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        w.conv(width, 3, stride=2)
        for _ in range(repeats):
            w.conv(width // 2, 1)
            w.conv(width, 3)
            w.shortcut(-3)
        stage_end[width] = w.n
    skip = {512: stage_end[512], 256: stage_end[256]}  # features re-joined after each upsample

    def head(width, mask):
        for _ in range(3):
            w.conv(width, 1)
            w.conv(width * 2, 3)
        w.conv(head_filters, 1, act='linear', bn=0)
        w.yolo(mask, anchors, classes, 9, random=1)

    head(512, (6, 7, 8))
    for width, mask in ((256, (3, 4, 5)), (128, (0, 1, 2))):
        w.route(-4)
        w.conv(width, 1)
        w.upsample(2)
        w.route(-1, skip[width * 2])
        head(width, mask)
    return w.text()


def yolov3_tiny(classes=80, size=416, anchors=TINY_ANCHORS, steps='400000,450000', scales='.1,.1'):
    w = CfgWriter()
    _net_v3(w, size, steps, scales)
    head_filters = 3 * (classes + 5)
    for i, width in enumerate((16, 32, 64, 128, 256, 512)):
        w.conv(width, 3)
        if width == 256:
            feat = w.n
        w.maxpool(2, 2 if width != 512 else 1)
    w.conv(1024, 3)
    w.conv(256, 1)
    w.conv(512, 3)
    w.conv(head_filters, 1, act='linear', bn=0)
    w.yolo((3, 4, 5), anchors, classes, 6, random=1)
    w.route(-4)
    w.conv(128, 1)
    w.upsample(2)
    w.route(-1, feat)
    w.conv(256, 3)
    w.conv(head_filters, 1, act='linear', bn=0)
    w.yolo((0, 1, 2), anchors, classes, 6, random=1)
    return w.text()


def yolov4_tiny(classes=80, size=416, anchors=TINY_ANCHORS):
    """CSPDarknet-tiny: three CSP stages whose inner branch is the second channel half of the stage input
    (``route groups=2 group_id=1``), 2/2 maxpools, two heads (stride 32, 16)."""
    w = CfgWriter()
    w.net(batch=64, subdivisions=1, width=size, height=size, channels=3, momentum=0.9, decay=0.0005, angle=0, saturation=1.5,
          exposure=1.5, hue='.1', learning_rate=0.00261, burn_in=1000, max_batches=500200, policy='steps',
          steps='400000,450000', scales='.1,.1')
    head_filters = 3 * (classes + 5)
    yolo_extra = dict(scale_x_y='1.05', cls_normalizer='1.0', iou_normalizer='0.07', iou_loss='ciou', ignore_thresh='.7',
                      truth_thresh=1, random=0, resize='1.5', nms_kind='greedynms', beta_nms='0.6')
    w.conv(32, 3, stride=2)
    w.conv(64, 3, stride=2)
    feat = None
    for width in (64, 128, 256):
        w.conv(width, 3)
        w.block('route', layers=-1, groups=2, group_id=1)
        w.conv(width // 2, 3)
        w.conv(width // 2, 3)
        w.route(-1, -2)
        w.conv(width, 1)
        if width == 256:
            feat = w.n
        w.route(-6, -1)
        w.maxpool(2, 2)
    w.conv(512, 3)
    w.conv(256, 1)
    w.conv(512, 3)
    w.conv(head_filters, 1, act='linear', bn=0)
    w.yolo((3, 4, 5), anchors, classes, 6, **yolo_extra)
    w.route(-4)
    w.conv(128, 1)
    w.upsample(2)
    w.route(-1, feat)
    w.conv(256, 3)
    w.conv(head_filters, 1, act='linear', bn=0)
    w.yolo((0, 1, 2), anchors, classes, 6, **yolo_extra)
    return w.text()


def yolov4(classes=80, size=608, anchors=COCO_ANCHORS_V4):
    """CSPDarknet-53 (Mish) + SPP + PANet neck (LeakyReLU) + 3 heads ordered stride 8, 16, 32."""
    w = CfgWriter()
    w.net(batch=64, subdivisions=8, width=size, height=size, channels=3, momentum=0.949, decay=0.0005, angle=0,
          saturation=1.5, exposure=1.5, hue='.1', learning_rate=0.00261, burn_in=1000, max_batches=500500,
          policy='steps', steps='400000,450000', scales='.1,.1', mosaic=1)
    head_filters = 3 * (classes + 5)
    m = 'mish'
    w.conv(32, 3, act=m)
    tap = {}
    # stage 1 is the odd one: the split keeps full width and the residual bottleneck halves it
    w.conv(64, 3, 2, act=m)
    w.conv(64, 1, act=m)
    w.route(-2)
    w.conv(64, 1, act=m)
    w.conv(32, 1, act=m)
    w.conv(64, 3, act=m)
    w.shortcut(-3)
    w.conv(64, 1, act=m)
    w.route(-1, -7)
    w.conv(64, 1, act=m)
    for width, repeats in ((128, 2), (256, 8), (512, 8), (1024, 4)):
        half = width // 2
        w.conv(width, 3, 2, act=m)
        w.conv(half, 1, act=m)
        w.route(-2)
        w.conv(half, 1, act=m)
        for _ in range(repeats):
            w.conv(half, 1, act=m)
            w.conv(half, 3, act=m)
            w.shortcut(-3)
        w.conv(half, 1, act=m)
        w.route(-1, -(4 + 3 * repeats))
        tap[width] = w.conv(width, 1, act=m)
    # SPP
    w.conv(512, 1)
    w.conv(1024, 3)
    w.conv(512, 1)
    w.maxpool(5, 1)
    w.route(-2)
    w.maxpool(9, 1)
    w.route(-4)
    w.maxpool(13, 1)
    w.route(-1, -3, -5, -6)
    w.conv(512, 1)
    w.conv(1024, 3)
    w.conv(512, 1)
    # top-down path
    for width, lateral in ((256, tap[512]), (128, tap[256])):
        w.conv(width, 1)
        w.upsample(2)
        w.route(lateral)
        w.conv(width, 1)
        w.route(-1, -3)
        for _ in range(2):
            w.conv(width, 1)
            w.conv(width * 2, 3)
        w.conv(width, 1)
    v4 = dict(iou_thresh=0.213, cls_normalizer=1.0, iou_normalizer=0.07, iou_loss='ciou', beta_nms=0.6)
    # heads + bottom-up path
    w.conv(256, 3)
    w.conv(head_filters, 1, act='linear', bn=0)
    w.yolo((0, 1, 2), anchors, classes, 9, scale_x_y=1.2, nms_kind='grpeedynms', **v4)
    for width, mask, back, sxy, extra in ((256, (3, 4, 5), -16, 1.1, {}), (512, (6, 7, 8), -37, 1.05, {'random': 1})):
        w.route(-4)
        w.conv(width, 3, 2)
        w.route(-1, back)
        for _ in range(2):
            w.conv(width, 1)
            w.conv(width * 2, 3)
        w.conv(width, 1)
        w.conv(width * 2, 3)
        w.conv(head_filters, 1, act='linear', bn=0)
        w.yolo(mask, anchors, classes, 9, scale_x_y=sxy, nms_kind='greedynms', **dict(extra, **v4))
    return w.text()


# MobileNetV3-large bottlenecks: kernel, expansion, out, squeeze-excite, nonlinearity, stride
MBV3_LARGE = [(3, 16, 16, 0, 'relu6', 1), (3, 64, 24, 0, 'relu6', 2), (3, 72, 24, 0, 'relu6', 1), (5, 72, 40, 1, 'relu6', 2),
              (5, 120, 40, 1, 'relu6', 1), (5, 120, 40, 1, 'relu6', 1), (3, 240, 80, 0, 'h_swish', 2),
              (3, 200, 80, 0, 'h_swish', 1), (3, 184, 80, 0, 'h_swish', 1), (3, 184, 80, 0, 'h_swish', 1),
              (3, 480, 112, 1, 'h_swish', 1), (3, 672, 112, 1, 'h_swish', 1), (5, 672, 160, 1, 'h_swish', 2),
              (5, 960, 160, 1, 'h_swish', 1), (5, 960, 160, 1, 'h_swish', 1)]


def yolov3_mobilenet(classes=80, size=416, anchors=COCO_ANCHORS_V3):
    """MobileNetV3-large backbone (depthwise + squeeze-excite bottlenecks) under the YOLOv3 neck and heads."""
    w = CfgWriter()
    _net_v3(w, size)
    head_filters = 3 * (classes + 5)
    w.conv(16, 3, stride=2, act='h_swish')
    cin, taps = 16, {}
    for k, exp, out, se, nl, stride in MBV3_LARGE:
        w.conv(exp, 1, act=nl)
        w.depthwise(exp, k, stride, act=nl)
        if se:
            w.se(exp)
        w.conv(out, 1, act='linear')
        if stride == 1 and cin == out:
            w.shortcut(-5 if se else -4)
        taps[out] = w.n
        cin = out
    w.conv(1024, 1, act='h_swish')

    def head(width, mask):
        for _ in range(3):
            w.conv(width, 1)
            w.conv(width * 2, 3)
        w.conv(head_filters, 1, act='linear', bn=0)
        w.yolo(mask, anchors, classes, 9, random=1)

    head(512, (6, 7, 8))
    for width, mask, skip in ((256, (3, 4, 5), taps[112]), (128, (0, 1, 2), taps[40])):
        w.route(-4)
        w.conv(width, 1)
        w.upsample(2)
        w.route(-1, skip)
        head(width, mask)
    return w.text()


GENERATED = {
    'yolov3-mobilenet/yolov3-mobilenet-coco.cfg': lambda: yolov3_mobilenet(80, 416),
    'yolov3/yolov3.cfg': lambda: yolov3(80, 416),
    'yolov3tiny/yolov3-tiny.cfg': lambda: yolov3_tiny(80, 416),
    'yolov3tiny/yolov3-tiny-hand.cfg': lambda: yolov3_tiny(1, 416, TINY_HAND_ANCHORS, '15,25,60,99,150,160,180',
                                                           '0.5,0.5,0.1,0.5,0.5,0.1,0.1'),
    'yolov4/yolov4.cfg': lambda: yolov4(80, 608),
    'yolov4tiny/yolov4-tiny.cfg': lambda: yolov4_tiny(80, 416),
}


def write_all(root=None):
    root = root or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cfg')
    for rel, make in GENERATED.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as fh:
            fh.write('# generated by tools/cfggen.py -- do not edit\n' + make() + '\n')
    return root


if __name__ == '__main__':
    print('wrote cfgs under', write_all())
