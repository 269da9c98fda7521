#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE implementation on CPU.

Run in the build container only (needs /root/reference; see tests/refharness.py for the two stubbed
I/O imports):

    python tests/golden/make_golden.py

Writes ``tests/golden/*.npz``.  Each network fixture stores the recipe (cfg, size, seeds), a SHA-256
of the seeded weights, a row subset of the reference's ``inf_out`` plus whole-tensor checksums, and
per-head raw-output checksums; NMS fixtures store the full inputs' recipe and the reference outputs.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401  (puts the package on sys.path)
import refharness
import synth

REFCFG = os.path.join(refharness.REF, 'cfg')

NET_CASES = [
    # name, cfg (relative to cfg/), size, batch, row stride of the stored subset
    ('tiny_hand_416', 'yolov3tiny/yolov3-tiny-hand.cfg', 416, 2, 1),
    ('yolov3_320', 'yolov3/yolov3.cfg', 320, 1, 16),
    ('yolov4_320', 'yolov4/yolov4.cfg', 320, 1, 16),
    ('yolov3_608', 'yolov3/yolov3.cfg', 608, 1, 64),
    ('mobilenet_224', 'yolov3-mobilenet/yolov3-mobilenet-coco.cfg', 224, 2, 4),
    ('yolov4tiny_416', 'yolov4tiny/yolov4-tiny.cfg', 416, 2, 2),
    # the remaining BASELINE.json shapes (configs 4 and 5): YOLOv4 at 640, YOLOv3-Mobilenetv3 at 416
    ('yolov4_640', 'yolov4/yolov4.cfg', 640, 1, 64),
    ('mobilenet_416', 'yolov3-mobilenet/yolov3-mobilenet-coco.cfg', 416, 1, 16),
]


def state_digest(state):
    h = hashlib.sha256()
    for k, v in state.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def checks(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), t.abs().max().item()], dtype=np.float64)


def build_reference(ref, rel, size):
    torch.manual_seed(0)
    model = ref.models.Darknet(os.path.join(REFCFG, rel), (size, size))
    state = model.state_dict()
    synth.randomize_bn_(state, seed=1)
    model.load_state_dict(state)
    return model.eval()


def net_fixture(ref, name, rel, size, batch, row_stride):
    model = build_reference(ref, rel, size)
    x = synth.image_batch(batch, size, seed=0)
    with torch.no_grad():
        inf, raws, _ = model(x)
        model.fuse()
        inf_fused = model(x)[0]
    out = dict(cfg=rel, size=size, batch=batch, row_stride=row_stride, weights_sha256=state_digest(model_state(ref, rel, size)),
               inf_rows=inf[:, ::row_stride].numpy().astype(np.float32), inf_checks=checks(inf),
               inf_shape=np.array(inf.shape), fused_vs_unfused_maxabs=(inf - inf_fused).abs().max().item())
    for i, r in enumerate(raws):
        out['raw%d_checks' % i] = checks(r)
        out['raw%d_shape' % i] = np.array(r.shape)
    np.savez_compressed(os.path.join(HERE, 'net_%s.npz' % name), **out)
    print(name, tuple(inf.shape), 'fused-vs-unfused drift %.3g' % out['fused_vs_unfused_maxabs'])


def model_state(ref, rel, size):
    torch.manual_seed(0)
    m = ref.models.Darknet(os.path.join(REFCFG, rel), (size, size))
    return synth.randomize_bn_(m.state_dict(), seed=1)


def nms_fixtures(ref):
    cases = [
        # name, n_img, rows, nc, seed, conf, iou, multi_label, agnostic
        ('detect_80', 2, 600, 80, 11, 0.3, 0.6, False, False),
        ('test_80', 2, 300, 80, 12, 0.001, 0.6, True, False),
        ('single_class', 1, 500, 1, 13, 0.3, 0.6, True, False),
        ('agnostic', 1, 400, 20, 14, 0.25, 0.45, False, True),
        ('empty', 2, 64, 80, 15, 0.999, 0.6, False, False),
        ('one_box', 1, 8, 3, 16, 0.9, 0.6, False, False),
    ]
    for name, n_img, rows, nc, seed, conf, iou, ml, agn in cases:
        pred = synth.nms_candidates(n_img, rows, nc, seed)
        res = ref.utils.non_max_suppression(pred.clone(), conf, iou, multi_label=ml, agnostic=agn)
        out = dict(n_img=n_img, rows=rows, nc=nc, seed=seed, conf=conf, iou=iou, multi_label=ml, agnostic=agn,
                   counts=np.array([0 if r is None else len(r) for r in res]))
        for i, r in enumerate(res):
            out['det%d' % i] = np.zeros((0, 6), np.float32) if r is None else r.numpy().astype(np.float32)
        np.savez_compressed(os.path.join(HERE, 'nms_%s.npz' % name), **out)
        print('nms', name, out['counts'])


def fuse_fixture(ref):
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(6, 10, 3, stride=2, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(10)
    g = torch.Generator().manual_seed(4)
    bn.weight.data = torch.rand(10, generator=g) + 0.5
    bn.bias.data = torch.randn(10, generator=g)
    bn.running_mean = torch.randn(10, generator=g)
    bn.running_var = torch.rand(10, generator=g) + 0.5
    fused = ref.torch_utils.fuse_conv_and_bn(conv, bn.eval())
    np.savez_compressed(os.path.join(HERE, 'fuse_conv_bn.npz'),
                        w=conv.weight.detach().numpy(), gamma=bn.weight.detach().numpy(), beta=bn.bias.detach().numpy(),
                        mean=bn.running_mean.numpy(), var=bn.running_var.numpy(), eps=bn.eps,
                        fused_w=fused.weight.detach().numpy(), fused_b=fused.bias.detach().numpy())
    print('fuse fixture ok')


def decode_fixture(ref):
    """YOLOLayer eval decode on a small seeded head map, both stride orders."""
    g = torch.Generator().manual_seed(5)
    anchors = np.array([[10, 13], [16, 30], [33, 23]], dtype=np.float64)
    p = torch.randn(2, 3 * 9, 5, 7, generator=g)
    layer = ref.models.YOLOLayer(anchors=anchors, nc=4, img_size=(224, 160), yolo_index=0, layers=[], stride=32,
                                 quantizer_output=False).eval()
    io, raw = layer(p.clone(), None)
    np.savez_compressed(os.path.join(HERE, 'yolo_decode.npz'), p=p.numpy(), anchors=anchors, stride=32, nc=4,
                        io=io.numpy(), raw=raw.numpy())
    print('decode fixture ok', tuple(io.shape))


HYP = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'fl_gamma': 0.0}  # train.py:25-35


def loss_fixture(ref):
    """compute_loss / build_targets of the reference on seeded raw heads and labels (tiny-hand cfg heads, nc = 1, and
    an 80-class yolov3 head set), incl. the gradient w.r.t. every raw head and the focal-loss variant."""
    out = {}
    for tag, rel, size, nc, gamma in (('hand', 'yolov3tiny/yolov3-tiny-hand.cfg', 416, 1, 0.0),
                                      ('coco', 'yolov3/yolov3.cfg', 320, 80, 0.0), ('focal', 'yolov3/yolov3.cfg', 320, 80, 1.5)):
        torch.manual_seed(0)
        model = ref.models.Darknet(os.path.join(REFCFG, rel), (size, size))
        model.nc, model.hyp, model.gr = nc, dict(HYP, fl_gamma=gamma), 0.7
        raws, targets = synth.loss_inputs(model, size, batch=3, seed=21)
        for r in raws:
            r.requires_grad_()
        loss, items = ref.utils.compute_loss(raws, targets, model)
        loss.backward()
        tcls, tbox, indices, av = ref.utils.build_targets(raws, targets, model)
        out[tag + '_items'] = items.numpy()
        for i, r in enumerate(raws):
            out['%s_grad%d_checks' % (tag, i)] = checks(r.grad)
            out['%s_grad%d_rows' % (tag, i)] = r.grad.reshape(-1)[::97].numpy()
            out['%s_idx%d' % (tag, i)] = torch.stack(indices[i]).numpy()
            out['%s_tbox%d' % (tag, i)] = tbox[i].numpy()
            out['%s_tcls%d' % (tag, i)] = tcls[i].numpy()
    np.savez_compressed(os.path.join(HERE, 'loss.npz'), **out)
    print('loss fixture ok', {k: v for k, v in out.items() if k.endswith('items')})


def train_fixture(ref):
    """One training step of the REFERENCE on CPU (row T): train-mode forward (batch-statistics BatchNorm), compute_loss,
    backward — yolov3-tiny-hand @128 batch 4 and the first stage of yolov3 is too slow for a fixture, so tiny + yolov4-tiny
    (maxpools, group routes).  Stored: raw-head checksums, loss items, per-parameter gradient checksums and a row subset,
    running statistics after the step."""
    out = {}
    for tag, rel, size, nc in (('tinyhand', 'yolov3tiny/yolov3-tiny-hand.cfg', 128, 1), ('v4tiny', 'yolov4tiny/yolov4-tiny.cfg', 128, 80)):
        torch.manual_seed(0)
        model = ref.models.Darknet(os.path.join(REFCFG, rel), (size, size))
        state = model.state_dict()
        synth.randomize_bn_(state, seed=1)
        model.load_state_dict(state)
        model.train()
        model.nc, model.hyp, model.gr = nc, dict(HYP), 1.0
        x = synth.image_batch(4, size, seed=0)
        targets = synth.loss_inputs(model, size, batch=4, seed=9, labels_per_image=5)[1]
        pred, _ = model(x)
        loss, items = ref.utils.compute_loss(pred, targets, model)
        loss.backward()
        out[tag + '_items'] = items.numpy()
        for i, p in enumerate(pred):
            out['%s_raw%d_checks' % (tag, i)] = checks(p)
        names, gsum = [], []
        for k, p in model.named_parameters():
            names.append(k)
            gsum.append(checks(p.grad))
            out['%s_grow_%s' % (tag, k)] = p.grad.reshape(-1)[::211].numpy()
        out[tag + '_param_names'] = np.array(names)
        out[tag + '_grad_checks'] = np.stack(gsum)
        rs = {k: v for k, v in model.state_dict().items() if 'running' in k}
        out[tag + '_running_names'] = np.array(list(rs))
        out[tag + '_running_checks'] = np.stack([checks(v) for v in rs.values()])
    np.savez_compressed(os.path.join(HERE, 'train_step.npz'), **out)
    print('train fixture ok', {k: v for k, v in out.items() if k.endswith('items')})


def main():
    ref = refharness.load()
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if only == ['loss']:
        return loss_fixture(ref)
    if only == ['train']:
        return train_fixture(ref)
    for case in NET_CASES:
        if only and case[0] not in only:
            continue
        net_fixture(ref, *case)
    if only:
        return
    nms_fixtures(ref)
    loss_fixture(ref)
    train_fixture(ref)
    fuse_fixture(ref)
    decode_fixture(ref)


if __name__ == '__main__':
    main()
