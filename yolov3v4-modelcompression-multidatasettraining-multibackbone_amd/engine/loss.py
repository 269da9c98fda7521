"""compute_loss on the HIP kernels (csrc/loss.hip): the autograd node behind ``utils.utils.compute_loss`` for CUDA tensors.

The reference evaluates the loss with ~200 small torch kernels per step, autograd then builds several full-size
zero-filled temporaries per head (index_put / select backward), and ``build_targets`` reads data-dependent sizes back to
the host three times per head.  Here the forward is three launches per head (target assignment, matched candidates,
objectness over every cell) and the backward two more that write the complete gradient of each raw head tensor, scaled
by autograd's incoming scalar on the device.  Nothing in the step waits for the GPU, so the host keeps enqueuing the
backward while the forward still runs.

The reference's label check (``build_targets`` asserts every class < nc, utils.py:757-760) needs the labels on the host;
here the kernels record a label error mask on the device, the mask is copied to pinned host memory asynchronously and the
same AssertionError is raised by the NEXT ``compute_loss`` call (or ``flush_label_check()``), after the copy has landed.
"""
import ctypes as C

import torch

from . import hiplib
from .hiplib import LossDesc

_LIB_OVERRIDE = None   # tests inject the host emulation here


def _lib():
    return _LIB_OVERRIDE if _LIB_OVERRIDE is not None else hiplib.load()


def usable(p, model):
    """Fused path: CUDA (or an injected emulator), fp32 raw heads with unit stride on the last axis, no focal loss."""
    if _LIB_OVERRIDE is None and not p[0].is_cuda:
        return False
    if model.hyp.get('fl_gamma', 0.0) > 0:
        return False
    return all(t.dtype == torch.float32 and t.dim() == 5 and t.stride(4) == 1 for t in p)


_pending = []   # (event or None, host int32 tensor (heads, 2), nc) of launched forwards whose label mask is unread


def flush_label_check():
    """Raise for label errors recorded by earlier compute_loss launches (waits for their kernels)."""
    while _pending:
        event, host, nc = _pending.pop(0)
        if event is not None:
            event.synchronize()
        mask = int(host[:, 1].max()) if host.numel() else 0
        if mask & 1:
            raise AssertionError('Model accepts %g classes labeled from 0-%g, however you labelled a class outside that '
                                 'range. See https://docs.ultralytics.com/yolov5/tutorials/train_custom_data' % (nc, nc - 1))
        if mask & 2:
            raise IndexError('a label refers to an image index outside the batch or a box centre outside [0, 1)')


def _head_bases(p):
    """The NHWC head tensors ``(bs, ny, nx, ld)`` the raw heads are views of, when every ``p[i]`` is exactly the view
    ``base[..., :na * no].view(bs, ny, nx, na, no).permute(0, 3, 1, 2, 4)`` the training forward hands out (models.py
    ``_forward_hip_train``); None otherwise.  With the bases as the autograd inputs the loss backward writes the gradient of the
    head tensor itself - autograd's slice / view backward (a zero fill plus a strided copy of every head, ~0.3 ms per step at
    YOLOv3-608 batch 64) disappears."""
    out = []
    for t in p:
        b = t._base
        if b is None or b.dim() != 4 or b.dtype != torch.float32 or not b.is_contiguous() or not b.requires_grad:
            return None
        bs, na, ny, nx, no = t.shape
        ld = b.shape[3]
        if tuple(b.shape[:3]) != (bs, ny, nx) or na * no > ld or t.data_ptr() != b.data_ptr():
            return None
        if tuple(t.stride()) != (ny * nx * ld, no, nx * ld, ld, 1):
            return None
        out.append((b, na, no))
    return out


class _YoloLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, *ps):
        lib = _lib()
        dev = ps[0].device
        nh = len(ps)
        ctx.on_bases = meta.get('geom') is not None
        if ctx.on_bases:     # inputs are the NHWC head tensors: rebuild the (bs, na, ny, nx, no) views the kernels address
            ps = tuple(b[..., :na * no].view(b.shape[0], b.shape[1], b.shape[2], na, no).permute(0, 3, 1, 2, 4)
                       for b, (na, no) in zip(ps, meta['geom']))
        targets = meta['targets']
        nt = int(targets.shape[0])
        sums = torch.zeros((nh, 3), device=dev, dtype=torch.float32)
        counts = torch.zeros((nh, 2), device=dev, dtype=torch.int32)
        heads, cells = [], []
        for i, p in enumerate(ps):
            bs, na, ny, nx, no = p.shape
            tobj = torch.zeros((bs, na, ny, nx), device=dev, dtype=torch.float32)
            winner = torch.full((bs, na, ny, nx), -1, device=dev, dtype=torch.int32) if nt else None
            anchors = meta['anchors'][i]
            d = LossDesc(p=hiplib.ptr(p), grad=None, tobj=hiplib.ptr(tobj), winner=hiplib.ptr(winner),
                         targets=hiplib.ptr(targets) if nt else None, anchors=hiplib.ptr(anchors), sums=hiplib.ptr(sums, 3 * i),
                         count=hiplib.ptr(counts, 2 * i), scale=None,
                         sb=p.stride(0), sa=p.stride(1), sy=p.stride(2), sx=p.stride(3), gb=0, ga=0, gy=0, gx=0,
                         bs=bs, na=na, ny=ny, nx=nx, no=no, nc=no - 5, nt=nt, iou_t=meta['iou_t'], gr=meta['gr'],
                         cp=meta['cp'], cn=meta['cn'], cls_pw=meta['cls_pw'], obj_pw=meta['obj_pw'],
                         g_box=meta['giou'], g_obj=meta['obj'], g_cls=meta['cls'])
            hiplib.check(lib.yh_yolo_loss_fwd(C.byref(d), hiplib.stream_ptr()), 'yh_yolo_loss_fwd')
            heads.append((d, tobj, p))
            cells.append(float(bs * na * ny * nx))
            del winner   # only the forward needs it (stream-ordered free)
        nc = ps[0].shape[4] - 5
        nb = counts[:, 0].clamp(min=1).to(torch.float32)
        lbox = (sums[:, 0] / nb).sum() * meta['giou']
        lobj = sum(sums[i, 1] / cells[i] for i in range(nh)) * meta['obj']
        lcls = (sums[:, 2] / (nb * nc)).sum() * meta['cls'] if nc > 1 else torch.zeros((), device=dev)
        if dev.type == 'cuda':
            host = torch.empty((nh, 2), dtype=torch.int32, pin_memory=True)
            host.copy_(counts, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            _pending.append((event, host, nc))
        else:
            _pending.append((None, counts, nc))
            flush_label_check()
        ctx.heads, ctx.keep_fwd = heads, (targets, counts, meta['anchors'])
        loss = lbox + lobj + lcls
        ctx.mark_non_differentiable(lbox, lobj, lcls)
        return loss, lbox, lobj, lcls

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, g_lbox, g_lobj, g_lcls):
        lib = _lib()
        scale = g_loss.detach().float().contiguous().reshape(1)
        grads = []
        for d, tobj, p in ctx.heads:
            if ctx.on_bases:
                bs, na, ny, nx, no = p.shape
                ld = p.stride(3)
                full = torch.empty((bs, ny, nx, ld), device=p.device, dtype=torch.float32)      # gradient of the head tensor itself
                if ld > na * no:
                    full[..., na * no:].zero_()                                                  # channel padding of the head conv
                g = full[..., :na * no].view(bs, ny, nx, na, no).permute(0, 3, 1, 2, 4)
            else:
                g = full = torch.empty_strided(p.shape, p.stride(), device=p.device, dtype=torch.float32)
            d.grad, d.scale = hiplib.ptr(g), hiplib.ptr(scale)
            d.gb, d.ga, d.gy, d.gx = g.stride(0), g.stride(1), g.stride(2), g.stride(3)
            with hiplib.on_device(p):
                hiplib.check(lib.yh_yolo_loss_bwd(C.byref(d), hiplib.stream_ptr()), 'yh_yolo_loss_bwd')
            grads.append(full)
        ctx.keep = scale
        return (None,) + tuple(grads)


def compute_loss(p, targets, model, yolo_modules, smooth_bce):
    """Same contract as utils.utils.compute_loss: ``(loss[1], detached [lbox, lobj, lcls, loss])``."""
    flush_label_check()
    h = model.hyp
    cp, cn = smooth_bce(eps=0.0)
    dev = p[0].device
    anchors = [layer.anchor_vec.to(device=dev, dtype=torch.float32).contiguous() for layer in yolo_modules(model)]
    meta = dict(targets=targets.to(device=dev, dtype=torch.float32).contiguous(), anchors=anchors, iou_t=float(h['iou_t']),
                gr=float(model.gr), cp=float(cp), cn=float(cn), cls_pw=float(h['cls_pw']),
                obj_pw=float(h['obj_pw']), giou=float(h['giou']), obj=float(h['obj']), cls=float(h['cls']))
    bases = _head_bases(p)
    with hiplib.on_device(p[0]):
        if bases is not None:
            meta['geom'] = [(na, no) for _, na, no in bases]
            loss, lbox, lobj, lcls = _YoloLoss.apply(meta, *[b for b, _, _ in bases])
        else:
            loss, lbox, lobj, lcls = _YoloLoss.apply(meta, *p)
    return loss.reshape(1), torch.stack((lbox, lobj, lcls, loss.detach())).detach()
