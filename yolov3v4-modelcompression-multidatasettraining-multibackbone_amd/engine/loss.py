"""compute_loss on the HIP kernels (csrc/loss.hip): the autograd node behind ``utils.utils.compute_loss`` for CUDA tensors.

The reference evaluates the loss with ~200 small torch kernels per step and autograd then builds several full-size
zero-filled temporaries per head (index_put / select backward).  Here the forward is two launches per head (matched
targets, objectness over every cell) and the backward two more that write the complete gradient of each raw head tensor,
scaled by autograd's incoming scalar on the device (no host sync).  Target assignment stays ``build_targets``.
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


class _YoloLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, *ps):
        lib = _lib()
        dev = ps[0].device
        heads = []
        sums = torch.zeros((len(ps), 3), device=dev, dtype=torch.float32)
        lbox = torch.zeros((), device=dev)
        lobj = torch.zeros((), device=dev)
        lcls = torch.zeros((), device=dev)
        for i, p in enumerate(ps):
            bs, na, ny, nx, no = p.shape
            idx, tbox, tcls, anchor = meta['matched'][i]
            nb = int(idx.shape[0])
            tobj = torch.zeros((bs, na, ny, nx), device=dev, dtype=torch.float32)
            winner = torch.full((bs, na, ny, nx), -1, device=dev, dtype=torch.int32) if nb else None
            cells = bs * na * ny * nx
            d = LossDesc(p=hiplib.ptr(p), grad=None, tobj=hiplib.ptr(tobj), winner=hiplib.ptr(winner), idx=hiplib.ptr(idx) if nb else None,
                         tbox=hiplib.ptr(tbox) if nb else None, tcls=hiplib.ptr(tcls) if nb else None,
                         anchor=hiplib.ptr(anchor) if nb else None, sums=hiplib.ptr(sums, 3 * i), scale=None,
                         sb=p.stride(0), sa=p.stride(1), sy=p.stride(2), sx=p.stride(3), gb=0, ga=0, gy=0, gx=0,
                         bs=bs, na=na, ny=ny, nx=nx, no=no, nc=no - 5, nb=nb, gr=meta['gr'], cp=meta['cp'], cn=meta['cn'],
                         cls_pw=meta['cls_pw'], obj_pw=meta['obj_pw'],
                         w_box=meta['giou'] / max(nb, 1), w_obj=meta['obj'] / cells,
                         w_cls=meta['cls'] / max(nb * (no - 5), 1))
            hiplib.check(lib.yh_yolo_loss_fwd(C.byref(d), hiplib.stream_ptr()), 'yh_yolo_loss_fwd')
            heads.append((d, tobj, p))
            del winner   # only the forward needs it (stream-ordered free)
            if nb:
                lbox = lbox + sums[i, 0] * d.w_box
                if no - 5 > 1:
                    lcls = lcls + sums[i, 2] * d.w_cls
            lobj = lobj + sums[i, 1] * d.w_obj
        ctx.heads, ctx.meta = heads, meta
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
            g = torch.empty_strided(p.shape, p.stride(), device=p.device, dtype=torch.float32)
            d.grad, d.scale = hiplib.ptr(g), hiplib.ptr(scale)
            d.gb, d.ga, d.gy, d.gx = g.stride(0), g.stride(1), g.stride(2), g.stride(3)
            hiplib.check(lib.yh_yolo_loss_bwd(C.byref(d), hiplib.stream_ptr()), 'yh_yolo_loss_bwd')
            grads.append(g)
        ctx.keep = scale
        return (None,) + tuple(grads)


def compute_loss(p, targets, model, build_targets, smooth_bce):
    """Same contract as utils.utils.compute_loss: ``(loss[1], detached [lbox, lobj, lcls, loss])``."""
    tcls, tbox, indices, anchor_vec = build_targets(p, targets, model)
    h = model.hyp
    cp, cn = smooth_bce(eps=0.0)
    matched = []
    for i in range(len(p)):
        b, a, gj, gi = indices[i]
        idx = torch.stack((b, a, gj, gi), 1).to(torch.int32).contiguous()
        matched.append((idx, tbox[i].float().contiguous(), tcls[i].to(torch.int32).contiguous(),
                        anchor_vec[i].float().contiguous()))
    meta = dict(matched=matched, gr=float(model.gr), cp=float(cp), cn=float(cn), cls_pw=float(h['cls_pw']),
                obj_pw=float(h['obj_pw']), giou=float(h['giou']), obj=float(h['obj']), cls=float(h['cls']))
    loss, lbox, lobj, lcls = _YoloLoss.apply(meta, *p)
    return loss.reshape(1), torch.stack((lbox, lobj, lcls, loss.detach())).detach()
