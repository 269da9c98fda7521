"""Training step of the cfg graph on the HIP kernels.

Replaces, for CUDA tensors in ``model.train()`` mode, what the reference gets from eager PyTorch + autograd:
``Darknet.forward`` in training mode (models.py:414-492: conv -> train-mode BatchNorm -> activation per block, shortcut
adds, route concats, upsample, raw ``p`` per yolo head, models.py:336-340) and the backward of all of it.  The loss
(``compute_loss``, utils.py:368) stays a torch function on the three raw head tensors; its gradient enters here as
fp32 head gradients and leaves as fp32 parameter gradients (fp16 compute with fp32 master weights and gradients, the
``-mpt`` mixed-precision recipe of train.py; or fp32 throughout).

Lowering is shared with the inference engine (``DarknetEngine._build_values`` / ``_place``): the same values, the same
zero-copy concat placement, the same fused residual / 2x-upsample stores.  Per conv block the forward plan runs

    conv (linear, MFMA implicit GEMM) -> z      yh_conv2d_fwd / yh_conv2d_stem_fwd
    batch statistics of z                      yh_bn_stats, yh_bn_finalize (also the running-stat update)
    y = act(bn(z)) [+ residual] [2x store]     yh_bn_act_fwd

and the backward plan, walking the values in reverse,

    dy of a fused upsample                     yh_upsample2_bwd
    residual: grad(res) += dy                  yh_add_channels / yh_copy_channels
    dgamma, dbeta (or conv-bias grad)          yh_bn_act_bwd_reduce   (written straight into the gradient arena)
    dz                                         yh_bn_act_bwd_apply
    dW                                         yh_conv2d_wgrad (first layer: through an 8-channel NHWC copy of the image)
    grad(input) (+)= conv(dz, W^T flipped)     yh_conv2d_fwd on the dgrad weight image (stride 2: four parity phases)

Every activation, z and gradient buffer is kept for the whole step (288 GB of HBM: YOLOv3-608 batch 64 needs ~40 GB).
Depthwise blocks, squeeze-excite, max-pools (incl. SPP) and CSP group-split routes are lowered too (csrc/depthwise.hip,
csrc/train.hip).  Graphs whose widths are not multiples of 8 (slim-pruned nets, GhostNet) train through the channel-padded twin
of ``engine/padded.py``; what neither form covers (weighted shortcuts, grouped convs other than depthwise) raises
NotImplementedError at plan build - ``models.Darknet`` has no eager fallback for training.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import hiplib
from .hiplib import (ConvDesc, StemDesc, CopyDesc, AddDesc, BnStatsDesc, BnFinalizeDesc, BnActFwdDesc, BnBwdReduceDesc,
                     BnBwdApplyDesc, WgradDesc, StemWgradDesc, StemBwdDesc, UpsampleBwdDesc, CastDesc, LayoutDesc, PoolDesc, PoolBwdDesc, PackItem, PackBatchDesc, DwDesc, SeDesc, DwWgradDesc,
                     DwDgradDesc, SeBwdDesc)
from .plan import DarknetEngine, ALIGN_C, _round_up

# stride-2 data gradients with at most this many input channels run as ONE four-phase pass (ups = 4: 16 tap-GEMMs instead of 9, dz read
# once); above it the four parity-phase launches.  Measured (profiles/r05_fused_dgrad_ab.txt): 64 input channels (152^2, 128 -> 64) 0.642 ms
# fused against 0.704 in four launches; 128 channels (76^2) 0.480 against 0.441 - the 78 % of extra tap work outweighs the dz re-reads from
# there on.  A/B knob: YOLO_HIP_FUSED_DGRAD_MAX
_FUSED_DGRAD_MAX_CIN = int(os.environ.get('YOLO_HIP_FUSED_DGRAD_MAX', '64'))

SLOT_INPUT = 0
SLOT_WS = 1          # shared fp32 workspace of the two-stage reductions
SLOT_WS2 = 63        # the same for the ops of the side lane (weight gradients): the two lanes run concurrently
SLOT_HEAD0 = 2       # forward: head output tensors; backward: head gradient tensors (fp32 NHWC)
LINEAR = hiplib.ACT_CODES['linear']


class ChannelAlignmentError(NotImplementedError):
    """The aligned lowering wants every tensor width to be a multiple of 8: ``engine/padded.py`` trains such graphs through a
    channel-padded twin."""


class _Arena:
    """One flat fp32 tensor handed out in 16-byte aligned pieces (per-channel statistics, parameter gradients)."""

    def __init__(self):
        self.size = 0
        self.items = []

    def reserve(self, n):
        off = self.size
        self.size += _round_up(n, 4)
        return off

    def allocate(self, device):
        self.buf = torch.zeros(max(self.size, 4), device=device, dtype=torch.float32)
        return self.buf

    def ptr(self, off):
        return self.buf.data_ptr() + 4 * off

    def view(self, off, n):
        return self.buf[off:off + n]


class TrainEngine(DarknetEngine):
    def __init__(self, model, precision='fp16', lib=None):
        if precision not in ('fp16', 'fp32'):
            raise ValueError("training precision must be 'fp16' or 'fp32'")
        super().__init__(model, precision, lib)
        self._tplans = {}
        self._current = None
        self.steps = 0   # forward count; the autograd node checks it so a stale backward fails loudly

    # --------------------------------------------------------------------------------- parameters
    def parameters(self):
        """Trainable tensors in the order the autograd function receives them (and returns gradients for)."""
        out = []
        for block in self.model.module_list:
            if isinstance(block, nn.Sequential) and len(block) and isinstance(block[0], nn.Conv2d):
                conv, bn = block[0], None
                for k in list(block.children())[1:]:
                    if isinstance(k, nn.modules.batchnorm.BatchNorm2d):
                        bn = k
                out.append(conv.weight)
                if conv.bias is not None:
                    out.append(conv.bias)
                if bn is not None:
                    out.extend((bn.weight, bn.bias))
            elif isinstance(block, nn.Sequential) and len(block) and block[0].__class__.__name__ == 'SE':
                out.extend((block[0].fc[0].weight, block[0].fc[2].weight))
        return out

    # --------------------------------------------------------------------------------- weights
    def _pack_items(self, plan):
        """Host image of the yh_pack_item table: every weight image of the step (forward, data gradient or its four
        stride-2 phases, first layer) straight from the live fp32 parameters."""
        items = []
        P = hiplib.ptr
        for v in plan['values']:
            if v.kind == 'dw':
                conv = v.conv
                for t in (conv.weight, conv.bias):
                    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
                        raise NotImplementedError('HIP training path: parameters must be contiguous fp32 (master weights)')
                items.append(PackItem(w=P(conv.weight), bias=P(conv.bias), packed=P(v.tpack['w']), bias_out=P(v.tpack['b']), mode=4,
                                      dtype=self.code, cout=v.C, cin=1, kh=v.k, kw=v.k, k_pad=v.c_phys, pad=v.pad))
                continue
            if v.kind != 'conv':
                continue
            conv, pk = v.conv, v.tpack
            for t in (conv.weight, conv.bias):
                if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
                    raise NotImplementedError('HIP training path: parameters must be contiguous fp32 (master weights)')
            w, cb = P(conv.weight), P(conv.bias)
            geo = dict(dtype=self.code, cout=v.C, cin=conv.in_channels, kh=v.k, kw=v.k, pad=v.pad)
            if v.src.kind == 'input':
                items.append(PackItem(w=w, bias=cb, packed=P(pk['w']), bias_out=P(pk['b']), mode=3, cout_pad=pk['cout_pad'], **geo))
                continue
            items.append(PackItem(w=w, bias=cb, packed=P(pk['w']), bias_out=P(pk['b']), mode=0, k_pad=pk['cin_k'], m_pad=pk['m_pad'],
                                  **geo))
            if v.stride == 2 and pk.get('stem_fused'):
                # its data gradient runs inside the first block's backward pass (csrc/stem_bwd.hip): the plain flipped-tap image
                items.append(PackItem(w=w, packed=P(pk['wt']), mode=1, k_pad=pk['cout_k'], m_pad=pk['dm_pad'], **geo))
            elif v.stride == 2 and 'wt_fused' in pk:
                items.append(PackItem(w=w, packed=P(pk['wt_fused']), mode=5, k_pad=pk['cout_k'], m_pad=pk['fused_m_pad'],
                                      cout_pad=v.src.c_phys, **geo))
            elif v.stride == 2:
                for (a, b), img in zip(((0, 0), (0, 1), (1, 0), (1, 1)), pk['wt_phase']):
                    items.append(PackItem(w=w, packed=P(img), mode=2, k_pad=pk['cout_k'], m_pad=pk['dm_pad'], pa=a, pb=b, **geo))
            else:
                items.append(PackItem(w=w, packed=P(pk['wt']), mode=1, k_pad=pk['cout_k'], m_pad=pk['dm_pad'], **geo))
        arr = (PackItem * len(items))(*items)
        return bytes(arr), len(items)

    def _refresh_pack_table(self, plan):
        """(Re)write the device table when a parameter moved (plan build, ``param.data = ...``)."""
        ptrs = tuple(t.data_ptr() for t in self.parameters())
        if plan.get('pack_ptrs') == ptrs:
            return
        raw, n = self._pack_items(plan)
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        if plan.get('pack_table') is None or plan['pack_table'].numel() != host.numel():
            raise RuntimeError('pack table size changed: rebuild the plan')
        plan['pack_table'].copy_(host)
        plan['pack_ptrs'] = ptrs

    # --------------------------------------------------------------------------------- plans
    def _check_supported(self, values):
        for v in values:
            if v.kind in ('dw', 'se') and v.c_phys != v.C:
                raise ChannelAlignmentError('HIP training path: depthwise / SE blocks need channel counts that are multiples of %d' % ALIGN_C)
            if v.kind in ('qadd',):
                raise NotImplementedError('HIP training path: %s blocks are not lowered yet (block %s)' % (v.kind, v.block))
            if v.kind == 'conv':
                if v.src.kind != 'input' and (v.src.C % ALIGN_C or v.src.c_phys != v.src.C):
                    raise ChannelAlignmentError('HIP training path: channel counts must be multiples of %d' % ALIGN_C)
                if not v.fp32 and v.C % ALIGN_C:
                    raise ChannelAlignmentError('HIP training path: channel counts must be multiples of %d' % ALIGN_C)
                if v.src.kind == 'input' and (v.src.C > 4 or v.src.C > ALIGN_C):
                    raise NotImplementedError('HIP training path: the first conv reads at most 4 image channels')
                if v.stride == 2 and not (v.k == 3 and v.pad == 1):
                    raise NotImplementedError('HIP training path: stride-2 convs must be 3x3 pad 1')
                if v.stride not in (1, 2):
                    raise NotImplementedError('HIP training path: conv stride %d' % v.stride)
                if v.bn is not None and v.fp32:
                    raise NotImplementedError('HIP training path: BatchNorm on a yolo-head conv')

    def _build_train_plan(self, N, Cin, H, W):
        values, heads, outs = self._build_values(N, Cin, H, W)
        self._check_supported(values)
        self._place(values)
        lib, dev, P = self.lib, self.device, hiplib.ptr
        plan = dict(values=values, heads=heads, N=N, storages=[], fwd_ops=[], bwd_ops=[], zero_list=[], ws_floats=0, ws2_floats=0)
        fwd = plan['fwd'] = lib.yh_plan_create()
        bwd = plan['bwd'] = lib.yh_plan_create()
        if not fwd or not bwd:
            raise MemoryError('yh_plan_create failed')
        stats = plan['stats'] = _Arena()      # zeroed before every forward: BN sums
        saved = plan['saved'] = _Arena()      # mean / invstd per BN (kept from forward to backward)
        grads = plan['grads'] = _Arena()      # parameter gradients (zeroed before every backward)

        def add(handle, log, desc, what):
            idx = lib.yh_plan_add(handle, hiplib.OP_KIND[type(desc)], C.byref(desc), C.sizeof(desc))
            if idx < 0:
                hiplib.check(idx, 'yh_plan_add(%s)' % what)
            log.append((what, desc))
            return idx

        def fixup(handle, op, desc_type, field, slot):
            hiplib.check(lib.yh_plan_add_fixup(handle, op, getattr(desc_type, field).offset, slot, 0), 'fixup')

        def add_reduction(handle, log, desc, what, side=False):
            """Reductions get the shared workspace (bound to SLOT_WS at run time) sized by the library's own query; ops of the
            side lane have their own (SLOT_WS2)."""
            query = lib.yh_conv2d_wgrad_workspace if isinstance(desc, WgradDesc) and not isinstance(desc, StemWgradDesc) \
                else lib.yh_dw_wgrad_workspace if isinstance(desc, DwWgradDesc) else lib.yh_bn_reduce_workspace
            need = int(query(C.byref(desc)))
            desc.ws_floats = need
            key = 'ws2_floats' if side else 'ws_floats'
            plan[key] = max(plan[key], need)
            op = add(handle, log, desc, what)
            if need:
                fixup(handle, op, type(desc), 'ws', SLOT_WS2 if side else SLOT_WS)
            return op

        def alloc(shape, fp32=False, zero=False):
            t = (torch.zeros if zero else torch.empty)(shape, device=dev, dtype=torch.float32 if fp32 else self.dtype)
            plan['storages'].append(t)
            return t

        def materialize(v):
            if v.storage is not None:
                return
            if v.parent is not None:
                materialize(v.parent)
                v.storage, v.c_off, v.ld = v.parent.storage, v.parent.c_off + v.parent_off, v.parent.ld
                v.gstorage = v.parent.gstorage
            else:
                v.storage = None if v.fp32 else alloc((N, v.H, v.W, v.c_phys))
                v.gstorage = None if v.fp32 else alloc((N, v.H, v.W, v.c_phys))
                v.c_off, v.ld = 0, v.c_phys

        # ---- pass 1: buffers, parameter slots
        kstep = self.kstep
        pidx = 0
        plan['param_slices'] = []   # (arena offset, numel, shape) in parameters() order
        for v in values:
            if v.kind == 'input':
                continue
            if v.kind == 'slice':   # a channel range of its source: same buffers, forward and gradient
                materialize(v.src)
                v.storage, v.gstorage, v.c_off, v.ld = v.src.storage, v.src.gstorage, v.src.c_off + v.first, v.src.ld
                continue
            materialize(v)
            if v.kind == 'dw':
                conv, bn = v.conv, v.bn
                v.p_first = len(plan['param_slices'])
                v.g_w = grads.reserve(conv.weight.numel())
                plan['param_slices'].append((v.g_w, conv.weight.numel(), tuple(conv.weight.shape)))
                v.g_b = None
                if conv.bias is not None:
                    v.g_b = grads.reserve(v.c_phys)
                    plan['param_slices'].append((v.g_b, v.C, (v.C,)))
                if bn is not None:
                    v.g_gamma, v.g_beta = grads.reserve(v.C), grads.reserve(v.C)
                    plan['param_slices'].append((v.g_gamma, v.C, (v.C,)))
                    plan['param_slices'].append((v.g_beta, v.C, (v.C,)))
                    v.s_sum, v.s_sumsq = stats.reserve(v.C), stats.reserve(v.C)
                    v.s_mean, v.s_invstd = saved.reserve(v.C), saved.reserve(v.C)
                v.tpack = dict(w=torch.empty(v.k * v.k * v.c_phys, device=dev, dtype=self.dtype),
                               b=torch.empty(v.c_phys, device=dev, dtype=torch.float32))
                v.plain = bn is None and v.act == LINEAR
                v.z = None if v.plain else alloc((N, v.H, v.W, v.c_phys))
                v.res, v.ups, v.Ho, v.Wo, v.fp32 = None, 1, v.H, v.W, False
                continue
            if v.kind == 'se':
                v.p_first = len(plan['param_slices'])
                v.g_w1 = grads.reserve(v.fc1.weight.numel())
                plan['param_slices'].append((v.g_w1, v.fc1.weight.numel(), tuple(v.fc1.weight.shape)))
                v.g_w2 = grads.reserve(v.fc2.weight.numel())
                plan['param_slices'].append((v.g_w2, v.fc2.weight.numel(), tuple(v.fc2.weight.shape)))
                for t in (v.fc1.weight, v.fc2.weight):
                    if t.dtype != torch.float32 or not t.is_contiguous():
                        raise NotImplementedError('HIP training path: parameters must be contiguous fp32 (master weights)')
                v.pooled = alloc((N, v.c_phys), fp32=True)
                v.gate = alloc((N, v.c_phys), fp32=True)
                v.scratch = alloc((N, v.c_phys), fp32=True)
                v.scratch2 = alloc((N, v.c_phys + 2 * v.fc1.weight.shape[0]), fp32=True)      # factors of the two weight gradients per image
                continue
            if v.kind != 'conv':
                continue
            conv, bn = v.conv, v.bn
            taps = v.k * v.k
            v.p_first = len(plan['param_slices'])
            v.g_w = grads.reserve(conv.weight.numel())
            plan['param_slices'].append((v.g_w, conv.weight.numel(), tuple(conv.weight.shape)))
            v.g_b = None
            if conv.bias is not None:
                v.g_b = grads.reserve(v.c_phys)
                plan['param_slices'].append((v.g_b, v.C, (v.C,)))
            if bn is not None:
                v.g_gamma, v.g_beta = grads.reserve(v.C), grads.reserve(v.C)
                plan['param_slices'].append((v.g_gamma, v.C, (v.C,)))
                plan['param_slices'].append((v.g_beta, v.C, (v.C,)))
                v.s_sum, v.s_sumsq = stats.reserve(v.C), stats.reserve(v.C)
                v.s_mean, v.s_invstd = saved.reserve(v.C), saved.reserve(v.C)
            if v.src.kind == 'input':
                cout_pad = _round_up(v.C, 32) if v.C % 32 == 0 else _round_up(v.C, 16)
                v.tpack = dict(w=torch.empty(taps * v.src.C * cout_pad, device=dev, dtype=torch.float32),
                               b=torch.empty(cout_pad, device=dev, dtype=torch.float32), cout_pad=cout_pad)
            else:
                cin_k = _round_up(v.src.c_phys, kstep)
                m_pad = _round_up(v.c_phys, 128)
                cout_k = _round_up(v.c_phys, kstep)          # dgrad: K runs over the output channels
                dm_pad = _round_up(v.src.c_phys, 128)
                v.tpack = dict(w=torch.empty(m_pad * taps * cin_k, device=dev, dtype=self.dtype),
                               b=torch.empty(m_pad, device=dev, dtype=torch.float32), cin_k=cin_k, m_pad=m_pad,
                               wt=None if v.stride == 2 else torch.empty(dm_pad * taps * cout_k, device=dev, dtype=self.dtype),
                               cout_k=cout_k, dm_pad=dm_pad)
                if self._dgrad_fused_into_stem(values, v):
                    v.tpack['stem_fused'] = True
                    v.tpack['wt'] = torch.empty(dm_pad * taps * cout_k, device=dev, dtype=self.dtype)
                elif v.stride == 2 and 16 <= v.src.c_phys <= _FUSED_DGRAD_MAX_CIN and v.src.c_phys % 4 == 0:
                    # few input channels: the data gradient is bound by reading dz and writing dx, so all four parity phases
                    # run as ONE 2x2-tap GEMM with 4 * cin rows (16 tap-GEMMs instead of 9, but dz is read once, not four
                    # times, and each workgroup writes whole rows of dx): yh_conv2d_fwd ups = 4
                    v.tpack['fused_m_pad'] = _round_up(4 * v.src.c_phys, 128)
                    v.tpack['wt_fused'] = torch.empty(v.tpack['fused_m_pad'] * 4 * cout_k, device=dev, dtype=self.dtype)
                elif v.stride == 2:
                    # parity (a, b) sees ((a + pad) // 2 + 1) x ((b + pad) // 2 + 1) taps
                    v.tpack['phase_taps'] = [((a + v.pad) // 2 + 1, (b + v.pad) // 2 + 1) for a in (0, 1) for b in (0, 1)]
                    v.tpack['wt_phase'] = [torch.empty(dm_pad * th * tw * cout_k, device=dev, dtype=self.dtype)
                                           for th, tw in v.tpack['phase_taps']]
            # z: the conv output before BN / activation.  Blocks with neither BN nor activation store straight to y.
            v.plain = bn is None and v.act == LINEAR and v.res is None and v.ups == 1
            if v.fp32:
                if not v.plain:
                    raise NotImplementedError('HIP training path: yolo-head conv with BN / activation / residual')
                v.z = None
            elif v.plain:
                v.z = None
            else:
                v.z = alloc((N, v.Ho, v.Wo, v.c_phys))
        plan['junk_n'] = max(v.c_phys for v in values if v.kind in ('conv', 'dw'))
        plan['junk'] = grads.reserve(plan['junk_n'])
        stats.allocate(dev)
        saved.allocate(dev)
        grads.allocate(dev)
        # one zero row serves every data gradient as its bias: as long as the longest m_pad among them, incl. the one-pass stride-2 form's
        # 4 * cin rows (a narrow / pruned net whose widest conv input is <= 128 but that has a stride-2 conv with 33 .. 64
        # inputs needs 256 there and failed at plan build)
        zero_bias = alloc((max([max(_round_up(v.src.c_phys, 128), v.tpack.get('fused_m_pad', 0))
                                for v in values if v.kind == 'conv' and v.src.kind != 'input'] + [128]),), fp32=True, zero=True)
        dz_elems = max(N * v.Ho * v.Wo * v.c_phys for v in values if v.kind in ('conv', 'dw'))
        # Two lanes in the backward plan (include/yolo_hip.h yh_plan_set_lane), OFF by default: the weight gradient of a layer only
        # feeds the optimizer, so it can run on the plan's side stream, issued after the layer's data gradient, while the main chain
        # goes on with the HBM-bound BatchNorm backward passes of the next layer.  What the two lanes share is the dz scratch
        # (written by a layer's BN-backward apply pass, read by its weight AND data gradient): two buffers alternate, and the writer
        # of a buffer waits for the weight gradient that read it two layers earlier (the host emulation replays the latest schedule
        # these dependencies allow, tests/test_train_emulated.py).  Measured on the MI355X (YOLOv3-608 b64 fp16): 63.3 vs 63.8 ms
        # per step - the two kernels do not co-reside (a weight-gradient workgroup pair holds the whole vector register file of its
        # CU), they time-slice: every kernel takes about twice as long while the other lane is busy.  YOLO_HIP_WGRAD_LANE=1 enables.
        # (round 6: 2 = only the weight gradients of the <= 38 x 38 stages, 3 = only their 1x1 ones - small grids that leave CUs idle)
        lane_mode = int(os.environ.get('YOLO_HIP_WGRAD_LANE', '0') or 0)
        two_lanes = lane_mode > 0
        lane_ok = lambda u: lane_mode == 1 or (lane_mode == 2 and u.Ho <= 38) or (lane_mode == 3 and u.Ho <= 38 and u.k == 1)
        # Round 6: the reduce launches of the weight gradients (partial tiles -> dW) on the plan's reduce stream, in the shadow of the ops that
        # follow (yh_plan_set_async_reduce).  For that the weight gradients get the second workspace to themselves: the shared one is
        # rewritten by the very next BatchNorm reduction.  Measured (profiles/r06_async_reduce_ab.txt, three alternating rounds): 52.18 against
        # 52.01 ms per step - the 1.3 ms of reduce launches do not hide in the following kernels' idle CUs, the events cost what little
        # overlap there is.  OFF by default; YOLO_HIP_ASYNC_REDUCE=1 enables (same bits either way).
        async_reduce = (not two_lanes and os.environ.get('YOLO_HIP_ASYNC_REDUCE', '0') == '1' and hasattr(lib, 'yh_plan_set_async_reduce'))
        dz_bufs = [alloc((dz_elems,)) for _ in range(2 if two_lanes else 1)]
        dz_state = dict(k=0, reader=[None] * len(dz_bufs))

        def next_dz():
            dz_state['k'] = (dz_state['k'] + 1) % len(dz_bufs)
            return dz_state['k']

        def dz_written_by(op, k):
            """op is the first writer of dz buffer k for a new layer: it must not overtake the side-lane reader of the old contents."""
            r = dz_state['reader'][k]
            if r is not None:
                hiplib.check(lib.yh_plan_add_dep(bwd, op, r), 'yh_plan_add_dep')
                dz_state['reader'][k] = None

        def on_side_lane(op, k):
            """op (a weight gradient reading dz buffer k) runs on the side lane, after everything emitted so far on the main lane."""
            if not two_lanes:
                return
            hiplib.check(lib.yh_plan_set_lane(bwd, op, 1), 'yh_plan_set_lane')
            hiplib.check(lib.yh_plan_add_dep(bwd, op, op - 1), 'yh_plan_add_dep')
            dz_state['reader'][k] = op
        head_index = {id(h.src): k for k, h in enumerate(heads)}

        raw, n_items = self._pack_items(plan)
        plan['pack_table'] = torch.zeros(len(raw), device=dev, dtype=torch.uint8)
        plan['storages'].append(plan['pack_table'])
        add(fwd, plan['fwd_ops'], PackBatchDesc(items=plan['pack_table'].data_ptr(), n_items=n_items), 'pack')

        # ---- pass 2: forward ops
        for v in values:
            if v.kind == 'input':
                continue
            y = None if v.storage is None else P(v.storage, v.c_off)
            if v.kind == 'conv':
                pk, s = v.tpack, v.src
                fused_stats = 0
                direct = v.plain                     # conv epilogue writes the block output itself
                zt = None if direct else v.z
                if s.kind == 'input':
                    d = StemDesc(x=None, w=P(pk['w']), bias=P(pk['b']), y=y if direct else P(zt), n=N, cin=s.C, h=s.H,
                                 w_in=s.W, ho=v.Ho, wo=v.Wo, cout=v.c_phys, cout_pad=pk['cout_pad'], kh=v.k, kw=v.k,
                                 stride=v.stride, pad=v.pad, ldy=v.ld if direct else v.c_phys, act=LINEAR, slope=0.0,
                                 dtype=self.code, out_scale=0.0)
                    if v.c_phys > pk['cout_pad'] or v.fp32:
                        raise NotImplementedError('HIP training path: stem geometry')
                    if v.bn is not None and not direct and os.environ.get('YOLO_HIP_STEM_STATS', '1') != '0':
                        # the MFMA first-layer kernel also emits the BatchNorm partial sums of what it stores (no yh_bn_stats pass)
                        fused_stats = int(lib.yh_conv2d_stem_stats_rows(C.byref(d)))
                        if fused_stats:
                            d.stats_ws_floats = fused_stats * 2 * v.c_phys
                            plan['ws_floats'] = max(plan['ws_floats'], d.stats_ws_floats)
                    op = add(fwd, plan['fwd_ops'], d, 'stem%d' % v.block)
                    fixup(fwd, op, StemDesc, 'x', SLOT_INPUT)
                    if fused_stats:
                        fixup(fwd, op, StemDesc, 'stats_ws', SLOT_WS)
                else:
                    d = ConvDesc(x=P(s.storage, s.c_off), w=P(pk['w']), bias=P(pk['b']), res=None,
                                 y=None if v.fp32 else (y if direct else P(zt)), n=N, h=s.H, w_in=s.W, cin=s.c_phys,
                                 ho=v.Ho, wo=v.Wo, cout=v.c_phys, kh=v.k, kw=v.k, stride=v.stride, pad=v.pad, ldx=s.ld,
                                 ldr=0, ldy=v.c_phys if (v.fp32 or not direct) else v.ld, cin_k=pk['cin_k'],
                                 m_pad=pk['m_pad'], act=LINEAR, slope=0.0, ups=1, out_f32=1 if v.fp32 else 0,
                                 dtype=self.code, tile=self.force_tile, acc_scale=0.0, out_scale=0.0)
                    if v.bn is not None and not direct:
                        # BatchNorm statistics ride in the conv epilogue: per-tile partial sums into the shared workspace
                        fused_stats = int(lib.yh_conv2d_stats_rows(C.byref(d)))
                        if fused_stats:
                            d.stats_ws_floats = fused_stats * 2 * v.c_phys
                            plan['ws_floats'] = max(plan['ws_floats'], d.stats_ws_floats)
                    op = add(fwd, plan['fwd_ops'], d, 'conv%d' % v.block)
                    if fused_stats:
                        fixup(fwd, op, ConvDesc, 'stats_ws', SLOT_WS)
                    if v.fp32:
                        fixup(fwd, op, ConvDesc, 'y', SLOT_HEAD0 + head_index[id(v)])
                if direct:
                    continue
                pixels = N * v.Ho * v.Wo
                bn = v.bn
                base = dict(z=P(zt), pixels=pixels, n=N, h=v.Ho, w_in=v.Wo, c=v.c_phys, ldz=v.c_phys, act=v.act,
                            slope=v.slope, dtype=self.code, ups=v.ups)
                if bn is not None:
                    bnp = dict(gamma=P(bn.weight), beta=P(bn.bias), mean=saved.ptr(v.s_mean), invstd=saved.ptr(v.s_invstd),
                               sum=stats.ptr(v.s_sum), sumsq=stats.ptr(v.s_sumsq), eps=float(bn.eps),
                               momentum=float(bn.momentum))
                    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
                        raise NotImplementedError('HIP training path: BatchNorm without momentum / running stats / affine')
                    for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var):
                        if t.dtype != torch.float32 or not t.is_contiguous():
                            raise NotImplementedError('HIP training path: BatchNorm tensors must be contiguous fp32')
                    fused = bool(fused_stats)
                    if not fused:
                        add_reduction(fwd, plan['fwd_ops'], BnStatsDesc(**base, **bnp), 'bnstat%d' % v.block)
                    fin = BnFinalizeDesc(**base, **bnp, running_mean=P(bn.running_mean), running_var=P(bn.running_var),
                                         nparts=fused_stats if fused else 0, ws_floats=(fused_stats * 2 * v.c_phys) if fused else 0)
                    op = add(fwd, plan['fwd_ops'], fin, 'bnfin%d' % v.block)
                    if fused:
                        fixup(fwd, op, BnFinalizeDesc, 'ws', SLOT_WS)
                else:
                    bnp = dict()
                v.bn_args = (base, bnp)
                add(fwd, plan['fwd_ops'],
                    BnActFwdDesc(**base, **bnp, out=y, ldo=v.ld, res=None if v.res is None else P(v.res.storage, v.res.c_off),
                                 ldr=0 if v.res is None else v.res.ld), 'bnact%d' % v.block)
            elif v.kind == 'dw':
                s, pk, bn = v.src, v.tpack, v.bn
                zt = None if v.plain else v.z
                add(fwd, plan['fwd_ops'],
                    DwDesc(x=P(s.storage, s.c_off), w=P(pk['w']), bias=P(pk['b']), y=y if v.plain else P(zt), n=N, h=s.H, w_in=s.W,
                           c=s.c_phys, ho=v.H, wo=v.W, k=v.k, stride=v.stride, pad=v.pad, ldx=s.ld, ldy=v.ld if v.plain else v.c_phys,
                           act=LINEAR, slope=0.0, dtype=self.code), 'dw%d' % v.block)
                if not v.plain:
                    pixels = N * v.H * v.W
                    base = dict(z=P(zt), pixels=pixels, n=N, h=v.H, w_in=v.W, c=v.c_phys, ldz=v.c_phys, act=v.act, slope=v.slope,
                                dtype=self.code, ups=1)
                    bnp = dict()
                    if bn is not None:
                        if bn.momentum is None or not bn.track_running_stats or not bn.affine:
                            raise NotImplementedError('HIP training path: BatchNorm without momentum / running stats / affine')
                        bnp = dict(gamma=P(bn.weight), beta=P(bn.bias), mean=saved.ptr(v.s_mean), invstd=saved.ptr(v.s_invstd),
                                   sum=stats.ptr(v.s_sum), sumsq=stats.ptr(v.s_sumsq), eps=float(bn.eps), momentum=float(bn.momentum))
                        add_reduction(fwd, plan['fwd_ops'], BnStatsDesc(**base, **bnp), 'bnstat%d' % v.block)
                        add(fwd, plan['fwd_ops'], BnFinalizeDesc(**base, **bnp, running_mean=P(bn.running_mean),
                                                                 running_var=P(bn.running_var)), 'bnfin%d' % v.block)
                    v.bn_args = (base, bnp)
                    add(fwd, plan['fwd_ops'], BnActFwdDesc(**base, **bnp, out=y, ldo=v.ld), 'bnact%d' % v.block)
            elif v.kind == 'se':
                s = v.src
                add(fwd, plan['fwd_ops'],
                    SeDesc(x=P(s.storage, s.c_off), y=y, w1=P(v.fc1.weight), w2=P(v.fc2.weight), pooled=P(v.pooled), gate=P(v.gate),
                           ch_map=None, n=N, h=s.H, w_in=s.W, c=v.C, c_phys=v.c_phys, cr=v.fc1.weight.shape[0], ldx=s.ld, ldy=v.ld,
                           dtype=self.code), 'se%d' % v.block)
            elif v.kind == 'pool':
                s = v.src
                add(fwd, plan['fwd_ops'], PoolDesc(x=P(s.storage, s.c_off), y=y, n=N, h=s.H, w_in=s.W, c=s.c_phys, ho=v.H, wo=v.W,
                                                   k=v.k, stride=v.stride, pad_lo=v.pad_lo, edge_zero=v.edge_zero, ldx=s.ld,
                                                   ldy=v.ld, dtype=self.code), 'pool%d' % v.block)
            elif v.kind == 'copy':
                s = v.src
                add(fwd, plan['fwd_ops'], CopyDesc(x=P(s.storage, s.c_off), y=y, n=N, h=s.H, w_in=s.W, c=s.c_phys, ups=v.ups,
                                                   ldx=s.ld, ldy=v.ld, dtype=self.code), 'ups%d' % v.block)
            elif v.kind == 'add':
                add(fwd, plan['fwd_ops'], AddDesc(a=P(v.a.storage, v.a.c_off), b=P(v.b.storage, v.b.c_off), y=y,
                                                  pixels=N * v.H * v.W, c=v.c_phys, lda=v.a.ld, ldb=v.b.ld, ldy=v.ld,
                                                  dtype=self.code), 'add%d' % v.block)
            elif v.kind == 'concat':
                for s, off, inplace in v.parts:
                    if not inplace:
                        add(fwd, plan['fwd_ops'], CopyDesc(x=P(s.storage, s.c_off), y=P(v.storage, v.c_off + off), n=N, h=s.H,
                                                           w_in=s.W, c=s.c_phys, ups=1, ldx=s.ld, ldy=v.ld, dtype=self.code),
                            'cat%d' % v.block)

        # ---- pass 3: backward ops.  A gradient storage's first writer must cover it entirely; otherwise the
        # storage is zeroed before the backward plan runs and every contribution accumulates.
        initialised = set()
        # gradient buffers addressed by more than one value (a concat and the parts produced into it, a tensor and its channel
        # slices): the residual-gradient aliasing below re-points ONE value's buffer and must leave these alone
        holders = {}
        for u in values:
            if getattr(u, 'gstorage', None) is not None:
                holders[id(u.gstorage)] = holders.get(id(u.gstorage), 0) + 1

        # Who reads each value (by identity), in forward order: the backward runs over `values` in reverse, so the LAST contribution to
        # a value's gradient comes from its consumer with the smallest index.  When that consumer is a 1x1 / stride-1 conv whose data
        # gradient the library can run on its persistent kernel, the gradient it completes belongs to a BatchNorm block whose
        # backward sums (sum g, sum g xhat) can be taken by that very launch - it has the finished dy rows in registers and reads z
        # on top - instead of by a reduction pass over dy and z (yh_conv_desc.bwd_z, round 6; YOLO_HIP_FUSE_DBN=0 switches it off).
        index = {id(u): i for i, u in enumerate(values)}
        consumers = {}
        for u in values:
            ins = [getattr(u, 'src', None), getattr(u, 'res', None), getattr(u, 'a', None), getattr(u, 'b', None)]
            ins += [part[0] for part in getattr(u, 'parts', None) or []]
            for t in ins:
                if t is not None and id(t) in index:
                    consumers.setdefault(id(t), []).append(u)
        fuse_dbn = os.environ.get('YOLO_HIP_FUSE_DBN', '1') != '0'

        def completes_bn_block(v, s):
            """conv `v` (1x1 / stride 1) writes the last contribution to grad(s), s a conv with BatchNorm whose gradient buffer is its own."""
            if not fuse_dbn or v.k != 1 or v.stride != 1 or v.pad != 0 or s.kind != 'conv' or s.bn is None or s.plain or s.fp32 or s.ups != 1:
                return False
            if s.src.kind == 'input' or getattr(s, 'parent', None) is not None or s.c_off != 0 or s.ld != s.c_phys:
                return False
            if getattr(s, 'z', None) is None or not hasattr(s, 'bn_args'):
                return False
            return min(index[id(u)] for u in consumers.get(id(s), [v])) == index[id(v)]

        def contribution_mode(t, full_width):
            key = id(t.gstorage)
            if key in initialised:
                return 'acc'
            initialised.add(key)
            if full_width and t.c_off == 0 and t.ld == t.c_phys:
                return 'write'
            plan['zero_list'].append(t.gstorage)
            return 'acc'

        def gptr(t):
            return P(t.gstorage, t.c_off)

        def contribute(t, src_ptr, src_ld, pixels_hw, what):
            """grad(t) (+)= rows at src_ptr (same spatial size and channel count as t)."""
            mode = contribution_mode(t, True)
            if mode == 'write':
                add(bwd, plan['bwd_ops'], CopyDesc(x=src_ptr, y=gptr(t), n=N, h=t.H, w_in=t.W, c=t.c_phys, ups=1, ldx=src_ld,
                                                   ldy=t.ld, dtype=self.code), what)
            else:
                add(bwd, plan['bwd_ops'], AddDesc(a=gptr(t), b=src_ptr, y=gptr(t), pixels=N * t.H * t.W, c=t.c_phys, lda=t.ld,
                                                  ldb=src_ld, ldy=t.ld, dtype=self.code), what)

        def has_grad(t):
            return t.fp32 or id(t.gstorage) in initialised

        bwd_pos = {}   # value -> index of its first backward op (ops of value i span [pos[i], pos[i-1]) in emission order)
        for v in reversed(values):
            bwd_pos[id(v)] = len(plan['bwd_ops'])
            if v.kind == 'input':
                continue
            if v.kind == 'concat':
                if not has_grad(v):
                    continue
                for s, off, inplace in v.parts:
                    if not inplace:
                        contribute(s, P(v.gstorage, v.c_off + off), v.ld, None, 'dcat%d' % v.block)
                continue
            if v.kind == 'add':
                if has_grad(v):
                    contribute(v.a, gptr(v), v.ld, None, 'dadd%d' % v.block)
                    contribute(v.b, gptr(v), v.ld, None, 'dadd%d' % v.block)
                continue
            if v.kind == 'slice':
                continue   # contributions to the slice were written straight into the source's gradient buffer
            if v.kind == 'se':
                if has_grad(v):
                    s = v.src
                    mode = contribution_mode(s, True)
                    hw = s.H * s.W
                    add(bwd, plan['bwd_ops'],
                        SeBwdDesc(x=P(s.storage, s.c_off), dy=gptr(v), dx=gptr(s), w1=P(v.fc1.weight), w2=P(v.fc2.weight),
                                  pooled=P(v.pooled), gate=P(v.gate), dw1=grads.ptr(v.g_w1), dw2=grads.ptr(v.g_w2), scratch=P(v.scratch),
                                  scratch2=P(v.scratch2), scratch2_floats=v.scratch2.numel(),
                                  n=N, h=s.H, w_in=s.W, c=v.c_phys, cr=v.fc1.weight.shape[0], ldx=s.ld, lddy=v.ld, lddx=s.ld,
                                  accumulate=1 if mode == 'acc' else 0, dtype=self.code), 'dse%d' % v.block)
                continue
            if v.kind == 'dw':
                if not has_grad(v):
                    continue
                s, pk = v.src, v.tpack
                pixels = N * v.H * v.W
                dyp, lddy = gptr(v), v.ld
                if v.plain:
                    dzp, lddz = dyp, lddy
                else:
                    kz = next_dz()
                    dzp, lddz = P(dz_bufs[kz]), v.c_phys
                    base, bnp = v.bn_args
                    if v.bn is not None:
                        acc = dict(bnp, sum=grads.ptr(v.g_beta), sumsq=grads.ptr(v.g_gamma))
                    else:
                        acc = dict(sum=grads.ptr(v.g_b if v.g_b is not None else self._junk(plan, grads, v.c_phys)),
                                   sumsq=grads.ptr(self._junk(plan, grads, v.c_phys)))
                    add_reduction(bwd, plan['bwd_ops'], BnBwdReduceDesc(**base, **acc, dy=dyp, lddy=lddy), 'dbn%d' % v.block)
                    op = add(bwd, plan['bwd_ops'], BnBwdApplyDesc(**base, **acc, dy=dyp, lddy=lddy, out=dzp, ldo=v.c_phys), 'dbnx%d' % v.block)
                    dz_written_by(op, kz)
                geo = dict(n=N, h=s.H, w_in=s.W, c=v.c_phys, ho=v.H, wo=v.W, k=v.k, stride=v.stride, pad=v.pad, ldx=s.ld, lddz=lddz,
                           dtype=self.code)
                add_reduction(bwd, plan['bwd_ops'], DwWgradDesc(x=P(s.storage, s.c_off), dz=dzp, dw=grads.ptr(v.g_w), lddx=0, accumulate=0, **geo),
                              'dwwgrad%d' % v.block)
                if s.kind != 'input':
                    mode = contribution_mode(s, True)
                    add(bwd, plan['bwd_ops'], DwDgradDesc(dz=dzp, w=P(pk['w']), dx=gptr(s), lddx=s.ld,
                                                          accumulate=1 if mode == 'acc' else 0, **geo), 'dwdgrad%d' % v.block)
                continue
            if v.kind == 'pool':
                if has_grad(v):
                    s = v.src
                    contribution_mode(s, False)   # scatter-add: the target must hold zeros or an earlier contribution
                    add(bwd, plan['bwd_ops'],
                        PoolBwdDesc(x=P(s.storage, s.c_off), dy=gptr(v), dx=gptr(s), n=N, h=s.H, w_in=s.W, c=s.c_phys, ho=v.H,
                                    wo=v.W, k=v.k, stride=v.stride, pad_lo=v.pad_lo, edge_zero=v.edge_zero, ldx=s.ld, lddy=v.ld,
                                    lddx=s.ld, dtype=self.code), 'dpool%d' % v.block)
                continue
            if v.kind == 'copy':
                if has_grad(v):
                    s = v.src
                    if contribution_mode(s, True) == 'write':
                        add(bwd, plan['bwd_ops'], UpsampleBwdDesc(x=gptr(v), y=gptr(s), n=N, h=s.H, w_in=s.W, c=s.c_phys, big_h=2 * s.H, big_w=2 * s.W,
                                                                  ldx=v.ld, ldy=s.ld, dtype=self.code), 'dups%d' % v.block)
                    else:
                        tmp = alloc((N, s.H, s.W, s.c_phys))
                        add(bwd, plan['bwd_ops'], UpsampleBwdDesc(x=gptr(v), y=P(tmp), n=N, h=s.H, w_in=s.W, c=s.c_phys, big_h=2 * s.H, big_w=2 * s.W,
                                                                  ldx=v.ld, ldy=s.c_phys, dtype=self.code), 'dups%d' % v.block)
                        add(bwd, plan['bwd_ops'], AddDesc(a=gptr(s), b=P(tmp), y=gptr(s), pixels=N * s.H * s.W, c=s.c_phys,
                                                          lda=s.ld, ldb=s.c_phys, ldy=s.ld, dtype=self.code), 'dups%d' % v.block)
                continue
            if v.kind != 'conv':
                raise NotImplementedError('HIP training path: backward of %s' % v.kind)
            if not has_grad(v):
                continue                      # dead branch: its parameters get zero gradients
            s, pk = v.src, v.tpack
            pixels = N * v.Ho * v.Wo
            kz = next_dz()
            dzp, lddz = P(dz_bufs[kz]), v.c_phys
            dz_private = True                 # dz is the scratch buffer (False: dz is dy itself, which the main lane may still change)
            if v.fp32:                        # head: fp32 gradient from autograd -> dtype
                op = add(bwd, plan['bwd_ops'], CastDesc(x=None, y=dzp, pixels=pixels, c=v.c_phys, ldx=v.c_phys, ldy=v.c_phys,
                                                        dtype=self.code), 'dhead%d' % v.block)
                fixup(bwd, op, CastDesc, 'x', SLOT_HEAD0 + head_index[id(v)])
                dz_written_by(op, kz)
                if v.g_b is not None:         # bias gradient = sum over pixels of dz
                    add_reduction(bwd, plan['bwd_ops'],
                        BnBwdReduceDesc(z=dzp, dy=dzp, pixels=pixels, n=N, h=v.Ho, w_in=v.Wo, c=v.c_phys, ldz=v.c_phys,
                                        lddy=v.c_phys, act=LINEAR, slope=0.0, dtype=self.code, ups=1,
                                        sum=grads.ptr(v.g_b), sumsq=grads.ptr(self._junk(plan, grads, v.c_phys))),
                        'dbias%d' % v.block)
            else:
                dyp, lddy = gptr(v), v.ld
                if v.ups == 2:
                    small = alloc((N, v.Ho, v.Wo, v.c_phys))
                    add(bwd, plan['bwd_ops'], UpsampleBwdDesc(x=dyp, y=P(small), n=N, h=v.Ho, w_in=v.Wo, c=v.c_phys, big_h=2 * v.Ho, big_w=2 * v.Wo, ldx=lddy,
                                                              ldy=v.c_phys, dtype=self.code), 'dups%d' % v.block)
                    dyp, lddy = P(small), v.c_phys
                if v.res is not None:
                    t = v.res
                    own = lambda u: u.parent is None and u.c_off == 0 and u.ld == u.c_phys and holders.get(id(u.gstorage), 0) == 1
                    if (id(t.gstorage) not in initialised and v.ups == 1 and own(t) and own(v) and not t.fp32
                            and (t.H, t.W, t.c_phys) == (v.H, v.W, v.c_phys) and not getattr(t, 'galias', False)):
                        # first contribution to the residual source is dy itself: share the buffer instead of copying it.
                        # Later contributions (the data gradient of the source's other consumer) accumulate in place, after
                        # this block's backward has consumed dy.
                        plan['storages'] = [u for u in plan['storages'] if u is not t.gstorage]   # free the unused buffer
                        t.gstorage, t.galias = v.gstorage, True
                    else:
                        contribute(t, dyp, lddy, None, 'dres%d' % v.block)
                if v.plain:                   # no BN, linear: dz is dy itself
                    dzp, lddz = dyp, lddy
                    dz_private = False
                    if v.g_b is not None:
                        add_reduction(bwd, plan['bwd_ops'],
                            BnBwdReduceDesc(z=dyp, dy=dyp, pixels=pixels, n=N, h=v.Ho, w_in=v.Wo, c=v.c_phys, ldz=lddy,
                                            lddy=lddy, act=LINEAR, slope=0.0, dtype=self.code, ups=1, sum=grads.ptr(v.g_b),
                                            sumsq=grads.ptr(self._junk(plan, grads, v.c_phys))), 'dbias%d' % v.block)
                else:
                    base, bnp = v.bn_args
                    base = dict(base, ups=1)
                    if v.bn is not None:
                        acc = dict(bnp, sum=grads.ptr(v.g_beta), sumsq=grads.ptr(v.g_gamma))
                    else:                     # activation without BN: z already holds the bias
                        acc = dict(sum=grads.ptr(v.g_b if v.g_b is not None else self._junk(plan, grads, v.c_phys)),
                                   sumsq=grads.ptr(self._junk(plan, grads, v.c_phys)))
                    if s.kind == 'input' and self._stem_bwd_fused(v, s):
                        # first block: BatchNorm backward + weight gradient as ONE pass over dy and z (csrc/stem_bwd.hip) - its dz
                        # has no other reader (no data gradient into the image), so it is never written
                        sd = StemBwdDesc(x=None, dy=dyp, z=base['z'], gamma=bnp['gamma'], beta=bnp['beta'], mean=bnp['mean'],
                                         invstd=bnp['invstd'], dgamma=grads.ptr(v.g_gamma), dbeta=grads.ptr(v.g_beta),
                                         dw=grads.ptr(v.g_w), n=N, cin=s.C, h=s.H, w_in=s.W, cout=v.c_phys, lddy=lddy,
                                         ldz=v.c_phys, act=v.act, slope=v.slope)
                        for key, val in getattr(v, 'fused_dgrad', {}).items():
                            setattr(sd, key, val)
                        need = int(lib.yh_stem_bwd_workspace(C.byref(sd)))
                        if need <= 0:
                            raise RuntimeError('yh_stem_bwd_workspace rejected a geometry _stem_bwd_fused accepted')
                        sd.ws_floats = need
                        plan['ws_floats'] = max(plan['ws_floats'], need)
                        op = add(bwd, plan['bwd_ops'], sd, ('stembwd%d' if not getattr(v, 'fused_dgrad', None) else 'stembwd_dgrad%d') % v.block)
                        fixup(bwd, op, StemBwdDesc, 'ws', SLOT_WS)
                        fixup(bwd, op, StemBwdDesc, 'x', SLOT_INPUT)
                        continue
                    if not getattr(v, 'dbn_done', False):     # else: the launch that completed dy took the sums (see dgrad below)
                        add_reduction(bwd, plan['bwd_ops'], BnBwdReduceDesc(**base, **acc, dy=dyp, lddy=lddy), 'dbn%d' % v.block)
                    op = add(bwd, plan['bwd_ops'], BnBwdApplyDesc(**base, **acc, dy=dyp, lddy=lddy, out=dzp, ldo=v.c_phys),
                             'dbnx%d' % v.block)
                    dz_written_by(op, kz)
            # weight gradient
            if s.kind == 'input':
                # the image as NHWC dtype with 8 channels (3..7 zero): the first layer then uses the MFMA wgrad kernel
                img = alloc((N, s.H, s.W, ALIGN_C))
                op = add(bwd, plan['bwd_ops'], LayoutDesc(x=None, y=P(img), n=N, c=s.C, h=s.H, w_in=s.W, c_pad=ALIGN_C, ldy=ALIGN_C,
                                                          dtype=self.code), 'image%d' % v.block)
                fixup(bwd, op, LayoutDesc, 'x', SLOT_INPUT)
                side = two_lanes and dz_private and lane_ok(v)
                op = add_reduction(bwd, plan['bwd_ops'],
                                   WgradDesc(x=P(img), dz=dzp, dw=grads.ptr(v.g_w), n=N, h=s.H, w_in=s.W, cin=ALIGN_C, ho=v.Ho, wo=v.Wo,
                                             cout=v.C, kh=v.k, kw=v.k, stride=v.stride, pad=v.pad, ldx=ALIGN_C, lddz=lddz,
                                             dtype=self.code, splits=0, cin_w=s.C), 'wgrad%d' % v.block, side=side)
                if side:
                    on_side_lane(op, kz)
                continue
            side = two_lanes and dz_private and lane_ok(v)

            def emit_wgrad(v=v, s=s, dzp=dzp, lddz=lddz, kz=kz, side=side):
                op = add_reduction(bwd, plan['bwd_ops'],
                                   WgradDesc(x=P(s.storage, s.c_off), dz=dzp, dw=grads.ptr(v.g_w), n=N, h=s.H, w_in=s.W, cin=s.C, ho=v.Ho,
                                             wo=v.Wo, cout=v.C, kh=v.k, kw=v.k, stride=v.stride, pad=v.pad, ldx=s.ld, lddz=lddz,
                                             dtype=self.code, splits=0), 'wgrad%d' % v.block, side=side or async_reduce)
                if side:
                    on_side_lane(op, kz)
            if not side:
                emit_wgrad()
            # data gradient into grad(src).  With two lanes the weight gradient is issued AFTER it and waits for it: two MFMA-bound
            # kernels gain nothing from sharing the CUs, the pairing that pays is weight gradient x the next layer's BatchNorm passes
            mode = contribution_mode(s, True)
            common = dict(bias=P(zero_bias), n=N, cin=v.c_phys, cout=s.c_phys, stride=1, ldx=lddz, ldr=s.ld if mode == 'acc' else 0,
                          ldy=s.ld, cin_k=pk['cout_k'], m_pad=pk['dm_pad'], act=LINEAR, slope=0.0, out_f32=0, dtype=self.code,
                          tile=self.force_tile if self.force_tile < 40 else 0, acc_scale=0.0, out_scale=0.0)
            if pk.get('stem_fused'):
                # no launch: the first block's backward computes this data gradient from dz on the fly (yh_stem_bwd dz1 / w1) and
                # never materialises it.  dz stays valid until then: the first block writes no dz of its own.
                if mode != 'write':
                    raise RuntimeError('fused first-block data gradient must be the only contribution to its gradient')
                s.fused_dgrad = dict(dz1=dzp, lddz1=lddz, w1=P(pk['wt']), h1=v.Ho, w1_in=v.Wo, k1=v.c_phys, k1_pad=pk['cout_k'])
            elif v.stride == 2 and 'wt_fused' in pk:
                fused = dict(common, cout=4 * s.c_phys, m_pad=pk['fused_m_pad'])
                if zero_bias.numel() < pk['fused_m_pad']:
                    raise RuntimeError('zero bias row is shorter than the fused data-gradient image')
                add(bwd, plan['bwd_ops'],
                    ConvDesc(x=dzp, w=P(pk['wt_fused']), res=gptr(s) if mode == 'acc' else None, y=gptr(s), h=v.Ho, w_in=v.Wo,
                             ho=(s.H + 1) // 2, wo=(s.W + 1) // 2, kh=2, kw=2, pad=0, ups=4, y_h=s.H, y_w=s.W, **fused),
                    'dgrad%d' % v.block)
            elif v.stride == 2:
                # input pixels of parity (a, b) only see the taps r = a + pad - 2t, s = b + pad - 2u: four small
                # correlations of dz scattered onto every other pixel, 9 tap-GEMMs in total (a dilated pass runs 36)
                for (a, b), (th, tw), img in zip(((0, 0), (0, 1), (1, 0), (1, 1)), pk['phase_taps'], pk['wt_phase']):
                    hp, wp = (s.H - a + 1) // 2, (s.W - b + 1) // 2
                    if hp <= 0 or wp <= 0:
                        continue
                    add(bwd, plan['bwd_ops'],
                        ConvDesc(x=dzp, w=P(img), res=gptr(s) if mode == 'acc' else None, y=gptr(s), h=v.Ho, w_in=v.Wo, ho=hp, wo=wp,
                                 kh=th, kw=tw, pad=0, ups=3, y_h=s.H, y_w=s.W, y_off_h=a, y_off_w=b, **common),
                        'dgrad%d' % v.block)
            else:
                dg = ConvDesc(x=dzp, w=P(pk['wt']), res=gptr(s) if mode == 'acc' else None, y=gptr(s), h=v.Ho, w_in=v.Wo, ho=s.H,
                              wo=s.W, kh=v.k, kw=v.k, pad=v.k - 1 - v.pad, ups=1, **common)
                rows = 0
                if completes_bn_block(v, s):
                    sbase, sbnp = s.bn_args
                    for key, val in dict(bwd_z=sbase['z'], bwd_ldz=s.c_phys, bwd_act=s.act, bwd_slope=s.slope, bwd_gamma=sbnp['gamma'],
                                         bwd_beta=sbnp['beta'], bwd_mean=sbnp['mean'], bwd_invstd=sbnp['invstd']).items():
                        setattr(dg, key, val)
                    rows = int(lib.yh_conv2d_bwd_stats_rows(C.byref(dg)))
                    if rows <= 0:
                        for key in ('bwd_z', 'bwd_gamma', 'bwd_beta', 'bwd_mean', 'bwd_invstd'):
                            setattr(dg, key, None)
                if rows > 0:
                    dg.stats_ws_floats = rows * 2 * s.c_phys
                    plan['ws_floats'] = max(plan['ws_floats'], dg.stats_ws_floats)
                op = add(bwd, plan['bwd_ops'], dg, 'dgrad%d' % v.block)
                if rows > 0:
                    # the rows it leaves in the shared workspace are added into dbeta / dgamma of `s` right away (the workspace is
                    # free game for the next op); s's own backward then starts at its apply pass
                    fixup(bwd, op, ConvDesc, 'stats_ws', SLOT_WS)
                    sacc = dict(sbnp, sum=grads.ptr(s.g_beta), sumsq=grads.ptr(s.g_gamma))
                    add_reduction(bwd, plan['bwd_ops'],
                                  BnBwdReduceDesc(**dict(sbase, ups=1), **sacc, dy=gptr(s), lddy=s.ld, nparts=rows), 'dbn%d' % s.block)
                    s.dbn_done = True
            if side:
                emit_wgrad()
        plan['head_shapes'] = [(N, h.src.H, h.src.W, h.src.c_phys) for h in heads]
        plan['segments'] = self._make_segments(plan, values, heads, bwd_pos)
        plan['ws'] = alloc((max(plan['ws_floats'], 4),), fp32=True)
        plan['ws2'] = alloc((max(plan['ws2_floats'], 4),), fp32=True)
        if async_reduce:
            hiplib.check(lib.yh_plan_set_async_reduce(bwd, 1), 'yh_plan_set_async_reduce')
        for handle in (fwd, bwd):
            lib.yh_plan_bind_slot(handle, SLOT_WS, plan['ws'].data_ptr())
        lib.yh_plan_bind_slot(bwd, SLOT_WS2, plan['ws2'].data_ptr())
        return plan

    def _stem_bwd_fused(self, v, s):
        """The one-pass backward of the first block applies: fp16 step, 3x3 / stride 1 / pad 1 on a 1- or 3-channel image, 16 or
        32 output channels, BatchNorm + activation, nothing fused into its store (YOLO_HIP_STEM_BWD=0: the four-launch form)."""
        if os.environ.get('YOLO_HIP_STEM_BWD', '1') == '0':
            return False
        return (self.code == hiplib.YH_F16 and v.bn is not None and v.k == 3 and v.stride == 1 and v.pad == 1 and s.C in (1, 3)
                and v.c_phys in (16, 32) and v.C == v.c_phys and v.ups == 1 and v.res is None and v.conv.bias is None)

    def _dgrad_fused_into_stem(self, values, v):
        """conv `v` is the ONLY consumer of the first block, a 3x3 / stride 2 / pad 1 conv with 64 output channels on its 32
        channels, and the first block takes the one-pass backward: then v's data gradient is computed inside that pass
        (YOLO_HIP_STEM_DGRAD=0 keeps it a launch of its own)."""
        s = v.src
        if os.environ.get('YOLO_HIP_STEM_DGRAD', '1') == '0' or os.environ.get('YOLO_HIP_WGRAD_LANE', '0') == '1':
            return False
        if v.kind != 'conv' or s.kind != 'conv' or s.src.kind != 'input' or not self._stem_bwd_fused(s, s.src):
            return False
        if not (v.k == 3 and v.stride == 2 and v.pad == 1 and v.C == 64 and v.c_phys == 64 and s.c_phys == 32 and not v.fp32):
            return False
        if s.parent is not None or s.c_off != 0 or s.ld != s.c_phys:
            return False
        users = 0
        for u in values:
            if u is v:
                continue
            refs = [getattr(u, 'src', None), getattr(u, 'res', None), getattr(u, 'a', None), getattr(u, 'b', None)]
            refs += [part[0] for part in getattr(u, 'parts', [])] if u.kind == 'concat' else []
            users += any(r is s for r in refs)
        return users == 0

    @staticmethod
    def _junk(plan, grads, n):
        """Scratch accumulator for a reduction output nobody reads (lives past the parameter slices)."""
        junk = plan.get('junk')
        if junk is None or plan['junk_n'] < n:
            raise RuntimeError('junk accumulator was not reserved')
        return junk

    # --------------------------------------------------------------------------------- execute
    def _get_plan(self, x):
        if x.dim() != 4:
            raise ValueError('expected an (N, C, H, W) batch')
        if self.device is None:
            self.device = x.device
        elif self.device != x.device:
            raise RuntimeError('engine was built on %s, input is on %s' % (self.device, x.device))
        key = tuple(x.shape)
        plan = self._tplans.get(key)
        if plan is None:
            # a plan owns every activation and gradient buffer of a step: keep only the two most recent input shapes
            # (multi-scale training walks through a dozen sizes)
            while len(self._tplans) >= 2:
                old = self._tplans.pop(next(iter(self._tplans)))
                for h in ('fwd', 'bwd'):
                    self.lib.yh_plan_destroy(old[h])
                if self._current is old:
                    self._current = None
            plan = self._tplans[key] = self._build_train_plan(*key)
        else:
            self._tplans[key] = self._tplans.pop(key)   # most recently used last
        return plan

    def forward(self, x):
        """Run the training forward; returns the fp32 NHWC head tensors [(N, ny, nx, c_phys)] (fresh tensors)."""
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        plan = self._get_plan(x)
        self._refresh_pack_table(plan)   # the packing itself is op 0 of the forward plan (one launch)
        lib = self.lib
        plan['stats'].buf.zero_()
        heads = [torch.empty(shape, device=x.device, dtype=torch.float32) for shape in plan['head_shapes']]
        lib.yh_plan_bind_slot(plan['fwd'], SLOT_INPUT, x.data_ptr())
        for k, h in enumerate(heads):
            lib.yh_plan_bind_slot(plan['fwd'], SLOT_HEAD0 + k, h.data_ptr())
        hiplib.check(lib.yh_plan_run(plan['fwd'], hiplib.stream_ptr()), 'yh_plan_run(train forward)')
        counters = [v.bn.num_batches_tracked for v in plan['values']
                    if v.kind == 'conv' and v.bn is not None and v.bn.num_batches_tracked is not None]
        if counters:
            torch._foreach_add_(counters, 1)
        plan['x'] = x
        self._current = plan
        self.steps += 1
        return heads

    def _make_segments(self, plan, values, heads, bwd_pos):
        """Cut the backward plan into ranges of about equal parameter size (forward order of the conv blocks).

        Each range is one autograd node in ``models.Darknet``: its parameter gradients are handed to autograd (and to
        DistributedDataParallel's bucket hooks) as soon as the range has run, so the RCCL all-reduce of a bucket overlaps
        with the backward kernels of the earlier layers instead of starting after the whole backward plan."""
        want = max(1, int(os.environ.get('YOLO_HIP_TRAIN_SEGMENTS', '8')))
        convs = [i for i, v in enumerate(values) if v.kind == 'conv']
        total = sum(values[i].conv.weight.numel() for i in convs)
        cuts, acc, target = [1], 0, total / want           # value index where each segment starts (value 0 is the input)
        for i in convs:
            if acc >= target * len(cuts) and i > cuts[-1]:
                cuts.append(i)
            acc += values[i].conv.weight.numel()
        cuts.append(len(values))
        nparams = len(plan['param_slices'])
        segs = []
        for k in range(len(cuts) - 1):
            lo, hi = cuts[k], cuts[k + 1]
            cv = [values[i] for i in range(lo, hi) if hasattr(values[i], 'p_first')]
            p_lo = cv[0].p_first if cv else nparams
            nxt = [values[i] for i in range(hi, len(values)) if hasattr(values[i], 'p_first')]
            p_hi = nxt[0].p_first if nxt else nparams
            segs.append(dict(values=(lo, hi), ops=(bwd_pos[id(values[hi - 1])], bwd_pos[id(values[lo - 1])]), params=(p_lo, p_hi),
                             heads=[j for j, h in enumerate(heads) if any(h.src is values[i] for i in range(lo, hi))]))
        assert segs[0]['ops'][1] == len(plan['bwd_ops']) and segs[-1]['ops'][0] == 0
        return segs

    def backward_segment(self, k, head_grads):
        """Run backward range ``k`` (the last range first).  ``head_grads``: fp32 gradients of this range's heads, in the
        order of ``plan['segments'][k]['heads']``.  Returns the fp32 gradients of the range's parameters."""
        plan = self._current
        if plan is None:
            raise RuntimeError('backward() without a preceding forward()')
        lib, x = self.lib, plan['x']
        seg, nseg = plan['segments'][k], len(plan['segments'])
        if k == nseg - 1:
            plan['grads'].buf.zero_()
            for t in plan['zero_list']:
                t.zero_()
            lib.yh_plan_bind_slot(plan['bwd'], SLOT_INPUT, x.data_ptr())
            plan['bwd_next'] = nseg - 1
            plan['bwd_keep'] = []
        if plan.get('bwd_next') != k:
            raise RuntimeError('backward ranges must run last to first (expected %s, got %d)' % (plan.get('bwd_next'), k))
        for j, g in zip(seg['heads'], head_grads):
            shape = plan['head_shapes'][j]
            if g is None:
                g = torch.zeros(shape, device=x.device, dtype=torch.float32)
            g = g.contiguous().float()
            if tuple(g.shape) != tuple(shape):
                raise ValueError('head gradient %d has shape %s, expected %s' % (j, tuple(g.shape), shape))
            plan['bwd_keep'].append(g)
            lib.yh_plan_bind_slot(plan['bwd'], SLOT_HEAD0 + j, g.data_ptr())
        first, last = seg['ops']
        if last > first:
            hiplib.check(lib.yh_plan_run_range(plan['bwd'], first, last, hiplib.stream_ptr()), 'yh_plan_run_range(train backward)')
        plan['bwd_next'] = k - 1
        p_lo, p_hi = seg['params']
        if p_hi == p_lo:
            return []
        off_lo = plan['param_slices'][p_lo][0]
        off_hi = plan['param_slices'][p_hi - 1][0] + _round_up(plan['param_slices'][p_hi - 1][1], 4)
        flat = plan['grads'].buf[off_lo:off_hi].clone()   # autograd may keep (and later accumulate into) what we return
        return [flat[off - off_lo:off - off_lo + n].view(shape) for off, n, shape in plan['param_slices'][p_lo:p_hi]]

    def backward(self, head_grads):
        """All ranges at once: fp32 head gradients (``forward``'s order) -> list of fp32 parameter gradients."""
        plan = self._current
        if plan is None:
            raise RuntimeError('backward() without a preceding forward()')
        out = [None] * len(plan['segments'])
        for k in reversed(range(len(plan['segments']))):
            out[k] = self.backward_segment(k, [head_grads[j] for j in plan['segments'][k]['heads']])
        return [g for seg in out for g in seg]

    def segment_parameters(self, plan):
        """The parameter list split per backward range (same order as ``parameters()``)."""
        params = self.parameters()
        return [params[seg['params'][0]:seg['params'][1]] for seg in plan['segments']]

    def _drop_plans(self):
        super()._drop_plans()
        for plan in getattr(self, '_tplans', {}).values():
            for key in ('fwd', 'bwd'):
                try:
                    self.lib.yh_plan_destroy(plan[key])
                except Exception:
                    pass
        self._tplans = {}
