"""Lowering of a ``models.Darknet`` module to a replayable launch plan on ``libyolo_hip.so``.

The reference executes the graph with a per-layer Python loop over ``nn.Module`` s
(models.py:524-545) on NCHW tensors.  Here the same cfg graph is *compiled* once per input shape:

1. **Values.**  Every block output that must exist in memory becomes a ``Value`` (NHWC, dtype fp16 or
   fp32, channel pitch ``ld``).  Blocks that only re-label data produce no value: a single-layer
   ``route`` is an alias, a ``shortcut`` following a conv is folded into that conv's epilogue as a
   residual operand, an ``upsample`` following a conv makes the conv write each pixel to its 2x2
   block, and a ``yolo`` block becomes a decode op writing straight into the concatenated output.
2. **Placement.**  A multi-input ``route`` owns one wide buffer; each input that is not already placed
   elsewhere is *produced directly into its channel slice* (zero-copy concat); consumers read slices
   through the pitch.  Channel counts that are not multiples of 8 are padded per segment and the
   consumer's packed weights get zero columns at the pad positions (``cin_map``), so pruned models
   with arbitrary widths lower the same way.
3. **Weights.**  BN is folded and weights are packed K-major by ``yh_conv_pack_weights`` on the device,
   straight from the live ``nn.Parameter`` storage.  The cache is keyed on every source tensor's
   ``(data_ptr, _version, shape)``; ``Darknet`` also drops the engine on ``load_state_dict`` /
   ``load_darknet_weights`` / ``fuse`` / ``.to()`` / ``train()``.
4. **Replay.**  ``yh_plan_run`` launches the recorded ops from native code; per-call pointers (input
   frames, output tensors) are slots patched at launch.

Nothing here computes activations with torch: torch supplies device memory and the stream.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import hiplib
from .hiplib import (ConvDesc, StemDesc, PoolDesc, CopyDesc, AddDesc, DecodeDesc, DwDesc, SeDesc, QCopyDesc, QPoolDesc,
                     QAddDesc)

SLOT_INPUT, SLOT_IO, SLOT_RAW0 = 0, 1, 2
ALIGN_C = 8  # physical channel granularity of every NHWC buffer (16 bytes of fp16)


def _round_up(a, b):
    return (a + b - 1) // b * b


class Value:
    """A tensor that exists in device memory during the forward."""

    ALIGN = ALIGN_C  # physical channel granularity; the engine raises it to 16 for int8 tensors

    def __init__(self, kind, C, H, W, block=None, **kw):
        self.kind = kind  # input | conv | pool | copy | add | concat | slice
        self.C, self.H, self.W = C, H, W
        self.block = block
        self.fp32 = False  # head convs store fp32 regardless of the engine precision
        self.segs = [(0, C)]  # (physical start relative to c_off, logical length)
        self.c_phys = _round_up(C, Value.ALIGN)
        self.scale = None  # int8 engines: real value = grid value * scale (power of two)
        self.parent = None  # concat value this one is produced into
        self.storage = None
        self.c_off = 0
        self.ld = None
        self.__dict__.update(kw)

    def channel_map(self):
        m = []
        for start, length in self.segs:
            m.extend(range(start, start + length))
        return m

    def is_dense(self):
        """Logical channel c sits at physical channel c (a concat of unpadded pieces counts)."""
        at = 0
        for start, length in self.segs:
            if start != at:
                return False
            at += length
        return at == self.C


class Head:
    def __init__(self, block, src, module):
        self.block, self.src = block, src
        self.na, self.no, self.nc = module.na, module.no, module.nc
        self.stride = float(module.stride)
        self.anchor_vec = module.anchor_vec.detach().float().cpu().numpy()  # anchors / stride, fp32 (models.py:362)


def _activation_of(block):
    """(code, slope) of a conv block's activation child, by class name like the reference's scripts."""
    for child in block.children():
        name = child.__class__.__name__
        if name == 'LeakyReLU':
            return hiplib.ACT_CODES['leaky'], float(child.negative_slope)
        if name == 'ReLU6':
            return hiplib.ACT_CODES['relu6'], 0.0
        if name == 'HardSwish':
            return hiplib.ACT_CODES['h_swish'], 0.0
        if name == 'ReLU':
            return hiplib.ACT_CODES['relu'], 0.0
        if name == 'Mish':
            return hiplib.ACT_CODES['mish'], 0.0
        if name in ('Swish', 'PReLU', 'Sigmoid'):
            raise NotImplementedError('HIP engine: activation %s is not lowered' % name)
    return hiplib.ACT_CODES['linear'], 0.0


_ACT_BY_NAME = {'leaky': 'leaky', 'relu6': 'relu6', 'h_swish': 'h_swish', 'relu': 'relu', 'mish': 'mish', 'linear': 'linear'}


def _quant_conv_info(conv):
    """(act code, slope, s_w, s_a) of a calibrated BNFold_COSPTQuantizedConv2d_For_FPGA (duck-typed)."""
    if not getattr(conv, 'quantized', False):
        raise RuntimeError('HIP int8 engine: quantised conv has not been calibrated (run the PTQ calibration first)')
    name = _ACT_BY_NAME.get(conv.activate)
    if name is None:
        raise NotImplementedError('HIP int8 engine: activation %r' % conv.activate)
    slope = 0.25 if getattr(conv, 'maxabsscaler', False) else 0.1
    return hiplib.ACT_CODES[name], slope, float(conv.weight_quantizer.scale), float(conv.activation_quantizer.scale)


def _conv_parts(block):
    kids = list(block.children())
    conv = kids[0]
    if not isinstance(conv, nn.Conv2d):
        raise NotImplementedError('HIP engine: block starts with %s, not nn.Conv2d' % conv.__class__.__name__)
    bn = None
    for k in kids[1:]:
        if isinstance(k, nn.modules.batchnorm.BatchNorm2d):
            bn = k
    return conv, bn


class DarknetEngine:
    def __init__(self, model, precision='fp16', lib=None):
        if precision not in ('fp16', 'fp32', 'int8'):
            raise ValueError("precision must be 'fp16', 'fp32' or 'int8'")
        # ``lib`` is only ever passed by the CPU test tier (tests/fakelib.py emulates the C ABI on host
        # memory to validate this lowering); the product path always loads the real library or raises.
        self.lib = hiplib.load() if lib is None else lib
        self.model = model
        self.precision = precision
        self.code = {'fp16': hiplib.YH_F16, 'fp32': hiplib.YH_F32, 'int8': hiplib.YH_I8}[precision]
        self.dtype = {'fp16': torch.float16, 'fp32': torch.float32, 'int8': torch.int8}[precision]
        self.kstep = {'fp16': 32, 'fp32': 16, 'int8': 64}[precision]
        self.align = 16 if precision == 'int8' else ALIGN_C
        self.q = precision == 'int8'  # PTQ eval graph: modules are the reference's COSPTQuantized* classes
        self.want_raw = True
        self.return_features = False
        self.force_tile = int(os.environ.get('YOLO_HIP_TILE', '0'))  # A/B profiling knob; 0 = library heuristic
        # batches up to this size replay through a captured hipGraph (one launch instead of ~100).  Off by default:
        # with the native plan executor the frame-at-a-time forward is GPU-bound (1.63 ms at batch 1, YOLOv3-608),
        # and the graph's staged input / cloned outputs make it ~8 % slower (profiles/r01_latency_*.txt).
        self.graph_max_batch = int(os.environ.get('YOLO_HIP_GRAPH_BATCH', '0'))
        self.max_plans = max(1, int(os.environ.get('YOLO_HIP_MAX_PLANS', '6')))   # input shapes kept resident
        self._plans = {}
        self._plans_with_features = False
        self._packed = {}  # block index -> dict(w=, b=, ...)
        self._signature = None
        self.device = None

    # ------------------------------------------------------------------------------------ graph
    def _build_values(self, N, Cin, H, W):
        model = self.model
        defs, mods, routs = model.module_defs, model.module_list, model.routs
        L = len(mods)
        Value.ALIGN = self.align
        x0 = Value('input', Cin, H, W)
        values, heads = [x0], []
        outs = [None] * L
        cur = x0
        skip = set()

        def ref(i, l):
            j = i + l if l < 0 else l
            v = outs[j]
            if v is None:
                raise NotImplementedError('HIP engine: block %d reads block %d, which has no tensor' % (i, j))
            return v

        for i, (mdef, module) in enumerate(zip(defs, mods)):
            kind = mdef['type']
            cname = module.__class__.__name__
            if i in skip:
                outs[i] = cur
                continue
            hidden = False  # True when this block's own output is fused away into its follower
            if kind in ('convolutional', 'depthwise'):
                conv, bn = _conv_parts(module)
                k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
                if conv.groups != 1:
                    if not (conv.groups == conv.in_channels == conv.out_channels == cur.C) or cur.kind == 'input':
                        raise NotImplementedError('HIP engine: grouped conv with groups != channels (block %d)' % i)
                    Ho, Wo = (cur.H + 2 * p - k) // s + 1, (cur.W + 2 * p - k) // s + 1
                    act, slope = _activation_of(module)
                    v = Value('dw', cur.C, Ho, Wo, block=i, src=cur, conv=conv, bn=bn, k=k, stride=s, pad=p, act=act,
                              slope=slope)
                    v.segs, v.c_phys = list(cur.segs), cur.c_phys
                    values.append(v)
                    cur = v
                    outs[i] = cur
                    continue
                if conv.kernel_size[0] != conv.kernel_size[1] or conv.in_channels != cur.C:
                    raise NotImplementedError('HIP engine: unsupported conv geometry at block %d' % i)
                Ho, Wo = (cur.H + 2 * p - k) // s + 1, (cur.W + 2 * p - k) // s + 1
                if self.q:
                    if conv.__class__.__name__ != 'BNFold_COSPTQuantizedConv2d_For_FPGA':
                        raise NotImplementedError('HIP int8 engine: block %d is not a COS-PTQ quantised conv' % i)
                    act, slope, s_w, s_a = _quant_conv_info(conv)
                else:
                    act, slope = _activation_of(module)
                v = Value('conv', conv.out_channels, Ho, Wo, block=i, src=cur, conv=conv, bn=bn, k=k, stride=s, pad=p,
                          act=act, slope=slope, res=None, ups=1)
                if self.q:
                    v.s_w, v.scale = s_w, s_a
                nxt = defs[i + 1]['type'] if i + 1 < L else None
                keep = self.return_features   # feature_out needs every conv block's own output: no epilogue fusion
                if nxt == 'shortcut' and not routs[i] and not keep and self._fusable_shortcut(i + 1, v, outs):
                    v.res = ref(i + 1, mods[i + 1].layers[0])
                    skip.add(i + 1)
                    hidden = True
                    if self.q:   # quantised shortcut in the conv epilogue: the value leaves on the SUM grid
                        sc = mods[i + 1]
                        v.qadd = (float(sc.scale_x), float(sc.scale_a), float(sc.scale_sum))
                        v.act_scale, v.scale = v.scale, float(sc.scale_sum)
                elif nxt == 'upsample' and not routs[i] and not keep and int(defs[i + 1]['stride']) == 2:
                    v.ups = 2
                    v.H, v.W = 2 * Ho, 2 * Wo
                    v.Ho, v.Wo = Ho, Wo
                    skip.add(i + 1)
                    hidden = True
                elif nxt == 'yolo':
                    v.fp32 = True
                if v.ups == 1:
                    v.Ho, v.Wo = Ho, Wo
                values.append(v)
                cur = v
            elif kind == 'maxpool':
                k, s = int(mdef['size']), int(mdef['stride'])
                if k == 2 and s == 1:
                    Ho, Wo, pad_lo, edge_zero = cur.H, cur.W, 0, 1
                else:
                    pad_lo, edge_zero = (k - 1) // 2, 0
                    Ho, Wo = (cur.H + 2 * pad_lo - k) // s + 1, (cur.W + 2 * pad_lo - k) // s + 1
                v = Value('pool', cur.C, Ho, Wo, block=i, src=cur, k=k, stride=s, pad_lo=pad_lo, edge_zero=edge_zero)
                v.segs, v.c_phys, v.scale = list(cur.segs), cur.c_phys, cur.scale
                values.append(v)
                cur = v
            elif kind == 'se':
                se = module[0] if isinstance(module, nn.Sequential) else module
                fc1, fc2 = se.fc[0], se.fc[2]
                if fc1.in_features != cur.C or fc2.out_features != cur.C or fc1.bias is not None or fc2.bias is not None:
                    raise NotImplementedError('HIP engine: unsupported SE geometry at block %d' % i)
                v = Value('se', cur.C, cur.H, cur.W, block=i, src=cur, fc1=fc1, fc2=fc2)
                v.segs, v.c_phys = list(cur.segs), cur.c_phys
                values.append(v)
                cur = v
            elif kind == 'upsample':
                s = int(mdef['stride'])
                if s != 2:
                    raise NotImplementedError('HIP engine: upsample stride %d' % s)
                v = Value('copy', cur.C, cur.H * 2, cur.W * 2, block=i, src=cur, ups=2)
                v.segs, v.c_phys, v.scale = list(cur.segs), cur.c_phys, cur.scale
                values.append(v)
                cur = v
            elif kind == 'route':
                layers = module.layers
                if len(layers) > 1:
                    srcs = [ref(i, l) for l in layers]
                    if any((s.H, s.W) != (srcs[0].H, srcs[0].W) for s in srcs):
                        raise ValueError('route %d joins tensors of different spatial size' % i)
                    v = Value('concat', sum(s.C for s in srcs), srcs[0].H, srcs[0].W, block=i, srcs=srcs)
                    # physical layout is fixed here (later pool/copy/add values inherit it): every source
                    # keeps its own padded span, logical channels map through the per-source segments
                    v.segs, v.offsets, off = [], [], 0
                    for s in srcs:
                        v.offsets.append(off)
                        v.segs.extend((off + st, ln) for st, ln in s.segs)
                        off += s.c_phys
                    v.c_phys = off
                    if self.q:  # COSPTQuantizedFeatureConcat: every input is re-quantised to one shared scale
                        v.scale = float(module.scale)
                        if not v.scale > 0:
                            raise RuntimeError('HIP int8 engine: route %d has no calibrated scale' % i)
                    values.append(v)
                    cur = v
                    if self.q:
                        # ... and the reference re-quantises the CACHED outputs in place (quantized_ptq_cos.py:1531-1545:
                        # `outputs[layer] = quantize(outputs[layer])`), so whatever reads block `layer` through `outputs` after
                        # this route - a later route or shortcut - sees it on THIS concat's grid (yolov4-tiny: block 23 feeds
                        # the routes 24 and 34).  Later references resolve to the re-quantised slice of the concat buffer.
                        for l, s_, off_ in zip(layers, srcs, v.offsets):
                            if s_.scale != v.scale:
                                j = i + l if l < 0 else l
                                sl = Value('slice', s_.C, s_.H, s_.W, block=j, src=v, first=off_)
                                sl.segs, sl.c_phys, sl.scale = list(s_.segs), s_.c_phys, v.scale
                                values.append(sl)
                                outs[j] = sl
                elif getattr(module, 'groups', False):
                    half = cur.C // 2
                    if not cur.is_dense() or half % self.align or (cur.C - half) % self.align:
                        raise NotImplementedError('HIP engine: unaligned group split at block %d' % i)
                    v = Value('slice', cur.C - half, cur.H, cur.W, block=i, src=cur, first=half)
                    v.scale = cur.scale
                    values.append(v)
                    cur = v
                else:
                    cur = ref(i, layers[0])
            elif kind == 'shortcut' and self.q:
                if cname not in ('COSPTQuantizedShortcut_min', 'COSPTQuantizedShortcut_max') or getattr(module, 'weight', False) \
                        or len(module.layers) != 1:
                    raise NotImplementedError('HIP int8 engine: shortcut form at block %d' % i)
                other = ref(i, module.layers[0])
                if other.C != cur.C or not cur.is_dense() or not other.is_dense():
                    raise NotImplementedError('HIP int8 engine: channel-mismatched shortcut (block %d)' % i)
                v = Value('qadd', cur.C, cur.H, cur.W, block=i, a=cur, b=other, scale_x=float(module.scale_x),
                          scale_a=float(module.scale_a))
                v.scale = float(module.scale_sum)
                values.append(v)
                cur = v
            elif kind == 'shortcut':
                if getattr(module, 'weight', False):
                    raise NotImplementedError('HIP engine: weighted shortcut (block %d)' % i)
                for l in module.layers:
                    other = ref(i, l)
                    if (other.H, other.W) != (cur.H, cur.W) or cur.fp32 or other.fp32:
                        raise NotImplementedError('HIP engine: shortcut over mismatched maps (block %d)' % i)
                    v = Value('add', cur.C, cur.H, cur.W, block=i, a=cur, b=other)
                    if other.C != cur.C or not cur.is_dense() or not other.is_dense():
                        # operands of different width sum over the leading min(Ca, Cb) channels (layers.py:65-70);
                        # concats of padded pieces are gathered through their channel maps
                        ma, mb = cur.channel_map(), other.channel_map()
                        v.amap = ma + [-1] * (v.c_phys - len(ma))
                        v.bmap = [mb[k] if k < len(mb) else -1 for k in range(cur.C)] + [-1] * (v.c_phys - cur.C)
                    values.append(v)
                    cur = v
            elif kind == 'yolo':
                if not getattr(cur, 'fp32', False):
                    raise NotImplementedError('HIP engine: yolo block %d is not fed by a conv block' % i)
                heads.append(Head(i, cur, module))
            elif kind == 'reorg3d':
                pass
            else:
                raise NotImplementedError('HIP engine: block type %r (block %d) is not lowered' % (kind, i))
            outs[i] = None if hidden else cur
        if not heads:
            raise ValueError('cfg has no yolo block')
        return values, heads, outs

    def _fusable_shortcut(self, j, conv_value, outs):
        module = self.model.module_list[j]
        names = ('COSPTQuantizedShortcut_min', 'COSPTQuantizedShortcut_max') if self.q else ('Shortcut',)
        if module.__class__.__name__ not in names or getattr(module, 'weight', False) or len(module.layers) != 1:
            return False
        l = module.layers[0]
        other = outs[j + l if l < 0 else l]
        ok = (other is not None and other.C == conv_value.C and (other.H, other.W) == (conv_value.H, conv_value.W)
              and other.is_dense())
        if ok and self.q:
            # the int8 epilogue carries the shortcut for the compile-time activations only; the routed tensor must be on a grid
            ok = (conv_value.act in (hiplib.ACT_CODES['linear'], hiplib.ACT_CODES['leaky'], hiplib.ACT_CODES['mish'])
                  and other.scale is not None and conv_value.C % 16 == 0 and os.environ.get('YOLO_HIP_FUSE_QADD', '1') != '0')
        return ok

    def _place(self, values):
        """Decide which values are produced directly into a concat buffer's channel slice."""
        for v in values:
            if v.kind != 'concat':
                continue
            seen = set()
            v.parts = []
            for s, off in zip(v.srcs, v.offsets):
                if s.kind == 'input':
                    raise NotImplementedError('HIP engine: route over the network input')
                inplace = (s.kind in ('conv', 'pool', 'copy', 'add', 'dw', 'se', 'qadd') and s.parent is None
                           and id(s) not in seen and not s.fp32 and s.storage is None
                           and (not self.q or s.scale == v.scale))  # int8: zero-copy only when no re-quantisation is due
                if inplace:
                    s.parent, s.parent_off = v, off
                seen.add(id(s))
                v.parts.append((s, off, inplace))

    # ---------------------------------------------------------------------------------- weights
    def _source_tensors(self):
        out = []
        for block in self.model.module_list:
            if self.q and isinstance(block, nn.Sequential) and len(block) and hasattr(block[0], 'q_weight'):
                c = block[0]
                out.extend((c.q_weight, c.q_bias, c.weight_quantizer.scale, c.activation_quantizer.scale))
            elif self.q and hasattr(block, 'scale_sum'):
                out.extend((block.scale_x, block.scale_a, block.scale_sum))
            elif self.q and hasattr(block, 'float_max_list'):
                out.append(block.scale)
            elif isinstance(block, nn.Sequential) and len(block) and isinstance(block[0], nn.Conv2d):
                conv, bn = _conv_parts(block)
                out.extend(t for t in (conv.weight, conv.bias) if t is not None)
                if bn is not None:
                    out.extend((bn.weight, bn.bias, bn.running_mean, bn.running_var))
            elif isinstance(block, nn.Sequential) and len(block) and block[0].__class__.__name__ == 'SE':
                out.extend((block[0].fc[0].weight, block[0].fc[2].weight))
        return out

    def _current_signature(self):
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in self._source_tensors())

    def _pack_conv(self, v):
        """(Re)build the packed weight image of conv value ``v``; buffers are reused when shapes allow."""
        conv, bn = v.conv, v.bn
        dev = conv.weight.device
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        if self.q:
            return self._pack_qconv(v)
        w = f32(conv.weight)
        cb = f32(conv.bias)
        g, be, mu, var = (f32(bn.weight), f32(bn.bias), f32(bn.running_mean), f32(bn.running_var)) if bn is not None \
            else (None, None, None, None)
        eps = float(bn.eps) if bn is not None else 0.0
        keep = [w, cb, g, be, mu, var]
        P = hiplib.ptr
        slot = self._packed.get(v.block)
        if v.src.kind == 'input':
            cout_pad = _round_up(v.C, 32) if v.C % 32 == 0 else _round_up(v.C, 16)
            taps = v.k * v.k
            if slot is None or slot['w'].numel() != taps * v.src.C * cout_pad:
                slot = dict(w=torch.empty(taps * v.src.C * cout_pad, device=dev, dtype=torch.float32),
                            b=torch.empty(cout_pad, device=dev, dtype=torch.float32), cout_pad=cout_pad)
            rc = self.lib.yh_stem_pack_weights(P(w), P(cb), P(g), P(be), P(mu), P(var), eps, v.C, v.src.C, v.k, v.k,
                                               cout_pad, P(slot['w']), P(slot['b']), hiplib.stream_ptr())
            hiplib.check(rc, 'yh_stem_pack_weights')
        else:
            cin_k = _round_up(v.src.c_phys, self.kstep)
            m_pad = _round_up(_round_up(v.C, ALIGN_C), 128)
            taps = v.k * v.k
            cmap = None
            if not v.src.is_dense():
                cmap = torch.tensor(v.src.channel_map(), dtype=torch.int32, device=dev)
                keep.append(cmap)
            if slot is None or slot['w'].numel() != m_pad * taps * cin_k or slot['w'].dtype != self.dtype:
                slot = dict(w=torch.empty(m_pad * taps * cin_k, device=dev, dtype=self.dtype),
                            b=torch.empty(m_pad, device=dev, dtype=torch.float32), cin_k=cin_k, m_pad=m_pad)
            rc = self.lib.yh_conv_pack_weights(self.code, P(w), P(cb), P(g), P(be), P(mu), P(var), eps, P(cmap), v.C,
                                               conv.in_channels, v.k, v.k, cin_k, m_pad, P(slot['w']), P(slot['b']),
                                               hiplib.stream_ptr())
            hiplib.check(rc, 'yh_conv_pack_weights')
        slot['keep'] = keep  # fp32 staging copies must outlive the async pack kernels
        self._packed[v.block] = slot
        return slot

    def _pack_qconv(self, v):
        """int8 image of a calibrated COS-PTQ conv: grid weights from q_weight / s_w, bias = q_bias in real units."""
        conv = v.conv
        dev = conv.q_weight.device
        qw = conv.q_weight.detach().float().contiguous()
        qb = conv.q_bias.detach().float().contiguous()
        P = hiplib.ptr
        slot = self._packed.get(v.block)
        if v.src.kind == 'input':  # float frames in: the stem kernel multiplies real-valued (de-quantised) weights
            cout_pad = v.C if v.C % 32 == 0 else _round_up(v.C, 16)
            taps = v.k * v.k
            if slot is None or slot['w'].numel() != taps * v.src.C * cout_pad:
                slot = dict(w=torch.empty(taps * v.src.C * cout_pad, device=dev, dtype=torch.float32),
                            b=torch.empty(cout_pad, device=dev, dtype=torch.float32), cout_pad=cout_pad)
            rc = self.lib.yh_stem_pack_weights(P(qw), P(qb), None, None, None, None, 0.0, v.C, v.src.C, v.k, v.k, cout_pad,
                                               P(slot['w']), P(slot['b']), hiplib.stream_ptr())
            hiplib.check(rc, 'yh_stem_pack_weights')
        else:
            cin_k = _round_up(v.src.c_phys, self.kstep)
            m_pad = _round_up(_round_up(v.C, self.align), 128)
            taps = v.k * v.k
            cmap = None
            if not v.src.is_dense():
                cmap = torch.tensor(v.src.channel_map(), dtype=torch.int32, device=dev)
            if slot is None or slot['w'].numel() != m_pad * taps * cin_k or slot['w'].dtype != torch.int8:
                slot = dict(w=torch.empty(m_pad * taps * cin_k, device=dev, dtype=torch.int8),
                            b=torch.zeros(m_pad, device=dev, dtype=torch.float32), cin_k=cin_k, m_pad=m_pad)
            rc = self.lib.yh_qconv_pack_weights(P(qw), v.s_w, P(cmap), v.C, conv.in_channels, v.k, v.k, cin_k, m_pad, P(slot['w']),
                                                hiplib.stream_ptr())
            hiplib.check(rc, 'yh_qconv_pack_weights')
            slot['b'].zero_()
            slot['b'][:v.C].copy_(qb)
            slot['cmap'] = cmap
        slot['keep'] = [qw, qb]
        self._packed[v.block] = slot
        return slot

    def _pack_dw(self, v):
        conv, bn = v.conv, v.bn
        dev = conv.weight.device
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        w, cb = f32(conv.weight), f32(conv.bias)
        g, be, mu, var = (f32(bn.weight), f32(bn.bias), f32(bn.running_mean), f32(bn.running_var)) if bn is not None \
            else (None, None, None, None)
        eps = float(bn.eps) if bn is not None else 0.0
        keep = [w, cb, g, be, mu, var]
        cmap = None
        if not v.src.is_dense():
            cmap = torch.tensor(v.src.channel_map(), dtype=torch.int32, device=dev)
            keep.append(cmap)
        P = hiplib.ptr
        slot = self._packed.get(v.block)
        taps = v.k * v.k
        if slot is None or slot['w'].numel() != taps * v.c_phys or slot['w'].dtype != self.dtype:
            slot = dict(w=torch.empty(taps * v.c_phys, device=dev, dtype=self.dtype),
                        b=torch.empty(v.c_phys, device=dev, dtype=torch.float32))
        rc = self.lib.yh_dw_pack_weights(self.code, P(w), P(cb), P(g), P(be), P(mu), P(var), eps, P(cmap), v.C, v.k, v.c_phys,
                                         P(slot['w']), P(slot['b']), hiplib.stream_ptr())
        hiplib.check(rc, 'yh_dw_pack_weights')
        slot['keep'] = keep
        self._packed[v.block] = slot
        return slot

    def _pack_se(self, v, N):
        dev = v.fc1.weight.device
        slot = self._packed.get(v.block)
        w1 = v.fc1.weight.detach().float().contiguous()
        w2 = v.fc2.weight.detach().float().contiguous()
        if slot is None or slot['w1'].shape != w1.shape:
            slot = dict(w1=torch.empty_like(w1), w2=torch.empty_like(w2), cmap=None)
            if not v.src.is_dense():
                slot['cmap'] = torch.tensor(v.src.channel_map(), dtype=torch.int32, device=dev)
        slot['w1'].copy_(w1)
        slot['w2'].copy_(w2)
        self._packed[v.block] = slot
        return slot

    # ------------------------------------------------------------------------------------- plan
    def _alloc(self, N, H, W, C, fp32=False):
        return torch.empty((N, H, W, C), device=self.device, dtype=torch.float32 if fp32 else self.dtype)

    def _build_plan(self, N, Cin, H, W):
        values, heads, outs = self._build_values(N, Cin, H, W)
        self._place(values)
        lib = self.lib
        handle = lib.yh_plan_create()
        if not handle:
            raise MemoryError('yh_plan_create failed')
        plan = dict(handle=handle, values=values, heads=heads, outs=outs, storages=[], N=N, ops=[])
        P = hiplib.ptr

        def add(desc, what):
            idx = lib.yh_plan_add(handle, hiplib.OP_KIND[type(desc)], C.byref(desc), C.sizeof(desc))
            if idx < 0:
                hiplib.check(idx, 'yh_plan_add(%s)' % what)
            plan['ops'].append((what, desc))
            return idx

        def fixup(op, desc_type, field, slot, byte_off=0):
            hiplib.check(lib.yh_plan_add_fixup(handle, op, getattr(desc_type, field).offset, slot, byte_off), 'fixup')

        def materialize(v):
            """Give ``v`` an address: its own buffer, or a slice of its parent concat buffer."""
            if v.storage is not None:
                return
            if v.parent is not None:
                materialize(v.parent)
                v.storage, v.c_off, v.ld = v.parent.storage, v.parent.c_off + v.parent_off, v.parent.ld
            else:
                v.storage = self._alloc(N, v.H, v.W, v.c_phys, v.fp32)
                v.c_off, v.ld = 0, v.c_phys
                plan['storages'].append(v.storage)

        for v in values:
            if v.kind == 'input':
                continue
            if v.kind == 'slice':
                v.storage, v.c_off, v.ld = v.src.storage, v.src.c_off + v.first, v.src.ld
                continue
            materialize(v)
            y = P(v.storage, v.c_off)
            if v.kind == 'conv':
                pk = self._packed.get(v.block) or self._pack_conv(v)
                if v.src.kind == 'input':
                    d = StemDesc(x=None, w=P(pk['w']), bias=P(pk['b']), y=y, n=N, cin=v.src.C, h=v.src.H, w_in=v.src.W,
                                 ho=v.Ho, wo=v.Wo, cout=v.c_phys, cout_pad=pk['cout_pad'], kh=v.k, kw=v.k,
                                 stride=v.stride, pad=v.pad, ldy=v.ld, act=v.act, slope=v.slope, dtype=self.code,
                                 out_scale=v.scale if self.q else 0.0)
                    if v.res is not None or v.ups != 1 or v.fp32:
                        raise NotImplementedError('HIP engine: fused epilogue on the first conv')
                    if v.c_phys > pk['cout_pad']:
                        raise NotImplementedError('HIP engine: stem channel padding')
                    op = add(d, 'stem%d' % v.block)
                    fixup(op, StemDesc, 'x', SLOT_INPUT)
                else:
                    s = v.src
                    if s.fp32 and self.code == hiplib.YH_F16:
                        raise NotImplementedError('HIP engine: block %d consumes a yolo-head tensor' % v.block)
                    if self.q and s.scale is None:
                        raise NotImplementedError('HIP int8 engine: block %d reads a tensor that is not on an int8 grid (%s output)'
                                                  % (v.block, s.kind))
                    tile = self.force_tile
                    if 10 <= tile < 20 and pk['cin_k'] % (2 * self.kstep):
                        tile -= 10  # the 8-unit K step needs cin_k to be a multiple of it
                    if tile >= 40 and not (v.k == 3 and v.stride == 1 and v.pad == 1 and s.W <= 160):
                        tile = 0    # halo kernels are 3x3 / stride 1 only
                    d = ConvDesc(x=P(s.storage, s.c_off), w=P(pk['w']), bias=P(pk['b']),
                                 res=None if v.res is None else P(v.res.storage, v.res.c_off), y=y,
                                 n=N, h=s.H, w_in=s.W, cin=s.c_phys, ho=v.Ho, wo=v.Wo, cout=v.c_phys,
                                 kh=v.k, kw=v.k, stride=v.stride, pad=v.pad, ldx=s.ld,
                                 ldr=0 if v.res is None else v.res.ld, ldy=v.ld, cin_k=pk['cin_k'], m_pad=pk['m_pad'],
                                 act=v.act, slope=v.slope, ups=v.ups, out_f32=1 if v.fp32 else 0, dtype=self.code, tile=tile,
                                 acc_scale=(v.s_w * s.scale) if self.q else 0.0, out_scale=v.scale if self.q else 0.0)
                    if self.q and v.res is not None:
                        sx, sa, ssum = v.qadd
                        d.out_scale = v.act_scale
                        d.q_rx, d.q_ra, d.q_scale_x, d.q_scale_a, d.q_inv_scale_sum = v.act_scale / sx, v.res.scale / sa, sx, sa, 1.0 / ssum
                    add(d, 'conv%d' % v.block)
            elif v.kind == 'dw':
                pk = self._packed.get(v.block) or self._pack_dw(v)
                s = v.src
                add(DwDesc(x=P(s.storage, s.c_off), w=P(pk['w']), bias=P(pk['b']), y=y, n=N, h=s.H, w_in=s.W, c=s.c_phys,
                           ho=v.H, wo=v.W, k=v.k, stride=v.stride, pad=v.pad, ldx=s.ld, ldy=v.ld, act=v.act, slope=v.slope,
                           dtype=self.code), 'dw%d' % v.block)
            elif v.kind == 'se':
                pk = self._packed.get(v.block) or self._pack_se(v, N)
                s = v.src
                pooled = torch.empty((N, v.c_phys), device=self.device, dtype=torch.float32)
                gate = torch.empty((N, v.c_phys), device=self.device, dtype=torch.float32)
                plan['storages'].extend((pooled, gate))
                add(SeDesc(x=P(s.storage, s.c_off), y=y, w1=P(pk['w1']), w2=P(pk['w2']), pooled=P(pooled), gate=P(gate),
                           ch_map=P(pk['cmap']), n=N, h=s.H, w_in=s.W, c=v.C, c_phys=v.c_phys, cr=pk['w1'].shape[0],
                           ldx=s.ld, ldy=v.ld, dtype=self.code), 'se%d' % v.block)
            elif v.kind == 'pool':
                s = v.src
                cls = QPoolDesc if self.q else PoolDesc
                add(cls(x=P(s.storage, s.c_off), y=y, n=N, h=s.H, w_in=s.W, c=s.c_phys, ho=v.H, wo=v.W, k=v.k,
                        stride=v.stride, pad_lo=v.pad_lo, edge_zero=v.edge_zero, ldx=s.ld, ldy=v.ld,
                        dtype=self.code), 'pool%d' % v.block)
            elif v.kind == 'copy' and self.q:
                s = v.src
                add(QCopyDesc(x=P(s.storage, s.c_off), y=y, n=N, h=s.H, w_in=s.W, c=s.c_phys, ups=v.ups, ldx=s.ld, ldy=v.ld,
                              ratio=1.0), 'ups%d' % v.block)
            elif v.kind == 'qadd':
                add(QAddDesc(x=P(v.a.storage, v.a.c_off), a=P(v.b.storage, v.b.c_off), y=y, pixels=N * v.H * v.W, c=v.c_phys,
                             ldx=v.a.ld, lda=v.b.ld, ldy=v.ld, rx=v.a.scale / v.scale_x, ra=v.b.scale / v.scale_a,
                             scale_x=v.scale_x, scale_a=v.scale_a, inv_scale_sum=1.0 / v.scale), 'qadd%d' % v.block)
            elif v.kind == 'copy':
                s = v.src
                add(CopyDesc(x=P(s.storage, s.c_off), y=y, n=N, h=s.H, w_in=s.W, c=s.c_phys, ups=v.ups, ldx=s.ld,
                             ldy=v.ld, dtype=self.code), 'ups%d' % v.block)
            elif v.kind == 'add':
                maps = [None, None]
                if getattr(v, 'amap', None) is not None:
                    maps = [torch.tensor(m, dtype=torch.int32).to(self.device) for m in (v.amap, v.bmap)]
                    plan['storages'].extend(maps)
                add(AddDesc(a=P(v.a.storage, v.a.c_off), b=P(v.b.storage, v.b.c_off), y=y, pixels=N * v.H * v.W,
                            c=v.c_phys, lda=v.a.ld, ldb=v.b.ld, ldy=v.ld, dtype=self.code,
                            amap=P(maps[0]) if maps[0] is not None else None,
                            bmap=P(maps[1]) if maps[1] is not None else None), 'add%d' % v.block)
            elif v.kind == 'concat':
                for s, off, inplace in v.parts:
                    if inplace:
                        continue  # its producer writes (or already wrote) the slice
                    if s.fp32:
                        raise NotImplementedError('HIP engine: route over a yolo-head tensor')
                    if self.q:
                        add(QCopyDesc(x=P(s.storage, s.c_off), y=P(v.storage, v.c_off + off), n=N, h=s.H, w_in=s.W, c=s.c_phys,
                                      ups=1, ldx=s.ld, ldy=v.ld, ratio=s.scale / v.scale), 'cat%d' % v.block)
                        continue
                    add(CopyDesc(x=P(s.storage, s.c_off), y=P(v.storage, v.c_off + off), n=N, h=s.H, w_in=s.W,
                                 c=s.c_phys, ups=1, ldx=s.ld, ldy=v.ld, dtype=self.code), 'cat%d' % v.block)

        rows = sum(h.na * h.src.H * h.src.W for h in heads)
        no = heads[0].no
        off = 0
        plan['raw_shapes'] = []
        plan['decode_descs'] = []                       # the heads as yh_yolo_decode_candidates takes them (detect())
        plan['first_decode_op'] = len(plan['ops'])      # the decode ops are the last ops of the plan
        for k, h in enumerate(heads):
            if h.no != no:
                raise ValueError('yolo heads disagree on the class count')
            s = h.src
            d = DecodeDesc(p=P(s.storage, s.c_off), io=None, raw=None, n=N, ny=s.H, nx=s.W, na=h.na, no=h.no, ldp=s.ld,
                           rows_total=rows, row_off=off, stride=h.stride)
            for a in range(h.na):
                d.anchor_w[a] = float(h.anchor_vec[a, 0])
                d.anchor_h[a] = float(h.anchor_vec[a, 1])
            op = add(d, 'yolo%d' % h.block)
            fixup(op, DecodeDesc, 'io', SLOT_IO)
            fixup(op, DecodeDesc, 'raw', SLOT_RAW0 + k)      # bound to NULL per call when the caller does not want the copies
            plan['raw_shapes'].append((N, h.na, s.H, s.W, h.no))
            plan['decode_descs'].append(DecodeDesc.from_buffer_copy(d))
            off += h.na * s.H * s.W
        plan['rows'], plan['no'] = rows, no
        return plan

    # ---------------------------------------------------------------------------------- execute
    def refresh_weights(self):
        """Re-pack every conv from the live parameters (buffers are reused; plans stay valid)."""
        for plan in self._plans.values():
            for v in plan['values']:
                if v.kind == 'conv':
                    self._pack_conv(v)
                elif v.kind == 'dw':
                    self._pack_dw(v)
                elif v.kind == 'se':
                    self._pack_se(v, plan['N'])
            break
        self._signature = self._current_signature()

    def __call__(self, x):
        x, plan = self._plan_for(x)
        N = x.shape[0]
        lib, handle = self.lib, plan['handle']
        if 0 < N <= self.graph_max_batch and x.is_cuda and hasattr(lib, 'yh_plan_graph_launch'):
            io, raws = self._run_graph(plan, x)
        else:
            io = torch.empty((N, plan['rows'], plan['no']), device=x.device, dtype=torch.float32)
            raws = [torch.empty(shape, device=x.device, dtype=torch.float32) for shape in plan['raw_shapes']] \
                if self.want_raw else []
            lib.yh_plan_bind_slot(handle, SLOT_INPUT, x.data_ptr())
            lib.yh_plan_bind_slot(handle, SLOT_IO, io.data_ptr())
            for k in range(len(plan['raw_shapes'])):
                lib.yh_plan_bind_slot(handle, SLOT_RAW0 + k, raws[k].data_ptr() if raws else hiplib.SLOT_NULL)
            hiplib.check(lib.yh_plan_run(handle, hiplib.stream_ptr()), 'yh_plan_run')
        feats = self._features(plan) if self.return_features else []
        return io, tuple(raws), feats

    def detect(self, x, conf_thres=0.3, iou_thres=0.6, multi_label=False, classes=None, agnostic=False):
        """``non_max_suppression(model(x)[0], ...)`` (reference detect.py:104-109) without the decoded (N, rows, 5 + nc) tensor: the plan
        runs up to the head convolutions, ``yh_yolo_decode_candidates`` decodes only the rows whose objectness passes ``conf_thres``
        straight into the NMS candidate records, the remaining NMS steps are ``engine/nms.py``'s.  Same list of (n_i, 6) tensors,
        bit for bit (tests/test_gpu_nms.py).  Small batches keep the hipGraph replay of ``__call__`` (launch-bound there)."""
        from . import nms as hnms
        x, plan = self._plan_for(x)
        N = x.shape[0]
        lib, handle = self.lib, plan['handle']
        if (0 < N <= self.graph_max_batch and x.is_cuda and hasattr(lib, 'yh_plan_graph_launch')) or self.return_features \
                or os.environ.get('YOLO_HIP_FUSED_DETECT', '1') == '0':
            keep = self.want_raw
            self.want_raw = False
            try:
                io = self(x)[0]
            finally:
                self.want_raw = keep
            return hnms.non_max_suppression(io, conf_thres, iou_thres, multi_label, classes, agnostic)
        lib.yh_plan_bind_slot(handle, SLOT_INPUT, x.data_ptr())
        hiplib.check(lib.yh_plan_run_range(handle, 0, plan['first_decode_op'], hiplib.stream_ptr()), 'yh_plan_run_range')
        return hnms.non_max_suppression_heads(plan['decode_descs'], N, plan['rows'], plan['no'] - 5, x.device, conf_thres, iou_thres,
                                              multi_label, classes, agnostic)

    def _plan_for(self, x):
        """The checks of a call and the plan of this input shape (built on first use)."""
        if x.dim() != 4:
            raise ValueError('expected an (N, C, H, W) batch')
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        if self.device is None:
            self.device = x.device
        elif self.device != x.device:
            raise RuntimeError('engine was built on %s, input is on %s' % (self.device, x.device))
        first = self.model.module_list[0]
        w0 = first[0].weight if isinstance(first, nn.Sequential) and len(first) else None
        if w0 is None or w0.device != x.device:
            raise RuntimeError('model parameters must live on the input device %s' % x.device)

        sig = self._current_signature()
        if self._signature is None:
            self._signature = sig
        elif sig != self._signature:
            old_shapes = [s[2] for s in self._signature]
            if old_shapes != [s[2] for s in sig]:  # graph widths changed (pruning): start over
                self._drop_plans()
                self._packed = {}
                self._signature = sig
            else:
                self._signature = sig
                if self._plans:
                    self.refresh_weights()
        elif self._plans and os.environ.get('YOLO_HIP_SAFE', '0') == '1':
            # edits through `param.data` views (`p.data.mul_(...)`) do not bump `param._version`: with YOLO_HIP_SAFE=1 every
            # eval call re-packs from the live tensors (one pack launch per conv) instead of trusting the signature
            self.refresh_weights()

        N, Cin, H, W = x.shape
        if bool(self.return_features) != self._plans_with_features:   # feature_out plans are built without epilogue fusion
            self._drop_plans()
            self._plans_with_features = bool(self.return_features)
        key = (N, Cin, H, W)
        plan = self._plans.get(key)
        if plan is None:
            # every plan owns the activation buffers of its input shape; rectangular evaluation (test.py rect=True) walks
            # through dozens of shapes, so only the most recently used few are kept
            while len(self._plans) >= self.max_plans:
                old = self._plans.pop(next(iter(self._plans)))
                self.lib.yh_plan_destroy(old['handle'])
            plan = self._plans[key] = self._build_plan(N, Cin, H, W)
        else:
            self._plans[key] = self._plans.pop(key)   # most recently used last
        return x, plan

    def _run_graph(self, plan, x):
        """Small-batch path: static I/O buffers + one hipGraph launch; results are copied out (fresh tensors).

        Stream capture is not allowed on the legacy default stream, so the graph lives on an engine-owned
        side stream that is ordered against the caller's current stream on both sides."""
        lib, handle = self.lib, plan['handle']
        cur = torch.cuda.current_stream()
        gs = self.__dict__.get('_graph_stream')
        if gs is None:
            gs = self._graph_stream = torch.cuda.Stream(device=x.device)
        gs.wait_stream(cur)
        with torch.cuda.stream(gs):
            sp = C.c_void_p(gs.cuda_stream)
            g = plan.get('graph')
            if g is None or g['want_raw'] != self.want_raw:
                dev = x.device
                g = dict(x=torch.empty_like(x),
                         io=torch.empty((x.shape[0], plan['rows'], plan['no']), device=dev, dtype=torch.float32),
                         raws=[torch.empty(sh, device=dev, dtype=torch.float32) for sh in plan['raw_shapes']]
                         if self.want_raw else [], want_raw=self.want_raw)
                lib.yh_plan_bind_slot(handle, SLOT_INPUT, g['x'].data_ptr())
                lib.yh_plan_bind_slot(handle, SLOT_IO, g['io'].data_ptr())
                for k in range(len(plan['raw_shapes'])):
                    lib.yh_plan_bind_slot(handle, SLOT_RAW0 + k, g['raws'][k].data_ptr() if g['raws'] else hiplib.SLOT_NULL)
                g['x'].copy_(x)
                hiplib.check(lib.yh_plan_run(handle, sp), 'yh_plan_run')   # warm (lazy code loads) before capturing
                hiplib.check(lib.yh_plan_graph_capture(handle, sp), 'yh_plan_graph_capture')
                plan['graph'] = g
            g['x'].copy_(x)
            hiplib.check(lib.yh_plan_graph_launch(handle, sp), 'yh_plan_graph_launch')
            io = g['io'].clone()
            raws = [r.clone() for r in g['raws']]
        cur.wait_stream(gs)
        for t in [io] + raws:
            t.record_stream(cur)
        return io, raws

    def _features(self, plan):
        """NCHW fp32 copies of the conv-block outputs the reference appends to ``feature_out``."""
        feats = []
        mods = self.model.module_list
        for i, m in enumerate(mods):
            if m.__class__.__name__ == 'Sequential' and i + 1 < len(mods) and \
                    mods[i + 1].__class__.__name__ != 'YOLOLayer':
                feats.append(self.block_output(plan, i))
        return feats

    def block_output(self, plan, i):
        """Output of cfg block ``i`` as an NCHW fp32 tensor (debug / feature_out); None if fused away."""
        v = plan['outs'][i]
        if v is None or v.kind == 'input' or v.storage is None:
            return None
        t = v.storage[..., v.c_off:v.c_off + v.c_phys]
        idx = torch.tensor(v.channel_map(), device=t.device)
        return t.index_select(3, idx).permute(0, 3, 1, 2).float().contiguous()

    def _drop_plans(self):
        for plan in self._plans.values():
            self.lib.yh_plan_destroy(plan['handle'])
        self._plans = {}

    def __del__(self):
        try:
            self._drop_plans()
        except Exception:
            pass
