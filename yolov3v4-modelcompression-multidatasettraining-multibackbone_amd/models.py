"""cfg-driven Darknet graph builder and module — the drop-in boundary of the hot path.

Interface mirror of the reference's ``models.py``: ``create_modules`` :11-347, ``YOLOLayer``
:350-437, ``Darknet`` :440-581, ``get_yolo_layers`` :583, ``load_darknet_weights`` :587-735,
``save_weights`` :738-782, ``convert`` :785, ``attempt_download`` :816.  The object contract that
the reference's scripts rely on is preserved:

* a conv block is ``nn.Sequential`` with children named ``Conv2d`` / ``DepthWise2d``,
  ``BatchNorm2d``, ``activation``; parameters stay live ``nn.Parameter`` s that prune scripts may
  slice or overwrite in place;
* ``forward_once`` dispatches on class *names* (``Shortcut``, ``FeatureConcat``, ``YOLOLayer`` ...);
* eval returns ``(inf_out, raw_p_tuple, feature_out)``, train returns ``(raw_p_list, feature_out)``.

What is new: for a CUDA input in eval mode ``forward`` hands the whole graph to the HIP engine
(``engine/plan.py`` -> ``libyolo_hip.so``).  The engine keeps a packed NHWC/MFMA-tiled copy of the
weights keyed on every source tensor's ``_version`` so in-place edits by the prune/PTQ scripts are
picked up.  There is no silent fallback on that path: if the library cannot be loaded, or the graph
holds a block the engine cannot lower, it raises.  A CUDA input in train mode runs the HIP training plans
(``engine/train.py``; widths that are not multiples of 8 through the channel-padded twin of ``engine/padded.py``) behind
one autograd node per backward range - again without an eager fallback.  CPU tensors run the eager module-by-module
semantics below (bit-equal to the reference; what the host-side scripts and the CPU test tier use).
"""
import copy
import math
import os
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from utils import torch_utils
from utils.layers import (FeatureConcat, Shortcut, Mish, ReLU6, HardSwish, HardSigmoid, SE, Swish,
                          Flatten, Concat, make_divisible)
from utils.parse_config import parse_model_cfg, parse_data_cfg

_MERGE_NAMES = ('Shortcut', 'FeatureConcat', 'QuantizedShortcut_max', 'QuantizedShortcut_min',
                'QuantizedFeatureConcat', 'COSPTQuantizedShortcut_min', 'COSPTQuantizedShortcut_max',
                'COSPTQuantizedFeatureConcat')


# ------------------------------------------------------------------------------------------- builder
def _activation(name, maxabsscaler):
    """cfg activation string -> module (None for linear / unknown)."""
    if name == 'leaky':
        return nn.LeakyReLU(0.25 if maxabsscaler else 0.1, inplace=True)
    if name == 'relu6':
        return ReLU6()
    if name == 'h_swish':
        return HardSwish()
    if name == 'relu':
        return nn.ReLU()
    if name == 'mish':
        return Mish()
    return None


def _takes_stream_pair(module):
    """Conv blocks of a quantised graph consume the calibration-mode [quantised, float] pair themselves."""
    return isinstance(module, nn.Sequential) and len(module) > 0 and hasattr(module[0], 'activation_quantizer')


def _quantized_namespace(quantized):
    """Lazy import of the reference's fake-quant operator classes (they run unmodified on top)."""
    if quantized == 1:
        import utils.quantized.quantized_google as q
    elif quantized == 2:
        import utils.quantized.quantized_TPSQ as q
    else:
        import utils.quantized.quantized_ptq_cos as q
    return q


def _float_conv_block(block, mdef, cin, depthwise, maxabsscaler):
    bn = int(mdef['batch_normalize'])
    cout = int(mdef['filters'])
    k = int(mdef['size'])
    pad = (k - 1) // 2 if int(mdef['pad']) else 0
    groups = cin if depthwise else (mdef['groups'] if 'groups' in mdef else 1)
    conv = nn.Conv2d(in_channels=cin, out_channels=cout, kernel_size=k, stride=int(mdef['stride']),
                     padding=pad, groups=groups, bias=not bn)
    block.add_module('DepthWise2d' if depthwise else 'Conv2d', conv)
    if bn:
        block.add_module('BatchNorm2d', nn.BatchNorm2d(cout, momentum=0.1))
    act = _activation(mdef['activation'], maxabsscaler)
    if act is not None:
        block.add_module('activation', act)
    return cout


def _quant_conv_block(block, mdef, cin, depthwise, i, q):
    """Quantised conv variants: constructor argument sets follow models.py:34-90,120-175."""
    bn = int(mdef['batch_normalize'])
    cout = int(mdef['filters'])
    k = int(mdef['size'])
    pad = (k - 1) // 2 if int(mdef['pad']) else 0
    groups = cin if depthwise else (mdef['groups'] if 'groups' in mdef else 1)
    common = dict(in_channels=cin, out_channels=cout, kernel_size=k, stride=int(mdef['stride']), padding=pad,
                  groups=groups, bias=not bn, a_bits=q['a_bit'], w_bits=q['w_bit'], bn=bn,
                  activate=mdef['activation'], quantizer_output=q['quantizer_output'],
                  maxabsscaler=q['maxabsscaler'])
    tag = "{:04d}".format(i) + "_" + mdef['type'][:4]
    ns = _quantized_namespace(q['quantized'])
    if q['quantized'] == 1:
        mod = ns.BNFold_QuantizedConv2d_For_FPGA(steps=q['steps'], reorder=q['reorder'], TM=q['TM'], TN=q['TN'],
                                                 name=tag, layer_idx=q['layer_idx'], **common)
    elif q['quantized'] == 2:
        mod = ns.TPSQ_BNFold_QuantizedConv2d_For_FPGA(steps=q['steps'], **common)
    else:
        mod = ns.BNFold_COSPTQuantizedConv2d_For_FPGA(reorder=q['reorder'], TM=q['TM'], TN=q['TN'], name=tag,
                                                      layer_idx=q['layer_idx'], **common)
    block.add_module('DepthWise2d' if depthwise else 'Conv2d', mod)
    return cout


def _graph_stride(scales, mdef, src):
    """Cumulative down-sampling factor of a block's output given its input's factor ``src``."""
    t = mdef['type']
    if t in ('convolutional', 'depthwise'):
        return src * int(mdef['stride'])
    if t == 'maxpool':
        return src * int(mdef['stride'])
    if t == 'upsample':
        return src / int(mdef['stride'])
    return src


def create_modules(module_defs, img_size, cfg, quantized, quantizer_output, layer_idx, reorder, TM, TN, a_bit=8,
                   w_bit=8, steps=0, is_gray_scale=False, maxabsscaler=False, shortcut_way=-1):
    """Walk the parsed cfg blocks and emit one module per block.

    Returns ``(nn.ModuleList, routs)`` where ``routs[i]`` says whether block i's output is read again
    by a later route/shortcut.  Pops the ``[net]`` block off ``module_defs`` (as the reference does).
    """
    img_size = [img_size] * 2 if isinstance(img_size, int) else img_size
    module_defs.pop(0)
    filters_hist = [1 if is_gray_scale else 3]  # channels of the network input, then of every block
    scale_hist = [1.0]  # cumulative stride of the input, then of every block's output
    module_list = nn.ModuleList()
    routed = []
    yolo_index = -1
    qopt = dict(quantized=quantized, quantizer_output=quantizer_output, layer_idx=layer_idx, reorder=reorder,
                TM=TM, TN=TN, a_bit=a_bit, w_bit=w_bit, steps=steps, maxabsscaler=maxabsscaler)
    filters = filters_hist[-1]

    for i, mdef in enumerate(module_defs):
        kind = mdef['type']
        modules = nn.Sequential()
        scale = _graph_stride(scale_hist, mdef, scale_hist[-1])

        if kind in ('convolutional', 'depthwise'):
            dw = kind == 'depthwise'
            if quantized in (1, 2, 3):
                filters = _quant_conv_block(modules, mdef, filters_hist[-1], dw, i, qopt)
            else:
                filters = _float_conv_block(modules, mdef, filters_hist[-1], dw, maxabsscaler)

        elif kind == 'BatchNorm2d':
            filters = filters_hist[-1]
            modules = nn.BatchNorm2d(filters, momentum=0.03, eps=1E-4)
            if i == 0 and filters == 3:  # imagenet statistics for an input-normalising first block
                modules.running_mean = torch.tensor([0.485, 0.456, 0.406])
                modules.running_var = torch.tensor([0.0524, 0.0502, 0.0506])

        elif kind == 'maxpool':
            k, s = mdef['size'], mdef['stride']
            pool = nn.MaxPool2d(kernel_size=k, stride=s, padding=(k - 1) // 2)
            if k == 2 and s == 1:  # yolov3-tiny: keep the grid size with a zero right/bottom border
                modules.add_module('ZeroPad2d', nn.ZeroPad2d((0, 1, 0, 1)))
                modules.add_module('MaxPool2d', pool)
            else:
                modules = pool

        elif kind == 'se':
            if 'filters' in mdef:
                filters = int(mdef['filters'])
            modules.add_module('se', SE(channel=filters))
            if 'reduction' in mdef:
                modules.add_module('se', SE(filters_hist[-1], reduction=int(mdef['reduction'])))

        elif kind == 'upsample':
            modules = nn.Upsample(scale_factor=mdef['stride'])

        elif kind == 'route':
            layers = mdef['layers']
            filters = sum(filters_hist[l + 1 if l > 0 else l] for l in layers)
            grouped = 'groups' in mdef
            if grouped:
                filters = filters // 2
            routed.extend([i + l if l < 0 else l for l in layers])
            first = layers[0]
            scale = scale_hist[first + 1 if first > 0 else first]
            if quantized == -1:
                modules = FeatureConcat(layers=layers, groups=grouped)
            else:
                ns = _quantized_namespace(quantized)
                cls = ns.COSPTQuantizedFeatureConcat if quantized == 3 else ns.QuantizedFeatureConcat
                modules = cls(layers=layers, groups=grouped, bits=a_bit, quantizer_output=quantizer_output,
                              reorder=reorder, TM=TM, TN=TN, name="{:04d}".format(i) + "_" + kind[:4],
                              layer_idx=layer_idx)

        elif kind == 'shortcut':
            layers = mdef['from']
            filters = filters_hist[-1]
            routed.extend([i + l if l < 0 else l for l in layers])
            weighted = 'weights_type' in mdef
            if quantized in (-1, 2):
                modules = Shortcut(layers=layers, weight=weighted)
            else:
                ns = _quantized_namespace(quantized)
                prefix = 'COSPTQuantizedShortcut_' if quantized == 3 else 'QuantizedShortcut_'
                suffix = {1: 'min', 2: 'max'}.get(shortcut_way)
                if suffix is not None:
                    modules = getattr(ns, prefix + suffix)(
                        layers=layers, weight=weighted, bits=a_bit, quantizer_output=quantizer_output,
                        reorder=reorder, TM=TM, TN=TN, name="{:04d}".format(i) + "_" + kind[:4],
                        layer_idx=layer_idx)

        elif kind == 'reorg3d':
            pass

        elif kind == 'yolo':
            yolo_index += 1
            layers = mdef['from'] if 'from' in mdef else []
            modules = YOLOLayer(anchors=mdef['anchors'][mdef['mask']], nc=mdef['classes'], img_size=img_size,
                                yolo_index=yolo_index, layers=layers,
                                stride=_head_stride(cfg, yolo_index, scale_hist[-1]),
                                quantizer_output=quantizer_output)
            _smart_bias_init(module_list, modules, layers, yolo_index, 'from' in mdef)

        else:
            print('Warning: Unrecognized Layer Type: ' + kind)

        module_list.append(modules)
        filters_hist.append(filters)
        scale_hist.append(scale)

    routs = [False] * len(module_defs)
    for r in routed:
        routs[r] = True
    return module_list, routs


def _head_stride(cfg, yolo_index, graph_scale):
    """Pixel stride of a yolo head.

    The reference picks it from a fixed table chosen by substring-matching the cfg *path*
    (models.py:312-315), which breaks when the cfg is a list of dicts (every prune script).  The
    graph's own cumulative down-sampling factor equals that table wherever the table is right, and
    stays right for list cfgs, so it is used whenever it is a positive integer.
    """
    g = int(round(graph_scale))
    if g >= 1 and abs(graph_scale - g) < 1e-9:
        return g
    table = [32, 16, 8]
    if isinstance(cfg, str) and any(tag in cfg for tag in ('panet', 'yolov4', 'cd53')) and 'yolov4-tiny' not in cfg:
        table = table[::-1]
    return table[yolo_index]


def _smart_bias_init(module_list, yolo, layers, yolo_index, has_from):
    """Focal-loss style prior on the conv that feeds a yolo head (arXiv 1708.02002 §3.3)."""
    try:
        with torch.no_grad():
            j = layers[yolo_index] if has_from else -1
            b = module_list[j][0].bias
            v = b[:yolo.no * yolo.na].view(yolo.na, -1)
            v[:, 4] -= 4.5
            v[:, 5:] += math.log(0.6 / (yolo.nc - 0.99))
            module_list[j][0].bias = torch.nn.Parameter(b, requires_grad=b.requires_grad)
    except Exception:
        print('WARNING: smart bias initialization failure.')


# ---------------------------------------------------------------------------------------------- head
class YOLOLayer(nn.Module):
    """Decode of one detection scale.

    Input ``p`` is (bs, na*no, ny, nx).  Train: returns the raw (bs, na, ny, nx, no) view.  Eval:
    ``xy = (sigmoid(t_xy) + cell) * stride``, ``wh = exp(t_wh) * anchor``, ``sigmoid`` on obj/cls;
    returns ``(io.view(bs, -1, no), raw)``.
    """

    def __init__(self, anchors, nc, img_size, yolo_index, layers, stride, quantizer_output=False):
        super().__init__()
        self.anchors = torch.Tensor(anchors)
        self.index = yolo_index
        self.layers = layers
        self.stride = stride
        self.nl = len(layers)
        self.na = len(anchors)
        self.nc = nc
        self.no = nc + 5
        self.nx, self.ny, self.ng = 0, 0, 0
        self.anchor_vec = self.anchors / self.stride
        self.anchor_wh = self.anchor_vec.view(1, self.na, 1, 1, 2)
        self.quantizer_output = quantizer_output

    def create_grids(self, ng=(13, 13), device='cpu'):
        self.nx, self.ny = ng
        self.ng = torch.tensor(ng, dtype=torch.float)
        if not self.training:
            ys = torch.arange(self.ny, device=device).view(self.ny, 1).expand(self.ny, self.nx)
            xs = torch.arange(self.nx, device=device).view(1, self.nx).expand(self.ny, self.nx)
            self.grid = torch.stack((xs, ys), 2).view(1, 1, self.ny, self.nx, 2).float()
        if self.anchor_vec.device != device:
            self.anchor_vec = self.anchor_vec.to(device)
            self.anchor_wh = self.anchor_wh.to(device)

    def forward(self, p, out):
        bs, _, ny, nx = p.shape
        self.create_grids((nx, ny), p.device)
        p = p.view(bs, self.na, self.no, self.ny, self.nx).permute(0, 1, 3, 4, 2).contiguous()
        if self.training:
            return p
        io = p.clone()
        io[..., :2] = torch.sigmoid(io[..., :2]) + self.grid
        io[..., 2:4] = torch.exp(io[..., 2:4]) * self.anchor_wh
        io[..., :4] *= self.stride
        torch.sigmoid_(io[..., 4:])
        return io.view(bs, -1, self.no), p


# ------------------------------------------------------------------------------------------- network
def _on_device(device):
    """Make ``device`` the current GPU for the launches inside (a no-op for the CPU tensors of the host-emulation tests)."""
    import contextlib
    if device is not None and torch.device(device).type == 'cuda':
        return torch.cuda.device(device)
    return contextlib.nullcontext()


class _HipTrainSegment(torch.autograd.Function):
    """Autograd node of one backward RANGE of a training step on the HIP engine.

    The step is a chain of these nodes (engine/train.py ``_make_segments``): node 0's forward runs the whole forward
    plan, the others only hand out the head tensors computed in their range; a dummy token links node k to node k+1
    so autograd runs the backward ranges last to first.  Each node returns its range's fp32 parameter gradients as
    soon as that range has run — optimizers, GradScaler, gradient accumulation and DistributedDataParallel's bucket
    hooks see ordinary ``.grad`` tensors, and DDP can all-reduce a bucket while earlier layers are still in backward."""

    @staticmethod
    def forward(ctx, engine, k, token, *params):
        if k == 0:
            with _on_device(token.device):                   # kernels launch on the current device: make it the batch's
                engine.step_heads = engine.forward(token)    # token of range 0 is the input batch
        plan = engine._current
        ctx.engine, ctx.k, ctx.step = engine, k, engine.steps
        heads = [engine.step_heads[j] for j in plan['segments'][k]['heads']]
        return (torch.zeros(1, device=heads[0].device if heads else engine.step_heads[0].device),) + tuple(heads)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_token, *head_grads):
        eng = ctx.engine
        if eng.steps != ctx.step:
            raise RuntimeError('HIP training path: backward() of a forward whose buffers were overwritten by a later '
                               'forward (one forward per backward, like gradient checkpointing-free eager training)')
        with _on_device(eng.device):
            grads = eng.backward_segment(ctx.k, head_grads)
        return (None, None, None) + tuple(grads)


_ENGINE_ATTRS = ('_hip_engine', '_hip_train_engine')   # ctypes handles + device buffers: never copied or pickled


class Darknet(nn.Module):
    """YOLOv3/v4 detector assembled from a darknet cfg (path or list of block dicts)."""

    def __init__(self, cfg, img_size=(416, 416), verbose=False, quantized=-1, a_bit=8, w_bit=8,
                 quantizer_output=False, layer_idx=-1, reorder=False, TM=32, TN=32, steps=0, is_gray_scale=False,
                 maxabsscaler=False, shortcut_way=-1, **ignored):
        # **ignored: detect.py:26 / convert_FPGA.py:18 pass FPGA=..., which the reference rejects
        super().__init__()
        if isinstance(cfg, str):
            self.module_defs = parse_model_cfg(cfg)
        elif isinstance(cfg, list):
            self.module_defs = cfg
        else:
            raise TypeError('cfg must be a path or a list of block dicts')
        self.quantized = quantized
        self.a_bit = a_bit
        self.w_bit = w_bit
        self.quantizer_output = quantizer_output
        self.layer_idx = layer_idx
        self.reorder = reorder
        self.TM = TM
        self.TN = TN
        self.is_gray_scale = is_gray_scale

        self.hyperparams = copy.deepcopy(self.module_defs[0])
        self.module_list, self.routs = create_modules(
            self.module_defs, img_size, cfg, quantized=quantized, quantizer_output=quantizer_output,
            reorder=reorder, TM=TM, TN=TN, layer_idx=layer_idx, a_bit=a_bit, w_bit=w_bit, steps=steps,
            is_gray_scale=is_gray_scale, maxabsscaler=maxabsscaler, shortcut_way=shortcut_way)
        self.yolo_layers = get_yolo_layers(self)

        self.version = np.array([0, 2, 5], dtype=np.int32)  # darknet file header: major, minor, revision
        self.seen = np.array([0], dtype=np.int64)  # darknet file header: images seen in training
        self._hip_engine = None  # built lazily on the first CUDA eval forward
        self.hip_precision = os.environ.get('YOLO_HIP_PRECISION', 'fp16')
        # feature_out (models.py:540-543: every conv block not feeding a yolo layer, used by the feature-distillation
        # losses KD4 / KD5) costs an NCHW fp32 copy of ~70 tensors per forward, so the HIP paths return [] unless asked:
        # eval copies them out of the engine's buffers; the training step does not carry them (the feature-distillation
        # strategies are out of scope) and raises when asked
        self.hip_return_features = False
        if quantized == -1:
            self.info(verbose)

    # -- execution ---------------------------------------------------------------------------------
    def forward(self, x, augment=False):
        if not augment:
            return self.forward_once(x)
        # test-time augmentation: original + (flip, 0.83x) + (0.67x), de-augmented and concatenated
        h, w = x.shape[-2:]
        s = [0.83, 0.67]
        y = []
        for xi in (x, torch_utils.scale_img(x.flip(3), s[0], same_shape=False),
                   torch_utils.scale_img(x, s[1], same_shape=False)):
            y.append(self.forward_once(xi)[0])
        y[1][..., :4] /= s[0]
        y[1][..., 0] = w - y[1][..., 0]
        y[2][..., :4] /= s[1]
        return torch.cat(y, 1), None

    def _use_hip(self, x):
        # float graphs and the calibrated COS-PTQ graph (quantized == 3, eval) are lowered; the QAT research
        # quantisers (1, 2) stay on the eager modules
        return x.is_cuda and not self.training and self.quantized in (-1, 3)

    def _use_hip_train(self, x):
        # training step on the HIP kernels (engine/train.py, engine/padded.py): every float graph on a GPU.  A cfg that path
        # cannot lower raises - there is no eager fallback and no switch that selects one (tests that want the eager modules
        # on a GPU call `_forward_eager` themselves).
        use = x.is_cuda and self.training and self.quantized == -1
        if use and self.__dict__.get('hip_return_features', False):
            raise NotImplementedError('feature_out is not produced by the HIP training step: the feature-distillation losses '
                                      '(reference train.py KDstr 2-5) are out of scope; clear hip_return_features')
        return use

    def forward_once(self, x, augment=False, verbose=False):
        if not verbose and not augment:
            if self._use_hip(x):
                return self._forward_hip(x)
            if self._use_hip_train(x):
                return self._forward_hip_train(x)
        return self._forward_eager(x, augment=augment, verbose=verbose)

    def _forward_hip_train(self, x):
        """Train-mode forward on the HIP engine; returns ``(raw_p_list, [])`` like the eager path (models.py:336-340:
        raw p is the (bs, na, ny, nx, no) view of the head conv).  fp16 compute under ``torch.autocast`` (the -mpt
        recipe), fp32 otherwise; ``YOLO_HIP_TRAIN_PRECISION`` overrides."""
        from engine.padded import make_train_engine  # raises if libyolo_hip.so is missing: no fallback
        precision = os.environ.get('YOLO_HIP_TRAIN_PRECISION') or \
            ('fp16' if torch.is_autocast_enabled() else 'fp32')
        eng = self.__dict__.get('_hip_train_engine')
        if eng is None or eng.precision != precision:
            # aligned widths: engine/train.py; widths that are not multiples of 8 (slim-pruned graphs): the same kernels through a
            # channel-padded twin (engine/padded.py).  NotImplementedError surfaces here, before any state changes
            eng = make_train_engine(self, precision, x)
            self.__dict__['_hip_train_engine'] = eng
        plan = eng._get_plan(x)
        heads = [None] * len(self.yolo_layers)
        token = x
        for k, params in enumerate(eng.segment_parameters(plan)):
            res = _HipTrainSegment.apply(eng, k, token, *params)
            token = res[0]
            for j, h in zip(plan['segments'][k]['heads'], res[1:]):
                heads[j] = h
        yolo_out = []
        for h, idx in zip(heads, self.yolo_layers):
            m = self.module_list[idx]
            bs, ny, nx, _ = h.shape
            yolo_out.append(h[..., :m.na * m.no].view(bs, ny, nx, m.na, m.no).permute(0, 3, 1, 2, 4))
        return yolo_out, []

    def _forward_hip(self, x):
        from engine.plan import DarknetEngine  # raises if libyolo_hip.so is missing: no fallback
        eng = self.__dict__.get('_hip_engine')
        precision = 'int8' if self.quantized == 3 else self.hip_precision
        if eng is None or eng.precision != precision:
            eng = DarknetEngine(self, precision=precision)
            self.__dict__['_hip_engine'] = eng
        eng.return_features = bool(self.__dict__.get('hip_return_features', False))
        # the second return value - the raw (bs, na, ny, nx, no) head maps, models.py:336-340 - is a copy the engine makes only for
        # callers that read it (test.py's validation loss); detect.py reads model(img)[0] alone and switches it off
        eng.want_raw = bool(self.__dict__.get('hip_return_raw', True))
        with _on_device(x.device):   # kernels launch on the current device / stream: make it the input's
            return eng(x)

    def hip_detect(self, x, conf_thres=0.3, iou_thres=0.6, multi_label=False, classes=None, agnostic=False):
        """``non_max_suppression(self(x)[0], conf_thres, iou_thres, ...)`` - what detect.py does with a batch (reference
        detect.py:104-109) - as ONE engine call: on the HIP path the decoded (N, rows, 5 + nc) tensor is never written, the yolo
        heads' outputs are decoded and filtered in one pass straight into the NMS candidate records (engine/plan.py ``detect``).
        Same result, bit for bit; on the eager path (CPU, training mode) it is literally the two calls."""
        from utils.utils import non_max_suppression
        if not self._use_hip(x):
            return non_max_suppression(self(x)[0], conf_thres, iou_thres, multi_label=multi_label, classes=classes, agnostic=agnostic)
        from engine.plan import DarknetEngine
        eng = self.__dict__.get('_hip_engine')
        precision = 'int8' if self.quantized == 3 else self.hip_precision
        if eng is None or eng.precision != precision:
            eng = DarknetEngine(self, precision=precision)
            self.__dict__['_hip_engine'] = eng
        # a caller that also reads the feature list through model(x) keeps its plans: flipping the flag would make _plan_for drop every
        # plan, buffer and hipGraph on each alternation; plan.detect takes its two-pass branch when features are on
        eng.return_features = bool(getattr(self, 'hip_return_features', False))
        with _on_device(x.device):
            return eng.detect(x, conf_thres, iou_thres, multi_label, classes, agnostic)

    def _forward_eager(self, x, augment=False, verbose=False):
        img_size = x.shape[-2:]
        yolo_out, out, feature_out = [], [], []
        if augment:
            nb = x.shape[0]
            s = [0.83, 0.67]
            x = torch.cat((x, torch_utils.scale_img(x.flip(3), s[0]), torch_utils.scale_img(x, s[1])), 0)
        last = len(self.module_list) - 1
        for i, module in enumerate(self.module_list):
            name = module.__class__.__name__
            if name in _MERGE_NAMES:
                x = module(x, out)
            elif name == 'YOLOLayer':
                yolo_out.append(module(x, out))
            else:
                if isinstance(x, list) and not _takes_stream_pair(module):
                    # COS-PTQ calibration: [quantised, float] streams through a module that knows nothing about pairs.  The
                    # reference does this for Upsample only (models.py:537-539) and raises on max-pool / zero-pad, which is
                    # why it cannot calibrate YOLOv4 or the tiny nets
                    x = [module(x[0]), module(x[1])]
                else:
                    x = module(x)
                if name == 'Sequential' and i < last and self.module_list[i + 1].__class__.__name__ != 'YOLOLayer':
                    feature_out.append(x)
            out.append(x if self.routs[i] else [])
            if verbose:
                print('%g/%g %s -' % (i, len(self.module_list), name), list(x.shape))
        if self.training:
            return yolo_out, feature_out
        x, p = zip(*yolo_out)
        x = torch.cat(x, 1)
        if augment:
            x = list(torch.split(x, nb, dim=0))
            x[1][..., :4] /= s[0]
            x[1][..., 0] = img_size[1] - x[1][..., 0]
            x[2][..., :4] /= s[1]
            x = torch.cat(x, 1)
        return x, p, feature_out

    # -- maintenance -------------------------------------------------------------------------------
    def fuse(self):
        """Fold every BatchNorm2d into its conv (eager modules; the HIP packer folds on its own)."""
        print('Fusing layers...')
        fused_list = nn.ModuleList()
        for block in self.module_list:
            if isinstance(block, nn.Sequential):
                kids = list(block.children())
                for k, child in enumerate(kids):
                    if isinstance(child, nn.modules.batchnorm.BatchNorm2d):
                        fused = torch_utils.fuse_conv_and_bn(kids[k - 1], child)
                        block = nn.Sequential(fused, *kids[k + 1:])
                        break
            fused_list.append(block)
        self.module_list = fused_list
        self.__dict__['_hip_engine'] = self.__dict__['_hip_train_engine'] = None

    def info(self, verbose=False):
        torch_utils.model_info(self, verbose)

    # -- keep the packed-weight cache honest ---------------------------------------------------------
    def hip_refresh(self):
        """Drop the HIP engine; the next CUDA eval forward re-packs from the live parameters.

        The engine already notices optimizer steps and ``param.data = ...`` rebinding (tensor version /
        address); call this after editing weights OR BatchNorm buffers through ``.data`` views in place
        (``bn.running_mean.data.zero_()`` moves no version counter: the training engine would pick it up
        only at its next unconditional buffer sync, engine/padded.py push)."""
        self.__dict__['_hip_engine'] = self.__dict__['_hip_train_engine'] = None

    def _apply(self, fn, *args, **kwargs):
        self.__dict__['_hip_engine'] = self.__dict__['_hip_train_engine'] = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__['_hip_engine'] = self.__dict__['_hip_train_engine'] = None
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        if mode:
            self.__dict__['_hip_engine'] = None  # weights are about to change; the train engine re-packs every step
        return super().train(mode)

    def __deepcopy__(self, memo):
        # the engine holds raw device pointers and ctypes handles: never copy it with the module
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in _ENGINE_ATTRS else copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        # torch.save(model) / pickling: the engines are rebuilt on the next CUDA forward
        state = dict(self.__dict__)
        for k in _ENGINE_ATTRS:
            if k in state:
                state[k] = None
        return state


def get_yolo_layers(model):
    return [i for i, m in enumerate(model.module_list) if m.__class__.__name__ == 'YOLOLayer']


# --------------------------------------------------------------------------------------- weights I/O
def _bn_targets(module, conv, quant):
    """The four tensors a darknet file stores per BN, in file order: beta, gamma, mean, var."""
    if quant:  # BN-folding quantised convs keep the BN parameters on the conv itself
        return [conv.beta, conv.gamma, conv.running_mean, conv.running_var]
    bn = module[1]
    return [bn.bias, bn.weight, bn.running_mean, bn.running_var]


class _FloatCursor:
    def __init__(self, flat):
        self.flat = flat
        self.pos = 0

    def fill(self, tensor):
        n = tensor.numel()
        chunk = torch.from_numpy(self.flat[self.pos:self.pos + n]).view_as(tensor)
        tensor.data.copy_(chunk)
        self.pos += n

    def skip(self, n):
        self.pos += n


def load_darknet_weights(self, weights, cutoff=-1, pt=False, quant=False, **ignored):
    """Read a darknet ``.weights`` file into ``self``.

    Layout: int32[3] version, int64 seen, then float32 per block in cfg order — conv: [BN beta, gamma,
    mean, var] or [conv bias], then conv weight; depthwise: same; se: fc1 then fc2 weights.
    """
    fname = Path(weights).name
    if fname == 'darknet53.conv.74':
        cutoff = 75
    elif fname == 'yolov3-tiny.conv.15':
        cutoff = 15

    with open(weights, 'rb') as fh:
        self.version = np.fromfile(fh, dtype=np.int32, count=3)
        self.seen = np.fromfile(fh, dtype=np.int64, count=1)
        cur = _FloatCursor(np.fromfile(fh, dtype=np.float32))

    prev_conv = None
    for i, (mdef, module) in enumerate(zip(self.module_defs[:cutoff], self.module_list[:cutoff])):
        kind = mdef['type']
        if kind == 'convolutional':
            conv = prev_conv = module[0]
            if mdef['batch_normalize']:
                for t in _bn_targets(module, conv, quant):
                    cur.fill(t)
                cur.fill(conv.weight)
            elif pt and fname.split('.')[-1] == 'weights':
                # COCO-pretrained file, head retargeted to another class count: skip the 255-wide head
                cur.skip(255 + int(self.module_defs[i - 1]['filters']) * 255)
            else:
                cur.fill(conv.bias)
                cur.fill(conv.weight)
        elif kind == 'depthwise':
            conv = module[0]
            if mdef['batch_normalize']:
                # reference quirk (models.py:676-694): with quant=True it fills the *previous*
                # convolutional block's BN tensors; kept so files round-trip the same way
                for t in _bn_targets(module, prev_conv if quant else conv, quant):
                    cur.fill(t)
            cur.fill(conv.weight)
        elif kind == 'se':
            fc = module[0].fc
            cur.fill(fc[0].weight)
            cur.fill(fc[2].weight)

    assert cur.pos == len(cur.flat), 'weights file holds %d floats, model consumed %d' % (len(cur.flat), cur.pos)
    if isinstance(self, Darknet):
        self.__dict__['_hip_engine'] = self.__dict__['_hip_train_engine'] = None


def save_weights(self, path='model.weights', cutoff=-1):
    """Write ``self`` as a darknet ``.weights`` file (not valid after ``fuse()``)."""
    def dump(fh, t):
        t.data.cpu().numpy().tofile(fh)

    with open(path, 'wb') as fh:
        self.version.tofile(fh)
        self.seen.tofile(fh)
        for mdef, module in zip(self.module_defs[:cutoff], self.module_list[:cutoff]):
            kind = mdef['type']
            if kind in ('convolutional', 'depthwise'):
                conv = module[0]
                if mdef['batch_normalize']:
                    bn = module[1]
                    for t in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                        dump(fh, t)
                else:
                    dump(fh, conv.bias)
                dump(fh, conv.weight)
            elif kind == 'se':
                fc = module[0].fc
                dump(fh, fc[0].weight)
                dump(fh, fc[2].weight)


def convert(cfg='cfg/yolov3-spp.cfg', weights='weights/yolov3-spp.weights'):
    """``.pt`` <-> ``.weights`` by extension."""
    model = Darknet(cfg)
    stem, ext = weights.rsplit('.', 1)
    if ext == 'pt':
        model.load_state_dict(torch.load(weights, map_location='cpu', weights_only=False)['model'])
        save_weights(model, path=stem + '.weights', cutoff=-1)
        print("Success: converted '%s' to '%s'" % (weights, stem + '.weights'))
    elif ext == 'weights':
        load_darknet_weights(model, weights)
        chkpt = {'epoch': -1, 'best_fitness': None, 'training_results': None, 'model': model.state_dict(),
                 'optimizer': None}
        torch.save(chkpt, stem + '.pt')
        print("Success: converted '%s' to '%s'" % (weights, stem + '.pt'))
    else:
        print('Error: extension not supported.')


def attempt_download(weights):
    """No network in this deployment: a no-op when the file exists, an explicit error otherwise."""
    weights = weights.strip().replace("'", '')
    if len(weights) > 0 and not os.path.isfile(weights):
        raise Exception(weights + ' missing and there is no network access to download it')
