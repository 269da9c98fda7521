"""HIP training for graphs whose channel counts are not multiples of 8 (every ``slim_prune.py`` output: reference
slim_prune.py:25-30 keeps any number of channels >= 1 per layer, utils/prune_utils.py:212-258 builds the compact model).

The training kernels move channels in 16-byte vectors (8 f16), so the lowering of ``engine/train.py`` wants every tensor width
to be a multiple of 8.  Instead of teaching each kernel about ragged widths, an odd-width model trains through a
**channel-padded twin**: the same cfg with every conv width rounded up to 8 (``Darknet(padded defs)``), living only inside the
engine.  Per step

    live parameters --(one gather launch)--> twin parameters (pad rows / columns / BN lanes are exact zeros)
    twin forward + backward on the HIP training plan (unchanged kernels; pad channels carry zeros end to end)
    twin gradients --(one gather per backward range)--> gradients of the live parameters
    twin BatchNorm running statistics --(one gather)--> the live buffers

so optimizers, GradScaler, EMA, DDP hooks and checkpoints keep seeing the model the user built.  Pad lanes are inert: a padded
output channel has zero weights and zero bias, hence z = 0, batch mean = var = 0, x_hat = 0, y = beta_pad = 0 and every
activation maps 0 -> 0; a padded input column multiplies zeros.  The result equals the unpadded computation exactly in exact
arithmetic and to summation order in floating point (tests/test_gpu_train.py, tests/test_train_emulated.py).

Layouts: each layer's output is described by ``pos`` (physical channel of every logical channel) and the physical width.  Dense
convs are free to place their output channels (default ``arange(C)`` padded at the end), depthwise convs / BN / SE / pools /
upsample keep their input's layout, routes concatenate layouts at 8-aligned offsets.  A shortcut needs both operands in one
layout: when a dense conv meets a concat (GhostNet adds a 24-channel conv to a 12 + 12 channel concat = 24 of 32 lanes) the conv
adopts the concat's layout (``plan_layouts``).  What the twin cannot express raises ``NotImplementedError`` (there is no eager
fallback): a conv that two shortcuts would need in two layouts, group-split routes of a padded tensor, grouped convs other than
depthwise.  All 39 cfgs of the reference tree that the reference itself can run lower (tests/test_reference_cfgs.py).
"""
import copy
import types

import torch
import torch.nn as nn

ALIGN = 8


def _ru(n, a=ALIGN):
    return (n + a - 1) // a * a


class _Layout:
    __slots__ = ('pos', 'phys', 'origin')

    def __init__(self, pos, phys, origin=None):
        self.pos, self.phys = pos, phys     # LongTensor [C_logical], int
        self.origin = origin                # block index of the dense conv whose (free) output layout this still is, or None

    @property
    def logical(self):
        return int(self.pos.numel())

    def dense(self):
        return self.phys == self.logical

    def same(self, other):
        return self.phys == other.phys and self.logical == other.logical and bool(torch.equal(self.pos, other.pos))


class _Relayout(Exception):
    def __init__(self, block, layout):
        self.block, self.layout = block, layout


def plan_layouts(module_defs, in_channels):
    """(padded block dicts, per-layer layouts).  ``module_defs``: ``Darknet.module_defs`` (the blocks after ``[net]``).

    A dense conv is free to put its output channels anywhere in a padded tensor, a concat is not (its parts sit at 8-aligned
    offsets).  When a shortcut joins the two (GhostNet: a 24-channel conv added to a 12 + 12 channel concat, 24 of 32 lanes), the
    conv is re-planned with the concat's layout and the walk restarts; a conv asked for two different layouts raises."""
    forced = {}
    while True:
        try:
            return _plan_layouts(module_defs, in_channels, forced)
        except _Relayout as r:
            if r.block in forced:
                raise NotImplementedError('HIP training path: conv block %d would need two different padded output layouts' % r.block)
            forced[r.block] = r.layout


def _plan_layouts(module_defs, in_channels, forced):
    defs = copy.deepcopy(list(module_defs))
    body = defs
    layouts = []
    prev = _Layout(torch.arange(in_channels), in_channels)
    for i, d in enumerate(body):
        t = d['type']
        src = layouts[i - 1] if i else prev
        if t == 'convolutional':
            filters, groups = int(d['filters']), int(d.get('groups', 1))
            if groups == 1:
                lay = _Layout(torch.arange(filters), _ru(filters), origin=i)
                if i in forced:
                    assert forced[i].logical == filters
                    lay = _Layout(forced[i].pos.clone(), forced[i].phys, origin=i)
                d['filters'] = lay.phys
            elif groups == src.logical and filters == src.logical:       # depthwise spelled as a grouped conv (GhostNet cfgs)
                lay = _Layout(src.pos.clone(), src.phys, src.origin)
                d['filters'], d['groups'] = src.phys, src.phys
            else:
                raise NotImplementedError('HIP training path: grouped conv (groups=%d) in a padded graph, block %d' % (groups, i))
        elif t == 'depthwise':
            lay = _Layout(src.pos.clone(), src.phys, src.origin)
            d['filters'] = src.phys
        elif t == 'se':
            lay = _Layout(src.pos.clone(), src.phys, src.origin)
            if 'filters' in d:
                d['filters'] = src.phys
        elif t == 'shortcut':
            lay = src
            for l in d['from']:
                other = layouts[i + l if l < 0 else l]
                if not other.same(src) and other.logical == src.logical:
                    # one side still carries a dense conv's own (free) layout: give that conv the other side's layout
                    if src.origin is not None and src.origin not in forced:
                        raise _Relayout(src.origin, other)
                    if other.origin is not None and other.origin not in forced:
                        raise _Relayout(other.origin, src)
                if not other.same(src):
                    raise NotImplementedError('HIP training path: shortcut (block %d) between tensors whose padded channel layouts '
                                              'differ (%d of %d vs %d of %d channels)' % (i, src.logical, src.phys, other.logical, other.phys))
        elif t == 'route':
            parts = [layouts[i + l if l < 0 else l] for l in d['layers']]
            if int(d.get('groups', 0) or 0):
                p = parts[0]
                if len(parts) != 1 or not p.dense() or (p.logical // 2) % ALIGN:
                    raise NotImplementedError('HIP training path: group-split route (block %d) of a padded tensor' % i)
                lay = _Layout(torch.arange(p.logical // 2), p.logical // 2)
            else:
                off, pos = 0, []
                for p in parts:
                    pos.append(p.pos + off)
                    off += p.phys
                lay = _Layout(torch.cat(pos), off)
        else:    # upsample, maxpool, yolo, ...: width and layout of the input
            lay = src
        layouts.append(lay)
    return defs, layouts


class PaddedTwin:
    """The padded twin of ``model`` plus the index maps between the two parameter / buffer sets."""

    def __init__(self, model, ordered_parameters):
        import models
        real_defs = model.module_defs
        in_ch = None
        for block in model.module_list:
            if isinstance(block, nn.Sequential) and len(block) and isinstance(block[0], nn.Conv2d):
                in_ch = block[0].in_channels
                break
        defs, layouts = plan_layouts(real_defs, in_ch)
        self.layouts = layouts
        dev = next(model.parameters()).device
        with torch.random.fork_rng(devices=[]):      # building the twin must not advance the user's RNG stream
            # a list cfg is [hyperparams] + blocks, the form the prune scripts use (slim_prune.py:147-149)
            twin = models.Darknet([copy.deepcopy(model.hyperparams)] + defs, verbose=False,
                                  is_gray_scale=bool(getattr(model, 'is_gray_scale', False)))
        first = next(b[0] for b in twin.module_list if isinstance(b, nn.Sequential) and len(b) and isinstance(b[0], nn.Conv2d))
        if first.in_channels != in_ch:
            raise NotImplementedError('HIP training path: the padded twin reads %d input channels, the model %d'
                                      % (first.in_channels, in_ch))
        self.twin = twin.to(dev).train()
        for attr in ('nc', 'hyp', 'gr'):
            if hasattr(model, attr):
                setattr(self.twin, attr, getattr(model, attr))
        self.real_params = ordered_parameters(model)
        self.twin_params = ordered_parameters(self.twin)
        assert len(self.real_params) == len(self.twin_params)
        self._build_maps(model, in_ch, dev)

    # ------------------------------------------------------------------------------------------------ index maps
    def _build_maps(self, model, in_ch, dev):
        real_off, off = {}, 0
        for p in self.real_params:
            real_off[id(p)] = off
            off += p.numel()
        zero_slot = off                                   # real flat = [all parameters | one zero]
        self.real_numel = off
        to_twin = {}                                      # id(twin param) -> LongTensor(twin.shape) of real-flat positions
        to_real = {}                                      # id(real param) -> LongTensor(real.shape) of positions inside the twin tensor
        buf_pairs = []                                    # (real buffer, twin buffer, pos)
        self._bn_pairs = []
        inputs = _Layout(torch.arange(in_ch), in_ch)

        def link(rp, tp, index):
            """index: tuple of broadcastable LongTensors addressing, inside ``tp``, the block that holds ``rp``."""
            src = torch.full(tuple(tp.shape), zero_slot, dtype=torch.long)
            src[index] = real_off[id(rp)] + torch.arange(rp.numel()).view(tuple(rp.shape))
            to_twin[id(tp)] = src
            to_real[id(rp)] = torch.arange(tp.numel()).view(tuple(tp.shape))[index].reshape(tuple(rp.shape))

        for i, (rb, tb) in enumerate(zip(model.module_list, self.twin.module_list)):
            lay = self.layouts[i]
            src = self.layouts[i - 1] if i else inputs
            if isinstance(rb, nn.Sequential) and len(rb) and isinstance(rb[0], nn.Conv2d):
                rc, tc = rb[0], tb[0]
                o = lay.pos
                if rc.groups == 1:
                    link(rc.weight, tc.weight, (o[:, None], src.pos[None, :]))
                else:                                      # depthwise: one filter per channel, rows follow the input layout
                    link(rc.weight, tc.weight, (o,))
                if rc.bias is not None:
                    link(rc.bias, tc.bias, (o,))
                for rk, tk in zip(list(rb.children())[1:], list(tb.children())[1:]):
                    if isinstance(rk, nn.modules.batchnorm.BatchNorm2d):
                        link(rk.weight, tk.weight, (o,))
                        link(rk.bias, tk.bias, (o,))
                        buf_pairs.append((rk.running_mean, tk.running_mean, o))
                        buf_pairs.append((rk.running_var, tk.running_var, o))
                        self._bn_pairs.append((rk, tk))
            elif isinstance(rb, nn.Sequential) and len(rb) and rb[0].__class__.__name__ == 'SE':
                r0, r2, t0, t2 = rb[0].fc[0], rb[0].fc[2], tb[0].fc[0], tb[0].fc[2]
                hid = torch.arange(r0.weight.shape[0])
                link(r0.weight, t0.weight, (hid[:, None], src.pos[None, :]))
                link(r2.weight, t2.weight, (src.pos[:, None], hid[None, :]))
        missing = [k for k, p in enumerate(self.real_params) if id(p) not in to_real]
        if missing:
            raise NotImplementedError('HIP training path: %d parameters of the padded graph have no layout rule' % len(missing))
        # twin parameters become views of one flat buffer filled by ONE gather per step
        total = sum(p.numel() for p in self.twin_params)
        self.twin_flat = torch.zeros(total, device=dev, dtype=torch.float32)
        gather, t_off = [], 0
        with torch.no_grad():
            for tp in self.twin_params:
                n = tp.numel()
                tp.data = self.twin_flat[t_off:t_off + n].view(tuple(tp.shape))
                gather.append(to_twin[id(tp)].reshape(-1))
                t_off += n
        self.push_index = torch.cat(gather).to(dev)
        self.real_flat = torch.zeros(self.real_numel + 1, device=dev, dtype=torch.float32)
        self.real_views = []
        r_off = 0
        for p in self.real_params:
            self.real_views.append(self.real_flat[r_off:r_off + p.numel()].view(tuple(p.shape)))
            r_off += p.numel()
        self.grad_index = {id(rp): to_real[id(rp)].reshape(-1) for rp in self.real_params}
        # BatchNorm running statistics: twin buffers as views of one flat tensor; one gather brings them home
        nbuf = sum(t.numel() for _, t, _ in buf_pairs)
        self.buf_flat = torch.zeros(max(nbuf, 1), device=dev, dtype=torch.float32)
        idx, b_off = [], 0
        self.real_bufs = []
        with torch.no_grad():
            for rbuf, tbuf, pos in buf_pairs:
                n = tbuf.numel()
                init = tbuf.detach().clone()
                tbuf.data = self.buf_flat[b_off:b_off + n]
                tbuf.copy_(init)          # pad lanes keep BatchNorm's defaults (mean 0, variance 1)
                tbuf[pos.to(dev)] = rbuf  # the twin starts from the live statistics and is authoritative from then on
                idx.append(pos + b_off)
                self.real_bufs.append(rbuf)
                b_off += n
        self.pull_index = torch.cat(idx).to(dev) if idx else torch.zeros(0, dtype=torch.long, device=dev)
        self.pull_sizes = [rb.numel() for rb in self.real_bufs]
        self.buf_pairs = buf_pairs

    # ------------------------------------------------------------------------------------------------ per step
    @torch.no_grad()
    def push(self):
        """Live parameters -> twin, one gather (pad rows / columns / lanes read the zero slot); live BatchNorm running
        statistics -> twin, one scatter, so that in-place edits of the live buffers between steps (DDP's per-forward
        ``broadcast_buffers`` on ranks > 0, a script resetting ``running_mean``) reach the kernels instead of being reverted by
        the next ``pull_stats``."""
        torch._foreach_copy_(self.real_views, [p.detach() for p in self.real_params])
        torch.index_select(self.real_flat, 0, self.push_index, out=self.twin_flat)
        if self.real_bufs:
            # only when somebody wrote to a live buffer since the last pull_stats (tensor version counters: ~150 integer reads instead
            # of a cat + scatter per step)
            # Writes through `.data` views (`bn.running_mean.data.zero_()`) do not move a version counter: rebinding shows
            # in the data pointers tracked with the versions, and every 64th step syncs unconditionally - after a pull_stats the live
            # buffers equal the twin's, so the forced sync changes nothing in a normal run and bounds how long such an edit can stay
            # unseen (Darknet.hip_refresh() is the immediate way).
            seen = self.__dict__.get('_buf_versions')
            now = self._live_versions()
            self._push_count = self.__dict__.get('_push_count', 0) + 1
            if seen != now or self._push_count % 64 == 0:
                self.buf_flat.index_copy_(0, self.pull_index, torch.cat([b.reshape(-1) for b in self.real_bufs]))
                pairs = [(rk.num_batches_tracked, tk.num_batches_tracked) for rk, tk in self._bn_pairs if rk.num_batches_tracked is not None]
                if pairs:      # a live reset of the step counter reaches the twin too
                    torch._foreach_copy_([b for _, b in pairs], [a for a, _ in pairs])

    @torch.no_grad()
    def pull_stats(self):
        if not self.real_bufs:
            return
        got = torch.index_select(self.buf_flat, 0, self.pull_index)
        torch._foreach_copy_(self.real_bufs, list(torch.split(got, self.pull_sizes)))
        pairs = [(rk.num_batches_tracked, tk.num_batches_tracked) for rk, tk in self._bn_pairs if rk.num_batches_tracked is not None]
        if pairs:
            torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
        self._buf_versions = self._live_versions()      # what the live buffers look like when WE wrote them last

    def _live_versions(self):
        return [(b._version, b.data_ptr()) for b in self.real_bufs] + [rk.num_batches_tracked._version for rk, _ in self._bn_pairs
                                                                        if rk.num_batches_tracked is not None]

    def map_grads(self, real_subset, twin_grads):
        """Gradients of the twin parameters of one backward range -> gradients of the matching live parameters."""
        if not twin_grads:
            return []
        flat = torch.cat([g.reshape(-1) for g in twin_grads])
        key = tuple(id(p) for p in real_subset)
        cache = self.__dict__.setdefault('_range_index', {})
        if key not in cache:
            idx, off = [], 0
            for rp, tg in zip(real_subset, twin_grads):
                idx.append(self.grad_index[id(rp)].to(flat.device) + off)
                off += tg.numel()
            cache[key] = torch.cat(idx)
        out = torch.index_select(flat, 0, cache[key])
        return [t.view(tuple(p.shape)) for t, p in zip(torch.split(out, [p.numel() for p in real_subset]), real_subset)]


class PaddedTrainEngine:
    """``TrainEngine`` of the padded twin behind the interface ``models.Darknet`` drives (forward / backward ranges)."""

    def __init__(self, model, precision='fp16', lib=None):
        from .train import TrainEngine
        order = lambda m: TrainEngine.parameters(types.SimpleNamespace(model=m))
        self.pad = PaddedTwin(model, order)
        self.inner = TrainEngine(self.pad.twin, precision, lib)
        self.model = model
        self.precision = precision
        self.step_heads = None
        self._twin_to_real = {id(t): r for t, r in zip(self.pad.twin_params, self.pad.real_params)}

    # the attributes models.Darknet / bench.py read
    device = property(lambda self: self.inner.device)
    steps = property(lambda self: self.inner.steps)
    _current = property(lambda self: self.inner._current)
    lib = property(lambda self: self.inner.lib)

    def _get_plan(self, x):
        return self.inner._get_plan(x)

    def parameters(self):
        return list(self.pad.real_params)

    def segment_parameters(self, plan):
        return [[self._twin_to_real[id(t)] for t in seg] for seg in self.inner.segment_parameters(plan)]

    def forward(self, x):
        self.pad.push()
        heads = self.inner.forward(x)
        self.pad.pull_stats()
        return heads

    def backward_segment(self, k, head_grads):
        plan = self.inner._current
        twin_grads = self.inner.backward_segment(k, head_grads)
        real_subset = self.segment_parameters(plan)[k]
        return self.pad.map_grads(real_subset, twin_grads)

    def backward(self, head_grads):
        plan = self.inner._current
        out = [None] * len(plan['segments'])
        for k in reversed(range(len(plan['segments']))):
            out[k] = self.backward_segment(k, [head_grads[j] for j in plan['segments'][k]['heads']])
        return [g for seg in out for g in seg]


def make_train_engine(model, precision, x, lib=None):
    """The training engine for ``model`` and input batch ``x``: the aligned lowering when every width is a multiple of 8, the
    padded twin otherwise.  Plan build errors (NotImplementedError) surface here, before any state changes."""
    from .train import TrainEngine, ChannelAlignmentError
    try:
        eng = TrainEngine(model, precision, lib)
        eng._get_plan(x)
    except ChannelAlignmentError:
        eng = PaddedTrainEngine(model, precision, lib)
        eng._get_plan(x)
    return eng
