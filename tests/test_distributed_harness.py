"""N > 1 path of bench.py on CPU: two gloo ranks, barrier-bracketed timing, max-over-ranks aggregation."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    from engine import distutil
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    delay = 0.02 if rank == 0 else 0.06  # rank 1 is the straggler
    elapsed = distutil.timed_region(lambda: time.sleep(delay), 5, dist, torch.device('cpu'))
    rate = distutil.aggregate_rate(8, 5, elapsed, world)
    out.put((rank, elapsed, rate))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_uses_slowest_rank():
    import conftest  # noqa: F401  (children re-import through the same sys.path via spawn + PYTHONPATH)
    os.environ['PYTHONPATH'] = os.pathsep.join([conftest.PKG, conftest.REPO, os.environ.get('PYTHONPATH', '')])
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, e0, r0), (_, e1, r1) = res
    assert abs(e0 - e1) < 1e-9, 'both ranks must report the same (max) time'
    assert e0 >= 5 * 0.06 * 0.95, 'time must be the straggler rank\'s'
    assert abs(r0 - 2 * 8 * 5 / e0) < 1e-6 and r0 == r1


# ---------------------------------------------------------------------------- data-parallel training
def _ddp_worker(rank, world, port, cfg_path, out):
    """One DDP rank on CPU: the HIP training path replayed on the host emulation of the C ABI, gradients all-reduced by
    DistributedDataParallel over gloo exactly as RCCL does on the GPUs (train.py:283-286 wraps the model the same way)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import conftest  # noqa: F401
    import fakelib
    import synth
    import train_harness as th
    from engine.train import TrainEngine
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    model = th.build(cfg_path, 64)                       # same seed on both ranks: identical initial weights
    model.__dict__['_hip_train_engine'] = TrainEngine(model, 'fp32', lib=fakelib.FakeLib())
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    x = synth.image_batch(2, 64, seed=10 + rank)          # each rank its own shard of the minibatch
    os.environ['YOLO_HIP_TRAIN_PRECISION'] = 'fp32'
    # DDP calls model.forward; route the CPU tensors through the HIP path (on a GPU box x.is_cuda does this)
    model._use_hip_train = lambda inp: True
    raws, _ = ddp(x)
    ws = th.loss_weights(raws, seed=5)
    th.toy_loss(raws, ws).backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    # local (un-averaged) gradients of the same shard, for the parent to average itself
    local = th.build(cfg_path, 64)
    local.__dict__['_hip_train_engine'] = TrainEngine(local, 'fp32', lib=fakelib.FakeLib())
    raws_l, _ = local._forward_hip_train(x)
    th.toy_loss(raws_l, ws).backward()
    out.put((rank, {k: v.numpy() for k, v in grads.items()}, {k: p.grad.numpy() for k, p in local.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_training_step_averages_hip_path_gradients():
    import numpy as np
    import conftest
    import train_harness as th
    os.environ['PYTHONPATH'] = os.pathsep.join([conftest.PKG, conftest.REPO, os.environ.get('PYTHONPATH', '')])
    cfg_path = th.write_cfg(th.mini_cfg_text())
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, cfg_path, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    os.unlink(cfg_path)
    (_, g0, l0), (_, g1, l1) = res
    for k in g0:
        assert np.array_equal(g0[k], g1[k]), 'ranks disagree on %s after the all-reduce' % k
        mean = (l0[k] + l1[k]) / 2
        assert np.abs(g0[k] - mean).max() <= 1e-5 * (np.abs(mean).max() + 1e-12), k


# ------------------------------------------------- the wrapper bench.py builds (bucket views, fp16 compression, overlap)
def _ddp_bench_worker(rank, world, port, cfg_path, out):
    """DDP exactly as bench.py:train_main builds it - gradient_as_bucket_view=True, a bucket cap that splits the gradients over
    several buckets, torch's fp16 compression hook - over the chain of `_HipTrainSegment` autograd nodes, and a log of WHEN each
    bucket's all-reduce is launched relative to the backward ranges (reference train.py:99-107,218-223 relies on DDP's default
    overlap of bucket all-reduces with the rest of backward)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import conftest  # noqa: F401
    import fakelib
    import synth
    import train_harness as th
    from engine.train import TrainEngine
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      YOLO_HIP_TRAIN_SEGMENTS='4', YOLO_HIP_TRAIN_PRECISION='fp32')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    model = th.build(cfg_path, 64)
    eng = TrainEngine(model, 'fp32', lib=fakelib.FakeLib())
    model.__dict__['_hip_train_engine'] = eng
    events = []
    inner = eng.backward_segment

    def logged_segment(k, head_grads):
        events.append(('range_start', k))
        res = inner(k, head_grads)
        events.append(('range_done', k))
        return res
    eng.backward_segment = logged_segment
    ddp = torch.nn.parallel.DistributedDataParallel(model, bucket_cap_mb=0.02, gradient_as_bucket_view=True)

    def hook(state, bucket):
        events.append(('bucket', bucket.index(), sum(g.numel() for g in bucket.gradients())))
        return default_hooks.fp16_compress_hook(state, bucket)
    ddp.register_comm_hook(None, hook)
    model._use_hip_train = lambda inp: True
    x = synth.image_batch(2, 64, seed=20 + rank)
    for step in range(2):      # DDP rebuilds its buckets in gradient-ready order after the first step; check the steady state
        del events[:]
        for p in model.parameters():
            p.grad = None
        raws, _ = ddp(x)
        ws = th.loss_weights(raws, seed=5)
        th.toy_loss(raws, ws).backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    plan = eng._current
    out.put((rank, list(events), len(plan['segments']), {k: v.numpy() for k, v in grads.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_as_bench_builds_it_launches_buckets_while_backward_is_still_running():
    import numpy as np
    import conftest
    import train_harness as th
    os.environ['PYTHONPATH'] = os.pathsep.join([conftest.PKG, conftest.REPO, os.environ.get('PYTHONPATH', '')])
    cfg_path = th.write_cfg(th.mini_cfg_text())
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_bench_worker, args=(r, 2, port, cfg_path, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    os.unlink(cfg_path)
    (_, ev0, nseg, g0), (_, ev1, _, g1) = res
    for k in g0:     # fp16-compressed all-reduce: both ranks end with the same averaged gradients
        assert np.array_equal(g0[k], g1[k]), k
        assert np.isfinite(g0[k]).all()
    assert nseg >= 3, 'the mini net must split into several backward ranges for this test to mean anything'
    for ev in (ev0, ev1):
        buckets = [i for i, e in enumerate(ev) if e[0] == 'bucket']
        ranges_done = {e[1]: i for i, e in enumerate(ev) if e[0] == 'range_done'}
        starts = {e[1]: i for i, e in enumerate(ev) if e[0] == 'range_start'}
        assert len(buckets) >= 3, 'bucket cap must split the gradients: %s' % ev
        # backward ranges run last -> first (range nseg-1 holds the heads' side of the net)
        assert [e[1] for e in ev if e[0] == 'range_start'] == list(range(nseg - 1, -1, -1))
        # overlap: the first bucket is launched before the LAST range (range 0) has even started, and every range but the
        # last has had at least one bucket launched by the time the next one starts
        assert buckets[0] < starts[0], 'no bucket was launched before the final backward range started: %s' % ev
        launched_before_last = sum(1 for b in buckets if b < starts[0])
        assert launched_before_last >= len(buckets) // 2, 'most buckets must be in flight while range 0 is still to run: %s' % ev
        assert max(buckets) > ranges_done[0] or buckets[-1] > starts[0], 'the last bucket closes after the last range'


# ------------------------------------------------- the padded twin under DDP's buffer broadcast; find_unused_parameters
def _pruned_cfg_text():
    """A slim-pruned-style graph (widths off the 8-channel grid): trains through engine/padded.py's channel-padded twin."""
    import train_harness as th
    c = lambda f, k, s: th._CONV % (f, k, s, 'leaky')
    sc = '[shortcut]\nfrom=-3\nactivation=linear\n\n'
    return ('[net]\nbatch=1\nwidth=64\nheight=64\nchannels=3\n\n'
            + c(5, 3, 1) + c(13, 3, 2) + c(3, 1, 1) + c(13, 3, 1) + sc
            + c(27, 3, 2) + c(9, 1, 1) + c(27, 3, 1) + sc
            + c(11, 1, 1) + c(30, 3, 1) + th._HEAD + th._YOLO % '3,4,5'
            + '[route]\nlayers = -4\n\n' + c(6, 1, 1) + '[upsample]\nstride=2\n\n'
            + '[route]\nlayers = -1, 4\n\n' + c(10, 1, 1) + c(19, 3, 1) + th._HEAD + th._YOLO % '0,1,2')


def _ddp_padded_worker(rank, world, port, cfg_path, use_engine, find_unused, out):
    """Two training steps under DistributedDataParallel with its DEFAULT broadcast_buffers=True (rank 0's BatchNorm buffers
    overwrite every other rank's live buffers at the start of each forward - reference train.py:218-223 builds DDP that way), on
    the padded-twin engine (host emulation) or on the eager modules; returns the running statistics and gradients per rank."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import conftest  # noqa: F401
    import fakelib
    import synth
    import train_harness as th
    from engine.padded import make_train_engine, PaddedTrainEngine
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), YOLO_HIP_TRAIN_PRECISION='fp32')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    model = th.build(cfg_path, 64)
    if rank == 1:      # make the ranks' buffers differ before the first broadcast: a stale twin would keep rank 1's values
        with torch.no_grad():
            for k, v in model.state_dict().items():
                if k.endswith('running_mean'):
                    v.add_(0.5)
    x = synth.image_batch(2, 64, seed=30 + rank)
    padded = False
    if use_engine:
        eng = make_train_engine(model, 'fp32', x, lib=fakelib.FakeLib())
        model.__dict__['_hip_train_engine'] = eng
        padded = isinstance(eng, PaddedTrainEngine)
        model._use_hip_train = lambda inp: True
    ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=find_unused)      # broadcast_buffers defaults to True
    for step in range(2):
        for p in model.parameters():
            p.grad = None
        raws = ddp(x)[0]
        ws = th.loss_weights(raws, seed=5)
        th.toy_loss(raws, ws).backward()
    stats = {k: v.clone().numpy() for k, v in model.state_dict().items() if 'running' in k or 'num_batches' in k}
    grads = {k: p.grad.clone().numpy() for k, p in model.named_parameters()}
    out.put((rank, padded, stats, grads))
    dist.barrier()
    dist.destroy_process_group()


def _run_ddp_padded(use_engine, find_unused=False):
    import conftest
    import train_harness as th
    os.environ['PYTHONPATH'] = os.pathsep.join([conftest.PKG, conftest.REPO, os.environ.get('PYTHONPATH', '')])
    cfg_path = th.write_cfg(_pruned_cfg_text())
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_padded_worker, args=(r, 2, port, cfg_path, use_engine, find_unused, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    os.unlink(cfg_path)
    return res


def test_padded_twin_under_ddp_buffer_broadcast_tracks_the_eager_modules():
    """ADVICE r2 / VERDICT r3 item 8: `pull_stats()` must not revert what DDP's per-forward buffer broadcast wrote into the live
    BatchNorm buffers.  Engine run and eager run of the same two-rank, two-step schedule: every rank ends with the same running
    statistics and the same averaged gradients (rank 1 started from shifted running means, which the first broadcast replaces)."""
    import numpy as np
    eng, ref = _run_ddp_padded(True), _run_ddp_padded(False)
    assert all(r[1] for r in eng), 'the odd-width graph must run on the padded twin'
    for (rank, _, st_e, g_e), (_, _, st_r, g_r) in zip(eng, ref):
        for k in st_r:
            assert np.abs(st_e[k].astype(np.float64) - st_r[k]).max() <= 1e-5 * (np.abs(st_r[k]).max() + 1), (rank, k)
        for k in g_r:
            assert np.abs(g_e[k] - g_r[k]).max() <= 2e-4 * (np.abs(g_r[k]).max() + 1e-6), (rank, k)
    # both ranks hold rank 0's momentum history: only the last step's own batch statistics differ, by at most momentum x the spread
    for k in eng[0][2]:
        if k.endswith('running_mean'):
            assert np.abs(eng[0][2][k] - eng[1][2][k]).max() < 0.2, k       # (the 0.5 shift of rank 1 is gone)


def test_ddp_with_find_unused_parameters_as_the_reference_passes_it():
    """Reference train.py:219 wraps the model with find_unused_parameters=True: DDP then walks the autograd graph from the outputs
    to mark unused parameters.  The `_HipTrainSegment` chain must present every parameter as used, and the averaged gradients must
    equal the plain wrapper's."""
    import numpy as np
    a, b = _run_ddp_padded(True, find_unused=True), _run_ddp_padded(True, find_unused=False)
    for (rank, _, _, g_a), (_, _, _, g_b) in zip(a, b):
        for k in g_b:
            assert np.isfinite(g_a[k]).all() and np.array_equal(g_a[k], g_b[k]), (rank, k)
    for k in a[0][3]:
        assert np.array_equal(a[0][3][k], a[1][3][k]), 'ranks disagree after the all-reduce: %s' % k
