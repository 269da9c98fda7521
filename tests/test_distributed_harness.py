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
