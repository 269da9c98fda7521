"""Single-rank RCCL probe for a 1-GPU box: the 8-GPU scaling run is the first time the data-parallel path meets the real backend,
so prove everything that can be proven with one device (reference train.py:99-107 initialises NCCL the same way):

* `init_process_group('nccl')` (= RCCL) comes up with HSA_ENABLE_IPC_MODE_LEGACY=0, world size 1, 127.0.0.1 rendezvous;
* an all-reduce, a broadcast and a barrier execute on the communicator;
* DistributedDataParallel built EXACTLY as bench.py builds it (gradient_as_bucket_view=True, bucket cap, optional fp16 compression
  hook) wraps the HIP training step: the bucket hooks fire from the `_HipTrainSegment` chain, the all-reduces run on RCCL's
  stream, and the gradients equal those of the unwrapped model.

    NCCL_DEBUG=INFO python tools/rccl_probe.py [--cfg cfg/yolov3tiny/yolov3-tiny.cfg --size 416 --batch 8]
"""
import argparse
import copy
import os
import socket
import sys

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, PKG)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default=os.path.join(PKG, 'cfg', 'yolov3', 'yolov3.cfg'))
    ap.add_argument('--size', type=int, default=320)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--bucket-mb', type=int, default=25)
    args = ap.parse_args()
    assert torch.cuda.is_available()
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
    print('backend', dist.get_backend(), 'world', dist.get_world_size(), 'HSA_ENABLE_IPC_MODE_LEGACY', os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])
    t = torch.arange(1 << 20, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    dist.broadcast(t, 0)
    dist.barrier()
    torch.cuda.synchronize()
    assert float(t[12345]) == 12345.0
    print('all_reduce / broadcast / barrier on RCCL: ok')

    from models import Darknet
    from utils.utils import compute_loss
    hyp = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'fl_gamma': 0.0}
    torch.manual_seed(0)
    model = Darknet(args.cfg, (args.size, args.size)).to(dev)
    plain = copy.deepcopy(model)
    for m in (model, plain):
        m.nc, m.hyp, m.gr = 80, hyp, 1.0
        m.train()
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    g = torch.Generator().manual_seed(3)
    x = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(dev)
    targets = torch.tensor([[b, b % 80, 0.3 + 0.04 * (b % 8), 0.5, 0.25, 0.3] for b in range(args.batch)], device=dev)

    def grads_of(net, steps=2):
        for _ in range(steps):
            for p in net.parameters():
                p.grad = None
            # fp32 step (no autocast): run to run it differs only by the order of the weight-gradient split sums, so a wiring
            # mistake in the DDP path cannot hide behind kink flips (an fp16 step of a random-weight net moves by several per cent
            # from one run to the next: packed-fp16 atomics in the max-pool backward, leaky kinks downstream).  The loss is scaled
            # by 16 so that the fp16-compressed buckets neither underflow nor overflow.
            pred, _ = net(x)
            loss, _ = compute_loss(pred, targets, net)
            (loss * 16.0).backward()
        torch.cuda.synchronize()
        core = net.module if hasattr(net, 'module') else net
        return [p.grad.detach().float().clone() for p in core.parameters()]

    def rel(a, b):
        num = sum((u - v).norm().item() ** 2 for u, v in zip(a, b))
        return (num / sum(v.norm().item() ** 2 for v in b)) ** 0.5

    want = grads_of(plain)
    again = grads_of(plain)
    noise = rel(again, want)
    print('run-to-run difference of the unwrapped HIP step (order of the split sums): %.2e' % noise)
    for compress in (False, True):
        net = copy.deepcopy(model)
        net.nc, net.hyp, net.gr = 80, hyp, 1.0
        net.train()
        fired = []
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], output_device=0, bucket_cap_mb=args.bucket_mb,
                                                        gradient_as_bucket_view=True)

        def hook(state, bucket, fired=fired, compress=compress):
            fired.append((bucket.index(), sum(t.numel() for t in bucket.gradients())))
            return (default_hooks.fp16_compress_hook if compress else default_hooks.allreduce_hook)(state, bucket)
        ddp.register_comm_hook(None, hook)
        ddp.yolo_layers = net.yolo_layers
        ddp.nc, ddp.hyp, ddp.gr = 80, hyp, 1.0
        got = grads_of(ddp)
        assert net.__dict__.get('_hip_train_engine') is not None, 'the HIP training path did not engage under DDP'
        r = rel(got, want)
        worst = sorted(((float((u - v).norm() / (v.norm() + 1e-20)), k) for (k, _), u, v in zip(net.named_parameters(), got, want)), reverse=True)[:3]
        print('DDP(HIP step) over RCCL, %s: %d bucket all-reduces in 2 steps %s; gradient difference to the unwrapped model %.2e; '
              'worst parameters %s' % ('fp16-compressed buckets' if compress else 'fp32 buckets', len(fired), fired, r,
                                       [(round(a, 4), k) for a, k in worst]))
        assert len(fired) >= 2
        assert r < (2e-3 if compress else 1e-4) + 4 * noise, (r, noise)
        del ddp, net
    dist.destroy_process_group()
    print('rccl probe ok')


if __name__ == '__main__':
    main()
