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
    ap.add_argument('--cfg', default=os.path.join(PKG, 'cfg', 'yolov3tiny', 'yolov3-tiny.cfg'))
    ap.add_argument('--size', type=int, default=416)
    ap.add_argument('--batch', type=int, default=8)
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
    fired = []

    def hook(state, bucket):
        fired.append((bucket.index(), sum(g.numel() for g in bucket.gradients())))
        return default_hooks.fp16_compress_hook(state, bucket)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, bucket_cap_mb=args.bucket_mb,
                                                    gradient_as_bucket_view=True)
    ddp.register_comm_hook(None, hook)
    ddp.yolo_layers = model.yolo_layers
    ddp.nc, ddp.hyp, ddp.gr = 80, hyp, 1.0
    g = torch.Generator().manual_seed(3)
    x = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(dev)
    targets = torch.tensor([[b, b % 80, 0.3 + 0.04 * (b % 8), 0.5, 0.25, 0.3] for b in range(args.batch)], device=dev)
    # a FIXED loss scale of 256: the fp16 compression hook casts the buckets to fp16, where GradScaler's initial 65536 overflows (in
    # training GradScaler then backs its scale off, as for any fp16 overflow) and unscaled fp16 gradients underflow
    for step in range(3):
        for net in (ddp, plain):
            for p in net.parameters():
                p.grad = None
            with torch.autocast('cuda', dtype=torch.float16):
                pred, _ = net(x)
            loss, _ = compute_loss(pred, targets, net)
            (loss * 256.0).backward()
    torch.cuda.synchronize()
    assert model.__dict__.get('_hip_train_engine') is not None, 'the HIP training path did not engage under DDP'
    num = den = 0.0
    for a, b in zip(model.parameters(), plain.parameters()):
        num += (a.grad.float() - b.grad.float()).norm().item() ** 2
        den += b.grad.float().norm().item() ** 2
    rel = (num / den) ** 0.5
    print('DDP(HIP step) over RCCL: %d bucket all-reduces in 3 steps (bucket index, elements): %s; fp16-compressed; gradient '
          'difference to the unwrapped model %.2e' % (len(fired), fired, rel))
    assert len(fired) >= 3 and rel < 2e-3, rel     # fp16 compression of the buckets rounds the gradients once
    dist.destroy_process_group()
    print('rccl probe ok')


if __name__ == '__main__':
    main()
