"""The first optimisation steps of the reference's training loop as a protocol - TEST INFRASTRUCTURE ONLY.

``run`` is /root/reference/train.py:112-144 (the three parameter groups, nesterov SGD) and :344-455 (per batch: burn-in
interpolation of lr / weight decay / momentum / giou ratio / accumulate count, forward, compute_loss, `loss *= batch / 64`,
backward, `optimizer.step()` every `accumulate` batches) written once, for whatever ``model`` / ``compute_loss`` pair it is
handed: tests/golden/make_golden_sgd608.py runs it on the REFERENCE's modules (fp32, CPU) and stores the trajectory,
tests/test_gpu_train.py runs it on this package's Darknet on the GPU (fp32 engine; fp16 engine under autocast + GradScaler, the
reference's --mpt branch, train.py:371-374,436-454) and compares step by step.

Recorded per step: the four loss items, the l2 norm of all parameters and the l2 norm of their displacement from the initial
point (the parameter norm alone moves by 1e-4 in ten steps; the displacement is what an update-rule or gradient error shows up
in).  Inputs: a fresh seeded batch and label set per step (synth.image_batch / synth.loss_inputs).
"""
import math

import numpy as np
import torch

import synth

HYP = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'lr0': 0.01, 'lrf': 0.0005,
       'momentum': 0.937, 'weight_decay': 0.0005, 'fl_gamma': 0.0}      # train.py:25-35
N_BURN = 500                                                             # train.py:311 max(3 * nb, 500) with a short epoch
EPOCHS = 300


def param_groups(model):
    pg0, pg1, pg2 = [], [], []
    for k, v in dict(model.named_parameters()).items():                  # train.py:125-133
        if '.bias' in k:
            pg2.append(v)
        elif 'Conv2d.weight' in k:
            pg1.append(v)
        else:
            pg0.append(v)
    return pg0, pg1, pg2


def run(model, compute_loss, steps, size, batch, device='cpu', mixed=False, nc=80, labels_per_image=8, on_step=None):
    """Returns dict(items (steps, 4), pnorm (steps,), dnorm (steps,), stepped (steps,) bool) after `steps` batches."""
    pg0, pg1, pg2 = param_groups(model)
    opt = torch.optim.SGD(pg0, lr=HYP['lr0'], momentum=HYP['momentum'], nesterov=True)     # train.py:142-144
    opt.add_param_group({'params': pg1, 'weight_decay': HYP['weight_decay']})
    opt.add_param_group({'params': pg2})
    for g in opt.param_groups:
        g['initial_lr'] = g['lr']                                        # what LambdaLR's constructor records (train.py:202)
    lf0 = (((1 + math.cos(0 * math.pi / EPOCHS)) / 2) ** 1.0) * 0.95 + 0.05       # train.py:201 at epoch 0
    model.nc, model.hyp = nc, dict(HYP)
    model.train()
    scaler = torch.amp.GradScaler('cuda', enabled=mixed)
    params = [p for g in (pg0, pg1, pg2) for p in g]
    start = [p.detach().clone() for p in params]
    items_all, pnorm, dnorm, stepped = [], [], [], []
    opt.zero_grad()
    for ni in range(steps):
        x = synth.image_batch(batch, size, seed=1000 + ni).to(device)
        targets = synth.loss_inputs(model, size, batch=batch, seed=2000 + ni, labels_per_image=labels_per_image)[1].to(device)
        xi = [0, N_BURN]                                                 # train.py:356-365
        model.gr = float(np.interp(ni, xi, [0.0, 1.0]))
        accumulate = max(1, np.interp(ni, xi, [1, 64 / batch]).round())
        for j, g in enumerate(opt.param_groups):
            g['lr'] = float(np.interp(ni, xi, [0.1 if j == 2 else 0.0, g['initial_lr'] * lf0]))
            g['weight_decay'] = float(np.interp(ni, xi, [0.0, HYP['weight_decay'] if j == 1 else 0.0]))
            g['momentum'] = float(np.interp(ni, xi, [0.9, HYP['momentum']]))
        if mixed:
            with torch.autocast('cuda', dtype=torch.float16):
                pred = model(x)[0]
        else:
            pred = model(x)[0]
        loss, items = compute_loss(pred, targets, model)
        assert torch.isfinite(loss)
        loss = loss * (batch / 64)                                       # train.py:435
        scaler.scale(loss).backward() if mixed else loss.backward()
        did = ni % accumulate == 0                                       # train.py:450-456
        if did:
            if mixed:
                scale = scaler.get_scale()
                scaler.step(opt)
                scaler.update()
                assert scaler.get_scale() >= scale, 'GradScaler skipped step %d (overflow)' % ni
            else:
                opt.step()
            opt.zero_grad()
        with torch.no_grad():
            pn = math.sqrt(sum(float(p.double().pow(2).sum()) for p in params))
            dn = math.sqrt(sum(float((p.double() - s.double()).pow(2).sum()) for p, s in zip(params, start)))
        items_all.append(items.detach().float().cpu().numpy())
        pnorm.append(pn)
        dnorm.append(dn)
        stepped.append(bool(did))
        if on_step is not None:
            on_step(ni, items_all[-1], pn, dn)
    return dict(items=np.stack(items_all), pnorm=np.array(pnorm), dnorm=np.array(dnorm), stepped=np.array(stepped))
