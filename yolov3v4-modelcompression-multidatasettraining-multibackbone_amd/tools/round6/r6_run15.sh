#!/bin/bash
# fp16 training kernels with the fast Mish forms (mish_fast forward, mish_grad_fast backward): YOLOv4-608 b32 training step + layer table, tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6p; mkdir -p $O
timeout 600 python bench.py --mode train --cfg $PKG/cfg/yolov4/yolov4.cfg --size 608 --batch 32 --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['metric'], d['value'], d['ms_per_step'], r.get('gpu_ms_per_step'), r.get('batchnorm_passes_ms'), r.get('by_role_ms'), d['config']['loss'])" > $O/v4_train.txt 2>&1
cat $O/v4_train.txt
( timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -30 ) > $O/t.txt 2>&1
cat $O/t.txt
