#!/bin/bash
# picker rules of r6_run26 (32-row tile only for <= 32 outputs over one K step; wide ring tiles from two workgroups per CU; fp16 64 -> 64 3x3 off the halo kernel;
# single tiny weight-gradient tile at -1024): training steps of the three families, detect legs, GPU tier
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
PKG=yolov3v4-modelcompression-multidatasettraining-multibackbone_amd
O=gpurun_out/r6aa; mkdir -p $O
( timeout 900 python bench.py --families --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench_families.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6aa/bench_families.json').read().strip().splitlines()[-1])
print('train', d['value'], d['ms_per_step'], d['roofline']['gpu_ms_per_step'])
for k in ('detect', 'detect_int8', 'detect_v4_640', 'detect_int8_v4_640', 'train_mobilenet_416', 'train_v4_608'):
    t = d.get(k) or {}
    print(k, t.get('value'), t.get('ms_per_step'), (t.get('roofline') or {}).get('gpu_ms_per_step'), d.get(k + '_error', ''))
PY
timeout 300 python $PKG/tools/profile_train.py --cfg $PKG/cfg/yolov3-mobilenet/yolov3-mobilenet-coco.cfg --size 416 --batch 64 2>&1 | tail -22 | head -8
timeout 300 python $PKG/tools/profile_train.py --cfg $PKG/cfg/yolov4/yolov4.cfg --size 608 --batch 32 2>&1 | tail -17 | head -4
( timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep "^E  \|passed\|failed\|FAILED" | cut -c1-300 | head -20 ) > $O/t.txt 2>&1
cat $O/t.txt
